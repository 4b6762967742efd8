#!/usr/bin/env python3
"""Single-layer timing of the 3x3 stride-1 convolution kernels at the CIFAR step's launch shapes (developer tool).

    python tools/bench_conv.py [--gen f4|auto|f2|igemm] [--reps N] [--check]

Prints ms per launch (HIP events around `reps` back-to-back launches, weights pre-transformed once through a throw-away
dlpm_conv2d_f32 call is NOT possible -- that entry point re-lays the weights out on every call, so the relayout kernels are
timed separately and subtracted), algorithmic TFLOP/s, and a digest of the output (to compare builds bit for bit)."""
import argparse
import ctypes as C
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from dlpm_amd import _lib

L = _lib.lib()
DEV = 'cuda'
# name, B, C0, C1, H(in), Cout, ups, coef+silu, res
SHAPES = [
    ('H32 128->128 gn res', 1024, 128, 0, 32, 128, 0, True, True),
    ('H16 256->256 gn res', 1024, 256, 0, 16, 256, 0, True, True),
    ('H32 128+128->128 gn', 1024, 128, 128, 32, 128, 0, True, False),
    ('H16 256+256->256 gn', 1024, 256, 256, 16, 256, 0, True, False),
    ('H32 256+128->128 gn', 1024, 256, 128, 32, 128, 0, True, False),
    ('H8  256->256 gn res', 1024, 256, 0, 8, 256, 0, True, True),
    ('H16->32 256->256 ups', 1024, 256, 0, 16, 256, 1, False, False),
    ('H16 128->256 gn', 1024, 128, 0, 16, 256, 0, True, False),
]


def run(name, B, C0, C1, H, Cout, ups, coef, res, gen, reps, zeros=False):
    Cin = C0 + C1
    g = torch.Generator(device=DEV).manual_seed(1)
    x0 = torch.randn(B, H, H, C0, device=DEV, generator=g)
    x1 = torch.randn(B, H, H, C1, device=DEV, generator=g) if C1 else None
    w = torch.randn(Cout, Cin, 3, 3, device=DEV, generator=g) * 0.05
    if zeros:
        x0.zero_()
        w.zero_()
        if x1 is not None:
            x1.zero_()
    bias = torch.randn(Cout, device=DEV, generator=g)
    Ho = H * 2 if ups else H
    out = torch.empty(B, Ho, Ho, Cout, device=DEV)
    a = _lib.ConvArgs()
    a.src0, a.C0 = x0.data_ptr(), C0
    if C1:
        a.src1, a.C1 = x1.data_ptr(), C1
    a.B, a.Hin, a.Win, a.Hout, a.Wout = B, H, H, Ho, Ho
    a.ksize, a.stride, a.upsample = 3, 1, ups
    a.weight, a.bias = w.data_ptr(), bias.data_ptr()
    keep = []
    if coef:
        cA = torch.rand(B, Cin, device=DEV, generator=g) + 0.5
        cB = torch.randn(B, Cin, device=DEV, generator=g) * 0.1
        keep += [cA, cB]
        a.coefA, a.coefB, a.act_silu = cA.data_ptr(), cB.data_ptr(), 1
    if res:
        r = torch.randn(B, Ho, Ho, Cout, device=DEV, generator=g)
        keep.append(r)
        a.res0, a.R0 = r.data_ptr(), Cout
    a.out, a.Cout = out.data_ptr(), Cout
    a.force_direct = {'auto': 0, 'f4': 8, 'f2': 0, 'igemm': 2}[gen]
    scratch = torch.empty(90 * w.numel() + 64 * 1024 * (1 + Cout // 32), device=DEV)
    a.scratch_floats = scratch.numel()
    st = _lib.stream_ptr()
    _lib.check(L.dlpm_conv2d_f32(C.byref(a), scratch.data_ptr(), st))
    torch.cuda.synchronize()
    _lib.check(L.dlpm_prof_enable(1))
    for _ in range(reps):
        _lib.check(L.dlpm_conv2d_f32(C.byref(a), scratch.data_ptr(), st))
    buf = C.create_string_buffer(1 << 16)
    _lib.check(L.dlpm_prof_report(buf, len(buf)))
    _lib.check(L.dlpm_prof_enable(0))
    ms, fl, cls = 0.0, 0.0, ''
    for line in buf.value.decode().strip().splitlines():
        nm, n, t, f, by = line.split()
        if nm.startswith('conv'):
            ms, fl, cls = float(t) / int(n), float(f) / int(n), nm
    dig = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]
    print('%-24s %-14s %8.4f ms  %7.1f alg TFLOP/s  mfma-util %5.3f  out %s' % (
        name, cls, ms, fl / ms / 1e9, (fl / ms / 1e9) * (0.25 if 'wino4' in cls else 16 / 36 if 'wino' in cls else 1) / 157.3, dig), flush=True)
    return ms


ap = argparse.ArgumentParser()
ap.add_argument('--gen', default='f4')
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--only', type=int, default=-1)
ap.add_argument('--zeros', action='store_true', help='all-zero operands: the same instruction stream at the clock the chip holds without data toggling (DVFS check)')
args = ap.parse_args()
tot = 0.0
for i, s in enumerate(SHAPES):
    if args.only >= 0 and i != args.only:
        continue
    tot += run(*s, gen=args.gen, reps=args.reps, zeros=args.zeros)
print('sum %.4f ms' % tot)
