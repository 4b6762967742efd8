#!/usr/bin/env python3
"""Single-launch timing of the 1x1 convolutions at the CIFAR step's launch shapes, fp32 MFMA against the bf16-split GEMM
(conv_split.hip), with each result's error against float64 on a sample of rows (developer tool).

    python tools/bench_gemm.py [--reps N] [--only I]

ms per launch from the library's own HIP events (dlpm_prof_enable); the per-call weight relayout of the test entry
point is a separate class and not counted."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from dlpm_amd import _lib

L = _lib.lib()
DEV = 'cuda'
# name, launches per step, B, C0, C1, H (input), Cout, coef (GroupNorm affine, no SiLU: the qkv input), res[, ksize, stride]
SHAPES = [
    ('3x3 s2 H32->16 128->128', 1, 1024, 128, 0, 32, 128, False, False, 3, 2),
    ('3x3 s2 H16->8 256->256', 1, 1024, 256, 0, 16, 256, False, False, 3, 2),
    ('3x3 s2 H8->4 256->256', 1, 1024, 256, 0, 8, 256, False, False, 3, 2),
    ('H8  256->768 gn (qkv)', 5, 1024, 256, 0, 8, 768, True, False),
    ('H32 128+128->128 skip', 2, 1024, 128, 128, 32, 128, False, False),
    ('H16 256+256->256 skip', 2, 1024, 256, 256, 16, 256, False, False),
    ('H32 256+128->128 skip', 1, 1024, 256, 128, 32, 128, False, False),
    ('H4  256->768 gn (qkv)', 6, 1024, 256, 0, 4, 768, True, False),
    ('H8  256->256 res (proj)', 5, 1024, 256, 0, 8, 256, False, True),
    ('H8  256+256->256 skip', 3, 1024, 256, 256, 8, 256, False, False),
    ('H16 256+128->256 skip', 1, 1024, 256, 128, 16, 256, False, False),
    ('H16 128->256 skip', 1, 1024, 128, 0, 16, 256, False, False),
    ('H4  256->256 res (proj)', 6, 1024, 256, 0, 4, 256, False, True),
    ('H4  256+256->256 skip', 3, 1024, 256, 256, 4, 256, False, False),
]


def run(name, n, B, C0, C1, H, Cout, coef, res, ks=1, stride=1, reps=10):
    Cin = C0 + C1
    Ho = (H - 1) // stride + 1
    g = torch.Generator(device=DEV).manual_seed(1)
    x0 = torch.randn(B, H, H, C0, device=DEV, generator=g)
    x1 = torch.randn(B, H, H, C1, device=DEV, generator=g) if C1 else None
    w = torch.randn(Cout, Cin, ks, ks, device=DEV, generator=g) / (Cin * ks * ks) ** 0.5
    bias = torch.randn(Cout, device=DEV, generator=g)
    if os.environ.get('BENCH_ZEROS') == '1':   # the same instruction stream without data toggling (DVFS check)
        x0.zero_(); w.zero_()
        if x1 is not None:
            x1.zero_()
    a = _lib.ConvArgs()
    a.src0, a.C0 = x0.data_ptr(), C0
    if C1:
        a.src1, a.C1 = x1.data_ptr(), C1
    a.B, a.Hin, a.Win, a.Hout, a.Wout = B, H, H, Ho, Ho
    a.ksize, a.stride, a.upsample = ks, stride, 0
    a.weight, a.bias = w.data_ptr(), bias.data_ptr()
    keep = []
    cA = cB = r = None
    if coef:
        cA = torch.rand(B, Cin, device=DEV, generator=g) + 0.5
        cB = torch.randn(B, Cin, device=DEV, generator=g) * 0.1
        a.coefA, a.coefB = cA.data_ptr(), cB.data_ptr()
    if res:
        r = torch.randn(B, Ho, Ho, Cout, device=DEV, generator=g)
        a.res0, a.R0 = r.data_ptr(), Cout
    a.Cout = Cout
    scratch = torch.empty(14 * w.numel() + 64 * 1024 * (1 + Cout // 32), device=DEV)
    a.scratch_floats = scratch.numel()
    st = _lib.stream_ptr()
    # float64 reference on the first 4096 rows
    rows = 4096
    xin = x0.reshape(-1, C0)[:rows] if not C1 else torch.cat([x0.reshape(-1, C0)[:rows], x1.reshape(-1, C1)[:rows]], 1)
    if coef:   # fmaf(x, A, B) in fp32, as both kernels stage it: the float64 expression rounded once
        bidx = torch.arange(rows, device=DEV) // (H * H)
        xin = (xin.double() * cA[bidx].double() + cB[bidx].double()).float()
    if ks == 1:
        want = xin.double() @ w.reshape(Cout, Cin).double().t() + bias.double()
        if res:
            want = want + r.reshape(-1, Cout)[:rows].double()
    else:   # the first samples through float64 conv2d
        nb = max(1, rows // (Ho * Ho))
        rows = nb * Ho * Ho
        want = torch.nn.functional.conv2d(x0[:nb].permute(0, 3, 1, 2).double(), w.double(), bias.double(), stride=stride, padding=1)
        want = want.permute(0, 2, 3, 1).reshape(rows, Cout)
    out = {}
    for mode, bit in (('f32', 2 if ks == 3 else 0), ('bf16x3', 16)):
        o = torch.empty(B, Ho, Ho, Cout, device=DEV)
        a.out = o.data_ptr()
        a.force_direct = bit
        _lib.check(L.dlpm_conv2d_f32(C.byref(a), scratch.data_ptr(), st))
        torch.cuda.synchronize()
        _lib.check(L.dlpm_prof_enable(1))
        for _ in range(reps):
            _lib.check(L.dlpm_conv2d_f32(C.byref(a), scratch.data_ptr(), st))
        buf = C.create_string_buffer(1 << 16)
        _lib.check(L.dlpm_prof_report(buf, len(buf)))
        _lib.check(L.dlpm_prof_enable(0))
        ms, fl, by, cls = 0.0, 0.0, 0.0, ''
        for line in buf.value.decode().strip().splitlines():
            nm, k, t, f, bb = line.split()
            if nm.startswith('conv'):
                ms, fl, by, cls = float(t) / int(k), float(f) / int(k), float(bb) / int(k), nm
        err = (o.reshape(-1, Cout)[:rows].double() - want).abs().max().item()
        out[mode] = (ms, err, o)
        print('%-26s %-16s %8.4f ms  %7.1f TFLOP/s  %6.0f GB/s  max err vs f64 %.2e' % (name, cls, ms, fl / ms / 1e9, by / ms / 1e6, err),
              flush=True)
    d = (out['f32'][2] - out['bf16x3'][2]).abs().max().item()
    print('%-26s f32 vs bf16x3 max |diff| %.2e   speedup %.2fx' % ('', d, out['f32'][0] / out['bf16x3'][0]), flush=True)
    return n * out['f32'][0], n * out['bf16x3'][0]


ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--only', type=int, default=-1)
args = ap.parse_args()
t32 = t16 = 0.0
for i, s in enumerate(SHAPES):
    if args.only >= 0 and i != args.only:
        continue
    a32, a16 = run(*s, reps=args.reps)
    t32 += a32
    t16 += a16
print('per step (launch counts of the CIFAR net): f32 %.3f ms   bf16x3 %.3f ms' % (t32, t16))
