#!/usr/bin/env python3
"""Where a convolution workgroup spends its cycles (developer tool).

Needs a library built with the phase counters compiled in:

    DLPM_BUILD_DEFS=-DDLPM_PHASE_TIMING python -m dlpm_amd.build
    python tools/phase_conv.py

Thread 0 of every workgroup adds clock64() deltas per phase (prologue / main loop / epilogue) into a device buffer;
this prints the mean per workgroup for a list of the CIFAR net's launch shapes next to the launch time.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from dlpm_amd import _lib

L = _lib.lib()
HAVE_PHASES = hasattr(L, 'dlpm_debug_phases')     # the product build has none: launch times only (PHASE_B sweeps of the product kernel)
if HAVE_PHASES:
    L.dlpm_debug_phases.restype = C.c_int
    L.dlpm_debug_phases.argtypes = [C.POINTER(C.c_ulonglong * 32)]
DEV = 'cuda'

# name, B, C0, C1, H, Cout, ks, coef+silu, res
SHAPES = [s for s in [
    ('1x1 skip H32 128+128->128', 1024, 128, 128, 32, 128, 1, False, False),
    ('1x1 skip H32 256+128->128', 1024, 256, 128, 32, 128, 1, False, False),
    ('1x1 skip H16 256+256->256', 1024, 256, 256, 16, 256, 1, False, False),
    ('1x1 qkv  H8  256->768', 1024, 256, 0, 8, 768, 1, True, False),
    ('1x1 proj H8  256->256 +res', 1024, 256, 0, 8, 256, 1, False, True),
    ('3x3 wino H32 128->128', 1024, 128, 0, 32, 128, 3, True, True),
    ('3x3 wino H16 256->256', 1024, 256, 0, 16, 256, 3, True, True),
    ('3x3 wino H32 128+128->128', 1024, 128, 128, 32, 128, 3, True, False),
    # the MNIST net's launch shapes (B = 256): PHASE_ONLY=mnist
    ('3x3 mnist H16 64->64', 256, 64, 0, 16, 64, 3, True, True),
    ('3x3 mnist H8 64->64', 256, 64, 0, 8, 64, 3, True, True),
    ('3x3 mnist H4 64->64', 256, 64, 0, 4, 64, 3, True, True),
    ('3x3 mnist H32 32->32', 256, 32, 0, 32, 32, 3, True, True),
    ('3x3 cifar8w H32 128->128', 1024, 128, 0, 32, 128, 3, True, True),     # the 8-wave kernel's worst layer: PHASE_ONLY=cifar8w PHASE_B=64 (256 workgroups)
] if os.environ.get('PHASE_ONLY', '') in s[0]]


def run(name, B, C0, C1, H, Cout, ks, coef, res, reps=int(os.environ.get('PHASE_REPS', '5'))):   # PHASE_REPS=4000: long enough for tools/power_probe.py
    B = int(os.environ.get('PHASE_B', B))    # e.g. 8: a launch of 32 workgroups, every one alone on its CU and (nearly) alone on the HBM
    Cin = C0 + C1
    x0 = torch.randn(B, H, H, C0, device=DEV)
    x1 = torch.randn(B, H, H, C1, device=DEV) if C1 else None
    w = torch.randn(Cout, Cin, ks, ks, device=DEV) * 0.05
    if os.environ.get('PHASE_ZEROS') == '1':     # all-zero operands: the same instruction stream without data toggling (power vs structure)
        x0.zero_()
        w.zero_()
        if x1 is not None:
            x1.zero_()
    bias = torch.randn(Cout, device=DEV)
    out = torch.empty(B, H, H, Cout, device=DEV)
    a = _lib.ConvArgs()
    a.src0, a.C0 = x0.data_ptr(), C0
    if C1:
        a.src1, a.C1 = x1.data_ptr(), C1
    a.B, a.Hin, a.Win, a.Hout, a.Wout = B, H, H, H, H
    a.ksize, a.stride, a.upsample = ks, 1, 0
    a.weight, a.bias = w.data_ptr(), bias.data_ptr()
    keep = []
    if coef:
        cA, cB = torch.rand(B, Cin, device=DEV) + 0.5, torch.randn(B, Cin, device=DEV) * 0.1
        keep += [cA, cB]
        a.coefA, a.coefB, a.act_silu = cA.data_ptr(), cB.data_ptr(), 1
    if res:
        r = torch.randn(B, H, H, Cout, device=DEV)
        keep.append(r)
        a.res0, a.R0 = r.data_ptr(), Cout
    a.out, a.Cout = out.data_ptr(), Cout
    a.force_direct = int(os.environ.get('PHASE_FORCE', '0'))
    scratch = torch.empty(9 * w.numel() + 16 * 1024 * (1 + Cout // 32), device=DEV)
    a.scratch_floats = scratch.numel()
    st = _lib.stream_ptr()
    ph = (C.c_ulonglong * 32)()
    _lib.check(L.dlpm_conv2d_f32(C.byref(a), scratch.data_ptr(), st))
    torch.cuda.synchronize()
    # the convolution kernel's own launch time (HIP events around it on the launch stream: the weight relayout launches of
    # dlpm_conv2d_f32 fall into classes of their own), min-free mean over `preps` launches
    preps = int(os.environ.get('PHASE_PROF_REPS', '200'))
    _lib.check(L.dlpm_prof_enable(1))
    for _ in range(preps):
        _lib.check(L.dlpm_conv2d_f32(C.byref(a), scratch.data_ptr(), st))
    pbuf = C.create_string_buffer(1 << 16)
    _lib.check(L.dlpm_prof_report(pbuf, len(pbuf)))
    _lib.check(L.dlpm_prof_enable(0))
    kern_us = None
    for line in pbuf.value.decode().strip().splitlines():
        nm, n_, t_, f_, by_ = line.split()
        if nm.startswith('conv'):
            kern_us = 1e3 * float(t_) / int(n_)
    if HAVE_PHASES:
        _lib.check(L.dlpm_debug_phases(C.byref(ph)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(L.dlpm_conv2d_f32(C.byref(a), scratch.data_ptr(), st))
    e1.record()
    torch.cuda.synchronize()
    if not HAVE_PHASES:
        print('%-30s B %5d  kernel %8.2f us/launch (events around the launch, %d launches)  %8.3f ms/call(+relayout)   [product build: no phase counters]'
              % (name, B, kern_us or -1, preps, e0.elapsed_time(e1) / reps))
        return
    _lib.check(L.dlpm_debug_phases(C.byref(ph)))
    base = 8 if ph[11] else 0
    n = ph[base + 3]
    pro, main, epi = (ph[base + i] / max(n, 1) for i in range(3))
    tot = pro + main + epi
    if tot == 0:
        print('%-30s %8.3f ms/call(+relayout)  (this shape\'s kernel carries no phase counters)' % (name, e0.elapsed_time(e1) / reps))
        return
    print('%-30s %8.3f ms/call(+relayout)  kernel %7.2f us  wgs/launch %6d  cycles/wg: prologue %7.0f (%4.1f%%)  loop %7.0f (%4.1f%%)  epilogue %7.0f (%4.1f%%)'
          % (name, e0.elapsed_time(e1) / reps, kern_us or -1, n // reps, pro, 100 * pro / tot, main, 100 * main / tot, epi, 100 * epi / tot))
    if any(ph[4 + i] for i in range(4)):
        print('    loop, per workgroup: stage+wait %7.0f  barrier %7.0f  MFMAs %7.0f  barrier %7.0f' % tuple(ph[4 + i] / max(n, 1) for i in range(4)))
    if ph[13]:
        print('    in-kernel clock: %.0f MHz (shader cycles / 100-MHz ticks over each workgroup\'s life)' % (100.0 * ph[12] / ph[13]))
    if any(ph[16 + w] for w in range(8)):
        print('    barrier wait cycles per workgroup, by wave: ' + ' '.join('%7.0f' % (ph[16 + w] / max(n, 1)) for w in range(8)))


for s in SHAPES:
    run(*s)
