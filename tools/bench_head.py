#!/usr/bin/env python3
"""The head convolution at the CIFAR step's shape ([1024, 32, 32, 128] -> 3 channels, fused GroupNorm affine + SiLU) through its three
forms (developer tool): the one-pass kernel (head_fused.hip, force_direct bit 64: bf16 x 3; bits 64 | 128: fp32 MFMA), the GEMM + gather
pair (bit 32), the VALU kernel.

    python tools/bench_head.py [--reps N] [--only fused|fused32|pair|valu] [--B 1024] [--H 32] [--C 128]

ms per launch from the library's own HIP events (dlpm_prof_enable); the per-call weight relayout is a separate class."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from dlpm_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--only', default='')
ap.add_argument('--B', type=int, default=1024)
ap.add_argument('--H', type=int, default=32)
ap.add_argument('--C', type=int, default=128)
args = ap.parse_args()
L, DEV = _lib.lib(), 'cuda'
B, H, Cc, Cout = args.B, args.H, args.C, 3
x = torch.randn(B, H, H, Cc, device=DEV)
w = torch.randn(Cout, Cc, 3, 3, device=DEV) * 0.03
bias = torch.randn(Cout, device=DEV)
cA, cB = torch.rand(B, Cc, device=DEV) + 0.5, torch.randn(B, Cc, device=DEV) * 0.1
out = torch.empty(B, Cout, H, H, device=DEV)
scratch = torch.empty(64 * Cc + B * H * H * 32 + 9 * Cc * 8 + 4096, device=DEV)
st = _lib.stream_ptr()
for name, bits in (('fused', 64), ('fused32', 64 | 128), ('pair', 32), ('valu', 0)):
    if args.only and args.only != name:
        continue
    a = _lib.ConvArgs()
    a.src0, a.C0, a.B, a.Hin, a.Win, a.Hout, a.Wout = x.data_ptr(), Cc, B, H, H, H, H
    a.ksize, a.stride, a.upsample, a.weight, a.bias = 3, 1, 0, w.data_ptr(), bias.data_ptr()
    a.coefA, a.coefB, a.act_silu, a.out, a.Cout, a.out_nchw = cA.data_ptr(), cB.data_ptr(), 1, out.data_ptr(), Cout, 1
    a.force_direct, a.scratch_floats = bits, scratch.numel()
    _lib.check(L.dlpm_conv2d_f32(C.byref(a), scratch.data_ptr(), st))
    torch.cuda.synchronize()
    _lib.check(L.dlpm_prof_enable(1))
    for _ in range(args.reps):
        _lib.check(L.dlpm_conv2d_f32(C.byref(a), scratch.data_ptr(), st))
    if hasattr(L, 'dlpm_debug_phases') and name.startswith('fused'):     # DLPM_PHASE_TIMING build: where a workgroup's life goes
        ph = (C.c_ulonglong * 32)()
        L.dlpm_debug_phases.argtypes = [C.POINTER(C.c_ulonglong * 32)]
        _lib.check(L.dlpm_debug_phases(C.byref(ph)))
        n = max(ph[3], 1)
        print('%-6s cycles per workgroup (wave 0): prologue %.0f  tile loop %.0f  barrier + gather %.0f;  barrier wait by wave: %s;  clock %.0f MHz'
              % (name, ph[0] / n, ph[1] / n, ph[2] / n, ' '.join('%.0f' % (ph[16 + w] / n) for w in range(8)), 100.0 * ph[12] / max(ph[13], 1)))
    buf = C.create_string_buffer(1 << 16)
    _lib.check(L.dlpm_prof_report(buf, len(buf)))
    _lib.check(L.dlpm_prof_enable(0))
    tot = 0.0
    for line in buf.value.decode().strip().splitlines():
        k, n, ms, fl, by = line.split()
        if 'relayout' in k:
            continue
        print('%-6s %-22s %3d launches  %.4f ms/launch' % (name, k, int(n), float(ms) / int(n)))
        tot += float(ms) / args.reps
    inp = 4.0 * B * H * H * Cc
    print('%-6s total %.4f ms per head  -> %.2f TB/s on the input bytes (%.0f MB)' % (name, tot, inp / tot / 1e9, inp / 1e6))
