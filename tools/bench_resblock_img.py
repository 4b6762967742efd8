#!/usr/bin/env python3
"""The whole-image ResBlock kernel alone (developer tool): launch time of k_resblock_wino4_img at the MNIST config's shapes through
dlpm_resblock_img_f32, and -- in a DLPM_PHASE_TIMING DLPM_PHASE_DEFER build (DLPM_LIB=...) -- cycles per workgroup up to the end of
pass 1 / of pass 2 / of the epilogue.

    python tools/bench_resblock_img.py [--batch 256] [--cin 32]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from dlpm_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--cin', type=int, default=32)
ap.add_argument('--reps', type=int, default=100)
ap.add_argument('--h16', action='store_true', help='the 16x16 / 64-channel shape (k_resblock_wino4_img16); --cin 64 | 128 | 96')
args = ap.parse_args()
L = _lib.lib()
HAVE = hasattr(L, 'dlpm_debug_phases')
if HAVE:
    L.dlpm_debug_phases.restype = C.c_int
    L.dlpm_debug_phases.argtypes = [C.POINTER(C.c_ulonglong * 32)]
B, cin, DEV = args.batch, args.cin, 'cuda'
HS, CO = (16, 64) if args.h16 else (32, 32)
c0 = ({64: 64, 128: 64, 96: 64} if args.h16 else {32: 32, 64: 32, 96: 64})[cin]
g = torch.Generator(device=DEV).manual_seed(1)
x0 = torch.randn(B, HS, HS, c0, device=DEV, generator=g)
x1 = torch.randn(B, HS, HS, cin - c0, device=DEV, generator=g) if cin > c0 else None
P = lambda *s: torch.randn(*s, device=DEV, generator=g)
a = _lib.ResBlockArgs()
a.x0, a.x1, a.C0, a.C1 = x0.data_ptr(), x1.data_ptr() if x1 is not None else None, c0, cin - c0
a.B, a.H, a.W = B, HS, HS
keep = dict(g1w=1 + 0.1 * P(cin), g1b=0.1 * P(cin), w1=P(CO, cin, 3, 3) / (9 * cin) ** 0.5, b1=0.1 * P(CO), ss=0.3 * P(B, 2 * CO), g2w=1 + 0.1 * P(CO),
            g2b=0.1 * P(CO), w2=P(CO, CO, 3, 3) / (9 * CO) ** 0.5, b2=0.1 * P(CO), sw=P(CO, cin, 1, 1) / cin ** 0.5, sb=0.1 * P(CO))
a.gn1_w, a.gn1_b, a.conv1_w, a.conv1_b = (keep[k].data_ptr() for k in ('g1w', 'g1b', 'w1', 'b1'))
a.ss, a.ss_stride = keep['ss'].data_ptr(), 2 * CO
a.gn2_w, a.gn2_b, a.conv2_w, a.conv2_b = (keep[k].data_ptr() for k in ('g2w', 'g2b', 'w2', 'b2'))
if cin != CO:
    a.skip_w, a.skip_b = keep['sw'].data_ptr(), keep['sb'].data_ptr()
out = torch.empty(B, HS, HS, CO, device=DEV)
stats = torch.empty(B, 4, 64, 2, device=DEV)
a.out, a.stats_out = out.data_ptr(), stats.data_ptr()
n = L.dlpm_resblock_img_scratch_floats(B, cin)
scratch = torch.empty(n, device=DEV)
st = _lib.stream_ptr()
_lib.check(L.dlpm_resblock_img_f32(C.byref(a), scratch.data_ptr(), n, st))
torch.cuda.synchronize()
ph = (C.c_ulonglong * 32)()
if HAVE:
    _lib.check(L.dlpm_debug_phases(C.byref(ph)))
_lib.check(L.dlpm_prof_enable(1))
for _ in range(args.reps):
    _lib.check(L.dlpm_resblock_img_f32(C.byref(a), scratch.data_ptr(), n, st))
buf = C.create_string_buffer(1 << 16)
_lib.check(L.dlpm_prof_report(buf, len(buf)))
_lib.check(L.dlpm_prof_enable(0))
for line in buf.value.decode().strip().splitlines():
    nm, n_, t_, f_, by_ = line.split()
    if nm.startswith('resblock_img'):
        print('resblock_img Cin %d B %d: %.2f us per launch (%d launches)' % (cin, B, 1e3 * float(t_) / int(n_), int(n_)))
if HAVE:
    _lib.check(L.dlpm_debug_phases(C.byref(ph)))
    nw = max(ph[11], 1)
    print('   cycles per workgroup: GroupNorm-1 + pass 1 %.0f | between + pass 2 %.0f | epilogue %.0f   (%d workgroups)' % (ph[8] / nw, ph[9] / nw, ph[10] / nw, nw))
