#!/usr/bin/env python3
"""k_attnblock16 alone (developer tool): launch time through dlpm_attnblock_small_f32 at H = W = 16 and -- in a DLPM_PHASE_TIMING build
(DLPM_LIB=...) -- wave 0's cycles per section.   python tools/bench_attn16.py [--batch 256]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from dlpm_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=256); ap.add_argument('--reps', type=int, default=100)
args = ap.parse_args()
L = _lib.lib(); HAVE = hasattr(L, 'dlpm_debug_phases')
if HAVE:
    L.dlpm_debug_phases.restype = C.c_int; L.dlpm_debug_phases.argtypes = [C.POINTER(C.c_ulonglong * 32)]
B, DEV = args.batch, 'cuda'
g = torch.Generator(device=DEV).manual_seed(1)
P = lambda *s: torch.randn(*s, device=DEV, generator=g)
x = P(B, 16, 16, 64)
k = dict(gw=1 + 0.1 * P(64), gb=0.1 * P(64), qw=P(192, 64, 1) / 8, qb=0.1 * P(192), pw=P(64, 64, 1) / 8, pb=0.1 * P(64))
a = _lib.AttnBlockArgs()
a.x, a.C, a.heads, a.B, a.H, a.W = x.data_ptr(), 64, 4, B, 16, 16
a.gn_w, a.gn_b, a.qkv_w, a.qkv_b, a.proj_w, a.proj_b = (k[n].data_ptr() for n in ('gw', 'gb', 'qw', 'qb', 'pw', 'pb'))
out = torch.empty(B, 16, 16, 64, device=DEV); stats = torch.empty(B, 64, 2, device=DEV)
a.out, a.stats_out = out.data_ptr(), stats.data_ptr()
n = 192 * 64 + 64 * 64; scratch = torch.empty(n, device=DEV); st = _lib.stream_ptr()
_lib.check(L.dlpm_attnblock_small_f32(C.byref(a), scratch.data_ptr(), n, st)); torch.cuda.synchronize()
ph = (C.c_ulonglong * 32)()
if HAVE: _lib.check(L.dlpm_debug_phases(C.byref(ph)))
_lib.check(L.dlpm_prof_enable(1))
for _ in range(args.reps): _lib.check(L.dlpm_attnblock_small_f32(C.byref(a), scratch.data_ptr(), n, st))
buf = C.create_string_buffer(1 << 16); _lib.check(L.dlpm_prof_report(buf, len(buf))); _lib.check(L.dlpm_prof_enable(0))
for line in buf.value.decode().strip().splitlines():
    nm, n_, t_, f_, by_ = line.split()
    if nm.startswith('attnblock16'): print('attnblock16 B %d: %.2f us per launch' % (B, 1e3 * float(t_) / int(n_)))
if HAVE:
    _lib.check(L.dlpm_debug_phases(C.byref(ph))); nw = max(ph[5], 1)
    print('   wave 0 cycles per workgroup: load + GroupNorm %.0f | q k v (4 heads) %.0f | attention %.0f | proj %.0f | epilogue %.0f' % tuple(ph[i] / nw for i in range(5)))
