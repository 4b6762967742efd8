#!/usr/bin/env python3
"""Non-isotropic (--non_iso) and history variants of the update kernel, and the per-element table build, on one
MI355X.  Algorithmic bytes per element: 12 (isotropic, Philox) + 8 for the two per-element coefficient reads
(DLPM_UPD_ELEMENTWISE) + 4 when a history row is written."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import dlpm_amd
from dlpm_amd import _lib

L = _lib.lib()
dev = 'cuda'
D = 3072
st = _lib.stream_ptr()


def bench(fn, n=100):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B in (1024, 4096, 16384):
    T = 8
    x = torch.randn(B, D, device=dev)
    e = torch.randn(B, D, device=dev)
    tabs = [torch.rand(T, device=dev) + 0.5 for _ in range(3)]
    t = torch.tensor([4], dtype=torch.int32, device=dev)
    hist = torch.empty(T, B, D, device=dev)
    cell = torch.tensor([hist.data_ptr()], dtype=torch.int64, device=dev)
    for elem in (False, True):
        n = B * D if elem else B
        ce, cn, A = (torch.rand(T, n, device=dev) for _ in range(3))
        for with_hist in (False, True):
            a = _lib.UpdateArgs()
            a.x_dev, a.eps_dev, a.z_dev, a.t_dev = x.data_ptr(), e.data_ptr(), None, t.data_ptr()
            a.g_dev, a.bg_dev, a.bs_dev = (v.data_ptr() for v in tabs)
            a.c_eps_dev, a.c_noise_dev, a.A_dev = ce.data_ptr(), cn.data_ptr(), A.data_ptr()
            a.B, a.D, a.T, a.flags, a.alpha, a.seed = B, D, T, (_lib.UPD_ELEMENTWISE if elem else 0), 1.7, 1
            a.hist_pp = cell.data_ptr() if with_hist else None
            us = bench(lambda: _lib.check(L.dlpm_update_f32(C.byref(a), st)))
            by = B * D * (12 + (8 if elem else 0) + (4 if with_hist else 0))
            print('update B=%6d %-11s %-8s %8.2f us/launch  %7.1f GB/s (%.1f%% of 8 TB/s)' % (
                B, 'per-element' if elem else 'per-sample', 'history' if with_hist else '', us, by / us / 1e3, by / us / 1e3 / 80))

# full-size prologue: A[T,B,D] draws (fp64 CMS per element) + in-place coefficient scan, CIFAR B=1024 T=1000
B, T = 1024, 1000
A = torch.empty(T, B * D, device=dev)
cn = torch.empty(T, B * D, device=dev)
g, s, bs = (torch.rand(T, device=dev) * 0.5 + 0.5 for _ in range(3))
torch.cuda.synchronize()
t0 = time.perf_counter()
_lib.check(L.dlpm_skewed_levy_elem_philox_f32(A.data_ptr(), T, B, D, 1.7, 10.0, 0, 0, st))
torch.cuda.synchronize()
t1 = time.perf_counter()
_lib.check(L.dlpm_coeff_tables_f32(A.data_ptr(), g.data_ptr(), s.data_ptr(), bs.data_ptr(), T, B * D, A.data_ptr(), cn.data_ptr(), None, st))
torch.cuda.synchronize()
t2 = time.perf_counter()
n = T * B * D
print('prologue [T=%d,B=%d,D=%d]: %.2f G skewed-Levy draws in %.3f s (%.1f G draws/s); coefficient scan %.3f s '
      '(%.0f GB/s over read A + write c_eps, c_noise); tables hold %.1f GB' % (
          T, B, D, n / 1e9, t1 - t0, n / (t1 - t0) / 1e9, t2 - t1, 12.0 * n / (t2 - t1) / 1e9, 8.0 * n / 1e9))
