#!/usr/bin/env python3
"""Micro-benchmark of the fused update kernel (dlpm_update_f32) alone: GB/s vs batch size.
Algorithmic bytes = 12 B/element (Philox) or 16 B/element (injected z)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from dlpm_amd import _lib

L = _lib.lib()
dev = 'cuda'
T, D = 1000, 3072
for B in (256, 1024, 4096, 16384):
    for inj in (False, True):
        x = torch.randn(B, D, device=dev)
        e = torch.randn(B, D, device=dev)
        z = torch.randn(B, D, device=dev) if inj else None
        tabs = [torch.rand(T, device=dev) + 0.5 for _ in range(3)]
        ce, cn, A = (torch.rand(T, B, device=dev) for _ in range(3))
        t = torch.tensor([500], dtype=torch.int32, device=dev)
        a = _lib.UpdateArgs()
        a.x_dev, a.eps_dev, a.z_dev, a.t_dev = x.data_ptr(), e.data_ptr(), z.data_ptr() if inj else None, t.data_ptr()
        a.g_dev, a.bg_dev, a.bs_dev = (v.data_ptr() for v in tabs)
        a.c_eps_dev, a.c_noise_dev, a.A_dev = ce.data_ptr(), cn.data_ptr(), A.data_ptr()
        a.B, a.D, a.T, a.flags, a.alpha, a.seed = B, D, T, 0, 1.7, 1
        st = _lib.stream_ptr()
        for _ in range(5):
            _lib.check(L.dlpm_update_f32(C.byref(a), st))
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            _lib.check(L.dlpm_update_f32(C.byref(a), st))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        by = B * D * (16 if inj else 12)
        print('B=%6d %-8s %8.2f us/launch  %7.1f GB/s (%.1f%% of 8 TB/s)' % (B, 'inject-z' if inj else 'philox', us, by / us / 1e3, by / us / 1e3 / 80))
