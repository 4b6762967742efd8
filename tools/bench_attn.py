#!/usr/bin/env python3
"""Attention kernel alone at the launch shapes of the BASELINE configs (developer tool): ms per launch, TFLOP/s, GB/s."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from dlpm_amd import _lib

L = _lib.lib()
SHAPES = [('cifar H8', 1024, 64, 256, 4), ('cifar H4', 1024, 16, 256, 4), ('celeba64 H16', 256, 256, 256, 4),
          ('celeba64 H8', 256, 64, 256, 4), ('mnist H16', 256, 256, 64, 4), ('mnist H8', 256, 64, 64, 4)]
for name, B, T, C, heads in SHAPES:
    qkv = torch.randn(B, T, 3 * C, device='cuda')
    out = torch.empty(B, T, C, device='cuda')
    st = _lib.stream_ptr()
    for _ in range(3):
        _lib.check(L.dlpm_attention_f32(qkv.data_ptr(), out.data_ptr(), B, T, C, heads, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        _lib.check(L.dlpm_attention_f32(qkv.data_ptr(), out.data_ptr(), B, T, C, heads, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 4.0 * B * T * T * C
    by = 16.0 * B * T * C
    print('%-14s B=%4d T=%3d C=%3d  %8.4f ms  %6.1f TFLOP/s  %6.0f GB/s' % (name, B, T, C, ms, fl / ms / 1e9, by / ms / 1e6))
