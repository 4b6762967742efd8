#!/usr/bin/env python3
"""BASELINE.json configs[0]: 2-D toy data, MLP score net, T=100, alpha=1.7, batch 512 -- full sample()
calls through the reference-shaped entry point, GPU (libdlpm_amd) vs the CPU oracle on this host."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import dlpm_amd
from oracle import nets, sampler as osampler

p = dlpm_amd.load_config('2d_data')
torch.manual_seed(1)
mlp = dlpm_amd.MLPModel(p)
B, T, alpha = 512, 100, 1.7
for graph in (True, False):
    m = dlpm_amd.GenerativeLevyProcess(alpha, 'cuda', T, rescale_timesteps=True, seed=0, use_graph=graph)
    ts = []
    for i in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = m.sample({'default': mlp}, [B, 1, 2], T)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print('GPU  sample() B=%d T=%d graph=%s: first %.2f ms, median of rest %.3f ms -> %.0f samples/s' % (
        B, T, graph, ts[0] * 1e3, np.median(ts[1:]) * 1e3, B / np.median(ts[1:])))
    m.close()
for B2 in (4096, 65536):
    m = dlpm_amd.GenerativeLevyProcess(alpha, 'cuda', T, rescale_timesteps=True, seed=0)
    ts = []
    for i in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = m.sample({'default': mlp}, [B2, 1, 2], T)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print('GPU  sample() B=%d: median %.3f ms -> %.0f samples/s' % (B2, np.median(ts[1:]) * 1e3, B2 / np.median(ts[1:])))
    m.close()
sd = {k: v.detach() for k, v in mlp.state_dict().items()}
ts = []
with torch.inference_mode():
    for i in range(4):
        t0 = time.perf_counter()
        osampler.sample(lambda x, t: nets.mlp_forward(sd, x, t), [B, 1, 2], T, alpha, osampler.Streams(i, i))
        ts.append(time.perf_counter() - t0)
print('CPU  oracle sample() B=%d T=%d (%d torch threads): median %.1f ms -> %.0f samples/s' % (
    B, T, torch.get_num_threads(), np.median(ts[1:]) * 1e3, B / np.median(ts[1:])))
