#!/usr/bin/env python3
"""Does splitting a small-image batch over concurrent streams help?  One sampler at B against K samplers at B / K whose graph replays
are enqueued back to back on K streams (developer experiment for the MNIST-shaped config: launch grids of 16-128 workgroups on 256 CUs).

    python tools/bench_concurrent.py [--workload mnist_unet_b256_T1000] [--steps 200]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import dlpm_amd
from dlpm_amd import _lib
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='mnist_unet_b256_T1000')
ap.add_argument('--steps', type=int, default=200)
args = ap.parse_args()
cfg_name, B, T, alpha = bench.WORKLOADS[args.workload]
L = _lib.lib()
p = dlpm_amd.load_config(cfg_name)
torch.manual_seed(1234)
net = dlpm_amd.rerandomize_(dlpm_amd.init_model_by_parameter(p), 4321)
net.set_conv_policy('auto', B)
ev = p['eval']['dlpm']
C, S = p['data']['channels'], p['data']['image_size']


def make(k, K):
    meth = dlpm_amd.GenerativeLevyProcess(alpha, 'cuda', T, rescale_timesteps=True, seed=0, sample_offset=k * (B // K))
    h = meth._native_sampler(net, [B // K, C, S, S], 0, 0.0, ev['clamp_a'], ev['clamp_eps'], 0, k * (B // K))
    return meth, h


for K in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(K)]
    ms = [make(k, K) for k in range(K)]
    # NOTE one live native sampler per method object; K method objects here
    for (m, h), st in zip(ms, streams):
        _lib.check(L.dlpm_sampler_begin(h, st.cuda_stream))
        _lib.check(L.dlpm_sampler_steps(h, 5, st.cuda_stream))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for (m, h), st in zip(ms, streams):
            _lib.check(L.dlpm_sampler_steps(h, 1, st.cuda_stream))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%d sampler(s) of B = %d on %d stream(s): %.3f ms per step of the whole batch' % (K, B // K, K, dt / args.steps * 1e3), flush=True)
    for m, h in ms:
        m.close()
