#!/usr/bin/env python3
"""Per-launch-shape timing of one reverse step (HIP events inside libdlpm_amd, DLPM_PROF_DETAIL=1).

    DLPM_PROF_DETAIL=1 python tools/prof_layers.py [--workload cifar10|mnist] [--batch B] [--steps N]
"""
import argparse
import ctypes as C
import os
import sys

os.environ.setdefault('DLPM_PROF_DETAIL', '1')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import dlpm_amd
from dlpm_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='cifar10')
ap.add_argument('--batch', type=int, default=1024)
ap.add_argument('--steps', type=int, default=3)
a = ap.parse_args()
p = dlpm_amd.load_config(a.workload)
torch.manual_seed(1234)
net = dlpm_amd.rerandomize_(dlpm_amd.init_model_by_parameter(p), 4321)
shape = [a.batch, p['data']['channels'], p['data']['image_size'], p['data']['image_size']]
m = dlpm_amd.GenerativeLevyProcess(1.7, 'cuda', 1000, rescale_timesteps=True, use_graph=False)
L, st = _lib.lib(), _lib.stream_ptr()
h = m._native_sampler(net, shape, 0, 0.0, 10.0, 50.0, 0)
_lib.check(L.dlpm_sampler_begin(h, st))
_lib.check(L.dlpm_sampler_steps(h, 2, st))
torch.cuda.synchronize()
_lib.check(L.dlpm_prof_enable(1))
_lib.check(L.dlpm_sampler_steps(h, a.steps, st))
buf = C.create_string_buffer(1 << 18)
_lib.check(L.dlpm_prof_report(buf, len(buf)))
rows = []
for line in buf.value.decode().strip().splitlines():
    name, n, ms, fl, by = line.split()
    rows.append((float(ms) / a.steps, name, int(n) // a.steps, float(fl) / a.steps, float(by) / a.steps))
tot = sum(r[0] for r in rows)
print('%-62s %5s %9s %8s %8s' % ('kernel class', 'n', 'ms/step', 'TFLOP/s', 'GB/s'))
for ms, name, n, fl, by in sorted(rows, reverse=True):
    print('%-62s %5d %9.3f %8.1f %8.0f' % (name, n, ms, fl / ms / 1e9 if ms else 0, by / ms / 1e6 if ms else 0))
print('total %.3f ms/step' % tot)
