#!/usr/bin/env python3
"""UNet forward (HIP, through the C ABI) against the reference's own outputs (tests/golden/f6_unet_*.npz).

    python tools/err_report.py                      # default kernels: Winograd F(4x4,3x3) where it applies
    DLPM_WINO_F4=0 python tools/err_report.py       # F(2x2,3x3) everywhere
    DLPM_WINO_F4=0 DLPM_NO_WINO=1 python tools/err_report.py   # implicit GEMM only
"""
import os, sys, numpy as np, torch
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
from conftest import golden
from test_host_mirror import build_unet
for name in ['tiny','tiny2','mnist','cifar']:
    f=golden('f6_unet_'+name); net,_=build_unet(name)
    x,t=torch.from_numpy(f['x']).cuda(), torch.from_numpy(f['t']).cuda()
    y=net(x,t).cpu().numpy()
    print('%-6s max |hip - reference| = %.3e   (|y| max %.3f)'%(name, np.abs(y-f['y']).max(), np.abs(f['y']).max()))
