import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import golden
from test_host_mirror import build_unet
for name in ['tiny','tiny2','mnist','cifar']:
    f=golden('f6_unet_'+name); net,_=build_unet(name)
    x,t=torch.from_numpy(f['x']).cuda(), torch.from_numpy(f['t']).cuda()
    y=net(x,t).cpu().numpy()
    print('%-6s max |hip - reference| = %.3e   (|y| max %.3f)'%(name, np.abs(y-f['y']).max(), np.abs(f['y']).max()))
