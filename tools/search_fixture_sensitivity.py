#!/usr/bin/env python3
"""Search the head scale of the bounded MNIST / CelebA-64 reference trajectories for an informative fixture:
sensitivity >= 0.4 (a network wrong by 2.5e-4 relative breaks the 1e-4 pixel contract) with >= 70 % of the final pixels strictly
inside (-1, 1).  Imports tools/make_fixtures.py (and through it the reference); writes nothing under tests/golden.

    python tools/search_fixture_sensitivity.py mnist 1 2 4 8        # head scales to try
    python tools/search_fixture_sensitivity.py celeba64 2 4
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_fixtures as M   # noqa: E402  (imports the reference)
import numpy as np          # noqa: E402
import torch                # noqa: E402

which = sys.argv[1]
scales = [float(v) for v in sys.argv[2:]]
torch.set_num_threads(int(os.environ.get('SEARCH_THREADS', '3')))
CASES = {'mnist': (1000, 2, 'mnist', 1, 32, 1.7, 20, 200), 'celeba64': (1000, 1, 'cifar', 3, 64, 1.8, 10, 50)}
ARCHS = {'cifar': (128, [1, 2, 2, 2], [4, 8, 16], 2), 'mnist': (32, [1, 2, 2, 2], [2, 4], 2)}
T, B, arch, chans, size, alpha, ca, ce = CASES[which]
T = int(os.environ.get('SEARCH_T', T))


class Perturbed(torch.nn.Module):
    def __init__(self, net, rel, seed):
        super().__init__()
        self.net, self.rel, self.g = net, rel, torch.Generator().manual_seed(seed)

    def forward(self, x, t, **kw):
        y = self.net(x, t, **kw)
        return y * (1 + self.rel * torch.randn(y.shape, generator=self.g))


def post(x):
    return (x.clamp(-1, 1) + 1) / 2


with torch.no_grad():
    for hs in scales:
        torch.manual_seed(1234)
        mc, mult, attn, res = ARCHS[arch]
        net = M.make_unet(chans, mc, mult, attn, 4, res).eval()
        M.rerandomize(net, 4321)
        net.out[2].weight.mul_(hs)
        net.out[2].bias.mul_(hs)
        shape = [B, chans, size, size]

        def run(model):
            np.random.seed(0)
            torch.manual_seed(0)
            meth = M.GenerativeLevyProcess(alpha=alpha, device='cpu', reverse_steps=T, rescale_timesteps=True)
            return meth.sample({'default': model}, shape, T, clamp_a=ca, clamp_eps=ce, clip_denoised=True, get_sample_history=True)

        x, hist = run(net)
        inside = float((x.abs() < 1).float().mean())
        xp, _ = run(Perturbed(net, 1e-4, 77))
        sens = float((post(xp) - post(x)).abs().max()) / 1e-4
        print('%s head_scale %g: inside %.1f %%, |x|max %.3g, max|state| %.3g, sensitivity %.3g (state %.3g)'
              % (which, hs, 100 * inside, float(x.abs().max()), float(hist.abs().max()), sens, float((xp - x).abs().max()) / 1e-4), flush=True)
