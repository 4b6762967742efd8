"""Synthetic ("random-init") weights for benchmarking and parity tests.

The reference zero-initialises the last conv of every ResBlock, every attention `proj_out` and the
output conv (`zero_module`, dlpm/models/unet.py:156-158,215,435), so a freshly constructed UNet
outputs exactly 0 and exercises nothing.  `rerandomize_` re-draws those tensors N(0, std^2) and
(optionally) perturbs the GroupNorm affine parameters.  THIS DEVIATES FROM THE REFERENCE INIT ON
PURPOSE and is only used for synthetic benchmarks/tests; trained checkpoints load unchanged.
"""
import hashlib

import torch

ZERO_INIT_SUFFIXES = ('out_layers.3.weight', 'out_layers.3.bias', 'proj_out.weight', 'proj_out.bias',
                      'out.2.weight', 'out.2.bias')
NORM_MARKERS = ('in_layers.0.', 'out_layers.0.', '.norm.', 'out.0.')


def rerandomize_(module, seed, std=0.02, perturb_norm=True):
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith(ZERO_INIT_SUFFIXES):
                p.copy_(torch.randn(p.shape, generator=gen) * std)
            elif perturb_norm and any(m in name for m in NORM_MARKERS):
                if name.endswith('weight'):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gen))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=gen))
    if hasattr(module, 'invalidate'):
        module.invalidate()
    return module


def state_digest(module):
    h = hashlib.sha256()
    for k, v in module.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()
