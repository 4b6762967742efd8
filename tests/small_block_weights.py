"""Weights of the F13 fixtures (tests/golden/f13_small_blocks.npz): tools/make_fixtures.py `seeded_state`, repeated verbatim over
the SAME key order as the reference modules' state_dict -- every tensor from its own seeded generator, so the fixture carries a
digest instead of the weights."""
import hashlib

import numpy as np
import torch

RES_KEYS = ['in_layers.0.weight', 'in_layers.0.bias', 'in_layers.2.weight', 'in_layers.2.bias', 'emb_layers.1.weight',
            'emb_layers.1.bias', 'out_layers.0.weight', 'out_layers.0.bias', 'out_layers.3.weight', 'out_layers.3.bias',
            'skip_connection.weight', 'skip_connection.bias']
ATTN_KEYS = ['norm.weight', 'norm.bias', 'qkv.weight', 'qkv.bias', 'proj_out.weight', 'proj_out.bias']


def _shapes_res(cin, emb=128, cout=64):
    s = {'in_layers.0.weight': (cin,), 'in_layers.0.bias': (cin,), 'in_layers.2.weight': (cout, cin, 3, 3), 'in_layers.2.bias': (cout,),
         'emb_layers.1.weight': (2 * cout, emb), 'emb_layers.1.bias': (2 * cout,), 'out_layers.0.weight': (cout,),
         'out_layers.0.bias': (cout,), 'out_layers.3.weight': (cout, cout, 3, 3), 'out_layers.3.bias': (cout,)}
    if cin != cout:
        s['skip_connection.weight'] = (cout, cin, 1, 1)
        s['skip_connection.bias'] = (cout,)
    return s


def _shapes_res_o(cin, cout):
    return _shapes_res(cin, 128, cout)


def _shapes_attn(c=64):
    return {'norm.weight': (c,), 'norm.bias': (c,), 'qkv.weight': (3 * c, c, 1), 'qkv.bias': (3 * c,), 'proj_out.weight': (c, c, 1),
            'proj_out.bias': (c,)}


def _draw(shapes, order, seed):
    sd = {}
    for i, k in enumerate([k for k in order if k in shapes]):
        gk = torch.Generator().manual_seed(seed * 1000 + i)
        r = torch.randn(shapes[k], generator=gk)
        if len(shapes[k]) == 1:
            norm = ('norm' in k or 'in_layers.0' in k or 'out_layers.0' in k) and k.endswith('weight')
            sd[k] = 1 + 0.1 * r if norm else 0.1 * r
        else:
            sd[k] = r / float(np.sqrt(int(np.prod(shapes[k][1:]))))
    return sd


def res_state(cin, hs):
    return _draw(_shapes_res(cin), RES_KEYS, cin + hs)


def attn_state(hs):
    return _draw(_shapes_attn(), ATTN_KEYS, 500 + hs)


# F14 (tests/golden/f14_blocks16.npz): the blocks of the MNIST-sized net's 16x16 / 32x32 levels; inputs from seeds as well
def attn16_state():
    return _draw(_shapes_attn(), ATTN_KEYS, 516)


def res_fine_state(cin, cout, hs):
    return _draw(_shapes_res_o(cin, cout), RES_KEYS, 7000 + cin + hs)


def seeded_input(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * 1.5 + 0.3


def attn16_input():
    return seeded_input((2, 64, 16, 16), 1416)


def res_fine_input(cin, hs, B):
    return seeded_input((B, cin, hs, hs), 1400 + cin + hs), torch.randn(B, 128, generator=torch.Generator().manual_seed(1500 + cin + hs))


def input_digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.numpy().tobytes())
    return h.digest().hex()


def digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()
