"""Oracle: score-network forwards (CPU, torch fp32 functional ops) -- TEST INFRASTRUCTURE ONLY.

`unet_forward` restates dlpm/models/unet.py:463-492 (UNetModel.forward) and the blocks it
calls, driven by a flat state_dict with the reference's key names, so it can be checked against
the reference (tests/golden/f6_*, f7_*) and then serve as the checker for the HIP path.
`mlp_forward` restates dlpm/models/Model.py:148-211 / DiffusionBlocks.py:125-136.
"""
import math

import torch
import torch.nn.functional as F


def silu(x):
    # dlpm/models/nn.py:12-14
    return x * torch.sigmoid(x)


def timestep_embedding(t, dim, max_period=10000):
    """[cos | sin] sinusoidal features of (possibly fractional) t: dlpm/models/nn.py:103-121."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    ang = t[:, None].float() * freqs[None]
    out = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    if dim % 2:
        out = torch.cat([out, torch.zeros_like(out[:, :1])], dim=-1)
    return out


def group_norm(x, w, b):
    # GroupNorm32 with min(32, C) groups, eps 1e-5: nn.py:17-19,93-100; unet.py:141
    return F.group_norm(x.float(), min(32, x.shape[1]), w, b, 1e-5)


def qkv_attention(qkv):
    """unet.py:230-250.  qkv: [N, 3*ch, T] with q|k|v stacked along dim 1."""
    ch = qkv.shape[1] // 3
    q, k, v = qkv[:, :ch], qkv[:, ch:2 * ch], qkv[:, 2 * ch:]
    sc = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum('bct,bcs->bts', q * sc, k * sc)
    w = torch.softmax(w.float(), dim=-1)
    return torch.einsum('bts,bcs->bct', w, v)


def res_block(sd, pre, x, emb):
    """ResBlock._forward with use_scale_shift_norm=True: unet.py:176-195."""
    h = F.conv2d(silu(group_norm(x, sd[pre + 'in_layers.0.weight'], sd[pre + 'in_layers.0.bias'])),
                 sd[pre + 'in_layers.2.weight'], sd[pre + 'in_layers.2.bias'], padding=1)
    e = F.linear(silu(emb), sd[pre + 'emb_layers.1.weight'], sd[pre + 'emb_layers.1.bias'])
    scale, shift = torch.chunk(e[:, :, None, None], 2, dim=1)
    h = group_norm(h, sd[pre + 'out_layers.0.weight'], sd[pre + 'out_layers.0.bias']) * (1 + scale) + shift
    h = F.conv2d(silu(h), sd[pre + 'out_layers.3.weight'], sd[pre + 'out_layers.3.bias'], padding=1)
    if pre + 'skip_connection.weight' in sd:
        x = F.conv2d(x, sd[pre + 'skip_connection.weight'], sd[pre + 'skip_connection.bias'])
    return x + h


def attention_block(sd, pre, x, heads):
    """AttentionBlock._forward: unet.py:220-228."""
    b, c = x.shape[:2]
    flat = x.reshape(b, c, -1)
    n = group_norm(flat, sd[pre + 'norm.weight'], sd[pre + 'norm.bias'])
    qkv = F.conv1d(n, sd[pre + 'qkv.weight'], sd[pre + 'qkv.bias'])
    a = qkv_attention(qkv.reshape(b * heads, -1, qkv.shape[2])).reshape(b, -1, qkv.shape[2])
    a = F.conv1d(a, sd[pre + 'proj_out.weight'], sd[pre + 'proj_out.bias'])
    return (flat + a).reshape(x.shape)


def _run_seq(sd, pre, h, emb, heads):
    """One TimestepEmbedSequential (unet.py:33-45): children are discovered from the key names."""
    idx = sorted({int(k[len(pre):].split('.')[0]) for k in sd if k.startswith(pre)})
    for i in idx:
        p = '%s%d.' % (pre, i)
        if p + 'in_layers.0.weight' in sd:
            h = res_block(sd, p, h, emb)
        elif p + 'qkv.weight' in sd:
            h = attention_block(sd, p, h, heads)
        elif p + 'op.weight' in sd:        # Downsample, stride-2 conv: unet.py:96
            h = F.conv2d(h, sd[p + 'op.weight'], sd[p + 'op.bias'], stride=2, padding=1)
        elif p + 'conv.weight' in sd:      # Upsample, nearest x2 then conv: unet.py:73-75
            h = F.conv2d(F.interpolate(h, scale_factor=2, mode='nearest'),
                         sd[p + 'conv.weight'], sd[p + 'conv.bias'], padding=1)
        elif p + 'weight' in sd:           # the stem conv input_blocks.0.0
            h = F.conv2d(h, sd[p + 'weight'], sd[p + 'bias'], padding=1)
        else:
            raise KeyError(p)
    return h


def unet_forward(sd, x, t, heads, return_feats=False):
    """UNetModel.forward: unet.py:463-492.  `sd` = reference-keyed state dict, t = i/T floats."""
    mc = sd['time_embed.0.weight'].shape[1]
    emb = F.linear(timestep_embedding(t, mc), sd['time_embed.0.weight'], sd['time_embed.0.bias'])
    emb = F.linear(silu(emb), sd['time_embed.2.weight'], sd['time_embed.2.bias'])
    n_in = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('input_blocks.'))
    n_out = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('output_blocks.'))
    feats = {'down': [], 'up': []}
    hs, h = [], x.float()
    for i in range(n_in):
        h = _run_seq(sd, 'input_blocks.%d.' % i, h, emb, heads)
        hs.append(h)
        feats['down'].append(h)
    h = _run_seq(sd, 'middle_block.', h, emb, heads)
    feats['middle'] = h
    for i in range(n_out):
        h = _run_seq(sd, 'output_blocks.%d.' % i, torch.cat([h, hs.pop()], dim=1), emb, heads)
        feats['up'].append(h)
    y = F.conv2d(silu(group_norm(h, sd['out.0.weight'], sd['out.0.bias'])),
                 sd['out.2.weight'], sd['out.2.bias'], padding=1)
    return (y, feats) if return_feats else y


def _mlp_block(sd, pre, x, temb):
    """DiffusionBlockConditioned.forward (time-conditioned, skip, LayerNorm): DiffusionBlocks.py:125-136."""
    nu = x.shape[-1]
    h = F.layer_norm(F.linear(x, sd[pre + 'mlp_1.1.weight'], sd[pre + 'mlp_1.1.bias']), [nu],
                     sd[pre + 'group_norm1.weight'], sd[pre + 'group_norm1.bias'])
    h = F.silu(h)
    h = h + F.silu(F.linear(temb, sd[pre + 't_proj.1.weight'], sd[pre + 't_proj.1.bias']))
    h = F.layer_norm(F.linear(h, sd[pre + 'mlp_2.1.weight'], sd[pre + 'mlp_2.1.bias']), [nu],
                     sd[pre + 'group_norm2.weight'], sd[pre + 'group_norm2.bias'])
    return F.silu(h + x)


def mlp_forward(sd, x, t):
    """MLPModel.forward for the 2d_data.yml architecture (learnable time embedding, no a_t input,
    LayerNorm, skip connections): dlpm/models/Model.py:148-211."""
    tt = t.reshape(-1, 1, 1).float()
    temb = F.silu(F.linear(tt, sd['time_emb.weight'], sd['time_emb.bias']))
    temb = F.silu(F.linear(temb, sd['time_mlp.2.weight'], sd['time_mlp.2.bias']))
    nu = sd['linear_in.weight'].shape[0]
    h = F.silu(F.layer_norm(F.linear(x, sd['linear_in.weight'], sd['linear_in.bias']), [nu],
                            sd['group_norm_in.weight'], sd['group_norm_in.bias']))
    nb = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('midblocks.'))
    for i in range(nb):
        h = _mlp_block(sd, 'midblocks.%d.' % i, h, temb)
    h = _mlp_block(sd, 'outblocks_mean.0.', h, temb)
    return F.linear(h, sd['outblocks_mean.1.weight'], sd['outblocks_mean.1.bias'])
