"""UNetModel: parameter container + HIP forward, drop-in for the reference's score network.

Mirrors the constructor signature of dlpm/models/unet.py:298-314 as used by
dlpm/dlpm_experiment.py:38-56 and exposes the SAME `state_dict` keys/shapes (so reference
checkpoints load with `load_state_dict`) and, because the torch modules are instantiated in the
reference's construction order (unet.py:334-436), the SAME default initialisation under a given
`torch.manual_seed`.  `forward(x, timesteps)` runs entirely in libdlpm_amd (HIP, gfx950); there is
no PyTorch compute path and it raises if the library is unavailable.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import _lib


def _holder(children):
    """A bare module whose children carry the reference's numeric / attribute names."""
    m = nn.Module()
    for name, child in children:
        m.add_module(name, child)
    return m


def _res_params(cin, emb_dim, cout):
    # order of creation = order of RNG draws in the reference ResBlock.__init__ (unet.py:140-168)
    in_gn = nn.GroupNorm(min(32, cin), cin)
    in_conv = nn.Conv2d(cin, cout, 3, padding=1)
    emb_lin = nn.Linear(emb_dim, 2 * cout)          # use_scale_shift_norm=True
    out_gn = nn.GroupNorm(min(32, cout), cout)
    out_conv = nn.Conv2d(cout, cout, 3, padding=1)
    for p in out_conv.parameters():                 # zero_module
        p.detach().zero_()
    kids = [('in_layers', _holder([('0', in_gn), ('2', in_conv)])),
            ('emb_layers', _holder([('1', emb_lin)])),
            ('out_layers', _holder([('0', out_gn), ('3', out_conv)]))]
    if cin != cout:
        kids.append(('skip_connection', nn.Conv2d(cin, cout, 1)))
    return _holder(kids)


def _attn_params(ch):
    norm = nn.GroupNorm(min(32, ch), ch)
    qkv = nn.Conv1d(ch, 3 * ch, 1)
    proj = nn.Conv1d(ch, ch, 1)
    for p in proj.parameters():
        p.detach().zero_()
    return _holder([('norm', norm), ('qkv', qkv), ('proj_out', proj)])


class UNetModel(nn.Module):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, num_heads=1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 image_size=None):
        super().__init__()
        if dims != 2 or num_classes is not None or not conv_resample or not use_scale_shift_norm:
            raise NotImplementedError('dlpm_amd.UNetModel implements the configuration DLPM instantiates '
                                      '(dims=2, unconditional, conv_resample, use_scale_shift_norm=True)')
        if num_heads_upsample not in (-1, num_heads):
            raise NotImplementedError('num_heads_upsample must equal num_heads')
        if dropout:
            raise NotImplementedError('sampling-only implementation: dropout is inactive in eval() anyway')
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        self.channel_mult = tuple(channel_mult)
        self.num_heads = num_heads
        self.image_size = image_size
        mc, ted = model_channels, 4 * model_channels

        self.time_embed = _holder([('0', nn.Linear(mc, ted)), ('2', nn.Linear(ted, ted))])
        blocks = [_holder([('0', nn.Conv2d(in_channels, mc, 3, padding=1))])]
        chans, ch, ds = [mc], mc, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(num_res_blocks):
                kids = [('0', _res_params(ch, ted, mult * mc))]
                ch = mult * mc
                if ds in self.attention_resolutions:
                    kids.append(('1', _attn_params(ch)))
                blocks.append(_holder(kids))
                chans.append(ch)
            if level != len(self.channel_mult) - 1:
                blocks.append(_holder([('0', _holder([('op', nn.Conv2d(ch, ch, 3, stride=2, padding=1))]))]))
                chans.append(ch)
                ds *= 2
        self.input_blocks = nn.ModuleList(blocks)
        self.middle_block = _holder([('0', _res_params(ch, ted, ch)), ('1', _attn_params(ch)),
                                     ('2', _res_params(ch, ted, ch))])
        ups = []
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                kids = [('0', _res_params(ch + chans.pop(), ted, mc * mult))]
                ch = mc * mult
                if ds in self.attention_resolutions:
                    kids.append((str(len(kids)), _attn_params(ch)))
                if level and i == num_res_blocks:
                    kids.append((str(len(kids)), _holder([('conv', nn.Conv2d(ch, ch, 3, padding=1))])))
                    ds //= 2
                ups.append(_holder(kids))
        self.output_blocks = nn.ModuleList(ups)
        out_conv = nn.Conv2d(mc, out_channels, 3, padding=1)
        for p in out_conv.parameters():
            p.detach().zero_()
        self.out = _holder([('0', nn.GroupNorm(min(32, ch), ch)), ('2', out_conv)])

        self._handle = None
        self._handle_size = None
        self._ws = None
        self._conv_policy = (_lib.CONV_AUTO, 0)
        self._gemm_policy = _lib.GEMM_AUTO
        self.handle_generation = 0          # bumped whenever the native handle is destroyed (sampler cache keys carry it)
        self._dependents = weakref.WeakSet()  # method objects holding native samplers built on this handle

    # ------------------------------------------------------------------ native handle management
    def invalidate(self):
        """Call after changing parameters in place; the next forward re-uploads them.  Native samplers captured
        against the old handle (their hipGraphs hold its weight pointers) are destroyed first."""
        for m in list(self._dependents):
            m._drop_samplers_of(self)
        if self._handle is not None:
            _lib.lib().dlpm_unet_destroy(self._handle)
            self.handle_generation += 1
        self._handle = None

    def set_conv_policy(self, generation='auto', dispatch_batch=0):
        """Which kernel generation the 3x3 stride-1 convolutions take (dlpm_unet_set_conv_policy): 'auto' (fastest per
        layer geometry), 'f4' (Winograd F(4x4,3x3)), 'f2' (Winograd F(2x2,3x3)), 'igemm'.  Never a function of the batch
        of a call, so samples do not depend on sharding / chunking; `dispatch_batch` lets 'auto' weigh grid occupancy
        for a batch the caller declares for the whole configuration."""
        gen = {'auto': _lib.CONV_AUTO, 'f4': _lib.CONV_F4, 'f2': _lib.CONV_F2, 'igemm': _lib.CONV_IGEMM}[generation]
        self._conv_policy = (gen, int(dispatch_batch))
        self._conv_policy_declared = True      # an explicit declaration: GenerationManager.generate leaves it alone
        if self._handle is not None:
            _lib.check(_lib.lib().dlpm_unet_set_conv_policy(self._handle, gen, int(dispatch_batch)))

    def declare_batch(self, nsamples):
        """The batch a caller that never declared one is about to sample (GenerationManager.generate: its `nsamples`): becomes the
        dispatch batch of the current generation, unless set_conv_policy was called explicitly.  At a declared batch <= 256 the
        16x16 / 8x8 levels of a 128-channel-multiple net run on 64- / 32-channel n-tiles instead of leaving CUs idle (round 6)."""
        if getattr(self, '_conv_policy_declared', False):
            return
        gen = self._conv_policy[0] if getattr(self, '_conv_policy', None) else _lib.CONV_AUTO
        self._conv_policy = (gen, int(nsamples))
        if self._handle is not None:
            _lib.check(_lib.lib().dlpm_unet_set_conv_policy(self._handle, gen, int(nsamples)))

    def set_gemm_policy(self, mode='auto'):
        """Which matrix pipe the 1x1 convolutions take (dlpm_unet_set_gemm_policy): 'bf16x3' (fp32 operands cut exactly
        into three bf16 planes, six partial products accumulated in fp32: fp32-grade results at 6/16 of the pipe time),
        'f32' (the fp32 MFMA), 'auto' = 'bf16x3' where the shape admits it.  Never a function of the batch."""
        mode = {'auto': _lib.GEMM_AUTO, 'f32': _lib.GEMM_F32, 'bf16x3': _lib.GEMM_BF16X3}[mode]
        self._gemm_policy = mode
        if self._handle is not None:
            _lib.check(_lib.lib().dlpm_unet_set_gemm_policy(self._handle, mode))

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def __del__(self):
        try:
            self.invalidate()
        except Exception:
            pass

    def native_handle(self, image_size):
        """Create (once per image size) the C handle and upload the weights."""
        if self._handle is not None and self._handle_size == image_size:
            return self._handle
        self.invalidate()
        L = _lib.lib()
        cfg = _lib.UNetConfig()
        cfg.in_channels, cfg.model_channels, cfg.out_channels = self.in_channels, self.model_channels, self.out_channels
        cfg.num_res_blocks, cfg.num_heads, cfg.image_size = self.num_res_blocks, self.num_heads, image_size
        cfg.n_mult = len(self.channel_mult)
        for i, m in enumerate(self.channel_mult):
            cfg.channel_mult[i] = m
        cfg.n_attn = len(self.attention_resolutions)
        for i, a in enumerate(self.attention_resolutions):
            cfg.attention_resolutions[i] = a
        h = C.c_void_p()
        _lib.check(L.dlpm_unet_create(C.byref(cfg), C.byref(h)))
        sd = self.state_dict()
        n = L.dlpm_unet_num_params(h)
        if n != len(sd):
            L.dlpm_unet_destroy(h)
            raise _lib.DlpmError('architecture mismatch: library expects %d tensors, module has %d' % (n, len(sd)))
        for k, v in sd.items():
            w = v.detach().to('cpu', torch.float32).contiguous()
            _lib.check(L.dlpm_unet_set_param(h, k.encode(), w.data_ptr(), w.numel()))
        _lib.check(L.dlpm_unet_finalize(h))
        _lib.check(L.dlpm_unet_set_conv_policy(h, self._conv_policy[0], self._conv_policy[1]))
        _lib.check(L.dlpm_unet_set_gemm_policy(h, self._gemm_policy))
        self._handle, self._handle_size = h, image_size
        return h

    def workspace(self, B, device):
        need = _lib.lib().dlpm_unet_workspace_bytes(self._handle, B)
        if need < 0:
            raise _lib.DlpmError(_lib.lib().dlpm_last_error().decode())
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def workspace_bytes(self, B, image_size):
        """Peak activation workspace of a forward at batch B (dlpm_unet_workspace_bytes: a dry run of the plan's arena)."""
        need = _lib.lib().dlpm_unet_workspace_bytes(self.native_handle(image_size), B)
        if need < 0:
            raise _lib.DlpmError(_lib.lib().dlpm_last_error().decode())
        return int(need)

    def flops_per_sample(self, image_size):
        return _lib.lib().dlpm_unet_flops_per_sample(self.native_handle(image_size))

    # ------------------------------------------------------------------ model(x, t) protocol
    def forward(self, x, timesteps, y=None):
        """eps = model(x[B,C,H,W] fp32 on the GPU, t[B] floats) -- GenerativeLevyProcess.py:180."""
        assert y is None, 'class conditioning is not part of the DLPM path'
        if not x.is_cuda:
            raise _lib.DlpmError('dlpm_amd.UNetModel.forward runs on the MI355X only (x is on %s); '
                                 'there is no CPU fallback' % x.device)
        B, Cc, H, W = x.shape
        assert Cc == self.in_channels and H == W, (x.shape, self.in_channels)
        h = self.native_handle(H)
        x = x.contiguous().float()
        t = timesteps.to(x.device, torch.float32).contiguous()
        assert t.shape == (B,)
        out = torch.empty((B, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        ws = self.workspace(B, x.device)
        _lib.check(_lib.lib().dlpm_unet_forward(h, x.data_ptr(), t.data_ptr(), out.data_ptr(), B, ws.data_ptr(),
                                               ws.numel(), _lib.stream_ptr()))
        return out

    def get_feature_vectors(self, x, timesteps, y=None):
        """Block outputs of a forward as NCHW tensors: {'down': [...], 'middle': t, 'up': [...]}
        (same structure as the reference's UNetModel.get_feature_vectors, unet.py:494-524)."""
        L = _lib.lib()
        _lib.check(L.dlpm_unet_keep_features(self.native_handle(x.shape[2]), 1))   # no arena recycling for this forward
        try:
            self.forward(x, timesteps)
            return self._collect_features(x)
        finally:
            _lib.check(L.dlpm_unet_keep_features(self._handle, 0))

    def _collect_features(self, x):
        L, h = _lib.lib(), self._handle
        B = x.shape[0]
        feats = []
        for i in range(L.dlpm_unet_num_features(h)):
            c, hh, ww = C.c_int32(), C.c_int32(), C.c_int32()
            _lib.check(L.dlpm_unet_feature_shape(h, i, C.byref(c), C.byref(hh), C.byref(ww)))
            f = torch.empty((B, c.value, hh.value, ww.value), dtype=torch.float32, device=x.device)
            _lib.check(L.dlpm_unet_get_feature(h, i, f.data_ptr(), B, _lib.stream_ptr()))
            feats.append(f)
        n_in = len(self.input_blocks)
        return {'down': feats[:n_in], 'middle': feats[n_in], 'up': feats[n_in + 1:]}


def unet_from_config(p):
    """_unet_model(p): dlpm/dlpm_experiment.py:24-57 (reference YAML schema)."""
    m = p['model']
    return UNetModel(in_channels=p['data']['channels'], model_channels=m['model_channels'],
                     out_channels=p['data']['channels'], num_res_blocks=m['num_res_blocks'],
                     attention_resolutions=m['attn_resolutions'], dropout=m['dropout'], channel_mult=m['channel_mult'],
                     dims=2, num_classes=None, use_checkpoint=False, num_heads=m['num_heads'], num_heads_upsample=-1,
                     use_scale_shift_norm=True, image_size=p['data'].get('image_size'))
