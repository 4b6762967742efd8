"""MLPModel: the 2-D toy score network (dlpm/models/Model.py:17-211) as a parameter container whose
forward runs in libdlpm_amd.  Same `state_dict` keys (including the aliases the reference's
nn.Sequential re-registrations create) and the same construction order, hence the same default
initialisation under a given torch.manual_seed.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import _lib


def _cond_block(nunits, temb):
    """Parameters of DiffusionBlockConditioned (time-conditioned, LayerNorm): DiffusionBlocks.py:88-123."""
    blk = nn.Module()
    blk.group_norm1 = nn.LayerNorm([nunits])
    blk.group_norm2 = nn.LayerNorm([nunits])
    drop = nn.Dropout(p=0.0)
    act = nn.SiLU()
    blk.mlp_1 = nn.Sequential(drop, nn.Linear(nunits, nunits), blk.group_norm1)
    blk.t_proj = nn.Sequential(drop, nn.Linear(temb, nunits), act)
    blk.mlp_2 = nn.Sequential(drop, nn.Linear(nunits, nunits), blk.group_norm2)
    return blk


class MLPModel(nn.Module):
    def __init__(self, p):
        super().__init__()
        m = p['model']
        self.nfeatures = p['data']['nfeatures']
        self.nunits, self.nblocks, self.time_emb_size = m['nunits'], m['nblocks'], m['time_emb_size']
        unsupported = []
        if not (m['no_a'] and p[p['method']]['isotropic']):
            unsupported.append('a_t inputs / non-isotropic (the reference asserts the same, Model.py:44)')
        if m['time_emb_type'] != 'learnable':
            unsupported.append("time_emb_type != 'learnable'")
        if m.get('a_pos_emb') or m.get('learn_variance'):
            unsupported.append('a_pos_emb / learn_variance')
        if not m['group_norm'] or not m['skip_connection'] or m.get('dropout_rate', 0.0):
            unsupported.append('group_norm=False / skip_connection=False / dropout')
        if unsupported:
            raise NotImplementedError('dlpm_amd.MLPModel implements the shipped 2d_data.yml architecture; got: '
                                      + '; '.join(unsupported))
        act = nn.SiLU()
        self.group_norm_in = nn.LayerNorm([self.nunits])
        self.time_emb = nn.Linear(1, self.time_emb_size)
        self.time_mlp = nn.Sequential(self.time_emb, act, nn.Linear(self.time_emb_size, self.time_emb_size), act)
        self.linear_in = nn.Linear(self.nfeatures, self.nunits)
        self.inblock = nn.Sequential(self.linear_in, self.group_norm_in, act)
        self.midblocks = nn.ModuleList([_cond_block(self.nunits, self.time_emb_size) for _ in range(self.nblocks)])
        self.outblocks_mean = nn.ModuleList([_cond_block(self.nunits, self.time_emb_size),
                                             nn.Linear(self.nunits, self.nfeatures)])
        self._handle = None
        self.handle_generation = 0          # bumped whenever the native handle is destroyed (sampler cache keys carry it)
        self._dependents = weakref.WeakSet()  # method objects holding native samplers built on this handle

    def invalidate(self):
        for m in list(self._dependents):
            m._drop_samplers_of(self)
        if self._handle is not None:
            _lib.lib().dlpm_mlp_destroy(self._handle)
            self.handle_generation += 1
        self._handle = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def __del__(self):
        try:
            self.invalidate()
        except Exception:
            pass

    def native_handle(self):
        if self._handle is not None:
            return self._handle
        L = _lib.lib()
        h = C.c_void_p()
        _lib.check(L.dlpm_mlp_create(self.nfeatures, self.nunits, self.nblocks, self.time_emb_size, C.byref(h)))
        for k, v in self.state_dict().items():
            w = v.detach().to('cpu', torch.float32).contiguous()
            _lib.check(L.dlpm_mlp_set_param(h, k.encode(), w.data_ptr(), w.numel()))
        _lib.check(L.dlpm_mlp_finalize(h))
        self._handle = h
        return h

    def forward(self, x, timestep, a_t_prime=None, a_t_1=None):
        """eps[B,1,F] = model(x[B,1,F], t[B]) on the GPU."""
        if not x.is_cuda:
            raise _lib.DlpmError('dlpm_amd.MLPModel.forward runs on the MI355X only; there is no CPU fallback')
        B = x.shape[0]
        assert x.shape[1:] == (1, self.nfeatures), x.shape
        x = x.contiguous().float()
        t = timestep.to(x.device, torch.float32).contiguous()
        out = torch.empty_like(x)
        _lib.check(_lib.lib().dlpm_mlp_forward(self.native_handle(), x.data_ptr(), t.data_ptr(), out.data_ptr(), B,
                                              _lib.stream_ptr()))
        return out
