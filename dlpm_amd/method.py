"""GenerativeLevyProcess: the sampling entry point, drop-in for the reference's method object.

Mirrors the constructor and `sample(...)` signature of
dlpm/methods/GenerativeLevyProcess.py:48-90,512-569 (what `GenerationManager.generate` and
`eval.py --generate` call).  The loop itself runs in libdlpm_amd:

  * `models['default']` is a dlpm_amd.UNetModel / MLPModel  -> dlpm_sampler_* (one reverse step
    captured as a hipGraph and replayed T-1 times);
  * any other callable `model(x, t)` on the GPU                -> the same noise / table / update
    kernels, with the model called from Python between them.

Two RNG modes:
  rng='philox'     device Philox4x32-10 keyed by (seed, GLOBAL sample index, step, element): the
                   samples do not depend on how a batch is sharded over GPUs (default);
  rng='reference'  the reference's CPU streams (numpy MT19937 -> scipy-style CMS, torch MT19937 ->
                   randn) regenerated on the host by libdlpm_amd and uploaded, for parity with the
                   reference CPU path on identical seeds.
"""
import contextlib
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib
from .process import DLPM


class ModelMeanType:
    """What the net predicts (dlpm/methods/dlpm.py:10-18)."""
    PREVIOUS_X = 'PREVIOUS_X'   # the anterior mean m_tilde_{t-1}
    START_X = 'START_X'         # x_0
    EPSILON = 'EPSILON'         # eps (every shipped config)
    Z = 'Z'                     # z_t


class ModelVarType:
    FIXED = 'FIXED'


class ReferenceStreams:
    """Stream N (numpy global RandomState, consumed by scipy's levy_stable.rvs) and stream P (torch's
    default CPU generator), as libdlpm_amd MT19937 states."""

    def __init__(self, np_seed=0, torch_seed=0):
        self.N, self.P = _lib.MT19937(), _lib.MT19937()
        L = _lib.lib()
        _lib.check(L.dlpm_mt19937_seed(C.byref(self.N), np_seed & 0xFFFFFFFF))
        _lib.check(L.dlpm_mt19937_seed(C.byref(self.P), torch_seed & 0xFFFFFFFF))

    @classmethod
    def from_global_numpy(cls, torch_seed):
        """Continue numpy's process-global stream exactly where it stands (write back with
        `store_global_numpy`)."""
        s = cls(0, torch_seed)
        kind, key, pos, _, _ = np.random.get_state()
        assert kind == 'MT19937'
        for i in range(624):
            s.N.key[i] = int(key[i])
        s.N.pos = int(pos)
        return s

    def store_global_numpy(self):
        key = np.array([self.N.key[i] for i in range(624)], dtype=np.uint32)
        np.random.set_state(('MT19937', key, int(self.N.pos), 0, 0.0))

    def skewed_levy(self, alpha, n, clamp_a=None):
        out = np.empty(n, np.float32)
        _lib.check(_lib.lib().dlpm_skewed_levy_host_f32(C.byref(self.N), float(alpha), n,
                                                       -1.0 if clamp_a is None else float(clamp_a), out.ctypes.data))
        return torch.from_numpy(out)

    def randn(self, shape):
        n = int(np.prod(shape))
        out = np.empty(n, np.float32)
        _lib.check(_lib.lib().dlpm_randn_host_f32(C.byref(self.P), n, out.ctypes.data))
        return torch.from_numpy(out).reshape(list(shape))


class GenerativeLevyProcess:
    def __init__(self, alpha, device, reverse_steps, model_mean_type=ModelMeanType.EPSILON,
                 model_var_type=ModelVarType.FIXED, time_spacing='linear', rescale_timesteps=False, isotropic=True,
                 LIM=False, scale='scale_preserving', input_scaling=False,
                 rng='philox', seed=0, sample_offset=0, use_graph=True, reference_streams=None, fused_mlp=True):
        # The reference constructor asserts EPSILON here (GenerativeLevyProcess.py:74-77) although p_mean_variance carries
        # the START_X / Z / PREVIOUS_X branches (:186-207), reachable there only by setting `model_mean_type` afterwards.
        # This build runs them (dlpm_predict_f32); the variance stays FIXED as in the reference.
        assert model_mean_type in _lib.MEAN_TYPES, 'unknown model_mean_type {}'.format(model_mean_type)
        assert model_var_type == ModelVarType.FIXED, 'Only fixed variance is supported for the moment'
        if LIM:
            assert (model_mean_type == ModelMeanType.EPSILON) and rescale_timesteps, \
                'LIM only supports epsilon prediction, fixed variance and rescaled timesteps'
            if not isotropic:
                raise NotImplementedError('the LIM sampler is isotropic only (its non-isotropic branch is commented '
                                          'out in the reference, LIM/functions/sampler.py:144-148)')
            from .lim import VPSDE
            self.sde = VPSDE(alpha, 'cosine')
        assert rng in ('philox', 'reference')
        self.alpha, self.device, self.reverse_steps = alpha, device, reverse_steps
        self.model_mean_type, self.model_var_type = model_mean_type, model_var_type
        self.time_spacing, self.rescale_timesteps, self.isotropic = time_spacing, rescale_timesteps, isotropic
        self.LIM, self.input_scaling = LIM, input_scaling
        self.rng, self.seed, self.sample_offset, self.use_graph = rng, seed, sample_offset, use_graph
        self.reference_streams = reference_streams
        self.fused_mlp = fused_mlp
        self.dlpm = DLPM(alpha, device, diffusion_steps=reverse_steps, time_spacing=time_spacing, isotropic=isotropic,
                         scale=scale)
        self._samplers = {}
        self.calls = 0          # number of sample() calls so far: folded into the Philox key
        self._dataset = None    # see dataset_stream()

    # -------------------------------------------------------------------------------- helpers
    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1.0 / self.reverse_steps)
        return t

    def get_timesteps(self, N, **kwargs):
        return self.dlpm.get_timesteps(N)

    def _philox_key(self):
        """(seed, first global sample index) of the next sample() call."""
        calls, first = (self._dataset['calls'], self._dataset['next']) if self._dataset else (self.calls, 0)
        return (self.seed + 0x9E3779B97F4A7C15 * calls) & 0xFFFFFFFFFFFFFFFF, self.sample_offset + first

    @contextlib.contextmanager
    def dataset_stream(self, first_index=0):
        """Inside this context every sample() call continues ONE Philox stream indexed by the running sample
        count, so the i-th generated sample does not depend on how the G samples are cut into chunks
        (EvaluationManager's eval.batch_size loop) or split over ranks.  Outside it each call draws a fresh
        stream, as successive reference calls do.  No effect with rng='reference' (the MT19937 streams
        simply continue, which is the reference's behaviour)."""
        assert self._dataset is None, 'dataset_stream() does not nest'
        self._dataset = dict(calls=self.calls, next=first_index)
        try:
            yield self
        finally:
            self._dataset = None
            self.calls += 1

    def _input_scale(self):
        """[T] host table 1/(1 + barsigma_t), or None: the factor the net input is multiplied by when input_scaling
        is on and the process was built scale_exploding (GenerativeLevyProcess.py:176-180)."""
        if self.input_scaling and self.dlpm.scale == 'scale_exploding':
            return (1 / (1 + self.dlpm.host_schedule[3])).contiguous()
        return None

    def _streams(self):
        if self.reference_streams is None:
            self.reference_streams = ReferenceStreams(self.seed, self.seed)
        return self.reference_streams

    def _native_sampler(self, model, shape, flags, eta, clamp_a, clamp_eps, seed, offset=None):
        offset = self.sample_offset if offset is None else offset
        lim = bool(flags & _lib.SMP_LIM)
        from .unet import UNetModel
        B = shape[0]
        if isinstance(model, UNetModel):
            assert len(shape) == 4 and shape[2] == shape[3], shape
            dims = (shape[1], shape[2], shape[3])
            handles = dict(unet=model.native_handle(shape[2]), mlp=None)
        else:
            assert len(shape) == 3 and shape[1] == 1, shape
            dims = (1, 1, shape[2])
            handles = dict(unet=None, mlp=model.native_handle())
        if not self.isotropic:
            flags |= _lib.UPD_ELEMENTWISE
        # the handle GENERATION, not its address: after load_state_dict / invalidate() a new handle is likely to get the
        # old one's address back from the allocator, and a cached sampler's hipGraph still points at the freed weights
        key = (id(model), model.handle_generation, tuple(shape),
               self.reverse_steps, self.alpha, flags, eta, clamp_a, clamp_eps, self.use_graph, self.fused_mlp,
               self.dlpm.host_schedule[3].data_ptr(), self._input_scale() is not None, self.model_mean_type)
        ent = self._samplers.get(key)
        if ent is not None:
            _lib.check(_lib.lib().dlpm_sampler_reseed(ent['h'], seed, offset))
            return ent['h']
        # one live native sampler per method object: they own activation workspaces sized for B
        for k in list(self._samplers):
            _lib.lib().dlpm_sampler_destroy(self._samplers.pop(k)['h'])
        cfg = _lib.SamplerConfig()
        cfg.unet, cfg.mlp = handles['unet'], handles['mlp']
        cfg.B, (cfg.C, cfg.H, cfg.W), cfg.T = B, dims, self.reverse_steps + (1 if lim else 0)
        cfg.alpha = float(self.alpha)
        cfg.clamp_a = -1.0 if clamp_a is None else float(clamp_a)
        cfg.clamp_eps = -1.0 if clamp_eps is None else float(clamp_eps)
        cfg.flags = flags | (0 if self.fused_mlp else _lib.SMP_NO_FUSED_MLP)
        cfg.dlim_eta, cfg.seed = float(eta), seed
        cfg.mean_type = _lib.MEAN_TYPES[self.model_mean_type]
        gs = 0
        if self.use_graph and self.rng == 'philox':
            # steps per captured graph: 1 for the UNets (~150 launches, ms-long steps), 33 for the
            # launch-bound MLP (4 launches per step) unless the caller asked for a specific count
            gs = self.use_graph if isinstance(self.use_graph, int) and not isinstance(self.use_graph, bool) else (
                1 if handles['unet'] else 33)
        cfg.sample_offset, cfg.use_graph = offset, gs
        sched = self.dlpm.host_schedule
        cfg.g, cfg.bg, cfg.s, cfg.bs = (v.data_ptr() for v in sched)
        isc = None if lim else self._input_scale()
        if isc is not None:
            cfg.in_scale = isc.data_ptr()
        if lim:
            from .lim import lim_tables
            tabs = lim_tables(self.sde, self.reverse_steps, bool(flags & _lib.UPD_DLIM))     # kept alive until create returns
            cfg.lim_ts, cfg.lim_tmp, cfg.lim_cx, cfg.lim_cs, cfg.lim_cn = (v.data_ptr() for v in tabs)
        h = C.c_void_p()
        _lib.check(_lib.lib().dlpm_sampler_create(C.byref(cfg), C.byref(h)))
        self._samplers[key] = dict(h=h, model=weakref.ref(model))
        model._dependents.add(self)
        return h

    def _drop_samplers_of(self, model):
        """Called by a model before it destroys its native handle: samplers built on it go first."""
        for k in list(self._samplers):
            m = self._samplers[k]['model']()
            if m is None or m is model:
                _lib.lib().dlpm_sampler_destroy(self._samplers.pop(k)['h'])

    def close(self):
        for k in list(self._samplers):
            _lib.lib().dlpm_sampler_destroy(self._samplers.pop(k)['h'])

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -------------------------------------------------------------------------------- loops
    def _host_noise_prologue(self, shape, clamp_a, clamp_eps, noise):
        """A[T,B] and x_T from the reference's CPU streams, in its draw order."""
        st = self._streams()
        T, B = self.reverse_steps, shape[0]
        n = B if self.isotropic else int(np.prod(shape))       # non-isotropic: one draw per element (Distributions.py:48)
        A = torch.stack([st.skewed_levy(self.alpha, n, clamp_a) for _ in range(T)])       # dlpm.py:226-227
        if noise is not None:
            xT = noise.detach().to('cpu', torch.float32)
        else:                                                                             # GLP.py:313
            a0 = st.skewed_levy(self.alpha, n, None)           # gen_sas draws its own UNclamped a
            a0 = a0.view(-1, *([1] * (len(shape) - 1))) if self.isotropic else a0.view(shape)
            e = torch.sqrt(a0) * st.randn(shape)
            if clamp_eps is not None:
                e = torch.clamp(e, -clamp_eps, clamp_eps)
            xT = self.dlpm.host_schedule[3][-1] * e
        return A.contiguous(), xT.contiguous()

    def _host_noise_prologue_lim(self, shape, clamp_eps):
        """x_0 = gen_eps.generate(shape) then one unclamped a per (step, sample), in the reference's draw order on
        stream N: a_0 first (GenerativeLevyProcess.py:464), then step by step (LIM/functions/sampler.py:143)."""
        st = self._streams()
        steps, B = self.reverse_steps, shape[0]
        a0 = st.skewed_levy(self.alpha, B, None) if self.alpha != 2.0 else torch.full((B,), 2.0)
        e = torch.sqrt(a0.view(-1, *([1] * (len(shape) - 1)))) * st.randn(shape)
        if clamp_eps is not None:
            e = torch.clamp(e, -clamp_eps, clamp_eps)
        return a0, e.contiguous()

    def _run_native(self, model, shape, flags, eta, clamp_a, clamp_eps, noise, history, progress):
        L, st = _lib.lib(), _lib.stream_ptr()
        lim = bool(flags & _lib.SMP_LIM)
        T = self.reverse_steps + (1 if lim else 0)      # LIM runs `reverse_steps` updates, DLPM reverse_steps - 1
        seed, offset = self._philox_key()
        h = self._native_sampler(model, shape, flags, eta, clamp_a, clamp_eps, seed, offset)
        dev = torch.device(self.device)
        x = torch.empty(shape, dtype=torch.float32, device=dev)
        # the update kernel stores every intermediate state into this buffer (row T - t), inside the graph
        hist = torch.empty([T] + list(shape), dtype=torch.float32, device=dev) if history else None
        _lib.check(L.dlpm_sampler_set_history(h, hist.data_ptr() if history else None, st))

        pbar = None
        if progress:
            from tqdm import tqdm
            pbar = tqdm(total=T)
        if lim and self.rng == 'reference':
            # stream N interleaves with the model calls in the reference but is independent of stream P, so all the
            # per-step a's can be drawn up front
            _, x0 = self._host_noise_prologue_lim(shape, clamp_eps)
            ode = bool(flags & _lib.UPD_DLIM)
            A = torch.ones((T - 1, shape[0])) if (ode or self.alpha == 2.0) else torch.stack(
                [self._streams().skewed_levy(self.alpha, shape[0], None) for _ in range(T - 1)])
            A_d, x0_d = A.contiguous().to(dev), x0.to(dev)
            _lib.check(L.dlpm_sampler_begin_injected(h, A_d.data_ptr(), x0_d.data_ptr(), st))
            for _ in range(T - 1):
                z_d = None if ode else self._streams().randn(shape).to(dev)
                _lib.check(L.dlpm_sampler_step_injected(h, z_d.data_ptr() if z_d is not None else None, st))
                if pbar:
                    pbar.update(1)
        elif self.rng == 'reference' or noise is not None:
            A, xT = self._host_noise_prologue(shape, clamp_a, clamp_eps, noise)
            A_d, xT_d = A.to(dev), xT.to(dev)
            _lib.check(L.dlpm_sampler_begin_injected(h, A_d.data_ptr(), xT_d.data_ptr(), st))
            need_z = not (flags & _lib.UPD_DLIM) or eta != 0.0
            for _ in range(T - 1):
                z_d = self._streams().randn(shape).to(dev) if (need_z and self.rng == 'reference') else None
                if z_d is None and need_z:
                    z_d = torch.randn(shape, device=dev)
                _lib.check(L.dlpm_sampler_step_injected(h, z_d.data_ptr() if z_d is not None else None, st))
                if pbar:
                    pbar.update(1)
        else:
            _lib.check(L.dlpm_sampler_begin(h, st))
            if pbar:
                for _ in range(T - 1):
                    _lib.check(L.dlpm_sampler_steps(h, 1, st))
                    pbar.update(1)
            else:
                _lib.check(L.dlpm_sampler_steps(h, T - 1, st))
        if pbar:
            pbar.close()
        _lib.check(L.dlpm_sampler_copy_state(h, x.data_ptr(), st))
        if history:
            _lib.check(L.dlpm_sampler_set_history(h, None, st))
        return (x, hist) if history else x

    def _run_callable(self, model, shape, flags, eta, clamp_a, clamp_eps, noise, history, progress,
                      denoised_fn=None, model_kwargs=None):
        """Generic `model(x, t, **model_kwargs)` (any torch callable on the GPU): same kernels, Python between them.
        Also the loop for a `denoised_fn` (a Python function applied to the x_0 prediction, p_mean_variance :162-167)."""
        model_kwargs = model_kwargs or {}
        L, st = _lib.lib(), _lib.stream_ptr()
        T, B = self.reverse_steps, shape[0]
        D = int(np.prod(shape[1:]))
        dev = torch.device(self.device)
        seed, offset = self._philox_key()
        g, bg, s, bs = (v.to(dev) for v in self.dlpm.host_schedule)
        ca = -1.0 if clamp_a is None else float(clamp_a)
        ce = -1.0 if clamp_eps is None else float(clamp_eps)
        host = self.rng == 'reference' or noise is not None
        if host:
            A, xT = self._host_noise_prologue(shape, clamp_a, clamp_eps, noise)
            A, x = A.to(dev), xT.to(dev).reshape(shape).contiguous()
        elif self.isotropic:
            A = torch.empty((T, B), dtype=torch.float32, device=dev)
            x = torch.empty(shape, dtype=torch.float32, device=dev)
            _lib.check(L.dlpm_skewed_levy_philox_f32(A.data_ptr(), T, B, float(self.alpha), ca, seed, offset, st))
            _lib.check(L.dlpm_init_state_philox_f32(x.data_ptr(), B, D, float(self.alpha), ce,
                                                   float(self.dlpm.host_schedule[3][-1]), seed, offset, st))
        else:
            A = torch.empty((T, B * D), dtype=torch.float32, device=dev)
            x = torch.empty(shape, dtype=torch.float32, device=dev)
            _lib.check(L.dlpm_skewed_levy_elem_philox_f32(A.data_ptr(), T, B, D, float(self.alpha), ca, seed, offset, st))
            _lib.check(L.dlpm_init_state_elem_philox_f32(x.data_ptr(), B, D, float(self.alpha), ce,
                                                        float(self.dlpm.host_schedule[3][-1]), seed, offset, st))
        if not self.isotropic:
            flags |= _lib.UPD_ELEMENTWISE
        c_eps, c_noise = torch.empty_like(A), torch.empty_like(A)
        _lib.check(L.dlpm_coeff_tables_f32(A.data_ptr(), g.data_ptr(), s.data_ptr(), bs.data_ptr(), T, A.shape[1],
                                          c_eps.data_ptr(), c_noise.data_ptr(), None, st))
        t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        tvec = torch.empty(B, dtype=torch.float32, device=dev)
        isc = self._input_scale()
        isc = None if isc is None else isc.to(dev)
        hist = [x.clone()] if history else []
        need_z = not (flags & _lib.UPD_DLIM) or eta != 0.0
        # p_mean_variance (:182-207): EPSILON without clipping bypasses everything (a denoised_fn is then never called, as in
        # the reference); EPSILON + clip without a denoised_fn is a flag of the update kernel; every other case goes
        # model output -> x_0 -> [denoised_fn] -> [clamp] -> eps through dlpm_predict_f32
        clip = bool(flags & _lib.UPD_CLIP)
        mean_type = _lib.MEAN_TYPES[self.model_mean_type]
        predict = (mean_type != 0) or (clip and denoised_fn is not None)
        pa = None
        if predict:
            flags &= ~_lib.UPD_CLIP
            pa = _lib.PredictArgs()
            pa.t_dev, pa.g_dev, pa.bg_dev, pa.bs_dev = t_dev.data_ptr(), g.data_ptr(), bg.data_ptr(), bs.data_ptr()
            pa.c_eps_dev, pa.A_dev = c_eps.data_ptr(), A.data_ptr()
            pa.B, pa.D, pa.T, pa.mean_type = B, D, T, mean_type
            el = _lib.PRED_ELEMENTWISE if not self.isotropic else 0
            tail = (_lib.PRED_CLIP if clip else 0) | _lib.PRED_TO_EPS | el
        args = _lib.UpdateArgs()
        args.t_dev, args.g_dev, args.bg_dev, args.bs_dev = t_dev.data_ptr(), g.data_ptr(), bg.data_ptr(), bs.data_ptr()
        args.c_eps_dev, args.c_noise_dev, args.A_dev = c_eps.data_ptr(), c_noise.data_ptr(), A.data_ptr()
        args.B, args.D, args.T, args.flags, args.dlim_eta, args.alpha = B, D, T, flags, float(eta), float(self.alpha)
        args.seed, args.sample_offset = seed, offset
        pbar = None
        if progress:
            from tqdm import tqdm
            pbar = tqdm(total=T)
        for i in range(T - 1, 0, -1):
            t_dev.fill_(i)
            _lib.check(L.dlpm_fill_scaled_t_f32(tvec.data_ptr(), t_dev.data_ptr(), T, B, st))
            xin = x if isc is None else x * isc[i]
            eps = model(xin, tvec if self.rescale_timesteps else torch.full((B,), i, device=dev), **model_kwargs)
            eps = eps.contiguous().float()
            if predict:
                pa.x_dev, pa.in_dev, pa.out_dev = x.data_ptr(), eps.data_ptr(), eps.data_ptr()
                if denoised_fn is None:
                    pa.flags = _lib.PRED_TO_XSTART | tail
                    _lib.check(L.dlpm_predict_f32(C.byref(pa), st))
                else:
                    pa.flags = _lib.PRED_TO_XSTART | el
                    _lib.check(L.dlpm_predict_f32(C.byref(pa), st))
                    eps = denoised_fn(eps.view(shape)).contiguous().float()
                    pa.in_dev, pa.out_dev, pa.flags = eps.data_ptr(), eps.data_ptr(), tail
                    _lib.check(L.dlpm_predict_f32(C.byref(pa), st))
            z = None
            if host and need_z:
                z = (self._streams().randn(shape) if self.rng == 'reference' else torch.randn(shape)).to(dev)
            args.x_dev, args.eps_dev = x.data_ptr(), eps.data_ptr()
            args.z_dev = z.data_ptr() if z is not None else None
            _lib.check(L.dlpm_update_f32(C.byref(args), st))
            if history:
                hist.append(x.clone())
            if pbar:
                pbar.update(1)
        if pbar:
            pbar.close()
        return (x, torch.stack(hist)) if history else x

    def _run_callable_lim(self, model, shape, flags, clamp_eps, history, progress):
        """LIM loop around a generic `model(x, t)` callable: same kernels, Python between them."""
        from .lim import lim_tables
        L, st = _lib.lib(), _lib.stream_ptr()
        steps, B = self.reverse_steps, shape[0]
        T = steps + 1
        D = int(np.prod(shape[1:]))
        dev = torch.device(self.device)
        ode = bool(flags & _lib.UPD_DLIM)
        seed, offset = self._philox_key()
        ts, tmp, cx, cs, cn = (v.to(dev) for v in lim_tables(self.sde, steps, ode))
        ce = -1.0 if clamp_eps is None else float(clamp_eps)
        host = self.rng == 'reference'
        gauss = self.alpha == 2.0
        A = None
        if host:
            _, x0 = self._host_noise_prologue_lim(shape, clamp_eps)
            x = x0.to(dev)
            if not (ode or gauss):
                A = torch.stack([self._streams().skewed_levy(self.alpha, B, None) for _ in range(steps)]).contiguous().to(dev)
        else:
            x = torch.empty(shape, dtype=torch.float32, device=dev)
            _lib.check(L.dlpm_init_state_philox_f32(x.data_ptr(), B, D, float(self.alpha), ce, 1.0, seed, offset, st))
            if not (ode or gauss):
                A = torch.empty((steps, B), dtype=torch.float32, device=dev)
                _lib.check(L.dlpm_skewed_levy_philox_f32(A.data_ptr(), steps, B, float(self.alpha), -1.0, seed, offset, st))
        t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        a = _lib.LimUpdateArgs()
        a.t_dev, a.tmp_dev, a.cx_dev, a.cs_dev, a.cn_dev = t_dev.data_ptr(), tmp.data_ptr(), cx.data_ptr(), cs.data_ptr(), cn.data_ptr()
        a.A_dev = A.data_ptr() if A is not None else None
        a.B, a.D, a.T, a.flags, a.clamp_eps = B, D, T, (_lib.UPD_DLIM if ode else 0), ce
        a.seed, a.sample_offset = seed, offset
        hist = [x.clone()] if history else []
        pbar = None
        if progress:
            from tqdm import tqdm
            pbar = tqdm(total=steps)
        for i in range(steps):
            t_dev.fill_(T - 1 - i)
            eps = model(x, torch.ones(B, device=dev) * ts[i]).contiguous().float()      # sampler.py:233
            z = self._streams().randn(shape).to(dev) if (host and not ode) else None
            a.x_dev, a.eps_dev, a.z_dev = x.data_ptr(), eps.data_ptr(), z.data_ptr() if z is not None else None
            _lib.check(L.dlpm_lim_update_f32(C.byref(a), st))
            if history:
                hist.append(x.clone())
            if pbar:
                pbar.update(1)
        if pbar:
            pbar.close()
        return (x, torch.stack(hist)) if history else x

    def _loop(self, model, shape, flags, eta, noise, denoised_fn, model_kwargs, history, progress):
        from .unet import UNetModel
        from .mlp import MLPModel
        if hasattr(model, 'eval'):
            model.eval()
        clamp_a = self.dlpm.gen_a.kwargs.get('clamp_a')
        clamp_eps = self.dlpm.gen_eps.kwargs.get('clamp_eps')
        native = isinstance(model, (UNetModel, MLPModel)) and self.rescale_timesteps and denoised_fn is None and not model_kwargs
        with torch.inference_mode():
            if native:
                out = self._run_native(model, list(shape), flags, eta, clamp_a, clamp_eps, noise, history, progress)
            else:
                out = self._run_callable(model, list(shape), flags, eta, clamp_a, clamp_eps, noise, history, progress,
                                         denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        if self._dataset is not None:
            self._dataset['next'] += shape[0]
        else:
            self.calls += 1
        return out

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=False, denoised_fn=None, model_kwargs=None, progress=False,
                      get_sample_history=False):
        """GenerativeLevyProcess.p_sample_loop (:241-289): the stochastic loop with the arguments `sample()` does not pass on --
        `noise` (x_T), `denoised_fn`, `model_kwargs`.  Clamps are the generators' current ones (set by the last sample())."""
        assert self.device is not None
        assert isinstance(shape, (tuple, list))
        return self._loop(model, shape, _lib.UPD_CLIP if clip_denoised else 0, 0.0, noise, denoised_fn, model_kwargs,
                          get_sample_history, progress)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=False, denoised_fn=None, model_kwargs=None, progress=False,
                         eta=0.0, get_sample_history=False):
        """GenerativeLevyProcess.ddim_sample_loop (:364-403): the DLIM loop."""
        assert self.device is not None
        assert isinstance(shape, (tuple, list))
        return self._loop(model, shape, _lib.UPD_DLIM | (_lib.UPD_CLIP if clip_denoised else 0), eta, noise, denoised_fn,
                          model_kwargs, get_sample_history, progress)

    # -------------------------------------------------------------------------------- BEM: SAMPLING
    def sample(self, models, shape, reverse_steps, time_spacing=None, initial_data=None, clip_denoised=False,
               deterministic=False, dlim_eta=1.0, print_progression=False, get_sample_history=False, clamp_a=None,
               clamp_eps=None):
        """GenerativeLevyProcess.sample: dlpm/methods/GenerativeLevyProcess.py:512-569."""
        from .unet import UNetModel
        from .mlp import MLPModel
        self.dlpm.gen_a.setParams(clamp_a=clamp_a)          # stateful, as in the reference (:526-527)
        self.dlpm.gen_eps.setParams(clamp_eps=clamp_eps)
        model = models['default']
        assert time_spacing is None, 'Specific time spacing is not yet supported for diffusion reverse sampling'
        if self.reverse_steps != reverse_steps:
            assert self.rescale_timesteps, 'Rescaling only works when rescale_timesteps is True'
            self.dlpm.rescale_diffusion(reverse_steps, time_spacing=time_spacing)
            self.reverse_steps = reverse_steps            # (the reference never restores it: SURVEY.md 3.4)
        if hasattr(model, 'eval'):
            model.eval()
        shape_arg = list(shape)
        shape = list(initial_data.shape) if (deterministic and initial_data is not None) else list(shape)
        noise = initial_data if deterministic else None
        flags = (_lib.UPD_DLIM if deterministic else 0) | (_lib.UPD_CLIP if clip_denoised else 0)
        eta = dlim_eta if deterministic else 0.0
        native = isinstance(model, (UNetModel, MLPModel)) and self.rescale_timesteps
        run = self._run_native if native else self._run_callable
        with torch.inference_mode():
            if self.LIM:
                # lim_sample (GenerativeLevyProcess.py:454-507): `deterministic` selects the ODE; clip_denoised and
                # initial_data are accepted and ignored, as in the reference
                shape, flags = shape_arg, _lib.SMP_LIM | (_lib.UPD_DLIM if deterministic else 0)
                if native:
                    out = self._run_native(model, shape, flags, 0.0, None, clamp_eps, None, get_sample_history, print_progression)
                else:
                    out = self._run_callable_lim(model, shape, flags, clamp_eps, get_sample_history, print_progression)
            else:
                out = run(model, shape, flags, eta, clamp_a, clamp_eps, noise, get_sample_history, print_progression)
        if self._dataset is not None:
            self._dataset['next'] += shape[0]
        else:
            self.calls += 1
        return out

    def training_losses(self, *a, **k):
        raise NotImplementedError('training is outside the sampling hot path this build covers (SURVEY.md 2b)')


def init_method_by_parameter(p, **kw):
    """dlpm/dlpm_experiment.py:103-131 for method == 'dlpm'."""
    m = p['method']
    assert m in ['dlpm', 'lim'], "chosen_gen_model should be in ['dlpm', 'lim'], got {}".format(m)
    q = p[m]
    return GenerativeLevyProcess(alpha=q['alpha'], device=p['device'], reverse_steps=q['reverse_steps'],
                                 model_mean_type=q.get('mean_predict', 'EPSILON'),
                                 model_var_type=q.get('var_predict', 'FIXED'),
                                 rescale_timesteps=q['rescale_timesteps'], isotropic=q['isotropic'], LIM=(m == 'lim'),
                                 scale=q.get('scale', 'scale_preserving'), input_scaling=q.get('input_scaling', False),
                                 **kw)
