"""DLPM: the discrete-time heavy-tailed forward/backward process (sampling half).

Host-side mirror of dlpm/methods/dlpm.py:56-297 for the pieces the reverse loop needs: the noise
schedule vectors, `rescale_diffusion`, and the stateful clamp parameters of the two noise generators
(`gen_a`, `gen_eps`: bem/datasets/Data.py:17-89).  The arithmetic of the loop itself (A draws,
Sigma recursion, x_{t-1} update) lives in libdlpm_amd and keeps [T,B] scalars instead of the
reference's [T,B,C,H,W] tensors (non-isotropic noise, `isotropic=False`, is the one case that really
needs a value per element: [T,B,D] tables).
"""
import numpy as np
import torch

from . import _lib


class NoiseParams:
    """The kwargs store of bem.datasets.Data.Generator (setParams mutates defaults: Data.py:60-65)."""

    def __init__(self, operation, **kwargs):
        if operation not in ('skewed_levy', 'sas'):
            raise Exception('Unknown distribution to sample from. Available distributions: {}'.format(
                ['skewed_levy', 'sas']))
        self.operation = operation
        self.kwargs = kwargs

    def setParams(self, *args, **kwargs):
        if args == () and kwargs == {}:
            raise Exception('Given void parameters')
        self.kwargs.update(kwargs)

    def get(self, name):
        return self.kwargs.get(name)


class DLPM:
    def __init__(self, alpha, device, diffusion_steps, time_spacing='linear', isotropic=True, clamp_a=None,
                 clamp_eps=None, scale='scale_preserving', native_schedule=False):
        if alpha > 2.0 or alpha <= 0.0:
            raise Exception('Wrong value of alpha ({}) for skewed levy r.v generation'.format(alpha))
        assert scale in ('scale_preserving', 'scale_exploding'), 'Unknown scale'
        self.alpha, self.device, self.time_spacing, self.isotropic, self.scale = alpha, device, time_spacing, isotropic, scale
        self.native_schedule = native_schedule
        self.gen_a = NoiseParams('skewed_levy', alpha=alpha, device=device, isotropic=isotropic, clamp_a=clamp_a)
        self.gen_eps = NoiseParams('sas', alpha=alpha, device=device, isotropic=isotropic, clamp_eps=clamp_eps)
        self._set_schedule(diffusion_steps, scale)

    def get_timesteps(self, steps):
        if self.time_spacing == 'linear':
            return torch.arange(0, steps, dtype=torch.float32)
        if self.time_spacing == 'quadratic':
            return steps * (torch.arange(0, steps, dtype=torch.float32) / steps) ** 2
        raise NotImplementedError(self.time_spacing)

    def gen_noise_schedule(self, diffusion_steps, scale='scale_preserving'):
        """[T] fp32 vectors gammas, bargammas, sigmas, barsigmas (dlpm.py:114-156).

        Several entries are differences of nearly equal fp32 numbers, so bit-level agreement with
        the reference needs the reference's own elementwise kernels: by default the T-vector is
        formed on the host with the same torch fp32 op sequence (host-side table setup, as in the
        reference, which also builds it on the CPU and copies it over, dlpm.py:76-77).
        `native_schedule=True` uses libdlpm_amd's dlpm_schedule_f32 instead (<= 1 ulp on gammas).
        """
        T = diffusion_steps
        if self.native_schedule:
            out = [np.empty(T, np.float32) for _ in range(4)]
            fn = _lib.lib().dlpm_schedule_f32 if scale == 'scale_preserving' else _lib.lib().dlpm_schedule_exploding_f32
            _lib.check(fn(T, float(self.alpha), *[o.ctypes.data for o in out]))
            return tuple(torch.from_numpy(o) for o in out)
        if scale == 'scale_exploding':                                              # dlpm.py:134-149
            ts = self.get_timesteps(T)
            sigma_min, sigma_max, rho = 0.002, 80, 7
            g, bg = torch.ones_like(ts), torch.ones_like(ts)
            bsig = (sigma_min ** (1 / rho) + (ts / (T - 1)) * (sigma_max ** (1 / rho) - sigma_min ** (1 / rho))) ** rho
            bsa = bsig ** self.alpha
            sa = torch.ones_like(bsig) * bsa[0]
            for i in range(1, len(bsig)):                 # the reference's running sums, summation order included
                sa[i] = bsa[i] - torch.sum(sa[:i])
            return g, bg, sa ** (1 / self.alpha), bsig
        assert scale == 'scale_preserving', 'Unknown scale'
        s = 0.008
        ts = self.get_timesteps(T)
        f = torch.cos((ts / T + s) / (1 + s) * torch.pi / 2) ** 2
        abar = f / f[0]
        beta = 1 - abar / torch.cat([abar[0:1], abar[0:-1]])
        g = (1 - beta) ** (1 / self.alpha)
        bg = torch.cumprod(g, dim=0)
        sig = (1 - g ** self.alpha) ** (1 / self.alpha)
        bsig = (1 - bg ** self.alpha) ** (1 / self.alpha)
        return g, bg, sig, bsig

    def _set_schedule(self, T, scale='scale_preserving'):
        self.host_schedule = tuple(v.contiguous() for v in self.gen_noise_schedule(T, scale))
        self.gammas, self.bargammas, self.sigmas, self.barsigmas = (v.to(self.device) for v in self.host_schedule)
        self.diffusion_steps = T

    def rescale_diffusion(self, diffusion_steps, time_spacing=None):
        assert isinstance(diffusion_steps, int), 'Diffusion steps must be an integer'
        if time_spacing is not None:
            self.time_spacing = time_spacing
        # as in the reference (dlpm.py:182-183) the regenerated schedule is ALWAYS scale_preserving, whatever
        # self.scale says: a scale_exploding process sampled with reverse_steps != its own switches schedule
        self._set_schedule(diffusion_steps)
