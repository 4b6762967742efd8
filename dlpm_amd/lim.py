"""LIM (continuous-time Levy-Ito model) sampling schedule: host-side mirror of VPSDE
(dlpm/methods/LIM/functions/sde.py:5-49, cosine schedule) and of the per-step scalars LIM_sampler derives
from it (dlpm/methods/LIM/functions/sampler.py:85-152, 217-218).

Like the DLPM schedule (process.py), the [steps] tables are formed on the host with the reference's own fp32 torch
op sequence so they agree bit for bit with what the reference computes per step; libdlpm_amd's
`dlpm_lim_tables_f32` is the native alternative (`native=True`, within a few ulp).
"""
import math

import numpy as np
import torch

from . import _lib


class VPSDE:
    def __init__(self, alpha, schedule='cosine', T=0.9946):
        if schedule != 'cosine':
            raise NotImplementedError("GenerativeLevyProcess builds VPSDE(alpha, 'cosine') only "
                                      '(dlpm/methods/GenerativeLevyProcess.py:80)')
        self.alpha, self.schedule, self.T = alpha, schedule, T
        self.cosine_s = 0.008
        self.cosine_log_alpha_0 = math.log(math.cos(self.cosine_s / (1. + self.cosine_s) * math.pi / 2.))

    def beta(self, t):
        return math.pi / 2 * self.alpha / (self.cosine_s + 1) * torch.tan((t + self.cosine_s) / (1 + self.cosine_s) * math.pi / 2)

    def marginal_log_mean_coeff(self, t):
        return torch.log(torch.cos((t + self.cosine_s) / (1. + self.cosine_s) * math.pi / 2.)) - self.cosine_log_alpha_0

    def diffusion_coeff(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.pow(1. - torch.exp(self.marginal_log_mean_coeff(t) * self.alpha), 1 / self.alpha)


def lim_tables(sde, steps, ode, native=False):
    """(ts[steps+1], tmp, cx, cs, cn [steps]) as contiguous fp32 CPU tensors."""
    if native:
        out = [np.empty(steps + 1, np.float32)] + [np.empty(steps, np.float32) for _ in range(4)]
        _lib.check(_lib.lib().dlpm_lim_tables_f32(float(sde.alpha), steps, int(bool(ode)), *[o.ctypes.data for o in out]))
        return tuple(torch.from_numpy(o) for o in out)
    al = sde.alpha
    ts = torch.linspace(sde.T, 1e-5, steps + 1)                                     # sampler.py:218
    s, t = ts[:-1], ts[1:]
    beta_step = sde.beta(s) * (s - t)
    if al == 2:
        tmp = torch.pow(sde.marginal_std(s) + 1e-5, -(al - 1))
        cx = 1 + beta_step / al
        cs = beta_step / 2 if ode else beta_step
        cn = torch.zeros_like(cx) if ode else torch.pow(beta_step, 1 / al)
    else:
        tmp = torch.pow(sde.marginal_std(s), -(al - 1))
        if ode:
            cx = sde.diffusion_coeff(t) * torch.pow(sde.diffusion_coeff(s), -1)
            a = sde.diffusion_coeff(t) * torch.pow(sde.diffusion_coeff(s), -1)
            cs = -al * (1 - a)
            cn = torch.zeros_like(cx)
        else:
            a = torch.exp(sde.marginal_log_mean_coeff(t) - sde.marginal_log_mean_coeff(s))
            cx = a
            cs = al ** 2 * (-1 + a)
            cn = torch.pow(-1 + torch.pow(a, al), 1 / al)
    return tuple(v.to(torch.float32).contiguous() for v in (ts, tmp, cx, cs, cn))
