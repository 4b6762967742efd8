"""Oracle: the LIM sampler (continuous-time Levy-Ito model, `method: lim`) -- TEST INFRASTRUCTURE ONLY.

Restates VPSDE (dlpm/methods/LIM/functions/sde.py:5-49, cosine schedule, T = 0.9946) and LIM_sampler's SDE / ODE
score updates (dlpm/methods/LIM/functions/sampler.py:85-181, 217-258) as called by
GenerativeLevyProcess.lim_sample (dlpm/methods/GenerativeLevyProcess.py:454-507).

RNG consumption of one sample() call (streams N = numpy/scipy, P = torch, SURVEY.md 8c-bis conventions):
  x_0 = gen_eps.generate(shape):  N: B uniforms + B exponentials (unclamped a), P: B*D normals; clamp_eps
  each SDE step, AFTER the model call: the same again (a fresh per-sample a every step)
  ODE steps draw nothing.
"""
import math

import torch

from .process import _b


class VPSDE:
    def __init__(self, alpha, T=0.9946):
        self.alpha, self.T, self.cosine_s = alpha, T, 0.008
        self.log_alpha_0 = math.log(math.cos(self.cosine_s / (1.0 + self.cosine_s) * math.pi / 2.0))

    def beta(self, t):                                                              # sde.py:27-33
        return math.pi / 2 * self.alpha / (self.cosine_s + 1) * torch.tan((t + self.cosine_s) / (1 + self.cosine_s) * math.pi / 2)

    def marginal_log_mean_coeff(self, t):                                           # sde.py:35-41
        return torch.log(torch.cos((t + self.cosine_s) / (1.0 + self.cosine_s) * math.pi / 2.0)) - self.log_alpha_0

    def diffusion_coeff(self, t):                                                   # sde.py:43-44
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):                                                      # sde.py:46-47
        return torch.pow(1.0 - torch.exp(self.marginal_log_mean_coeff(t) * self.alpha), 1 / self.alpha)


def timesteps(sde, steps):
    return torch.linspace(sde.T, 1e-5, steps + 1)                                   # sampler.py:218


def step_coefficients(sde, s, t, ode):
    """(tmp, x_coeff, score_coeff, noise_coeff) of one update from time s to time t (vectors allowed)."""
    al = sde.alpha
    beta_step = sde.beta(s) * (s - t)
    if al == 2:
        tmp = torch.pow(sde.marginal_std(s) + 1e-5, -(al - 1))                      # sampler.py:95-96,133-134
        x_coeff = 1 + beta_step / al
        if ode:
            return tmp, x_coeff, beta_step / 2, None                                # :99-100
        return tmp, x_coeff, beta_step, torch.pow(beta_step, 1 / al)               # :138-140
    tmp = torch.pow(sde.marginal_std(s), -(al - 1))                                 # :86,120
    if ode:
        x_coeff = sde.diffusion_coeff(t) * torch.pow(sde.diffusion_coeff(s), -1)    # :92
        a = sde.diffusion_coeff(t) * torch.pow(sde.diffusion_coeff(s), -1)          # :108
        return tmp, x_coeff, -al * (1 - a), None                                    # :109
    a = torch.exp(sde.marginal_log_mean_coeff(t) - sde.marginal_log_mean_coeff(s))  # :126
    return tmp, a, al ** 2 * (-1 + a), torch.pow(-1 + torch.pow(a, al), 1 / al)     # :127-129,152


def gen_eps(streams, alpha, shape, clamp_eps, isotropic=True):
    """gen_sas: bem/datasets/Distributions.py:57-73 (own unclamped a; alpha == 2 gives a = 2)."""
    n = shape[0] if isotropic else int(torch.tensor(shape).prod())
    a = streams.skewed_levy(alpha, n, None)
    a = _b(a, torch.empty(shape)) if isotropic else a.reshape(shape)
    e = torch.sqrt(a) * streams.randn(shape)
    if clamp_eps is not None:
        e = torch.clamp(e, -clamp_eps, clamp_eps)
    return e


def sample(model, shape, steps, alpha, streams, ode=False, clamp_eps=None, get_sample_history=False, trace=None):
    sde = VPSDE(alpha)
    B = shape[0]
    x = gen_eps(streams, alpha, shape, clamp_eps)                                   # GenerativeLevyProcess.py:464
    ts = timesteps(sde, steps)
    hist = [x]
    if trace is not None:
        trace.update(x0=x, noise=[])
    for i in range(steps):
        s = torch.ones(B) * ts[i]
        t = torch.ones(B) * ts[i + 1]
        tmp, cx, cs, cn = step_coefficients(sde, s, t, ode)
        score = model(x, s) * _b(tmp, x)
        if ode:
            x = _b(cx, x) * x + _b(cs, x) * score
        else:
            e = streams.randn(shape) if alpha == 2 else gen_eps(streams, alpha, shape, clamp_eps)   # sampler.py:135,143
            if trace is not None:
                trace['noise'].append(e)
            x = _b(cx, x) * x + _b(cs, x) * score + _b(cn, x) * e
        hist.append(x)
    return (x, torch.stack(hist)) if get_sample_history else x
