"""Oracle: the reverse-time sampling loop (CPU) -- TEST INFRASTRUCTURE ONLY.

Restates GenerativeLevyProcess.sample -> p_sample_loop / ddim_sample_loop
(dlpm/methods/GenerativeLevyProcess.py:241-330, 365-452, 512-569) with the RNG
consumption order of SURVEY.md 8c-bis:

  1. A[k], k = 0..T-1     stream N: B uniforms then B exponentials per k, clamp_a applied
                          (non-isotropic, Distributions.py:47-48: B*D of each, A[k] has the state's shape)
  2. x_T = bs[T-1] * sqrt(a) * randn, a unclamped (Distributions.py:63-65), clamp_eps applied
  3. i = T-1..1           stream P: B*D normals per step, also at i == 1 (masked)

`model(x, t)` is any callable taking fp32 x and t = i/T floats.
"""
import numpy as np
import torch

from . import process as P
from .rng import MT


class Streams:
    """The two process-global generators of the reference, as oracle MT19937 streams."""

    def __init__(self, np_seed=0, torch_seed=0):
        self.N = MT(np_seed)
        self.P = MT(torch_seed)

    def skewed_levy(self, alpha, B, clamp_a=None):
        # gen_skewed_levy, isotropic: bem/datasets/Distributions.py:33-51
        if alpha == 2.0:
            return torch.full((B,), 2.0)
        a = torch.tensor(self.N.skewed_levy(alpha, B), dtype=torch.float32)
        if clamp_a is not None:
            a = torch.clamp(a, 0.0, clamp_a)
        return a

    def randn(self, shape):
        n = int(np.prod(shape))
        return torch.from_numpy(self.P.torch_randn(n)).reshape(list(shape))


def sample(model, shape, T, alpha, streams, deterministic=False, dlim_eta=0.0, clip_denoised=False,
           clamp_a=None, clamp_eps=None, get_sample_history=False, trace=None, isotropic=True,
           scale='scale_preserving', input_scaling=False, mean_type='EPSILON', denoised_fn=None, model_kwargs=None, noise=None):
    B = shape[0]
    model_kwargs = model_kwargs or {}
    n = B if isotropic else int(np.prod(shape))
    g, bg, s, bs = P.schedule(T, alpha, scale)
    A = torch.stack([streams.skewed_levy(alpha, n, clamp_a) for _ in range(T)])    # dlpm.py:226-227
    if not isotropic:
        A = A.reshape([T] + list(shape))
    Sig = P.sigma_table(A, g, s)                                                    # dlpm.py:230-239
    # x_T: GenerativeLevyProcess.py:313 -> gen_sas (own unclamped a, then randn, then clamp_eps)
    if noise is not None:                                                           # :311-312: a given x_T draws nothing
        x = noise
    else:
        a0 = streams.skewed_levy(alpha, n, None)
        a0 = P._b(a0, torch.empty(shape)) if isotropic else a0.reshape(shape)
        e = torch.sqrt(a0) * streams.randn(shape)
        if clamp_eps is not None:
            e = torch.clamp(e, -clamp_eps, clamp_eps)
        x = bs[-1] * e
    hist = [x]
    if trace is not None:
        trace.update(A=A, Sigmas=Sig, xT=x, z=[])
    for i in range(T - 1, 0, -1):
        t = torch.full((B,), i, dtype=torch.int64)
        xin = x * (1 / (1 + bs[i])) if input_scaling else x                         # :176-179 (scale_exploding only)
        eps = model(xin, t.float() * (1.0 / T), **model_kwargs)                     # :92-96,180
        eps = P.model_eps(x, eps, i, mean_type, clip_denoised, denoised_fn, g, bg, bs, Sig=Sig, A=A)   # :182-207
        if deterministic:
            z = streams.randn(shape) if dlim_eta != 0.0 else None
            x = P.dlim_step(x, eps, i, g, bs, eta=dlim_eta, alpha=alpha, A=A, z=z)
        else:
            z = streams.randn(shape)
            if trace is not None:
                trace['z'].append(z)
            x, _, _ = P.dlpm_step(x, eps, i, Sig, g, bs, z)
        hist.append(x)
    if get_sample_history:
        return x, torch.stack(hist)
    return x


def sample_with_tables(model, shape, T, alpha, A, xT, zs, clip_denoised=False):
    """Same loop with injected noise (A [T,B], x_T, z per step): the checker for the HIP sampler
    when both are fed identical noise."""
    g, bg, s, bs = P.schedule(T, alpha)
    Sig = P.sigma_table(A, g, s)
    x = xT
    k = 0
    for i in range(T - 1, 0, -1):
        t = torch.full((shape[0],), float(i)) * (1.0 / T)
        eps = model(x, t)
        if clip_denoised:
            eps = P.clipped_eps(x, eps, i, bg, bs)
        x, _, _ = P.dlpm_step(x, eps, i, Sig, g, bs, zs[k])
        k += 1
    return x


# ------------------------------------------------------------------------------------------------------------------
# The reference's own data layout, for the CPU baseline of bench.py (SURVEY.md 8d "reference-faithful" variant): A and
# Sigma as full-size [T,B,C,H,W] tensors, the schedule re-broadcast to [T,C,H,W] by `repeat` TWICE per step (the
# update_constants cache compares a tuple of tensors' shape with a torch.Size and misses every time it is asked for a
# different shape: compute_Gamma_t asks with Sigma's shape, anterior_mean_variance_dlpm with x's -- dlpm.py:171-174,
# 250-254, 272-278), full-size element-wise arithmetic.  Bit-identical to the scalar-table functions above.
# ------------------------------------------------------------------------------------------------------------------
def match_last_dims(data, size):
    """dlpm.py:47-51."""
    for _ in range(len(size) - 1):
        data = data.unsqueeze(-1)
    return data.repeat(1, *(size[1:]))


def full_size_tables(A_tb, shape, g, s):
    """sample_A's expand (Distributions.py:45-46: one draw per sample, expanded and copied to the state's shape) and
    compute_Sigmas (dlpm.py:230-239) over the full-size tensors: T sequential passes over [B,C,H,W]."""
    T = A_tb.shape[0]
    A = torch.stack([A_tb[k].view(-1, *([1] * (len(shape) - 1))).expand(shape).clone() for k in range(T)])
    G = match_last_dims(g, shape)
    S = match_last_dims(s, shape)
    Sig = [S[0] ** 2 * A[0]]
    for t in range(1, T):
        Sig.append(S[t] ** 2 * A[t] + G[t] ** 2 * Sig[-1])
    return A, torch.stack(Sig)


def full_size_step(x, eps, i, Sig, g, bs, z):
    """p_mean_variance's tail + p_sample with the reference's tensors (GenerativeLevyProcess.py:208-239, dlpm.py:250-278)."""
    shape = list(x.shape)
    G1 = match_last_dims(g, shape)                                   # compute_Gamma_t -> update_constants (miss)
    Gam = 1 - (G1[i] ** 2 * Sig[i - 1]) / Sig[i]
    G2, BS = match_last_dims(g, shape), match_last_dims(bs, shape)   # anterior_mean_variance_dlpm -> update_constants (miss)
    t = torch.full((shape[0],), i, dtype=torch.int64)
    mean = (x - BS[t] * Gam * eps) / G2[t]
    var = Gam * Sig[i - 1]
    mask = (t != 1).float().view(-1, *([1] * (len(shape) - 1)))
    return mean + mask * torch.sqrt(var) * z
