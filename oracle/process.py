"""Oracle: DLPM noise schedule, Sigma recursion and the x_{t-1} update (CPU, torch fp32).

TEST INFRASTRUCTURE ONLY.  Restates dlpm/methods/dlpm.py with per-sample scalars
([T,B] tables) instead of the reference's fully expanded [T,B,C,H,W] tensors; for
isotropic noise the arithmetic per element is identical (same fp32 op order).  Non-isotropic
noise passes [T,B,C,H,W] tables through the same functions.
"""
import math

import numpy as np
import torch


def schedule(T, alpha, scale='scale_preserving'):
    """gammas, bargammas, sigmas, barsigmas: dlpm/methods/dlpm.py:114-156 (linear time spacing :103-105)."""
    ts = torch.arange(0, T, dtype=torch.float32)
    if scale == 'scale_preserving':
        f = torch.cos((ts / T + 0.008) / 1.008 * torch.pi / 2) ** 2
        abar = f / f[0]
        beta = 1 - abar / torch.cat([abar[:1], abar[:-1]])
        g = (1 - beta) ** (1 / alpha)
        bg = torch.cumprod(g, dim=0)
        s = (1 - g ** alpha) ** (1 / alpha)
        bs = (1 - bg ** alpha) ** (1 / alpha)
        return g, bg, s, bs
    if scale == 'scale_exploding':                                                  # dlpm.py:134-149
        sigma_min, sigma_max, rho = 0.002, 80, 7
        g, bg = torch.ones_like(ts), torch.ones_like(ts)
        bs = (sigma_min ** (1 / rho) + (ts / (T - 1)) * (sigma_max ** (1 / rho) - sigma_min ** (1 / rho))) ** rho
        bsa = bs ** alpha
        sa = torch.ones_like(bs) * bsa[0]
        for i in range(1, T):
            sa[i] = bsa[i] - torch.sum(sa[:i])
        return g, bg, sa ** (1 / alpha), bs
    raise NotImplementedError(scale)


def sigma_table(A, g, s):
    """Sigma_0 = s_0^2 A_0 ; Sigma_t = s_t^2 A_t + g_t^2 Sigma_{t-1}  (dlpm.py:230-239).  A: [T,B]."""
    rows = [s[0] ** 2 * A[0]]
    for t in range(1, A.shape[0]):
        rows.append(s[t] ** 2 * A[t] + g[t] ** 2 * rows[-1])
    return torch.stack(rows)


def gamma_var(t, Sig, g):
    """Gamma_t = 1 - g_t^2 Sigma_{t-1} / Sigma_t (dlpm.py:250-254); var = Gamma_t Sigma_{t-1} (:256-257)."""
    Gam = 1 - (g[t] ** 2 * Sig[t - 1]) / Sig[t]
    return Gam, Gam * Sig[t - 1]


def _b(v, x):
    """per-sample [B] -> broadcastable over x; per-element tables (non-isotropic noise) pass through."""
    if v.dim() == x.dim():
        return v
    return v.view(-1, *([1] * (x.dim() - 1)))


def dlpm_step(x, eps, t, Sig, g, bs, z):
    """mean = (x - bs_t Gamma_t eps)/g_t (dlpm.py:272-278); x' = mean + 1[t!=1] sqrt(var) z
    (GenerativeLevyProcess.py:236-238)."""
    Gam, var = gamma_var(t, Sig, g)
    mean = (x - (bs[t] * _b(Gam, x)) * eps) / g[t]
    mask = 0.0 if t == 1 else 1.0
    return mean + (mask * torch.sqrt(_b(var, x))) * z, mean, var


def dlim_step(x, eps, t, g, bs, eta=0.0, alpha=None, A=None, z=None):
    """dlpm.py:281-297.  eta > 0 uses the per-sample A[t] (the reference's A[t] indexing with a
    [B] tensor is shape-broken, see tools/make_fixtures.py f5 note); the diagonal is restated."""
    if eta == 0.0:
        return (x - bs[t] * eps) / g[t] + bs[t - 1] * eps
    sig = eta * bs[t - 1]
    out = (x - bs[t] * eps) / g[t]
    out = out + (bs[t - 1] ** alpha - sig ** alpha) ** (1 / alpha) * eps
    mask = 0.0 if t == 1 else 1.0
    var = mask * sig ** 2 * A[t]
    return out + torch.sqrt(_b(var, x)) * z


def clipped_eps(x, eps, t, bg, bs):
    """clip_denoised branch: xstart = ((x - eps bs_t)/bg_t).clamp(-1,1); eps' = (x - xstart bg_t)/bs_t
    (GenerativeLevyProcess.py:186-207, dlpm.py:191-202)."""
    xs = ((x - eps * bs[t]) / bg[t]).clamp(-1, 1)
    return (x - xs * bg[t]) / bs[t]


def model_eps(x, out, t, mean_type, clip, denoised_fn, g, bg, bs, Sig=None, A=None):
    """p_mean_variance between the model call and anterior_mean_variance_* (GenerativeLevyProcess.py:182-207):
    EPSILON without clipping bypasses everything (a denoised_fn is then never called); otherwise the output becomes x_0
    per mean type, goes through process_xstart (denoised_fn, then clamp) and back to eps (predict_eps, dlpm.py:198-202).
    Z / PREVIOUS_X index A[t] / Sigmas[t] with the [B] tensor t in the reference, which raises for every B (shape
    [B,B,...] against [B,...]); the per-sample reading -- the same helpers called with an integer t
    (predict_eps_from_m_tilde, dlpm.py:204-209) -- is restated here and pinned by tests/golden/f12_mean_types.npz."""
    if mean_type == 'EPSILON' and not clip:
        return out
    if mean_type == 'START_X':
        xs = out
    else:
        if mean_type == 'EPSILON':
            e = out
        elif mean_type == 'Z':
            e = torch.sqrt(_b(A[t], x)) * out
        elif mean_type == 'PREVIOUS_X':
            Gam, _ = gamma_var(t, Sig, g)
            e = (x - out * g[t]) / (bs[t] * _b(Gam, x))
        else:
            raise NotImplementedError(mean_type)
        xs = (x - e * bs[t]) / bg[t]
    if denoised_fn is not None:
        xs = denoised_fn(xs)
    if clip:
        xs = xs.clamp(-1, 1)
    return (x - xs * bg[t]) / bs[t]


def generation_postprocess(x, is_image):
    """bem/GenerationManager.py:50-63 + bem/datasets/__init__.py:108-109."""
    c = 1.0 if is_image else 6.0
    y = x.clamp(-c, c).cpu()
    return (y + 1) / 2 if is_image else y
