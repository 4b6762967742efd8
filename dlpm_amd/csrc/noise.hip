// noise.hip -- everything on the sampling path that is not the score network:
//   * Philox4x32-10 counter-based noise keyed by the GLOBAL sample index (sharding-invariant)
//   * device Chambers-Mallows-Stuck draws of the skewed alpha/2-stable a_t  (dlpm.py:226-227)
//   * x_T initialisation                                                     (GenerativeLevyProcess.py:313)
//   * Sigma recursion -> per-(t, sample) coefficient tables                  (dlpm.py:230-257)
//   * the fused x_{t-1} update                                               (dlpm.py:272-297, GLP.py:225-239)
// All kernels are HBM/latency bound: coalesced 16-byte accesses over the flattened (B, D) state,
// one coefficient pair per sample fetched through the scalar/L1 path.
#include "common.h"
#include "philox.h"

using namespace dlpm;

namespace {

constexpr int kPurposeA = 1;       // rows of A[T,B]
constexpr int kPurposeInitA = 2;   // the unclamped a of gen_sas
constexpr int kPurposeInitZ = 3;   // x_T normals
constexpr int kPurposeAElem = 5;     // non-isotropic: A[T,B,D], counter = (sample, element, purpose | t << 8)
constexpr int kPurposeInitAElem = 6; // non-isotropic: the unclamped per-element a of gen_sas

// CMS in fp64, S1 parameterisation, beta = 1, stability a = alpha/2, as scipy's _rvs_Z1 "otherwise"
// branch; U uniform(0,1), W standard exponential.
__device__ inline double cms_skewed(double a, double zeta, double th0, double scale, double U, double W) {
    const double pi = 3.141592653589793;
    double th = U * pi + (-pi / 2.0);
    double ath = a * th, c = cos(th), tg = tan(th);
    double lead = W / (c / tan(a * (th0 + th)) + sin(th));
    double core = (cos(ath) + sin(ath) * tg - zeta * (sin(ath) - cos(ath) * tg)) / W;
    return lead * pow(core, 1.0 / a) * scale;
}

__device__ inline float draw_skewed(uint64_t seed, uint64_t gidx, uint32_t row, uint32_t purpose, double a,
                                    double zeta, double th0, double scale, uint32_t step = 0) {
    uint4 r = philox4x32_10(make_uint4((uint32_t)gidx, (uint32_t)(gidx >> 32), row, purpose | (step << 8)), seed);
    // 53-bit uniforms in (0,1): never 0 or 1, so th stays inside (-pi/2, pi/2) and W is finite
    double U = ((double)(((uint64_t)r.x << 21) ^ (r.y >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
    double V = ((double)(((uint64_t)r.z << 21) ^ (r.w >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
    return (float)cms_skewed(a, zeta, th0, scale, U, -log(V));
}

__global__ void k_skewed_levy(float *A, int T, int64_t B, double alpha, float clamp_a, int has_clamp, uint64_t seed,
                              int64_t off) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)T * B) return;
    int t = (int)(i / B);
    int64_t b = i - (int64_t)t * B;
    float v;
    if (alpha == 2.0) {
        v = 2.0f;
    } else {
        const double pi = 3.141592653589793;
        double a = alpha * 0.5, zeta = tan(pi * a * 0.5), th0 = atan(zeta) / a;
        double scale = 2.0 * pow(cos(pi * alpha * 0.25), 2.0 / alpha);
        v = draw_skewed(seed, (uint64_t)(off + b), (uint32_t)t, kPurposeA, a, zeta, th0, scale);
        if (has_clamp) v = fminf(fmaxf(v, 0.0f), clamp_a);
    }
    A[i] = v;
}

// x_T[b, :] = bs_last * clamp(sqrt(a0[b]) * z)
__global__ void k_init_state(float *x, int64_t B, int64_t D, double alpha, float clamp_eps, int has_clamp,
                             float bs_last, uint64_t seed, int64_t off) {
    int64_t nq = (D + 3) / 4;  // quads per sample
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nq) return;
    int64_t b = i / nq, q = i - b * nq;
    float a0;
    if (alpha == 2.0) {
        a0 = 2.0f;
    } else {
        const double pi = 3.141592653589793;
        double a = alpha * 0.5, zeta = tan(pi * a * 0.5), th0 = atan(zeta) / a;
        double scale = 2.0 * pow(cos(pi * alpha * 0.25), 2.0 / alpha);
        a0 = draw_skewed(seed, (uint64_t)(off + b), 0u, kPurposeInitA, a, zeta, th0, scale);
    }
    float sa = sqrtf(a0);
    float4 z = philox_normal4(seed, (uint64_t)(off + b), (uint32_t)q, kPurposeInitZ, 0u);
    float zz[4] = {z.x, z.y, z.z, z.w};
    for (int j = 0; j < 4; j++) {
        int64_t e = q * 4 + j;
        if (e < D) {
            float v = sa * zz[j];
            if (has_clamp) v = fminf(fmaxf(v, -clamp_eps), clamp_eps);
            x[b * D + e] = bs_last * v;
        }
    }
}

// Non-isotropic noise (`--non_iso`): one independent skewed-Levy draw per ELEMENT (Distributions.py:47-48).
// A[T,B,D]; thread i -> (t, b, e), coalesced stores; keyed by (sample, element, t) so shards agree.
__global__ void k_skewed_levy_elem(float *A, int T, int64_t B, int64_t D, double alpha, float clamp_a, int has_clamp,
                                   uint64_t seed, int64_t off) {
    const int64_t n = (int64_t)T * B * D;
    const double pi = 3.141592653589793;
    const double a = alpha * 0.5, zeta = tan(pi * a * 0.5), th0 = atan(zeta) / a;
    const double scale = 2.0 * pow(cos(pi * alpha * 0.25), 2.0 / alpha);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tb = i / D, e = i - tb * D;
        const int t = (int)(tb / B);
        const int64_t b = tb - (int64_t)t * B;
        float v = 2.0f;
        if (alpha != 2.0) {
            v = draw_skewed(seed, (uint64_t)(off + b), (uint32_t)e, kPurposeAElem, a, zeta, th0, scale, (uint32_t)t);
            if (has_clamp) v = fminf(fmaxf(v, 0.0f), clamp_a);
        }
        A[i] = v;
    }
}

// x_T[b, e] = bs_last * clamp(sqrt(a0[b, e]) * z[b, e]) with a per-element unclamped a0
__global__ void k_init_state_elem(float *x, int64_t B, int64_t D, double alpha, float clamp_eps, int has_clamp,
                                  float bs_last, uint64_t seed, int64_t off) {
    int64_t nq = (D + 3) / 4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nq) return;
    int64_t b = i / nq, q = i - b * nq;
    const double pi = 3.141592653589793;
    const double a = alpha * 0.5, zeta = tan(pi * a * 0.5), th0 = atan(zeta) / a;
    const double scale = 2.0 * pow(cos(pi * alpha * 0.25), 2.0 / alpha);
    float4 z = philox_normal4(seed, (uint64_t)(off + b), (uint32_t)q, kPurposeInitZ, 0u);
    float zz[4] = {z.x, z.y, z.z, z.w};
    for (int j = 0; j < 4; j++) {
        int64_t e = q * 4 + j;
        if (e < D) {
            float a0 = alpha == 2.0 ? 2.0f
                                    : draw_skewed(seed, (uint64_t)(off + b), (uint32_t)e, kPurposeInitAElem, a, zeta, th0, scale);
            float v = sqrtf(a0) * zz[j];
            if (has_clamp) v = fminf(fmaxf(v, -clamp_eps), clamp_eps);
            x[b * D + e] = bs_last * v;
        }
    }
}

// One thread per sample walks t = 0..T-1 (a scan over t; T*B FMAs in total).
// c_eps may alias A (each thread reads A[t,b] before it writes c_eps[t,b]).
__global__ void k_coeff_tables(const float *A, const float *g, const float *s, const float *bs, int T, int64_t B,
                               float *c_eps, float *c_noise, float *sig_out) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float prev = (s[0] * s[0]) * A[b];  // Sigma_0 = s_0^2 A_0
    if (sig_out) sig_out[b] = prev;
    c_eps[b] = 0.f;
    c_noise[b] = 0.f;
    for (int t = 1; t < T; t++) {
        float gt = g[t], st = s[t];
        float g2 = gt * gt;
        // no contraction: the reference rounds each product (dlpm.py:238,253)
        float cur = __fadd_rn(__fmul_rn(st * st, A[(int64_t)t * B + b]), __fmul_rn(g2, prev));
        float Gam = 1.0f - __fdiv_rn(__fmul_rn(g2, prev), cur);
        float var = __fmul_rn(Gam, prev);
        c_eps[(int64_t)t * B + b] = __fmul_rn(bs[t], Gam);
        c_noise[(int64_t)t * B + b] = (t == 1) ? 0.0f : __fsqrt_rn(var);
        if (sig_out) sig_out[(int64_t)t * B + b] = cur;
        prev = cur;
    }
}

struct StepScalars {
    float g, bg, bs, bs_prev;
};

__device__ inline float clip_eps(float x, float e, const StepScalars &c) {
    // predict_xstart -> clamp -> predict_eps (dlpm.py:191-202)
    float xs = __fdiv_rn(x - __fmul_rn(e, c.bs), c.bg);
    xs = fminf(fmaxf(xs, -1.0f), 1.0f);
    return __fdiv_rn(x - __fmul_rn(xs, c.bg), c.bs);
}

// One reverse step, fused.  VEC: D % 4 == 0 -> each thread owns UNROLL float4 quads spaced a whole
// grid apart, issues all of their loads (x, eps, coefficients, optional z) before any arithmetic,
// then computes (Philox normals in registers) and stores: enough bytes in flight per CU to stream
// at HBM rate even though the whole tensor is only tens of MB.
template <bool VEC>
__global__ void __launch_bounds__(256) k_update(dlpm_update_args p) {
    constexpr int UNROLL = 4;
    if (p.key_dev) { p.seed = p.key_dev[0]; p.sample_offset = (int64_t)p.key_dev[1]; }
    const int t = *p.t_dev;
    StepScalars c{p.g_dev[t], p.bg_dev[t], p.bs_dev[t], p.bs_dev[t > 0 ? t - 1 : 0]};
    const bool dlim = p.flags & DLPM_UPD_DLIM, clip = p.flags & DLPM_UPD_CLIP;
    const int64_t D = p.D;
    const int64_t nq = VEC ? D / 4 : D;          // work items per sample
    const int64_t total = p.B * nq;
    // DLIM eta > 0 extra mean coefficient: (bs_{t-1}^alpha - (eta bs_{t-1})^alpha)^(1/alpha)
    float dl_mean = c.bs_prev, dl_sig = 0.f;
    if (dlim && p.dlim_eta != 0.0f) {
        dl_sig = p.dlim_eta * c.bs_prev;
        dl_mean = powf(powf(c.bs_prev, p.alpha) - powf(dl_sig, p.alpha), 1.0f / p.alpha);
    }
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float *hist = p.hist_pp ? *p.hist_pp : nullptr;
    if (hist) hist += (int64_t)(p.T - t) * p.B * D;
    if (VEC) {
        for (int64_t base_i = tid0; base_i < total; base_i += nthreads * UNROLL) {
            float4 x[UNROLL], e[UNROLL], z[UNROLL];
            float ce[UNROLL], cn[UNROLL];
            int64_t b[UNROLL], q[UNROLL];
            bool ok[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const int64_t i = base_i + u * nthreads;
                ok[u] = i < total;
                const int64_t ii = ok[u] ? i : tid0;          // clamp: loads stay unconditional
                b[u] = ii / nq;
                q[u] = ii - b[u] * nq;
                const int64_t off = b[u] * D + q[u] * 4;
                x[u] = *reinterpret_cast<const float4 *>(p.x_dev + off);
                e[u] = *reinterpret_cast<const float4 *>(p.eps_dev + off);
                if (p.z_dev) z[u] = *reinterpret_cast<const float4 *>(p.z_dev + off);
                if (!dlim) {
                    ce[u] = p.c_eps_dev[(int64_t)t * p.B + b[u]];
                    cn[u] = p.c_noise_dev[(int64_t)t * p.B + b[u]];
                } else {
                    ce[u] = c.bs;
                    cn[u] = (p.dlim_eta != 0.0f && t != 1) ? dl_sig * sqrtf(p.A_dev[(int64_t)t * p.B + b[u]]) : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                if (!p.z_dev)
                    z[u] = (cn[u] != 0.0f) ? philox_normal4(p.seed, (uint64_t)(p.sample_offset + b[u]), (uint32_t)q[u],
                                                            kPurposeStepZ, (uint32_t)t)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                float xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, ev[4] = {e[u].x, e[u].y, e[u].z, e[u].w};
                float zv[4] = {z[u].x, z[u].y, z[u].z, z[u].w}, o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float ee = clip ? clip_eps(xv[j], ev[j], c) : ev[j];
                    float m = __fdiv_rn(xv[j] - __fmul_rn(ce[u], ee), c.g);
                    if (dlim) m = __fadd_rn(m, __fmul_rn(dl_mean, ee));
                    o[j] = __fadd_rn(m, __fmul_rn(cn[u], zv[j]));
                }
                if (ok[u]) {
                    *reinterpret_cast<float4 *>(p.x_dev + b[u] * D + q[u] * 4) = make_float4(o[0], o[1], o[2], o[3]);
                    if (hist) *reinterpret_cast<float4 *>(hist + b[u] * D + q[u] * 4) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    } else {
        for (int64_t i = tid0; i < total; i += nthreads) {
            const int64_t b = i / nq, q = i - b * nq;
            float ce, cn;
            if (!dlim) {
                ce = p.c_eps_dev[(int64_t)t * p.B + b];
                cn = p.c_noise_dev[(int64_t)t * p.B + b];
            } else {
                ce = c.bs;
                cn = (p.dlim_eta != 0.0f && t != 1) ? dl_sig * sqrtf(p.A_dev[(int64_t)t * p.B + b]) : 0.0f;
            }
            const int64_t idx = b * D + q;
            float xv = p.x_dev[idx], ev = p.eps_dev[idx], zv;
            if (p.z_dev) zv = p.z_dev[idx];
            else if (cn == 0.0f) zv = 0.0f;
            else {
                float4 z = philox_normal4(p.seed, (uint64_t)(p.sample_offset + b), (uint32_t)(q >> 2), kPurposeStepZ, (uint32_t)t);
                float zz[4] = {z.x, z.y, z.z, z.w};
                zv = zz[q & 3];
            }
            float ee = clip ? clip_eps(xv, ev, c) : ev;
            float m = __fdiv_rn(xv - __fmul_rn(ce, ee), c.g);
            if (dlim) m = __fadd_rn(m, __fmul_rn(dl_mean, ee));
            const float o = __fadd_rn(m, __fmul_rn(cn, zv));
            p.x_dev[idx] = o;
            if (hist) hist[idx] = o;
        }
    }
}

// Non-isotropic variant (DLPM_UPD_ELEMENTWISE): c_eps / c_noise / A are [T,B,D] and are streamed like x and eps
// (20 B/element with Philox noise).  Same arithmetic as k_update, element by element.
template <bool VEC>
__global__ void __launch_bounds__(256) k_update_elem(dlpm_update_args p) {
    constexpr int W = VEC ? 4 : 1;
    if (p.key_dev) { p.seed = p.key_dev[0]; p.sample_offset = (int64_t)p.key_dev[1]; }
    const int t = *p.t_dev;
    StepScalars c{p.g_dev[t], p.bg_dev[t], p.bs_dev[t], p.bs_dev[t > 0 ? t - 1 : 0]};
    const bool dlim = p.flags & DLPM_UPD_DLIM, clip = p.flags & DLPM_UPD_CLIP;
    const int64_t D = p.D, BD = p.B * D;
    float dl_mean = c.bs_prev, dl_sig = 0.f;
    if (dlim && p.dlim_eta != 0.0f) {
        dl_sig = p.dlim_eta * c.bs_prev;
        dl_mean = powf(powf(c.bs_prev, p.alpha) - powf(dl_sig, p.alpha), 1.0f / p.alpha);
    }
    float *hist = p.hist_pp ? *p.hist_pp : nullptr;
    if (hist) hist += (int64_t)(p.T - t) * BD;
    const float *ce_t = dlim ? nullptr : p.c_eps_dev + (int64_t)t * BD;
    const float *cn_t = dlim ? nullptr : p.c_noise_dev + (int64_t)t * BD;
    const float *a_t = (dlim && p.dlim_eta != 0.0f) ? p.A_dev + (int64_t)t * BD : nullptr;
    const int64_t items = BD / W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t off = i * W;
        const int64_t b = off / D, e0 = off - b * D;
        float xv[W], ev[W], zv[W], ce[W], cn[W];
        if (VEC) {
            *reinterpret_cast<float4 *>(xv) = *reinterpret_cast<const float4 *>(p.x_dev + off);
            *reinterpret_cast<float4 *>(ev) = *reinterpret_cast<const float4 *>(p.eps_dev + off);
            if (p.z_dev) *reinterpret_cast<float4 *>(zv) = *reinterpret_cast<const float4 *>(p.z_dev + off);
            if (ce_t) {
                *reinterpret_cast<float4 *>(ce) = *reinterpret_cast<const float4 *>(ce_t + off);
                *reinterpret_cast<float4 *>(cn) = *reinterpret_cast<const float4 *>(cn_t + off);
            } else if (a_t) {
                *reinterpret_cast<float4 *>(cn) = *reinterpret_cast<const float4 *>(a_t + off);
            }
        } else {
            xv[0] = p.x_dev[off];
            ev[0] = p.eps_dev[off];
            if (p.z_dev) zv[0] = p.z_dev[off];
            if (ce_t) { ce[0] = ce_t[off]; cn[0] = cn_t[off]; }
            else if (a_t) cn[0] = a_t[off];
        }
        if (dlim) {
#pragma unroll
            for (int j = 0; j < W; j++) {
                ce[j] = c.bs;
                cn[j] = (a_t && t != 1) ? dl_sig * sqrtf(cn[j]) : 0.0f;
            }
        }
        if (!p.z_dev) {
            const bool need = dlim ? (a_t && t != 1) : (t != 1);
            float4 z = need ? philox_normal4(p.seed, (uint64_t)(p.sample_offset + b), (uint32_t)(e0 >> 2), kPurposeStepZ, (uint32_t)t)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
            float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
            for (int j = 0; j < W; j++) zv[j] = zz[VEC ? j : (int)(e0 & 3)];
        }
        float o[W];
#pragma unroll
        for (int j = 0; j < W; j++) {
            float ee = clip ? clip_eps(xv[j], ev[j], c) : ev[j];
            float m = __fdiv_rn(xv[j] - __fmul_rn(ce[j], ee), c.g);
            if (dlim) m = __fadd_rn(m, __fmul_rn(dl_mean, ee));
            o[j] = __fadd_rn(m, __fmul_rn(cn[j], zv[j]));
        }
        if (VEC) {
            *reinterpret_cast<float4 *>(p.x_dev + off) = *reinterpret_cast<float4 *>(o);
            if (hist) *reinterpret_cast<float4 *>(hist + off) = *reinterpret_cast<float4 *>(o);
        } else {
            p.x_dev[off] = o[0];
            if (hist) hist[off] = o[0];
        }
    }
}

// Fast path of the fused update (stochastic DLPM step, no clipping, D % 4 == 0): one workgroup per
// sample, so the two per-sample coefficients are wave-uniform scalars, the Philox counter is just
// (sample, quad), and there is no per-element index division.  Three quads per thread are loaded
// before any arithmetic.  The division by gamma_t is a reciprocal multiply plus one Newton residual
// step (correctly rounded except for ties; the reference divides).

__global__ void __launch_bounds__(256) k_update_rows(dlpm_update_args p) {
    const int t = *p.t_dev;
    const float g = p.g_dev[t], rg = 1.0f / g;
    const int64_t b = blockIdx.x;
    const float ce = p.c_eps_dev[(int64_t)t * p.B + b], cn = p.c_noise_dev[(int64_t)t * p.B + b];
    const int nq = (int)(p.D >> 2);
    float *xr = p.x_dev + b * p.D;
    const float *er = p.eps_dev + b * p.D;
    const float *zr = p.z_dev ? p.z_dev + b * p.D : nullptr;
    const uint64_t seed = p.key_dev ? p.key_dev[0] : p.seed;
    const uint64_t gidx = (uint64_t)((p.key_dev ? (int64_t)p.key_dev[1] : p.sample_offset) + b);
    float *hr = p.hist_pp ? *p.hist_pp : nullptr;
    if (hr) hr += ((int64_t)(p.T - t) * p.B + b) * p.D;
    for (int q0 = threadIdx.x; q0 < nq; q0 += 3 * 256) {
        float4 x[3], e[3], z[3];
        int q[3];
        bool ok[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            q[u] = q0 + u * 256;
            ok[u] = q[u] < nq;
            const int qq = ok[u] ? q[u] : q0;
            x[u] = reinterpret_cast<const float4 *>(xr)[qq];
            e[u] = reinterpret_cast<const float4 *>(er)[qq];
            if (zr) z[u] = reinterpret_cast<const float4 *>(zr)[qq];
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            if (!zr) z[u] = (cn != 0.0f) ? philox_normal4(seed, gidx, (uint32_t)q[u], kPurposeStepZ, (uint32_t)t)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            o.x = fmaf(cn, z[u].x, div_by(x[u].x - ce * e[u].x, g, rg));
            o.y = fmaf(cn, z[u].y, div_by(x[u].y - ce * e[u].y, g, rg));
            o.z = fmaf(cn, z[u].z, div_by(x[u].z - ce * e[u].z, g, rg));
            o.w = fmaf(cn, z[u].w, div_by(x[u].w - ce * e[u].w, g, rg));
            if (ok[u]) {
                reinterpret_cast<float4 *>(xr)[q[u]] = o;
                if (hr) reinterpret_cast<float4 *>(hr)[q[u]] = o;
            }
        }
    }
}

// p_mean_variance for the model mean types other than "EPSILON without clipping" (GenerativeLevyProcess.py:182-207):
// the model output becomes x_0 (per mean type), optionally clamped, and is turned back into the eps the step formulas
// take (predict_eps, dlpm.py:198-202).  Element-wise, the reference's rounding order; ELEMENTWISE: [T,B,D] tables.
__global__ void __launch_bounds__(256) k_predict(dlpm_predict_args p) {
    const int t = *p.t_dev;
    const float g = p.g_dev[t], bg = p.bg_dev[t], bs = p.bs_dev[t];
    const bool elem = p.flags & DLPM_PRED_ELEMENTWISE;
    const int64_t n = p.B * p.D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = elem ? i : i / p.D;
        const int64_t ti = (int64_t)t * (elem ? n : p.B) + col;
        const float x = p.x_dev[i];
        float v = p.in_dev[i];
        if (p.flags & DLPM_PRED_TO_XSTART) {
            if (p.mean_type != DLPM_MEAN_START_X) {
                float e = v;
                if (p.mean_type == DLPM_MEAN_Z) e = __fmul_rn(__fsqrt_rn(p.A_dev[ti]), v);                       // eps = sqrt(A[t]) * out (:193)
                else if (p.mean_type == DLPM_MEAN_PREVIOUS_X) e = __fdiv_rn(x - __fmul_rn(v, g), p.c_eps_dev[ti]);   // dlpm.py:204-209
                v = __fdiv_rn(x - __fmul_rn(e, bs), bg);                                                        // predict_xstart, dlpm.py:191-196
            }
        }
        if (p.flags & DLPM_PRED_CLIP) v = fminf(fmaxf(v, -1.0f), 1.0f);
        if (p.flags & DLPM_PRED_TO_EPS) v = __fdiv_rn(x - __fmul_rn(v, bg), bs);                                  // predict_eps
        p.out_dev[i] = v;
    }
}

// LIM update (sde_score_update / ode_score_update, dlpm/methods/LIM/functions/sampler.py:85-152): per step the
// coefficients are four scalars; the SDE noise is gen_sas's clamp(sqrt(a_b) z) with a fresh per-sample a.
// One workgroup per sample (wave-uniform coefficients, Philox counter = (sample, quad)); VEC needs D % 4 == 0.
template <bool VEC>
__global__ void __launch_bounds__(256) k_update_lim(dlpm_lim_update_args p) {
    const int t = *p.t_dev;
    const int i = (p.T - 1) - t;
    const float tmp = p.tmp_dev[i], cx = p.cx_dev[i], cs = p.cs_dev[i], cn = p.cn_dev[i];
    const bool ode = p.flags & DLPM_UPD_DLIM;
    const int64_t b = blockIdx.x;
    const float sa = p.A_dev ? sqrtf(p.A_dev[(int64_t)i * p.B + b]) : 1.0f;
    const bool clamp = p.A_dev && p.clamp_eps >= 0.0f;
    const uint64_t seed = p.key_dev ? p.key_dev[0] : p.seed;
    const uint64_t gidx = (uint64_t)((p.key_dev ? (int64_t)p.key_dev[1] : p.sample_offset) + b);
    float *xr = p.x_dev + b * p.D;
    const float *er = p.eps_dev + b * p.D;
    const float *zr = p.z_dev ? p.z_dev + b * p.D : nullptr;
    float *hr = p.hist_pp ? *p.hist_pp : nullptr;
    if (hr) hr += ((int64_t)(p.T - t) * p.B + b) * p.D;
    constexpr int W = VEC ? 4 : 1;
    const int n = (int)(p.D / W);
    for (int q = threadIdx.x; q < n; q += blockDim.x) {
        float xv[W], ev[W], zv[W], o[W];
        if (VEC) {
            *reinterpret_cast<float4 *>(xv) = reinterpret_cast<const float4 *>(xr)[q];
            *reinterpret_cast<float4 *>(ev) = reinterpret_cast<const float4 *>(er)[q];
            if (zr) *reinterpret_cast<float4 *>(zv) = reinterpret_cast<const float4 *>(zr)[q];
        } else {
            xv[0] = xr[q];
            ev[0] = er[q];
            if (zr) zv[0] = zr[q];
        }
        if (!ode && !zr) {
            float4 z = philox_normal4(seed, gidx, (uint32_t)(VEC ? q : q >> 2), kPurposeStepZ, (uint32_t)t);
            float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
            for (int j = 0; j < W; j++) zv[j] = zz[VEC ? j : (q & 3)];
        }
#pragma unroll
        for (int j = 0; j < W; j++) {
            const float score = __fmul_rn(ev[j], tmp);
            float v = __fadd_rn(__fmul_rn(cx, xv[j]), __fmul_rn(cs, score));
            if (!ode) {
                float e = __fmul_rn(sa, zv[j]);
                if (clamp) e = fminf(fmaxf(e, -p.clamp_eps), p.clamp_eps);
                v = __fadd_rn(v, __fmul_rn(cn, e));
            }
            o[j] = v;
        }
        if (VEC) {
            reinterpret_cast<float4 *>(xr)[q] = *reinterpret_cast<float4 *>(o);
            if (hr) reinterpret_cast<float4 *>(hr)[q] = *reinterpret_cast<float4 *>(o);
        } else {
            xr[q] = o[0];
            if (hr) hr[q] = o[0];
        }
    }
}

__global__ void k_fill_table_t(float *tv, const int32_t *t, const float *ts, int32_t T, int64_t B) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) tv[i] = ts[(T - 1) - *t];
}

__global__ void k_advance(int32_t *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *t = *t - 1;
}

__global__ void k_fill_t(float *tv, const int32_t *t, int32_t T, int64_t B) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) tv[i] = __fmul_rn((float)(*t), 1.0f / (float)T);  // t.float() * (1.0 / T)
}

__global__ void k_scale_by_table(const float *x, float *o, int64_t n, const int32_t *t, const float *tab) {
    const float sc = tab[*t];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = __fmul_rn(x[i], sc);
}

__global__ void k_postprocess(const float *x, float *o, int64_t n, float c, int affine) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = fminf(fmaxf(x[i], -c), c);
    o[i] = affine ? __fdiv_rn(v + 1.0f, 2.0f) : v;
}

__global__ void k_nchw_to_nhwc(const float *src, float *dst, int B, int C, int HW) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = (int64_t)B * C * HW;
    if (i >= n) return;
    int c = (int)(i % C);
    int64_t r = i / C;
    int p = (int)(r % HW);
    int64_t b = r / HW;
    dst[i] = src[(b * C + c) * HW + p];
}

__global__ void k_nhwc_to_nchw(const float *src, float *dst, int B, int C, int HW) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = (int64_t)B * C * HW;
    if (i >= n) return;
    int p = (int)(i % HW);
    int64_t r = i / HW;
    int c = (int)(r % C);
    int64_t b = r / C;
    dst[i] = src[(b * HW + p) * C + c];
}

}  // namespace

extern "C" int dlpm_skewed_levy_philox_f32(float *A_dev, int T, int64_t B, double alpha, double clamp_a, uint64_t seed,
                                           int64_t sample_offset, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(A_dev && T > 0 && B > 0, "dlpm_skewed_levy_philox_f32: bad shape T=%d B=%lld", T, (long long)B);
    DLPM_CHECK_ARG(alpha > 0.0 && alpha <= 2.0, "Wrong value of alpha (%g) for skewed levy r.v generation", alpha);
    int64_t n = (int64_t)T * B;
    k_skewed_levy<<<(unsigned)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(A_dev, T, B, alpha, (float)clamp_a,
                                                                            clamp_a >= 0.0, seed, sample_offset);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_skewed_levy_elem_philox_f32(float *A_dev, int T, int64_t B, int64_t D, double alpha, double clamp_a,
                                                uint64_t seed, int64_t sample_offset, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(A_dev && T > 0 && B > 0 && D > 0, "dlpm_skewed_levy_elem_philox_f32: bad shape");
    DLPM_CHECK_ARG(D < (1ll << 32), "dlpm_skewed_levy_elem_philox_f32: D must fit 32 bits");
    DLPM_CHECK_ARG(T < (1 << 24), "dlpm_skewed_levy_elem_philox_f32: T must fit 24 bits");
    DLPM_CHECK_ARG(alpha > 0.0 && alpha <= 2.0, "Wrong value of alpha (%g) for skewed levy r.v generation", alpha);
    int64_t n = (int64_t)T * B * D;
    unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n, 256), 256 * 64);
    k_skewed_levy_elem<<<grid, 256, 0, as_stream(stream)>>>(A_dev, T, B, D, alpha, (float)clamp_a, clamp_a >= 0.0, seed,
                                                           sample_offset);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_init_state_elem_philox_f32(float *x_dev, int64_t B, int64_t D, double alpha, double clamp_eps,
                                               float barsigma_last, uint64_t seed, int64_t sample_offset,
                                               dlpm_stream_t stream) {
    DLPM_CHECK_ARG(x_dev && B > 0 && D > 0 && D < (1ll << 32), "dlpm_init_state_elem_philox_f32: bad shape");
    DLPM_CHECK_ARG(alpha > 0.0 && alpha <= 2.0, "Wrong value of alpha (%g) for skewed levy r.v generation", alpha);
    int64_t n = B * ((D + 3) / 4);
    k_init_state_elem<<<(unsigned)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(x_dev, B, D, alpha, (float)clamp_eps,
                                                                                clamp_eps >= 0.0, barsigma_last, seed,
                                                                                sample_offset);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_init_state_philox_f32(float *x_dev, int64_t B, int64_t D, double alpha, double clamp_eps,
                                          float barsigma_last, uint64_t seed, int64_t sample_offset,
                                          dlpm_stream_t stream) {
    DLPM_CHECK_ARG(x_dev && B > 0 && D > 0, "dlpm_init_state_philox_f32: bad shape");
    DLPM_CHECK_ARG(alpha > 0.0 && alpha <= 2.0, "Wrong value of alpha (%g) for skewed levy r.v generation", alpha);
    int64_t n = B * ((D + 3) / 4);
    k_init_state<<<(unsigned)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(x_dev, B, D, alpha, (float)clamp_eps,
                                                                           clamp_eps >= 0.0, barsigma_last, seed,
                                                                           sample_offset);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_coeff_tables_f32(const float *A_dev, const float *g_dev, const float *s_dev, const float *bs_dev,
                                     int T, int64_t B, float *c_eps_dev, float *c_noise_dev, float *sigmas_out_dev,
                                     dlpm_stream_t stream) {
    DLPM_CHECK_ARG(A_dev && g_dev && s_dev && bs_dev && c_eps_dev && c_noise_dev, "dlpm_coeff_tables_f32: null pointer");
    DLPM_CHECK_ARG(T >= 2 && B > 0, "dlpm_coeff_tables_f32: bad shape T=%d B=%lld", T, (long long)B);
    k_coeff_tables<<<(unsigned)ceil_div(B, 64), 64, 0, as_stream(stream)>>>(A_dev, g_dev, s_dev, bs_dev, T, B, c_eps_dev,
                                                                          c_noise_dev, sigmas_out_dev);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_update_f32(const dlpm_update_args *a, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(a && a->x_dev && a->eps_dev && a->t_dev && a->g_dev && a->bg_dev && a->bs_dev,
                   "dlpm_update_f32: null pointer");
    DLPM_CHECK_ARG(a->B > 0 && a->D > 0 && a->T >= 2, "dlpm_update_f32: bad shape");
    if (!(a->flags & DLPM_UPD_DLIM))
        DLPM_CHECK_ARG(a->c_eps_dev && a->c_noise_dev, "dlpm_update_f32: DLPM step needs the coefficient tables");
    else if (a->dlim_eta != 0.0f)
        DLPM_CHECK_ARG(a->A_dev, "dlpm_update_f32: DLIM with eta > 0 needs A");
    const bool vec = (a->D % 4 == 0) && ((reinterpret_cast<uintptr_t>(a->x_dev) | reinterpret_cast<uintptr_t>(a->eps_dev) |
                                          reinterpret_cast<uintptr_t>(a->z_dev)) % 16 == 0);
    int64_t items = a->B * (vec ? a->D / 4 : a->D);
    // memory-bound: cap the grid at 8 blocks per CU and grid-stride the rest
    // 4 quads per thread (see k_update); at most 8 blocks per CU, the rest is grid-strided
    unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(items, 256 * (vec ? 4 : 1)), 256 * 8));
    // algorithmic bytes: read x, read eps, write x (+ read z when injected)
    const bool elem = a->flags & DLPM_UPD_ELEMENTWISE;
    ProfScope ps(elem ? "update_elem" : "update", 0.0,
                 4.0 * (double)a->B * a->D * ((a->z_dev ? 4 : 3) + (elem ? 2 : 0) + ((a->hist_pp && (a->flags & DLPM_UPD_HIST_ON)) ? 1 : 0)), as_stream(stream));
    if (elem) {
        const bool evec = vec && ((reinterpret_cast<uintptr_t>(a->c_eps_dev) | reinterpret_cast<uintptr_t>(a->c_noise_dev) |
                                   reinterpret_cast<uintptr_t>(a->A_dev)) % 16 == 0);
        unsigned egrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(a->B * a->D / (evec ? 4 : 1), 256), 256 * 16));
        if (evec) k_update_elem<true><<<egrid, 256, 0, as_stream(stream)>>>(*a);
        else k_update_elem<false><<<egrid, 256, 0, as_stream(stream)>>>(*a);
    } else if (vec && !(a->flags & (DLPM_UPD_DLIM | DLPM_UPD_CLIP)) && a->B < (1 << 30))
        k_update_rows<<<(unsigned)a->B, 256, 0, as_stream(stream)>>>(*a);
    else if (vec) k_update<true><<<grid, 256, 0, as_stream(stream)>>>(*a);
    else k_update<false><<<grid, 256, 0, as_stream(stream)>>>(*a);
    DLPM_LAUNCH_CHECK();
    if (a->flags & DLPM_UPD_ADVANCE) {
        k_advance<<<1, 64, 0, as_stream(stream)>>>(const_cast<int32_t *>(a->t_dev));
        DLPM_LAUNCH_CHECK();
    }
    return DLPM_OK;
}

extern "C" int dlpm_predict_f32(const dlpm_predict_args *a, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(a && a->x_dev && a->in_dev && a->out_dev && a->t_dev && a->g_dev && a->bg_dev && a->bs_dev, "dlpm_predict_f32: null pointer");
    DLPM_CHECK_ARG(a->B > 0 && a->D > 0 && a->T >= 2, "dlpm_predict_f32: bad shape");
    DLPM_CHECK_ARG(a->mean_type >= DLPM_MEAN_EPSILON && a->mean_type <= DLPM_MEAN_PREVIOUS_X, "dlpm_predict_f32: unknown mean type %d", a->mean_type);
    if (a->flags & DLPM_PRED_TO_XSTART) {
        DLPM_CHECK_ARG(a->mean_type != DLPM_MEAN_Z || a->A_dev, "dlpm_predict_f32: mean type Z needs A");
        DLPM_CHECK_ARG(a->mean_type != DLPM_MEAN_PREVIOUS_X || a->c_eps_dev, "dlpm_predict_f32: mean type PREVIOUS_X needs c_eps");
    }
    const int64_t n = a->B * a->D;
    ProfScope ps("predict", 0.0, 12.0 * (double)n, as_stream(stream));
    k_predict<<<(unsigned)std::min<int64_t>(ceil_div(n, 256), 256 * 16), 256, 0, as_stream(stream)>>>(*a);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

namespace dlpm {
int launch_step_advance(int32_t *t_dev, hipStream_t st) {
    k_advance<<<1, 64, 0, st>>>(t_dev);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}
}  // namespace dlpm

extern "C" int dlpm_lim_update_f32(const dlpm_lim_update_args *a, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(a && a->x_dev && a->eps_dev && a->t_dev && a->tmp_dev && a->cx_dev && a->cs_dev && a->cn_dev,
                   "dlpm_lim_update_f32: null pointer");
    DLPM_CHECK_ARG(a->B > 0 && a->B < (1ll << 31) && a->D > 0 && a->D < (1ll << 31) && a->T >= 2, "dlpm_lim_update_f32: bad shape");
    const bool vec = (a->D % 4 == 0) && ((reinterpret_cast<uintptr_t>(a->x_dev) | reinterpret_cast<uintptr_t>(a->eps_dev) |
                                          reinterpret_cast<uintptr_t>(a->z_dev)) % 16 == 0);
    const bool ode = a->flags & DLPM_UPD_DLIM;
    ProfScope ps("lim_update", 0.0, 4.0 * (double)a->B * a->D * ((a->z_dev && !ode ? 4 : 3) + ((a->hist_pp && (a->flags & DLPM_UPD_HIST_ON)) ? 1 : 0)),
                 as_stream(stream));
    const int64_t items = vec ? a->D / 4 : a->D;
    const unsigned threads = items >= 256 ? 256 : 64;
    if (vec) k_update_lim<true><<<(unsigned)a->B, threads, 0, as_stream(stream)>>>(*a);
    else k_update_lim<false><<<(unsigned)a->B, threads, 0, as_stream(stream)>>>(*a);
    DLPM_LAUNCH_CHECK();
    if (a->flags & DLPM_UPD_ADVANCE) {
        k_advance<<<1, 64, 0, as_stream(stream)>>>(const_cast<int32_t *>(a->t_dev));
        DLPM_LAUNCH_CHECK();
    }
    return DLPM_OK;
}

extern "C" int dlpm_fill_table_t_f32(float *tvec_dev, const int32_t *t_dev, const float *ts_dev, int32_t T, int64_t B,
                                     dlpm_stream_t stream) {
    DLPM_CHECK_ARG(tvec_dev && t_dev && ts_dev && T >= 2 && B > 0, "dlpm_fill_table_t_f32: bad argument");
    k_fill_table_t<<<(unsigned)ceil_div(B, 256), 256, 0, as_stream(stream)>>>(tvec_dev, t_dev, ts_dev, T, B);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_fill_scaled_t_f32(float *tvec_dev, const int32_t *t_dev, int32_t T, int64_t B, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(tvec_dev && t_dev && T > 0 && B > 0, "dlpm_fill_scaled_t_f32: bad argument");
    k_fill_t<<<(unsigned)ceil_div(B, 256), 256, 0, as_stream(stream)>>>(tvec_dev, t_dev, T, B);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_scale_by_table_f32(const float *x_dev, float *out_dev, int64_t n, const int32_t *t_dev,
                                       const float *table_dev, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(x_dev && out_dev && t_dev && table_dev && n > 0, "dlpm_scale_by_table_f32: bad argument");
    unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n, 256), 256 * 16);
    k_scale_by_table<<<grid, 256, 0, as_stream(stream)>>>(x_dev, out_dev, n, t_dev, table_dev);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_postprocess_f32(const float *x_dev, float *out_dev, int64_t n, float clamp, int affine,
                                    dlpm_stream_t stream) {
    DLPM_CHECK_ARG(x_dev && out_dev && n > 0, "dlpm_postprocess_f32: bad argument");
    k_postprocess<<<(unsigned)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(x_dev, out_dev, n, clamp, affine);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_nchw_to_nhwc_f32(const float *src, float *dst, int32_t B, int32_t C, int32_t H, int32_t W,
                                     dlpm_stream_t stream) {
    DLPM_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "dlpm_nchw_to_nhwc_f32: bad argument");
    int64_t n = (int64_t)B * C * H * W;
    k_nchw_to_nhwc<<<(unsigned)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(src, dst, B, C, H * W);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_nhwc_to_nchw_f32(const float *src, float *dst, int32_t B, int32_t C, int32_t H, int32_t W,
                                     dlpm_stream_t stream) {
    DLPM_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "dlpm_nhwc_to_nchw_f32: bad argument");
    int64_t n = (int64_t)B * C * H * W;
    k_nhwc_to_nchw<<<(unsigned)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(src, dst, B, C, H * W);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}
