// groupnorm.hip -- GroupNorm32 statistics folded into per-(sample, channel) affine coefficients.
//
// Replaces GroupNorm32.forward (dlpm/models/nn.py:17-19; 32 groups or C if smaller, eps 1e-5) and
// the scale-shift modulation of ResBlock._forward (unet.py:187-191):
//     y = GN(x) * (1 + scale) + shift  ==  x * A[b,c] + Bc[b,c]
// The consumer convolution applies (A, Bc) (+ SiLU) while staging its input tile, so the
// normalised activation is never written to HBM: this kernel only READS x (4 B/element, HBM/L2
// bound) and writes 2*B*C floats.  x is NHWC and may be a virtual concat of two tensors (the
// UNet skip connection, unet.py:489), whose groups can straddle the two sources.
// Statistics in fp32, one workgroup per sample, parallel over channels (coalesced rows) x pixel
// slices: a one-pass shifted-data kernel for channel counts divisible by 4 (every shipped config),
// a two-pass scalar kernel otherwise.
#include "conv.h"
#include "gn_stats.h"

namespace dlpm {
namespace {

__global__ void __launch_bounds__(512) k_gn_coeffs(const float *__restrict__ src0, const float *__restrict__ src1, int C0,
                                                   int C1, int HW, int G, const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, const float *__restrict__ ss,
                                                   int64_t ss_stride, int64_t ss_offset, float *__restrict__ coefA,
                                                   float *__restrict__ coefB, float eps) {
    extern __shared__ float sh[];
    const int C = C0 + C1, cg = C / G;
    const int nt = blockDim.x, tid = threadIdx.x;
    const int nsl = (C <= nt) ? nt / C : 1;  // pixel slices
    const int nwork = nsl * C;
    float *partial = sh;                               // [max(nt, C)]
    float *chan = sh + (nwork > nt ? nwork : nt);      // [C]
    float *mean = chan + C;                            // [G]
    float *rstd = mean + G;                            // [G]
    const int b = blockIdx.x;
    const float *x0 = src0 + (int64_t)b * HW * C0;
    const float *x1 = src1 ? src1 + (int64_t)b * HW * C1 : nullptr;
    const float inv_n = 1.0f / (float)((int64_t)cg * HW);

    for (int pass = 0; pass < 2; pass++) {
        for (int idx = tid; idx < nwork; idx += nt) {
            const int c = idx % C, sl = idx / C;
            const float *xp = (c < C0) ? x0 + c : x1 + (c - C0);
            const int ld = (c < C0) ? C0 : C1;
            const float mu = pass ? mean[c / cg] : 0.f;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int p = sl;
            for (; p + 3 * nsl < HW; p += 4 * nsl) {
                float v0 = xp[(int64_t)p * ld] - mu, v1 = xp[(int64_t)(p + nsl) * ld] - mu;
                float v2 = xp[(int64_t)(p + 2 * nsl) * ld] - mu, v3 = xp[(int64_t)(p + 3 * nsl) * ld] - mu;
                if (pass) { a0 = fmaf(v0, v0, a0); a1 = fmaf(v1, v1, a1); a2 = fmaf(v2, v2, a2); a3 = fmaf(v3, v3, a3); }
                else { a0 += v0; a1 += v1; a2 += v2; a3 += v3; }
            }
            for (; p < HW; p += nsl) {
                float v = xp[(int64_t)p * ld] - mu;
                a0 = pass ? fmaf(v, v, a0) : a0 + v;
            }
            partial[idx] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        for (int c = tid; c < C; c += nt) {
            float s = 0.f;
            for (int sl = 0; sl < nsl; sl++) s += partial[sl * C + c];
            chan[c] = s;
        }
        __syncthreads();
        for (int g = tid; g < G; g += nt) {
            float s = 0.f;
            for (int c = g * cg; c < (g + 1) * cg; c++) s += chan[c];
            if (pass) rstd[g] = 1.0f / sqrtf(s * inv_n + eps);
            else mean[g] = s * inv_n;
        }
        __syncthreads();
    }
    for (int c = tid; c < C; c += nt) {
        const int g = c / cg;
        float a = rstd[g] * gamma[c];
        float bb = beta[c] - mean[g] * a;
        if (ss) {
            const float sc = 1.0f + ss[(int64_t)b * ss_stride + ss_offset + c];
            const float sft = ss[(int64_t)b * ss_stride + ss_offset + C + c];
            a = a * sc;
            bb = fmaf(bb, sc, sft);
        }
        coefA[(int64_t)b * C + c] = a;
        coefB[(int64_t)b * C + c] = bb;
    }
}

// Vectorised ONE-pass variant (C0 % 4 == 0 and C1 % 4 == 0): the activation is read exactly once
// (4 B/element, the algorithmic minimum).  One thread owns 4 consecutive channels and a pixel slice
// and accumulates sum and sum of squares of (x - K_c), K_c = the channel's first pixel -- the
// shifted-data formulation, whose cancellation is governed by |K_c - mean_c| / std_c (O(1) when the
// pivot is a sample of the data) instead of |mean| / std.  Per-channel (mean, M2) are then merged
// into groups with Chan's parallel update, which is exact for unequal channel means.
__global__ void __launch_bounds__(512) k_gn_coeffs_v4(const float *__restrict__ src0, const float *__restrict__ src1,
                                                      int C0, int C1, int HW, int G, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, const float *__restrict__ ss,
                                                      int64_t ss_stride, int64_t ss_offset, float *__restrict__ coefA,
                                                      float *__restrict__ coefB, float eps) {
    extern __shared__ float sh[];
    const int C = C0 + C1, cg = C / G, Cq = C / 4;
    const int nt = blockDim.x, tid = threadIdx.x;
    const int nsl = (Cq <= nt) ? nt / Cq : 1;
    const int nwork = nsl * Cq;
    float *p1 = sh;                          // [nsl * C] partial sums of (x - K)
    float *p2 = p1 + (size_t)nsl * C;        // [nsl * C] partial sums of (x - K)^2
    float *cmean = p2 + (size_t)nsl * C;     // [C] channel means
    float *cm2 = cmean + C;                  // [C] channel centred sums of squares
    float *mean = cm2 + C, *rstd = mean + G; // [G]
    const int b = blockIdx.x;
    const float *x0 = src0 + (int64_t)b * HW * C0;
    const float *x1 = src1 ? src1 + (int64_t)b * HW * C1 : nullptr;

    for (int idx = tid; idx < nwork; idx += nt) {
        const int c = (idx % Cq) * 4, sl = idx / Cq;
        const float *xp = (c < C0) ? x0 + c : x1 + (c - C0);
        const int ld = (c < C0) ? C0 : C1;
        const float4 K = *reinterpret_cast<const float4 *>(xp);  // pivot: pixel 0
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, t1 = s1, t2 = s1;
        int p = sl;
        for (; p + nsl < HW; p += 2 * nsl) {
            const float4 v = *reinterpret_cast<const float4 *>(xp + (int64_t)p * ld);
            const float4 w = *reinterpret_cast<const float4 *>(xp + (int64_t)(p + nsl) * ld);
            float d;
            d = v.x - K.x; s1.x += d; s2.x = fmaf(d, d, s2.x); d = v.y - K.y; s1.y += d; s2.y = fmaf(d, d, s2.y);
            d = v.z - K.z; s1.z += d; s2.z = fmaf(d, d, s2.z); d = v.w - K.w; s1.w += d; s2.w = fmaf(d, d, s2.w);
            d = w.x - K.x; t1.x += d; t2.x = fmaf(d, d, t2.x); d = w.y - K.y; t1.y += d; t2.y = fmaf(d, d, t2.y);
            d = w.z - K.z; t1.z += d; t2.z = fmaf(d, d, t2.z); d = w.w - K.w; t1.w += d; t2.w = fmaf(d, d, t2.w);
        }
        for (; p < HW; p += nsl) {
            const float4 v = *reinterpret_cast<const float4 *>(xp + (int64_t)p * ld);
            float d;
            d = v.x - K.x; s1.x += d; s2.x = fmaf(d, d, s2.x); d = v.y - K.y; s1.y += d; s2.y = fmaf(d, d, s2.y);
            d = v.z - K.z; s1.z += d; s2.z = fmaf(d, d, s2.z); d = v.w - K.w; s1.w += d; s2.w = fmaf(d, d, s2.w);
        }
        *reinterpret_cast<float4 *>(p1 + (size_t)sl * C + c) = make_float4(s1.x + t1.x, s1.y + t1.y, s1.z + t1.z, s1.w + t1.w);
        *reinterpret_cast<float4 *>(p2 + (size_t)sl * C + c) = make_float4(s2.x + t2.x, s2.y + t2.y, s2.z + t2.z, s2.w + t2.w);
    }
    __syncthreads();
    const float fn = (float)HW;
    for (int c = tid; c < C; c += nt) {
        float a = 0.f, q = 0.f;
        for (int sl = 0; sl < nsl; sl++) { a += p1[(size_t)sl * C + c]; q += p2[(size_t)sl * C + c]; }
        const float K = (c < C0) ? x0[c] : x1[c - C0];
        const float dm = a / fn;
        cmean[c] = K + dm;
        cm2[c] = fmaxf(q - a * dm, 0.f);   // sum (x - mean_c)^2 = sum (x-K)^2 - (sum (x-K))^2 / n
    }
    __syncthreads();
    for (int g = tid; g < G; g += nt) {
        float m = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; c++) m += cmean[c];
        m /= (float)cg;
        float M2 = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; c++) {
            const float d = cmean[c] - m;
            M2 += cm2[c] + fn * d * d;
        }
        mean[g] = m;
        rstd[g] = 1.0f / sqrtf(M2 / (fn * (float)cg) + eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += nt) {
        const int g = c / cg;
        float a = rstd[g] * gamma[c];
        float bb = beta[c] - mean[g] * a;
        if (ss) {
            const float sc = 1.0f + ss[(int64_t)b * ss_stride + ss_offset + c];
            const float sft = ss[(int64_t)b * ss_stride + ss_offset + C + c];
            a = a * sc;
            bb = fmaf(bb, sc, sft);
        }
        coefA[(int64_t)b * C + c] = a;
        coefB[(int64_t)b * C + c] = bb;
    }
}

// Coefficients from the per-tile channel statistics the MFMA conv epilogues emit (ConvLaunch::
// stats_out): the activation itself is never re-read, so GroupNorm costs O(B*C) instead of a full
// 4 B/element pass.  Tiles, then channels of a group, are merged with Chan's parallel update.
__global__ void __launch_bounds__(256) k_gn_coeffs_stats(const float2 *__restrict__ st0, const float2 *__restrict__ st1, int C0,
                                                         int C1, int nt0, int nt1, int HW, int G, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, const float *__restrict__ ss,
                                                         int64_t ss_stride, int64_t ss_offset, float *__restrict__ coefA,
                                                         float *__restrict__ coefB, float eps) {
    extern __shared__ float sh[];
    const int C = C0 + C1;
    const int b = blockIdx.x;
    // (the arithmetic lives in gn_stats.h: the whole-image ResBlock kernel runs the same function inside its workgroups)
    gn_coeffs_from_stats_image(st0 + (int64_t)b * nt0 * C0, st1 ? st1 + (int64_t)b * nt1 * C1 : nullptr, C0, C1, nt0, nt1, HW, G, gamma, beta,
                               ss ? ss + (int64_t)b * ss_stride + ss_offset : nullptr, sh, coefA + (int64_t)b * C, coefB + (int64_t)b * C, eps,
                               (int)threadIdx.x, (int)blockDim.x, [] { __syncthreads(); });
}

}  // namespace

int launch_gn_coeffs_from_stats(const float2 *st0, const float2 *st1, int C0, int C1, int B, int nt0, int nt1, int HW, int groups,
                                const float *gamma, const float *beta, const float *ss, int64_t ss_stride,
                                int64_t ss_offset, float *coefA, float *coefB, hipStream_t st) {
    const int C = C0 + C1;
    ProfScope ps("groupnorm_from_stats", 0.0, 8.0 * (double)B * (nt0 * C0 + nt1 * C1) + 8.0 * B * C, st);
    k_gn_coeffs_stats<<<(unsigned)B, 256, (size_t)(2 * C + 2 * groups) * sizeof(float), st>>>(
        st0, st1, C0, C1, nt0, nt1 > 0 ? nt1 : 1, HW, groups, gamma, beta, ss, ss_stride, ss_offset, coefA, coefB, 1e-5f);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int launch_gn_coeffs(const float *src0, const float *src1, int C0, int C1, int B, int HW, int groups, const float *gamma,
                     const float *beta, const float *ss, int64_t ss_stride, int64_t ss_offset, float *coefA,
                     float *coefB, hipStream_t st) {
    const int C = C0 + C1;
    const int nt = 512;
    const int nsl = (C <= nt) ? nt / C : 1;
    const int nwork = nsl * C;
    const size_t shmem = (size_t)((nwork > nt ? nwork : nt) + C + 2 * groups) * sizeof(float);
    ProfScope ps("groupnorm_coeffs", 0.0, 4.0 * ((double)B * HW * C + 2.0 * B * C), st);  // reads x once (algorithmic)
    if (C0 % 4 == 0 && C1 % 4 == 0 && C / 4 <= nt) {
        const int nslv = nt / (C / 4);
        const size_t shv = (size_t)((size_t)2 * nslv * C + 2 * C + 2 * groups) * sizeof(float);
        k_gn_coeffs_v4<<<(unsigned)B, nt, shv, st>>>(src0, src1, C0, C1, HW, groups, gamma, beta, ss, ss_stride, ss_offset,
                                                   coefA, coefB, 1e-5f);
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
    k_gn_coeffs<<<(unsigned)B, nt, shmem, st>>>(src0, src1, C0, C1, HW, groups, gamma, beta, ss, ss_stride, ss_offset,
                                               coefA, coefB, 1e-5f);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
