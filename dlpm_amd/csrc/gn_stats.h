// gn_stats.h -- GroupNorm coefficients of ONE image from the per-tile channel statistics (mean, centred sum of squares) its producers
// emitted: the body of k_gn_coeffs_stats (groupnorm.hip), shared with the kernels that compute their own coefficients inside a
// workgroup (round 6: the whole-image ResBlock kernel, conv_wino4.hip).  One source, one arithmetic: a block that folds the
// coefficient step into its prologue gets the bits the separate launch would have written.
//
// Replaces GroupNorm32.forward (dlpm/models/nn.py:17-19) + the scale-shift of ResBlock._forward (unet.py:187-191), like groupnorm.hip.
#pragma once
#include "common.h"

namespace dlpm {

// Called by EVERY thread of the block (tid of nthr).  st0 / st1: this image's first partial of the two concat sources ([nt][C0] /
// [nt][C1] float2, any address space), nt0 / nt1 partials of HW / nt pixels each; ss_row: this image's emb row (scale at [c], shift
// at [C + c]) or null; sh: 2 C + 2 G floats of shared scratch; outA / outB: C coefficients each (global or shared).  `bar` is the
// block barrier to use between the three steps (and the caller synchronises before reading outA / outB from other threads).
template <typename Bar>
__device__ __forceinline__ void gn_coeffs_from_stats_image(const float2 *st0, const float2 *st1, int C0, int C1, int nt0, int nt1, int HW, int G,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *ss_row, float *sh, float *outA, float *outB, float eps, int tid, int nthr,
                                                           Bar bar) {
    const int C = C0 + C1, cg = C / G;
    float *cmean = sh, *cm2 = sh + C, *mean = cm2 + C, *rstd = mean + G;
    for (int c = tid; c < C; c += nthr) {
        const int nt = (c < C0) ? nt0 : nt1;
        const float npt = (float)(HW / nt);  // pixels per tile
        const float2 *sp = (c < C0) ? st0 + c : st1 + (c - C0);
        const int ld = (c < C0) ? C0 : C1;
        float m = sp[0].x, M2 = sp[0].y, na = npt;
        for (int k = 1; k < nt; k++) {
            const float2 q = sp[(int64_t)k * ld];
            const float d = q.x - m, N = na + npt;
            m += d * (npt / N);
            M2 += q.y + d * d * (na * npt / N);
            na = N;
        }
        cmean[c] = m;
        cm2[c] = M2;
    }
    bar();
    const float fn = (float)HW;
    for (int g = tid; g < G; g += nthr) {
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
    bar();
    for (int c = tid; c < C; c += nthr) {
        const int g = c / cg;
        float a = rstd[g] * gamma[c];
        float bb = beta[c] - mean[g] * a;
        if (ss_row) {
            const float sc = 1.0f + ss_row[c];
            const float sft = ss_row[C + c];
            a = a * sc;
            bb = fmaf(bb, sc, sft);
        }
        outA[c] = a;
        outB[c] = bb;
    }
}

}  // namespace dlpm
