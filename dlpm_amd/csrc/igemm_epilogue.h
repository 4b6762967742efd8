// igemm_epilogue.h -- the full-tile epilogue of the 128-pixel MFMA GEMM tiles, shared by the fp32 implicit-GEMM kernel
// (conv_igemm.hip) and the split-operand 1x1 kernel (conv_split.hip).  Both accumulate in the C/D layout of the 32x32
// MFMAs: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
#pragma once
#include "conv.h"

namespace dlpm {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128;     // pixels per workgroup tile

// Full-tile form of epilogue_rows (M % 128 == 0, Cout % BN == 0: no bounds tests).  The co-resident workgroup's MFMAs
// share this SIMD's issue port, so every VALU instruction here is paid for in matrix-pipe time (phase counters: the
// generic epilogue took 30 % of a 1x1 workgroup's life): thread -> (channel quad, row group) is fixed, rows advance by
// pointer increments, bias is loaded once and the residual rows of a phase are fetched before its barrier.
template <int BN, int WAVES_M, int WAVES_N, int RM, int RN>
__device__ __forceinline__ void epilogue_rows_full(const ConvLaunch &p, floatx16 (&acc)[RM][RN], float *lds, int64_t m0, int n0,
                                                   int tid, int wm, int wn, int l31, int kh) {
    constexpr int LD = BN + 4, PR = RM * 32, C4 = BN / 4, RG = 256 / C4, NP = PR / RG;
    static_assert(PR % RG == 0, "row groups must tile a phase");
    const int c4 = tid % C4, rg = tid / C4;
    const int n = n0 + c4 * 4;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias = *reinterpret_cast<const float4 *>(p.bias + n);
    const bool has_res = p.res0 != nullptr;
    const int R1 = p.Cout - p.R0;
    const int rs = (n < p.R0) ? p.R0 : R1;                       // residual row stride of this thread's source
    const float *rp = has_res ? ((n < p.R0) ? p.res0 + (m0 + rg) * p.R0 + n : p.res1 + (m0 + rg) * R1 + (n - p.R0)) : p.out;
    float *op = p.out + (m0 + rg) * p.Cout + n;
    const float *lp = lds + rg * LD + c4 * 4;
    const bool do_stats = p.stats_out != nullptr;
    float4 K = make_float4(0.f, 0.f, 0.f, 0.f), s1 = K, s2 = K;
#pragma unroll
    for (int ph = 0; ph < WAVES_M; ph++) {
        float4 q[NP];
        if (has_res) {
#pragma unroll
            for (int k = 0; k < NP; k++) q[k] = *reinterpret_cast<const float4 *>(rp + (int64_t)(ph * PR + k * RG) * rs);
        }
        if (wm == ph) {
#pragma unroll
            for (int i = 0; i < RM; i++)
#pragma unroll
                for (int j = 0; j < RN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        lds[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * LD + (wn * RN + j) * 32 + l31] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NP; k++) {
            float4 v = *reinterpret_cast<const float4 *>(lp + k * RG * LD);
            v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
            if (has_res) { v.x += q[k].x; v.y += q[k].y; v.z += q[k].z; v.w += q[k].w; }
            if (do_stats) {
                if (ph == 0 && k == 0) K = v;  // pivot = this thread's first value per channel
                float d;
                d = v.x - K.x; s1.x += d; s2.x = fmaf(d, d, s2.x);
                d = v.y - K.y; s1.y += d; s2.y = fmaf(d, d, s2.y);
                d = v.z - K.z; s1.z += d; s2.z = fmaf(d, d, s2.z);
                d = v.w - K.w; s1.w += d; s2.w = fmaf(d, d, s2.w);
            }
            *reinterpret_cast<float4 *>(op + (int64_t)(ph * PR + k * RG) * p.Cout) = v;
        }
        __syncthreads();
    }
    if (do_stats) {
        float2 *part = reinterpret_cast<float2 *>(lds);  // the row image is dead now
        const float fc = (float)(NP * WAVES_M);
        const float mx = s1.x / fc, my = s1.y / fc, mz = s1.z / fc, mw = s1.w / fc;
        part[rg * BN + c4 * 4 + 0] = make_float2(K.x + mx, fmaxf(s2.x - s1.x * mx, 0.f));
        part[rg * BN + c4 * 4 + 1] = make_float2(K.y + my, fmaxf(s2.y - s1.y * my, 0.f));
        part[rg * BN + c4 * 4 + 2] = make_float2(K.z + mz, fmaxf(s2.z - s1.z * mz, 0.f));
        part[rg * BN + c4 * 4 + 3] = make_float2(K.w + mw, fmaxf(s2.w - s1.w * mw, 0.f));
        __syncthreads();
        if (tid < BN) {
            const float npart = (float)(BM / RG);
            float mean = part[tid].x, M2 = part[tid].y, na = npart;
            for (int g = 1; g < RG; g++) {
                const float2 qq = part[g * BN + tid];
                const float d = qq.x - mean, N = na + npart;
                mean += d * (npart / N);
                M2 += qq.y + d * d * (na * npart / N);
                na = N;
            }
            p.stats_out[(m0 / BM) * p.Cout + n0 + tid] = make_float2(mean, M2);
        }
    }
}

}  // namespace
}  // namespace dlpm
