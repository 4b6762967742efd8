// conv_direct.hip -- direct convolution for the shapes the MFMA tiling does not cover: the stem
// (Cin = image channels, reads the caller's NCHW state), the head (Cout = image channels, writes
// NCHW eps, fused GroupNorm affine + SiLU on its input) and any channel count that is not a
// multiple of 32.  These are ~0.1 % of the path's FLOPs (unet.py:347,435); the kernel is a plain
// one-thread-per-output FMA loop with channel-fastest (coalesced) weight and output accesses.
#include <algorithm>

#include "conv.h"
#include "philox.h"

namespace dlpm {
namespace {

__global__ void __launch_bounds__(256) k_conv_direct(ConvLaunch p) {
    const int Cin = p.C0 + p.C1;
    const int HWo = p.Hout * p.Wout;
    const int64_t total = (int64_t)p.B * HWo * p.Cout;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i % p.Cout);
    const int64_t m = i / p.Cout;
    const int b = (int)(m / HWo);
    const int rem = (int)(m - (int64_t)b * HWo);
    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
    const int pad = p.ks >> 1;
    const int Hi = p.ups ? p.Hin * 2 : p.Hin, Wi = p.ups ? p.Win * 2 : p.Win;
    const float *cA = p.coefA ? p.coefA + (int64_t)b * Cin : nullptr;
    const float *cB = p.coefB ? p.coefB + (int64_t)b * Cin : nullptr;

    float acc = 0.f;
    for (int ky = 0; ky < p.ks; ky++) {
        const int iy = oy * p.stride + ky - pad;
        if (iy < 0 || iy >= Hi) continue;
        const int sy = p.ups ? (iy >> 1) : iy;
        for (int kx = 0; kx < p.ks; kx++) {
            const int ix = ox * p.stride + kx - pad;
            if (ix < 0 || ix >= Wi) continue;
            const int sx = p.ups ? (ix >> 1) : ix;
            const float *w = p.w + ((int64_t)(ky * p.ks + kx) * Cin) * p.Cout + n;
            for (int c = 0; c < Cin; c++) {
                float v;
                if (p.in_nchw) {
                    v = p.src0[(((int64_t)b * Cin + c) * p.Hin + sy) * p.Win + sx];
                } else {
                    const int64_t pix = ((int64_t)b * p.Hin + sy) * p.Win + sx;
                    v = (c < p.C0) ? p.src0[pix * p.C0 + c] : p.src1[pix * p.C1 + (c - p.C0)];
                }
                if (cA) v = fmaf(v, cA[c], cB[c]);
                if (p.act_silu) v = silu_f(v);
                acc = fmaf(v, w[(int64_t)c * p.Cout], acc);
            }
        }
    }
    if (p.bias) acc += p.bias[n];
    if (p.res0) acc += (n < p.R0) ? p.res0[m * p.R0 + n] : p.res1[m * (p.Cout - p.R0) + (n - p.R0)];
    if (p.out_nchw) p.out[(((int64_t)b * p.Cout + n) * p.Hout + oy) * p.Wout + ox] = acc;
    else p.out[m * p.Cout + n] = acc;
}

// ---------------------------------------------------------------------------------------------
// Head convolution (unet.py:435, `out.2`): 3x3, Cin = model_channels -> Cout = image channels (<= 4), fused GroupNorm
// affine + SiLU on the input, NCHW output -- and, in the sampler, the reverse update x <- (x - c_eps eps) / gamma +
// c_noise z applied to its own result (GenerativeLevyProcess.py:225-239, dlpm.py:272-278): eps never goes to HBM.
//
// On the MFMA path it ran as a 32-wide N tile with 29 of 32 columns empty.  Here a thread owns FOUR consecutive pixels of
// a row and all <= 4 output channels (16 accumulators): per (input row, channel quad) it reads 6 activated input pixels
// and 12 weight quads from LDS for 192 FMAs, so the loop is VALU-bound (the one-pixel-per-thread generation of this
// kernel issued 360 LDS reads per pixel and 32-channel chunk).  Four consecutive pixels of one channel are
// also exactly one Philox counter of the update (element quad), so the fused epilogue draws the same normals as
// k_update_rows and reproduces it bit for bit.  A workgroup owns TH whole rows of one image (512 pixels, 128 threads);
// per 8-channel chunk the activated halo goes through LDS once (12 floats per pixel, odd row pitch: lanes walk down
// the rows, 16 rows = 16 distinct bank groups), weights [tap][cin][4] with uniform (broadcast) reads.  What bounds it
// now is VALU work: 36 FMAs (a quarter of them on the padding channel when Cout = 3) and one SiLU per input element.
// ---------------------------------------------------------------------------------------------
// HEAD_C channels per chunk; HEAD_C + 4 floats per halo pixel in LDS (an odd number of 16-byte units)
template <int HEAD_C>
__global__ void __launch_bounds__(128) k_conv3x3_head(ConvLaunch p, HeadUpdate u, int TH, int Wp) {
    constexpr int HEAD_LD = HEAD_C + 4, NQ = HEAD_C / 4;
    extern __shared__ __attribute__((aligned(16))) float hs[];
    const int W = p.Wout, H = p.Hout, W2 = W + 2, hp = (TH + 2) * Wp, nthr = blockDim.x;
    float *Cf = hs + hp * HEAD_LD;   // this chunk's GroupNorm coefficients [A 16 | B 16]
    float *Ws = Cf + 2 * HEAD_C;     // this chunk's weights [tap][16][4]
    const int tpi = H / TH;
    const int b = blockIdx.x / tpi, y0 = (blockIdx.x % tpi) * TH;
    const int tid = threadIdx.x, ty = tid % TH, tx4 = tid / TH;
    const int Cin = p.C0, nch = Cin / HEAD_C;
    const float4 *__restrict__ w4 = reinterpret_cast<const float4 *>(p.w_small);
    float acc[4][4];   // [pixel][output channel]
#pragma unroll
    for (int co = 0; co < 4; co++) {
        const float bv = (co < p.Cout && p.bias) ? p.bias[co] : 0.f;
#pragma unroll
        for (int px = 0; px < 4; px++) acc[px][co] = bv;
    }
    const bool has_coef = p.coefA != nullptr;
    for (int chunk = 0; chunk < nch; chunk++) {
        __syncthreads();   // the previous chunk's tile is no longer read
        if (has_coef && tid < 2 * NQ) {
            const int isb = tid / NQ, q = tid % NQ;
            *reinterpret_cast<float4 *>(Cf + isb * HEAD_C + q * 4) =
                *reinterpret_cast<const float4 *>((isb ? p.coefB : p.coefA) + (int64_t)b * Cin + chunk * HEAD_C + q * 4);
        }
        for (int idx = tid; idx < 9 * HEAD_C; idx += nthr) {
            const int tap = idx / HEAD_C, ci = idx - tap * HEAD_C;
            *reinterpret_cast<float4 *>(Ws + idx * 4) = w4[(int64_t)tap * Cin + chunk * HEAD_C + ci];
        }
        __syncthreads();
        {
            // halo pixels flat over the threads (channel quad fixed per thread: its two coefficient quads are loop invariants);
            // pixel -> (row, column) by a float reciprocal, exact for these few hundred pixels, instead of an integer division
            const int q = tid % NQ, pstep = nthr / NQ;
            float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_coef) {
                ca = *reinterpret_cast<const float4 *>(Cf + q * 4);
                cb = *reinterpret_cast<const float4 *>(Cf + HEAD_C + q * 4);
            }
            const float rW2 = 1.0f / (float)W2;
            const float *sbase = p.src0 + (int64_t)b * H * W * Cin + chunk * HEAD_C + q * 4;
            for (int pix = tid / NQ; pix < (TH + 2) * W2; pix += pstep) {
                const int hy = (int)(((float)pix + 0.5f) * rW2), hx = pix - hy * W2;
                const int iy = y0 + hy - 1, ix = hx - 1;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    x = *reinterpret_cast<const float4 *>(sbase + (int64_t)(iy * W + ix) * Cin);
                    if (has_coef) {
                        x.x = fmaf(x.x, ca.x, cb.x);
                        x.y = fmaf(x.y, ca.y, cb.y);
                        x.z = fmaf(x.z, ca.z, cb.z);
                        x.w = fmaf(x.w, ca.w, cb.w);
                    }
                    if (p.act_silu) {
                        x.x = silu_f(x.x);
                        x.y = silu_f(x.y);
                        x.z = silu_f(x.z);
                        x.w = silu_f(x.w);
                    }
                }
                *reinterpret_cast<float4 *>(hs + (hy * Wp + hx) * HEAD_LD + q * 4) = x;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const float *row = hs + ((ty + ky) * Wp + 4 * tx4) * HEAD_LD;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                float4 a[6];
#pragma unroll
                for (int j = 0; j < 6; j++) a[j] = *reinterpret_cast<const float4 *>(row + j * HEAD_LD + q * 4);
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const float4 *wt = reinterpret_cast<const float4 *>(Ws) + ((ky * 3 + kx) * HEAD_C + q * 4);
                    const float4 w0 = wt[0], w1 = wt[1], w2 = wt[2], w3 = wt[3];
#pragma unroll
                    for (int px = 0; px < 4; px++) {
                        const float4 v = a[px + kx];
                        acc[px][0] = fmaf(v.x, w0.x, acc[px][0]); acc[px][1] = fmaf(v.x, w0.y, acc[px][1]);
                        acc[px][2] = fmaf(v.x, w0.z, acc[px][2]); acc[px][3] = fmaf(v.x, w0.w, acc[px][3]);
                        acc[px][0] = fmaf(v.y, w1.x, acc[px][0]); acc[px][1] = fmaf(v.y, w1.y, acc[px][1]);
                        acc[px][2] = fmaf(v.y, w1.z, acc[px][2]); acc[px][3] = fmaf(v.y, w1.w, acc[px][3]);
                        acc[px][0] = fmaf(v.z, w2.x, acc[px][0]); acc[px][1] = fmaf(v.z, w2.y, acc[px][1]);
                        acc[px][2] = fmaf(v.z, w2.z, acc[px][2]); acc[px][3] = fmaf(v.z, w2.w, acc[px][3]);
                        acc[px][0] = fmaf(v.w, w3.x, acc[px][0]); acc[px][1] = fmaf(v.w, w3.y, acc[px][1]);
                        acc[px][2] = fmaf(v.w, w3.z, acc[px][2]); acc[px][3] = fmaf(v.w, w3.w, acc[px][3]);
                    }
                }
            }
        }
    }
    const int64_t HW = (int64_t)H * W;
    const int64_t pix = (int64_t)(y0 + ty) * W + 4 * tx4;
    if (u.x) {
        // the reverse update on this thread's element quads: same arithmetic, same Philox counters as k_update_rows (noise.hip)
        const int t = *u.t;
        const float g = u.g[t], rg = 1.0f / g;
        const float ce = u.c_eps[(int64_t)t * u.B + b], cn = u.c_noise[(int64_t)t * u.B + b];
        const uint64_t seed = u.key ? u.key[0] : u.seed;
        const uint64_t gidx = (uint64_t)((u.key ? (int64_t)u.key[1] : u.sample_offset) + b);
        float *hr = u.hist_pp ? *u.hist_pp : nullptr;
        const int64_t D = (int64_t)p.Cout * HW;
        if (hr) hr += ((int64_t)(u.T - t) * u.B + b) * D;
#pragma unroll
        for (int co = 0; co < 4; co++)
            if (co < p.Cout) {
                const int64_t e0 = (int64_t)co * HW + pix;       // element index inside the sample (NCHW); e0 % 4 == 0
                const float4 x = *reinterpret_cast<const float4 *>(u.x + (int64_t)b * D + e0);
                float4 z;
                if (u.z) z = *reinterpret_cast<const float4 *>(u.z + (int64_t)b * D + e0);
                else z = (cn != 0.0f) ? philox_normal4(seed, gidx, (uint32_t)(e0 >> 2), kPurposeStepZ, (uint32_t)t) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 o;
                o.x = fmaf(cn, z.x, div_by(x.x - ce * acc[0][co], g, rg));
                o.y = fmaf(cn, z.y, div_by(x.y - ce * acc[1][co], g, rg));
                o.z = fmaf(cn, z.z, div_by(x.z - ce * acc[2][co], g, rg));
                o.w = fmaf(cn, z.w, div_by(x.w - ce * acc[3][co], g, rg));
                *reinterpret_cast<float4 *>(u.x + (int64_t)b * D + e0) = o;
                if (hr) *reinterpret_cast<float4 *>(hr + e0) = o;
                if (u.eps_out) *reinterpret_cast<float4 *>(u.eps_out + (int64_t)b * D + e0) = make_float4(acc[0][co], acc[1][co], acc[2][co], acc[3][co]);
            }
        return;
    }
#pragma unroll
    for (int co = 0; co < 4; co++)
        if (co < p.Cout) {
            if (p.out_nchw) {
                *reinterpret_cast<float4 *>(p.out + ((int64_t)b * p.Cout + co) * HW + pix) = make_float4(acc[0][co], acc[1][co], acc[2][co], acc[3][co]);
            } else {
#pragma unroll
                for (int px = 0; px < 4; px++) p.out[((int64_t)b * HW + pix + px) * p.Cout + co] = acc[px][co];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Head convolution as a GEMM plus a gather (round 3; k_conv3x3_head above stays as the fallback for shapes this does not take).
//
// y[p, co] = b[co] + sum_tap sum_ci act(x)[p + off(tap), ci] w[co, ci, tap]  =  b[co] + sum_tap P[p + off(tap), tap * Cout + co]
// with P = act(x) W',  W'[ci][n = tap * Cout + co] = w[co, ci, tap]:  a 1x1 convolution with 9 Cout (27, padded to 32) output
// channels -- the fp32 MFMA kernel with the fused GroupNorm affine + SiLU, every input line fetched exactly once, 27 of 32 MFMA
// columns live -- followed by a 9-point gather-add over P.  (k_conv3x3_head stages the 128-channel input in 8-channel chunks:
// 32 bytes of every 128-byte line per pass, 4x the input in HBM/MALL traffic, 0.42 ms for a 0.09-ms read; as an N = 32 MFMA
// tile of the 3x3 implicit GEMM 29 of 32 columns were empty.)  Zero padding applies to the ACTIVATED input, i.e. taps whose source
// pixel lies outside the image are simply absent from the sum.
//
// The gather kernel is the sampler's fused update kernel: it reads P (4 * 9 Cout B per pixel) and x, applies
// x <- (x - c_eps eps) / gamma + c_noise z with the Philox counters of k_update_rows (a thread owns four consecutive pixels of
// one row = element quads of the NCHW state), and writes x: eps never reaches HBM.  HBM-bound.
// ---------------------------------------------------------------------------------------------
constexpr int HG_LD = 29;   // LDS words per halo pixel (odd: the 9 Cout values of neighbouring pixels fall into different banks; >= 4 ceil(27 / 4) = 28)

// Persistent workgroups (HG_MINBLOCKS = 3 per CU) walk the tiles: the P rows of tile i + 1 are in flight into registers while tile i is summed,
// updated and stored, so the HBM latency of the 134-MB read is off the critical path.  Tile ids are dealt so that the row tiles of one
// image run on ONE XCD back to back (ids t, t + 8, t + 16, ...: the halo rows a tile shares with its neighbours are L2 hits).
#ifndef HG_MINBLOCKS
#define HG_MINBLOCKS 3   // workgroups per CU the register budget is set for (4 fits the LDS but needs 128 VGPRs: 14 spill, 88 instead of 38 us)
#endif
template <int COUT>
__global__ void __launch_bounds__(256, HG_MINBLOCKS) k_head_gather(const float *__restrict__ P, int Np, const float *__restrict__ bias, float *out,
                                                     int out_nchw, HeadUpdate u, int B, int H, int W, int TH) {
    extern __shared__ __attribute__((aligned(16))) float hg[];
    constexpr int NV = 9 * COUT, NQ4 = (NV + 3) / 4, NPF = 10;   // NPF: float4 a thread holds of the next tile (launcher: items <= 256 NPF)
    static_assert(4 * NQ4 <= HG_LD, "a pixel's tap values must fit its LDS slot");
    const int tpi = H / TH, W2 = W + 2, tid = threadIdx.x;
    const int items = (TH + 2) * W2 * NQ4;
    const int ntiles = ((B + 7) / 8) * 8 * tpi;                   // ids beyond the batch are skipped
    const int nq = W >> 2, per_co = TH * nq;
    const bool computes = tid < COUT * per_co;
    const int co = computes ? tid / per_co : 0, rq = tid - co * per_co, r = rq / nq, q = rq - r * nq;
    const float bv = bias ? bias[co] : 0.f;
    const int64_t HW = (int64_t)H * W, D = (int64_t)COUT * HW;
    // per-thread geometry of its load items (the same for every tile), one packed word each: hy [31:24] | hx [23:16] | k [15:12] | valid [0]
    uint32_t geo[NPF];
#pragma unroll
    for (int j = 0; j < NPF; j++) {
        const int it = tid + j * 256;
        const int k = it % NQ4, pix = it / NQ4;
        const int hy = pix / W2, hx = pix - hy * W2;
        geo[j] = ((uint32_t)hy << 24) | ((uint32_t)hx << 16) | ((uint32_t)k << 12) | (it < items ? 1u : 0u);
    }
    auto tile_of = [&](int t, int &b, int &y0) {
        const int grp = t / (8 * tpi), rem = t - grp * 8 * tpi;
        b = grp * 8 + (rem & 7);
        y0 = (rem >> 3) * TH;
    };
    float4 pf[NPF];
    auto prefetch = [&](int t) {
        int b, y0;
        tile_of(t, b, y0);
        const float *Pb = P + (int64_t)min(b, B - 1) * HW * Np;
#pragma unroll
        for (int j = 0; j < NPF; j++) {
            const int hy = geo[j] >> 24, hx = (geo[j] >> 16) & 255, k = (geo[j] >> 12) & 15;
            const int iy = y0 + hy - 1, ix = hx - 1;
            const bool ok = (geo[j] & 1u) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && b < B;
            const float4 v = *reinterpret_cast<const float4 *>(Pb + (int64_t)(ok ? iy * W + ix : 0) * Np + 4 * k);   // unconditional load
            pf[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int t = blockIdx.x;
    if (t < ntiles) prefetch(t);
    for (; t < ntiles; t += gridDim.x) {
        int b, y0;
        tile_of(t, b, y0);
#pragma unroll
        for (int j = 0; j < NPF; j++)
            if (geo[j] & 1u) {
                const int hy = geo[j] >> 24, hx = (geo[j] >> 16) & 255, k = (geo[j] >> 12) & 15;
                float *d = hg + (hy * W2 + hx) * HG_LD + 4 * k;
                d[0] = pf[j].x; d[1] = pf[j].y; d[2] = pf[j].z; d[3] = pf[j].w;   // (HG_LD is odd: no 16-byte stores)
            }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) prefetch(t + gridDim.x);   // in flight while this tile is summed and stored
        if (computes && b < B) {
            float acc[4] = {bv, bv, bv, bv};
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const float *src = hg + ((r + ky) * W2 + 4 * q + kx) * HG_LD + (ky * 3 + kx) * COUT + co;
#pragma unroll
                    for (int px = 0; px < 4; px++) acc[px] += src[px * HG_LD];
                }
            const int64_t pix = (int64_t)(y0 + r) * W + 4 * q;
            const int64_t e0 = (int64_t)co * HW + pix;       // element index inside the sample (NCHW); e0 % 4 == 0
            if (u.x) {
                // the reverse update on this thread's element quad: same arithmetic, same Philox counters as k_update_rows (noise.hip)
                const int tt = *u.t;
                const float g = u.g[tt], rg = 1.0f / g;
                const float ce = u.c_eps[(int64_t)tt * u.B + b], cn = u.c_noise[(int64_t)tt * u.B + b];
                const uint64_t seed = u.key ? u.key[0] : u.seed;
                const uint64_t gidx = (uint64_t)((u.key ? (int64_t)u.key[1] : u.sample_offset) + b);
                float *hr = u.hist_pp ? *u.hist_pp : nullptr;
                if (hr) hr += ((int64_t)(u.T - tt) * u.B + b) * D;
                const float4 x = *reinterpret_cast<const float4 *>(u.x + (int64_t)b * D + e0);
                float4 z;
                if (u.z) z = *reinterpret_cast<const float4 *>(u.z + (int64_t)b * D + e0);
                else z = (cn != 0.0f) ? philox_normal4(seed, gidx, (uint32_t)(e0 >> 2), kPurposeStepZ, (uint32_t)tt) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 o;
                o.x = fmaf(cn, z.x, div_by(x.x - ce * acc[0], g, rg));
                o.y = fmaf(cn, z.y, div_by(x.y - ce * acc[1], g, rg));
                o.z = fmaf(cn, z.z, div_by(x.z - ce * acc[2], g, rg));
                o.w = fmaf(cn, z.w, div_by(x.w - ce * acc[3], g, rg));
                *reinterpret_cast<float4 *>(u.x + (int64_t)b * D + e0) = o;
                if (hr) *reinterpret_cast<float4 *>(hr + e0) = o;
                if (u.eps_out) *reinterpret_cast<float4 *>(u.eps_out + (int64_t)b * D + e0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else if (out_nchw) {
                *reinterpret_cast<float4 *>(out + (int64_t)b * D + e0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else {
#pragma unroll
                for (int px = 0; px < 4; px++) out[((int64_t)b * HW + pix + px) * COUT + co] = acc[px];
            }
        }
        __syncthreads();   // the tile's LDS image is dead: the next one may overwrite it
    }
}

// OIHW (3x3) -> W' [Np][Cin] (the 1x1 kernel's [Cout][Cin] layout), row n = tap * Cout + co, rows beyond 9 Cout zero
__global__ void k_relayout_weight_head_taps(const float *oihw, float *dst, int Cout, int Cin, int Np) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Np * Cin) return;
    const int n = i / Cin, ci = i - n * Cin;
    const int tap = n / Cout, co = n - tap * Cout;
    dst[i] = n < 9 * Cout ? oihw[((int64_t)co * Cin + ci) * 9 + tap] : 0.f;
}

// OIHW (3x3) -> [tap][cin][4] with the output channel padded to 4
__global__ void k_relayout_weight_head(const float *oihw, float *dst, int Cout, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * Cin * 4) return;
    const int co = i & 3, ci = (i >> 2) % Cin, tap = (i >> 2) / Cin;
    dst[i] = co < Cout ? oihw[((int64_t)co * Cin + ci) * 9 + tap] : 0.f;
}

}  // namespace

int launch_conv_direct(const ConvLaunch &c, hipStream_t st) {
    const int64_t total = (int64_t)c.B * c.Hout * c.Wout * c.Cout;
    const double K = (double)(c.C0 + c.C1) * c.ks * c.ks;
    ProfScope ps("conv_direct", 2.0 * total * K, 4.0 * ((double)c.B * c.Hin * c.Win * (c.C0 + c.C1) + K * c.Cout + total), st);
    k_conv_direct<<<(unsigned)ceil_div(total, 256), 256, 0, st>>>(c);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm

namespace dlpm {

static int head_rows(const ConvLaunch &c) { return std::min(c.Hout, 512 / c.Wout); }   // rows of a workgroup's tile (<= 512 pixels)

bool head_conv_ok(const ConvLaunch &c) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("DLPM_NO_HEADK"); off = (e && e[0] == '1') ? 1 : 0; }
    if (off || !c.w_small || c.ks != 3 || c.stride != 1 || c.ups || c.in_nchw || c.C1 != 0 || c.res0) return false;
    if (c.Cout < 1 || c.Cout > 4 || c.C0 % 16 != 0 || c.Hin != c.Hout || c.Win != c.Wout) return false;
    const int W = c.Wout, H = c.Hout;
    if (W < 8 || W > 64 || (W & 3) || 512 % W != 0) return false;
    const int TH = head_rows(c), nthr = TH * (W / 4);
    return H % TH == 0 && nthr % 64 == 0 && nthr <= 128;
}

int launch_conv_head(const ConvLaunch &c, const HeadUpdate *hu, hipStream_t st) {
    const int W = c.Wout, H = c.Hout, TH = head_rows(c);
    const int Wp = (W + 2) | 1;   // odd halo row pitch: 5 Wp 16-byte units per row, odd -> lanes walking down the rows spread over the banks
    const int hp = (TH + 2) * Wp;
    const int64_t M = (int64_t)c.B * H * W;
    // algorithmic bytes: the input once, the output once -- or, with the update fused, the state read and written
    const double bytes = 4.0 * ((double)M * c.C0 + (double)M * c.Cout * (hu ? 2 + (hu->z ? 1 : 0) + (hu->eps_out ? 1 : 0) : 1));
    ProfScope ps(hu ? "conv3x3_head+update" : "conv3x3_head", 2.0 * M * c.Cout * 9.0 * c.C0, bytes, st);
    HeadUpdate none{};
    // 8 channels per chunk: 31 KB of LDS per workgroup, five workgroups (ten waves) on a CU; 16 channels per chunk (53 KB,
    // six waves) measured 19 % slower on the CIFAR head
    constexpr int HC = 8;
    const size_t shmem = (size_t)(hp * (HC + 4) + 2 * HC + 9 * HC * 4) * sizeof(float);
    int r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_conv3x3_head<HC>), 64 * 1024);
    if (r != DLPM_OK) return r;
    k_conv3x3_head<HC><<<(unsigned)(c.B * (H / TH)), TH * (W / 4), shmem, st>>>(c, hu ? *hu : none, TH, Wp);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int head_taps_rows(int Cout) { return (9 * Cout + 31) / 32 * 32; }

static int head_gather_rows(const ConvLaunch &c) {   // rows of a gather workgroup's tile: (TH + 2) (W + 2) HG_LD floats of LDS <= 48 KB,
    int th = c.Hout;                                     // Cout TH W / 4 compute threads <= 256, (TH + 2) (W + 2) ceil(9 Cout / 4) load items <= 2560
    const int nq4 = (9 * c.Cout + 3) / 4;                // (without the last limit the 64x64 head stopped at TH = 4 = 2772 items, head_gemm_ok
    while (th > 1 && ((th + 2) * (c.Wout + 2) * HG_LD * 4 > 48 * 1024 || c.Cout * th * (c.Wout / 4) > 256 ||   //  said no and the net fell
                      (th + 2) * (c.Wout + 2) * nq4 > 256 * 10))                                                   //  back to k_conv3x3_head: ADVICE r03)
        th >>= 1;
    return th;
}

bool head_gemm_ok(const ConvLaunch &c) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("DLPM_NO_HEAD_GEMM"); off = (e && e[0] == '1') ? 1 : 0; }
    if (off || !c.w_taps || c.ks != 3 || c.stride != 1 || c.ups || c.in_nchw || c.C1 != 0 || c.res0) return false;
    if (c.Cout < 1 || c.Cout > 3 || c.C0 % 32 != 0 || c.Hin != c.Hout || c.Win != c.Wout) return false;
    const int W = c.Wout, H = c.Hout;
    if (W < 16 || W > 64 || (W & 3) || (H & (H - 1)) || ((int64_t)H * W) % 128 != 0) return false;
    const int TH = head_gather_rows(c);
    const int items = (TH + 2) * (W + 2) * ((9 * c.Cout + 3) / 4);
    return H % TH == 0 && (TH + 2) * (W + 2) * HG_LD * 4 <= 48 * 1024 && c.Cout * TH * (W / 4) <= 256 && items <= 256 * 10;
}

int64_t head_gemm_scratch_floats(const ConvLaunch &c) { return (int64_t)c.B * c.Hout * c.Wout * head_taps_rows(c.Cout); }

int launch_conv_head_gemm(const ConvLaunch &c, const HeadUpdate *hu, float *P, hipStream_t st) {
    const int Np = head_taps_rows(c.Cout);
    ConvLaunch g = c;
    g.ks = 1; g.w = c.w_taps; g.w_frag = nullptr; g.w_wino = nullptr; g.w_wino4 = nullptr; g.w_small = nullptr; g.w_split = nullptr;
    g.w_taps = nullptr; g.bias = nullptr; g.out = P; g.out_nchw = 0; g.Cout = Np; g.stats_out = nullptr; g.gemm = DLPM_GEMM_F32;
    int r = launch_conv_igemm(g, st);
    if (r != DLPM_OK) return r;
    const int W = c.Wout, H = c.Hout, TH = head_gather_rows(c);
    const int64_t M = (int64_t)c.B * H * W;
    // algorithmic bytes of the gather: the 9 Cout partial products per pixel once + the output -- or the state read and written
    const double bytes = 4.0 * ((double)M * 9 * c.Cout + (double)M * c.Cout * (hu ? 2 + (hu->z ? 1 : 0) + (hu->eps_out ? 1 : 0) : 1));
    ProfScope ps(hu ? "head_gather+update" : "head_gather", 2.0 * M * c.Cout * 9.0, bytes, st);
    HeadUpdate none{};
    const size_t shmem = (size_t)(TH + 2) * (W + 2) * HG_LD * sizeof(float);
    const int ntiles = ((c.B + 7) / 8) * 8 * (H / TH);
    const unsigned grid = (unsigned)std::min(ntiles, 256 * HG_MINBLOCKS), nthr = 256;   // persistent: four workgroups per CU walk the tiles
#define DLPM_HG(CO)                                                                                                          \
    do {                                                                                                                     \
        r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_head_gather<CO>), 64 * 1024);                               \
        if (r != DLPM_OK) return r;                                                                                          \
        k_head_gather<CO><<<grid, nthr, shmem, st>>>(P, Np, c.bias, c.out, c.out_nchw, hu ? *hu : none, c.B, H, W, TH);      \
    } while (0)
    if (c.Cout == 1) DLPM_HG(1);
    else if (c.Cout == 2) DLPM_HG(2);
    else DLPM_HG(3);
#undef DLPM_HG
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int relayout_weight_head_taps(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st) {
    const int Np = head_taps_rows(Cout);
    k_relayout_weight_head_taps<<<(unsigned)ceil_div(Np * Cin, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin, Np);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int relayout_weight_head(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st) {
    k_relayout_weight_head<<<(unsigned)ceil_div(9 * Cin * 4, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
