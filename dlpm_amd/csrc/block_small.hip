// block_small.hip -- whole UNet blocks of SMALL images in one launch, activations resident in LDS (round 4).
//
// Replaces, for 64-output-channel blocks on 8x8 / 4x4 images (the MNIST net's two coarsest levels and its middle block;
// dlpm/models/unet.py:105-196 ResBlock with use_scale_shift_norm, :199-250 AttentionBlock + QKVAttention):
//
//   ResBlock      h = conv3x3(silu(GN(x)));  h = conv3x3(silu(GN(h) * (1 + scale) + shift));  out = skip(x) + h
//                 (x may be the virtual concat [x0 | x1] of the output blocks; skip = identity or a 1x1 convolution)
//   AttentionBlock  qkv = conv1(GN(x));  a = softmax(q k^T / sqrt(ch)) v per head;  out = x + conv1(a)
//
// Why a different kernel family: at these sizes a launch of the tiled kernels is ONE workgroup's serial life -- prologue,
// 8 K-chunks with their barriers, epilogue: 23 of a launch's 26 us (profiles/r03/mnist_where_the_step_goes/) -- and a ResBlock
// is two of them plus two GroupNorm launches plus (output blocks) a 1x1 launch.  An 8x8 x 64-channel image is 16 KB; with its
// halo, both activated copies, the concat input and the hand-over buffer it is 131 KB: the whole block fits one CU's LDS, so ONE workgroup per image
// walks the block with no HBM round trip, no launch gap and no grid-wide dependency (GroupNorm is per image).  A sample's
// result depends on nothing but the sample: bits are independent of the batch.
//
// Work split (8 waves, 2 per SIMD): a convolution is the implicit GEMM  D[pixel][cout] = sum_{tap, cin} act[pixel + tap][cin] W
// on v_mfma_f32_16x16x4_f32; wave w (0..3) owns output channels 16 w .. 16 w + 15 for ALL pixels (MT = HW / 16 accumulator tiles),
// waves 4..7 own the same channels for the SECOND HALF of K (fragments) and hand their partial sums over through LDS (fixed
// order).  Weights are pre-arranged in B-fragment order Wf[cout / 16][fragment = tap * Cin / 16 + j][lane][4] and stream
// L2 -> registers through a ring, every fragment read by exactly one wave; the MFMA's free k-permutation (slot lk of MFMA e is
// channel 16 j + 4 lk + e) lets one ds_read_b128 of four consecutive channels feed four MFMAs.  The activated images live in LDS
// as zero-bordered halo tiles [(H + 2)(W + 2)][C + 4] (row pitch = 4 mod 32 words: the 16 pixels of an A read hit distinct
// bank quads).
#include "conv.h"

namespace dlpm {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int RS_NT = 512;     // threads per workgroup
constexpr int RS_CO = 64;      // output channels of a block
constexpr int RS_LDO = RS_CO + 4;

// The first R weight fragments of a wave's stream (wp already points at its first fragment + lane), requested as early as the
// caller can: the L2 round trip is then behind the phase that precedes the pass.
template <int R>
__device__ __forceinline__ void ring_fill(float4 (&ring)[R], const float4 *__restrict__ wp, int nf) {
#pragma unroll
    for (int r = 0; r < R; r++) ring[r] = wp[(int64_t)min(r, nf - 1) * 64];
}

// One pass of the implicit GEMM over `nf` weight fragments starting at global fragment index f_first (this wave's share):
// acc[mt] += A[pixels of tile mt][k] B[k][this wave's 16 channels].  halo: abuf is a halo tile and fragment (tap, j) reads the
// pixel shifted by the tap; otherwise (1x1) fragment j reads the pixel itself.  R = weight fragments in flight (divides nf); the ring
// arrives FILLED (ring_fill).  RELOAD = false: the ring holds ALL nf = R fragments and is left intact (the caller runs several
// passes over the same weights).
template <int MT, int R, bool RELOAD = true>
__device__ __forceinline__ void mfma_pass(floatx4 (&acc)[MT], float4 (&ring)[R], const float *abuf, const int (&abase)[MT], int LD, int WP,
                                          bool halo, const float4 *__restrict__ wp, int f_first, int nf, int jc_shift) {
    auto a_off = [&](int fg) {
        const int tap = fg >> jc_shift, j = fg & ((1 << jc_shift) - 1);
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;          // tap / 3, tap % 3 for tap < 9
        return (halo ? (ky * WP + kx) * LD : 0) + 16 * j;
    };
    // A fragments are read ONE weight fragment ahead (left to the compiler every ds_read sits in front of its MFMAs behind a wait)
    float4 an[MT];
    {
        const int o = a_off(f_first);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) an[mt] = *reinterpret_cast<const float4 *>(abuf + abase[mt] + o);
    }
    for (int f = 0; f < nf; f += R) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float4 bq = ring[r];
            if (RELOAD) ring[r] = wp[(int64_t)min(f + r + R, nf - 1) * 64];        // unconditional (clamped): no branch around a load
            float4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) a[mt] = an[mt];
            const int o = a_off(f_first + min(f + r + 1, nf - 1));
#pragma unroll
            for (int mt = 0; mt < MT; mt++) an[mt] = *reinterpret_cast<const float4 *>(abuf + abase[mt] + o);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, bq.x, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, bq.y, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, bq.z, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, bq.w, acc[mt], 0, 0, 0);
        }
    }
}

// sum over the lanes that share (channel pair, all pixel groups): xor 1 (the group's other channel), 16, 32 (the four row groups)
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

template <int HS>
__global__ void __launch_bounds__(RS_NT, 1) k_resblock_small(ResSmallLaunch p) {
    constexpr int HW = HS * HS, MT = HW / 16, WP = HS + 2, HP = WP * WP;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Cin = p.C0 + p.C1, LDI = Cin + 4, JC = Cin >> 4, jc_shift = 31 - __builtin_clz(JC);
    float *raw = sm;                         // [HW][LDI]       x as it arrives (skip path, GroupNorm statistics)
    float *a1 = raw + HW * LDI;              // [HP][LDI]       silu(GN1(x)), zero border
    float *a2 = a1 + HP * LDI;               // [HP][RS_LDO]    silu(GN2(h) (1 + scale) + shift), zero border
    float *red = a2 + HP * RS_LDO;           // [4][MT][4][64]  partial sums of waves 4..7
    float *cf = red + 4 * MT * 256;          // [2][Cin]        GroupNorm-1 coefficients
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w4 = wave & 3, half = wave >> 2, ln = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x;
    // weight streams of this wave: output channels 16 w4 .., fragment half `half`; the first ring of conv1 is requested now
    const int F1 = 9 * JC, nf1 = F1 >> 1;
    const float4 *wp1 = reinterpret_cast<const float4 *>(p.w1f) + ((int64_t)w4 * F1 + half * nf1) * 64 + lane;
    const float4 *wp2 = reinterpret_cast<const float4 *>(p.w2f) + ((int64_t)w4 * 36 + half * 18) * 64 + lane;
    float4 ring6[6];
    ring_fill<6>(ring6, wp1, nf1);
    // every per-channel parameter this thread will need, requested NOW: left where they are used, each is a dependent global load
    // at the head of a phase (GroupNorm weights after the statistics, biases in the epilogues, the emb row before a2): six exposed
    // L2 / HBM round trips on a workgroup whose whole life is ~20 us
    const int c_out = 16 * w4 + ln;      // this lane's output channel (waves 0..3 and 4..7 alike)
    const int cg1 = Cin >> 5, g1 = tid >> 4, i1 = tid & 15, c_gn1 = min(g1 * cg1 + i1, Cin - 1);
    const float pr_g1w = p.gn1_w[c_gn1], pr_g1b = p.gn1_b[c_gn1];
    const float pr_b1 = p.b1[c_out], pr_g2w = p.gn2_w[c_out], pr_g2b = p.gn2_b[c_out];
    const float *ssr = p.emb + (int64_t)b * p.emb_stride + p.emb_off;
    const float pr_sc = ssr[c_out], pr_sft = ssr[RS_CO + c_out];
    const float pr_bo = p.b2[c_out] + (p.wsf ? p.bs[c_out] : 0.f);

    // ---- phase 0: zero the halo tiles (their borders stay zero), x -> raw
    {
        float4 *z = reinterpret_cast<float4 *>(a1);
        for (int i = tid; i < (HP * LDI + HP * RS_LDO) / 4; i += RS_NT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int q4 = Cin >> 2;
        for (int i = tid; i < HW * q4; i += RS_NT) {
            const int pix = i / q4, c = (i - pix * q4) * 4;
            const float *src = c < p.C0 ? p.x0 + ((int64_t)b * HW + pix) * p.C0 + c : p.x1 + ((int64_t)b * HW + pix) * p.C1 + (c - p.C0);
            *reinterpret_cast<float4 *>(raw + pix * LDI + c) = *reinterpret_cast<const float4 *>(src);
        }
    }
    __syncthreads();
    // ---- phase 1: GroupNorm-1 statistics, two passes over the LDS copy: 32 groups x 16 threads
    {
        const int cg = Cin >> 5, g = tid >> 4, i = tid & 15;
        const float inv_n = 1.0f / (float)(cg * HW);
        float s = 0.f;
        for (int pp = i; pp < HW; pp += 16)
            for (int k = 0; k < cg; k++) s += raw[pp * LDI + g * cg + k];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        const float mean = s * inv_n;
        float v = 0.f;
        for (int pp = i; pp < HW; pp += 16)
            for (int k = 0; k < cg; k++) {
                const float d = raw[pp * LDI + g * cg + k] - mean;
                v = fmaf(d, d, v);
            }
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        const float rstd = 1.0f / sqrtf(v * inv_n + 1e-5f);
        if (i < cg) {
            const int c = g * cg + i;
            const float a = rstd * pr_g1w;
            cf[c] = a;
            cf[Cin + c] = pr_g1b - mean * a;
        }
    }
    __syncthreads();
    // ---- phase 2: a1 = silu(x A + B) into the halo tile's interior
    {
        const int q4 = Cin >> 2;
        for (int i = tid; i < HW * q4; i += RS_NT) {
            const int pix = i / q4, c = (i - pix * q4) * 4;
            const float4 x = *reinterpret_cast<const float4 *>(raw + pix * LDI + c);
            const float4 A = *reinterpret_cast<const float4 *>(cf + c), Bc = *reinterpret_cast<const float4 *>(cf + Cin + c);
            float4 v;
            v.x = silu_f(fmaf(x.x, A.x, Bc.x));
            v.y = silu_f(fmaf(x.y, A.y, Bc.y));
            v.z = silu_f(fmaf(x.z, A.z, Bc.z));
            v.w = silu_f(fmaf(x.w, A.w, Bc.w));
            const int y = pix / HS, xx = pix - y * HS;
            *reinterpret_cast<float4 *>(a1 + ((y + 1) * WP + xx + 1) * LDI + c) = v;
        }
    }
    __syncthreads();
    // ---- phase 3: conv1 (all waves, K split over the wave halves), partial sums of waves 4..7 through LDS
    floatx4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[mt][r] = 0.f;
    int ab1[MT], ab2[MT], abr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int pix = 16 * mt + ln, y = pix / HS, xx = pix - y * HS;
        ab1[mt] = (y * WP + xx) * LDI + 4 * lq;
        ab2[mt] = (y * WP + xx) * RS_LDO + 4 * lq;
        abr[mt] = pix * LDI + 4 * lq;
    }
    mfma_pass<MT, 6>(acc, ring6, a1, ab1, LDI, WP, true, wp1, half * nf1, nf1, jc_shift);
    float4 ring4[4];
    if (half) {
        if (p.wsf) ring_fill<4>(ring4, reinterpret_cast<const float4 *>(p.wsf) + ((int64_t)w4 * JC) * 64 + lane, JC);
    } else {
        ring_fill<6>(ring6, wp2, 18);       // conv2's first fragments travel while GroupNorm-2 runs
    }
    if (half) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[((w4 * MT + mt) * 4 + r) * 64 + lane] = acc[mt][r];
    }
    __syncthreads();
    if (!half) {
        // ---- phase 4a (waves 0..3): h = conv1 + bias; GroupNorm-2 (groups of two channels = lane pairs) with scale-shift; a2
        const float b1 = pr_b1;
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[mt][r] = (acc[mt][r] + red[((w4 * MT + mt) * 4 + r) * 64 + lane]) + b1;
                s += acc[mt][r];
            }
        const float inv_n = 1.0f / (float)(2 * HW);
        const float mean = group_sum(s) * inv_n;
        float v = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float d = acc[mt][r] - mean;
                v = fmaf(d, d, v);
            }
        const float rstd = 1.0f / sqrtf(group_sum(v) * inv_n + 1e-5f);
        float a = rstd * pr_g2w;
        float bb = pr_g2b - mean * a;
        const float sc = 1.0f + pr_sc, sft = pr_sft;
        a = a * sc;
        bb = fmaf(bb, sc, sft);
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pix = 16 * mt + 4 * lq + r, y = pix / HS, xx = pix - y * HS;
                a2[((y + 1) * WP + xx + 1) * RS_LDO + c_out] = silu_f(fmaf(acc[mt][r], a, bb));
                acc[mt][r] = 0.f;
            }
    } else {
        // ---- phase 4b (waves 4..7, meanwhile): the 1x1 skip convolution of the raw input opens their conv2 accumulators
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[mt][r] = 0.f;
        if (p.wsf) {
            const float4 *wp = reinterpret_cast<const float4 *>(p.wsf) + ((int64_t)w4 * JC) * 64 + lane;
            mfma_pass<MT, 4>(acc, ring4, raw, abr, LDI, WP, false, wp, 0, JC, jc_shift);
        }
        ring_fill<6>(ring6, wp2, 18);
    }
    __syncthreads();
    // ---- phase 5: conv2 over a2 (64 channels: 36 fragments, 18 per wave half)
    mfma_pass<MT, 6>(acc, ring6, a2, ab2, RS_LDO, WP, true, wp2, half * 18, 18, 2);
    if (half) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[((w4 * MT + mt) * 4 + r) * 64 + lane] = acc[mt][r];
    }
    __syncthreads();
    if (!half) {
        // ---- epilogue (waves 0..3): + bias (+ skip bias | + x), store NHWC, optional per-image GroupNorm statistics of the output
        const float bias = pr_bo;
        float *o = p.out + (int64_t)b * HW * RS_CO + c_out;
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pix = 16 * mt + 4 * lq + r;
                float v = (acc[mt][r] + red[((w4 * MT + mt) * 4 + r) * 64 + lane]) + bias;
                if (!p.wsf) v += raw[pix * LDI + c_out];
                acc[mt][r] = v;
                s += v;
                o[pix * RS_CO] = v;
            }
        if (p.stats_out) {
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / (float)HW);
            float m2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float d = acc[mt][r] - mean;
                    m2 = fmaf(d, d, m2);
                }
            m2 += __shfl_xor(m2, 16);
            m2 += __shfl_xor(m2, 32);
            if (lq == 0) p.stats_out[(int64_t)b * RS_CO + c_out] = make_float2(mean, m2);
        }
    }
}

// AttentionBlock of a small image (unet.py:199-250): GroupNorm -> qkv 1x1 -> per-head softmax(q k^T) v -> proj 1x1 -> + x, one
// workgroup per image, T = HS^2 tokens, 64 channels, 4 heads of 16 channels.  qkv channel order is the reference's head-major
// [head][q | k | v][16] (unet.py:224,243-244), i.e. 16-channel output tile nt = 3 head + {q, k, v}.  The attention core is
// attention.hip's (S^T = K Q^T tile by tile with the query on the lane, softmax in registers, probabilities already in the
// A-operand layout of P V), with q, k, v read from the LDS copy of qkv; q and k are pre-scaled by ch^(-1/4) as the reference does.
constexpr int AB_C = 64, AB_LD = AB_C + 4, AB_QLD = 3 * AB_C + 4, AB_HEADS = 4, AB_CH = 16;

// QKV_ONLY (16x16 images, T = 256: qkv would not fit LDS beside x): the kernel stops after qkv = conv1(GN(x)) and writes it
// (unscaled, NHWC [B][T][192]) for attention.hip's kernel -- still one launch instead of GroupNorm coefficients + a 1x1 GEMM of
// 128-pixel tiles that ran at 40 TFLOP/s (latency-bound: 40 us for 1.6 GFLOP; profiles/r04/mnist_fused_blocks/).
template <int HS, bool QKV_ONLY>
__global__ void __launch_bounds__(RS_NT, 1) k_attnblock_small(AttnSmallLaunch p) {
    constexpr int T = HS * HS, MT = T / 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *raw = sm;                       // [T][AB_LD]   x
    float *xn = QKV_ONLY ? raw : raw + T * AB_LD;      // [T][AB_LD]   GN(x) (in place when x is not needed again); later the attention output a
    float *qkvs = xn + T * AB_LD;          // [T][AB_QLD]  qkv (q, k scaled)
    float *cf = QKV_ONLY ? xn + T * AB_LD : qkvs + T * AB_QLD;         // [2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x;
    // parameters and the first weight fragments, requested at entry (see k_resblock_small)
    const int c_gn = min(2 * (tid >> 4) + (tid & 15), AB_C - 1);
    const float pr_gw = p.gn_w[c_gn], pr_gb = p.gn_b[c_gn];
    const float pr_bq0 = p.bqkv[16 * wave + ln], pr_bq1 = p.bqkv[16 * min(wave + 8, 3 * AB_HEADS - 1) + ln];
    const float pr_bp = p.bproj[16 * (wave & 3) + ln];
    const float4 *wq0 = reinterpret_cast<const float4 *>(p.wqkv) + (int64_t)wave * 4 * 64 + lane;
    float4 ringq[4];
    ring_fill<4>(ringq, wq0, 4);
    for (int i = tid; i < T * (AB_C / 4); i += RS_NT) {
        const int pix = i >> 4, c = (i & 15) * 4;
        *reinterpret_cast<float4 *>(raw + pix * AB_LD + c) = *reinterpret_cast<const float4 *>(p.x + ((int64_t)b * T + pix) * AB_C + c);
    }
    __syncthreads();
    {   // GroupNorm: 32 groups of two channels x 16 threads, two passes over the LDS copy
        const int g = tid >> 4, i = tid & 15;
        const float inv_n = 1.0f / (float)(2 * T);
        float s = 0.f;
        for (int pp = i; pp < T; pp += 16) s += raw[pp * AB_LD + 2 * g] + raw[pp * AB_LD + 2 * g + 1];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        const float mean = s * inv_n;
        float v = 0.f;
        for (int pp = i; pp < T; pp += 16) {
            const float d0 = raw[pp * AB_LD + 2 * g] - mean, d1 = raw[pp * AB_LD + 2 * g + 1] - mean;
            v = fmaf(d0, d0, v);
            v = fmaf(d1, d1, v);
        }
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        const float rstd = 1.0f / sqrtf(v * inv_n + 1e-5f);
        if (i < 2) {
            const int c = 2 * g + i;
            const float a = rstd * pr_gw;
            cf[c] = a;
            cf[AB_C + c] = pr_gb - mean * a;
        }
    }
    __syncthreads();
    for (int i = tid; i < T * (AB_C / 4); i += RS_NT) {
        const int pix = i >> 4, c = (i & 15) * 4;
        const float4 x = *reinterpret_cast<const float4 *>(raw + pix * AB_LD + c);
        const float4 A = *reinterpret_cast<const float4 *>(cf + c), Bc = *reinterpret_cast<const float4 *>(cf + AB_C + c);
        *reinterpret_cast<float4 *>(xn + pix * AB_LD + c) = make_float4(fmaf(x.x, A.x, Bc.x), fmaf(x.y, A.y, Bc.y), fmaf(x.z, A.z, Bc.z), fmaf(x.w, A.w, Bc.w));
    }
    __syncthreads();
    // (16x16 images: the 16 pixel tiles are walked in groups of four, so the accumulators and the A read-ahead stay at 4 tiles)
    constexpr int MG = MT > 4 ? 4 : MT;
    int abx[MG];
#pragma unroll
    for (int mt = 0; mt < MG; mt++) abx[mt] = (16 * mt + ln) * AB_LD + 4 * lq;
    // ---- qkv = conv1(GN(x)): 12 output tiles of 16 channels over the 8 waves
    const float scale = 0.5f;   // ch^(-1/4), ch = 16 (exact)
#pragma unroll 1
    for (int nt = wave; nt < 3 * AB_HEADS; nt += 8) {
        const float4 *wp = reinterpret_cast<const float4 *>(p.wqkv) + (int64_t)nt * 4 * 64 + lane;
        const int co = 16 * nt + ln;
        const float bias = nt < 8 ? pr_bq0 : pr_bq1;
        const bool scaled = (nt % 3) != 2;      // q and k tiles
#pragma unroll 1      // (unrolled over the four pixel-tile groups hipcc keeps 64 accumulators live: 256 VGPRs + 32 spills, 29 -> 43 us)
        for (int m0 = 0; m0 < MT; m0 += MG) {
            floatx4 acc[MG];
#pragma unroll
            for (int mt = 0; mt < MG; mt++) acc[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (nt >= 8 && m0 == 0) ring_fill<4>(ringq, wp, 4);       // (the first tile's fragments were requested at entry)
            mfma_pass<MG, 4, false>(acc, ringq, xn + 16 * m0 * AB_LD, abx, AB_LD, 0, false, wp, 0, 4, 2);
            if (QKV_ONLY) {
                float *o = p.out + ((int64_t)b * T + 16 * m0) * (3 * AB_C) + co;
#pragma unroll
                for (int mt = 0; mt < MG; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) o[(16 * mt + 4 * lq + r) * (3 * AB_C)] = acc[mt][r] + bias;
            } else {
#pragma unroll
                for (int mt = 0; mt < MG; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float v = acc[mt][r] + bias;
                        if (scaled) v *= scale;
                        qkvs[(16 * (m0 + mt) + 4 * lq + r) * AB_QLD + co] = v;
                    }
            }
        }
    }
    if constexpr (!QKV_ONLY) {
    __syncthreads();
    // ---- attention: units (head, query tile) over the waves; a -> xn
    for (int u = wave; u < AB_HEADS * MT; u += 8) {
        const int h = u / MT, t0 = (u - h * MT) * 16;
        const float *qb = qkvs + h * 3 * AB_CH, *kb = qb + AB_CH, *vb = qb + 2 * AB_CH;
        const float4 qv = *reinterpret_cast<const float4 *>(qb + (t0 + ln) * AB_QLD + 4 * lq);
        const float qf[4] = {qv.x, qv.y, qv.z, qv.w};
        floatx4 acc[MT];
#pragma unroll
        for (int j = 0; j < MT; j++) {
            acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
            const float4 kv = *reinterpret_cast<const float4 *>(kb + (16 * j + ln) * AB_QLD + 4 * lq);
            const float kf[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kk], qf[kk], acc[j], 0, 0, 0);
        }
        // acc[j][r] = S[query t0 + ln][key 16 j + 4 lq + r]: softmax over the keys = registers (j, r) and the four lane groups
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < MT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) mx = fmaxf(mx, acc[j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < MT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[j][r] = __expf(acc[j][r] - mx);
                sum += acc[j][r];
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float rsum = 1.0f / sum;
        floatx4 o = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float pv = acc[j][r] * rsum;
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vb[(16 * j + 4 * lq + r) * AB_QLD + ln], o, 0, 0, 0);
            }
        // D: row = query 4 lq + r of the tile, column = channel ln of the head
#pragma unroll
        for (int r = 0; r < 4; r++) xn[(t0 + 4 * lq + r) * AB_LD + h * AB_CH + ln] = o[r];
    }
    __syncthreads();
    // ---- out = x + proj(a): four output tiles on waves 0..3
    if (wave < 4) {
        floatx4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
        const float4 *wp = reinterpret_cast<const float4 *>(p.wproj) + (int64_t)wave * 4 * 64 + lane;
        float4 ring[4];
        ring_fill<4>(ring, wp, 4);
        mfma_pass<MT, 4>(acc, ring, xn, abx, AB_LD, 0, false, wp, 0, 4, 2);
        const int co = 16 * wave + ln;
        const float bias = pr_bp;
        float *o = p.out + (int64_t)b * T * AB_C + co;
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pix = 16 * mt + 4 * lq + r;
                const float v = raw[pix * AB_LD + co] + (acc[mt][r] + bias);
                acc[mt][r] = v;
                s += v;
                o[pix * AB_C] = v;
            }
        if (p.stats_out) {
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / (float)T);
            float m2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float d = acc[mt][r] - mean;
                    m2 = fmaf(d, d, m2);
                }
            m2 += __shfl_xor(m2, 16);
            m2 += __shfl_xor(m2, 32);
            if (lq == 0) p.stats_out[(int64_t)b * AB_C + co] = make_float2(mean, m2);
        }
    }
    }   // !QKV_ONLY
}

// ---- AttentionBlock of a 16x16 image (T = 256) in ONE launch (round 6).  Round 4 stopped at GroupNorm + qkv for this size (qkv of all four
// heads is 196 KB) and left attention.hip's kernel and the proj launch behind it: 25 + 60 + 14 us per block on the MNIST config, 0.50 of
// its 2.25 ms step, with qkv (50 MB) and the attention output (17 MB) crossing HBM between them.  Here a workgroup walks the HEADS:
//   xn  = GN(x) in LDS (70 KB);  per head h:  q | k | v of that head = xn Wqkv[3h .. 3h+2] (53 KB, q and k pre-scaled) -> barrier ->
//   a_h = softmax(q k^T) v for 16 query tiles (two per wave; the core of k_attnblock_small with 16 key tiles) -> barrier ->
//   out += a_h Wproj[:, 16h .. 16h+15] -- the proj GEMM accumulates over the heads in REGISTERS (a wave owns 32 pixels x 64 channels:
//   8 accumulator tiles), in the same channel order as a single K = 64 pass.  Nothing but x and the result touches HBM.
// (second form: q | k | v of a head in ONE pass over the normalised image -- every A fragment feeds six independent accumulators instead
//  of being re-read per 16-channel tile; v is kept TRANSPOSED, vT[channel][key], so that the P V products read one 16-byte word per key
//  tile where the first form read four scalars; S and P V run four independent accumulator chains instead of one: 90 -> see profiles/r06.)
constexpr int A16_T = 256, A16_QLD = 2 * AB_CH + 4, A16_VLD = A16_T + 4, A16_ALD = AB_CH + 4;

__global__ void __launch_bounds__(RS_NT, 1) k_attnblock16(AttnSmallLaunch p) {
    constexpr int T = A16_T, MT = T / 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *xn = sm;                        // [T][AB_LD]    x, then GN(x) in place
    float *qh = xn + T * AB_LD;            // [T][A16_QLD]  q | k of the current head (scaled)
    float *vT = qh + T * A16_QLD;          // [16][A16_VLD] v of the current head, transposed
    float *ah = vT + AB_CH * A16_VLD;      // [T][A16_ALD]  attention output of the current head
    float *cf = ah + T * A16_ALD;          // [2][64]
    float *red = qh;                       // epilogue: [2][8 waves][64] partial statistics (qh is dead by then)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x;
#ifdef DLPM_PHASE_TIMING      // developer builds: cycles of wave 0 per section (load + GroupNorm | q k v | attention | proj | epilogue), added at the end
    long long _t = clock64();
    unsigned long long _sec[5] = {0, 0, 0, 0, 0};
#define A16_MARK(i) do { const long long _n = clock64(); _sec[i] += (unsigned long long)(_n - _t); _t = _n; } while (0)
#else
#define A16_MARK(i)
#endif
    const int c_gn = min(2 * (tid >> 4) + (tid & 15), AB_C - 1);
    const float pr_gw = p.gn_w[c_gn], pr_gb = p.gn_b[c_gn];
    float pr_bp[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) pr_bp[nt] = p.bproj[16 * nt + ln];
    for (int i = tid; i < T * (AB_C / 4); i += RS_NT) {
        const int pix = i >> 4, c = (i & 15) * 4;
        *reinterpret_cast<float4 *>(xn + pix * AB_LD + c) = *reinterpret_cast<const float4 *>(p.x + ((int64_t)b * T + pix) * AB_C + c);
    }
    __syncthreads();
    {   // GroupNorm: 32 groups of two channels x 16 threads, two passes over the LDS copy (k_attnblock_small's)
        const int g = tid >> 4, i = tid & 15;
        const float inv_n = 1.0f / (float)(2 * T);
        float s = 0.f;
        for (int pp = i; pp < T; pp += 16) s += xn[pp * AB_LD + 2 * g] + xn[pp * AB_LD + 2 * g + 1];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        const float mean = s * inv_n;
        float v = 0.f;
        for (int pp = i; pp < T; pp += 16) {
            const float d0 = xn[pp * AB_LD + 2 * g] - mean, d1 = xn[pp * AB_LD + 2 * g + 1] - mean;
            v = fmaf(d0, d0, v);
            v = fmaf(d1, d1, v);
        }
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        const float rstd = 1.0f / sqrtf(v * inv_n + 1e-5f);
        if (i < 2) {
            const int c = 2 * g + i;
            const float a = rstd * pr_gw;
            cf[c] = a;
            cf[AB_C + c] = pr_gb - mean * a;
        }
    }
    __syncthreads();
    for (int i = tid; i < T * (AB_C / 4); i += RS_NT) {
        const int pix = i >> 4, c = (i & 15) * 4;
        const float4 x = *reinterpret_cast<const float4 *>(xn + pix * AB_LD + c);
        const float4 A = *reinterpret_cast<const float4 *>(cf + c), Bc = *reinterpret_cast<const float4 *>(cf + AB_C + c);
        *reinterpret_cast<float4 *>(xn + pix * AB_LD + c) = make_float4(fmaf(x.x, A.x, Bc.x), fmaf(x.y, A.y, Bc.y), fmaf(x.z, A.z, Bc.z), fmaf(x.w, A.w, Bc.w));
    }
    __syncthreads();
    A16_MARK(0);
    // this wave's rows of the two GEMMs: pixels 32 wave .. 32 wave + 31 (two pixel tiles)
    const int row0 = 32 * wave;
    int abx[2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) abx[mt] = (row0 + 16 * mt + ln) * AB_LD + 4 * lq;
    floatx4 pacc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) pacc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int h = 0; h < AB_HEADS; h++) {
        // ---- (1) q | k | v of head h for this wave's 32 pixels in one pass (qkv channel order is head-major [head][q | k | v][16]:
        // output tile nt = 3 h + qi): 12 weight fragments, six accumulator tiles
        {
            float4 wq[3][4];
#pragma unroll
            for (int qi = 0; qi < 3; qi++)
#pragma unroll
                for (int j = 0; j < 4; j++) wq[qi][j] = reinterpret_cast<const float4 *>(p.wqkv)[((int64_t)(3 * h + qi) * 4 + j) * 64 + lane];
            float bias[3];
#pragma unroll
            for (int qi = 0; qi < 3; qi++) bias[qi] = p.bqkv[16 * (3 * h + qi) + ln];
            floatx4 acc[2][3];
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int qi = 0; qi < 3; qi++) acc[mt][qi] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float4 a[2];
#pragma unroll
                for (int mt = 0; mt < 2; mt++) a[mt] = *reinterpret_cast<const float4 *>(xn + abx[mt] + 16 * j);
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int qi = 0; qi < 3; qi++) acc[mt][qi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, wq[qi][j].x, acc[mt][qi], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int qi = 0; qi < 3; qi++) acc[mt][qi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, wq[qi][j].y, acc[mt][qi], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int qi = 0; qi < 3; qi++) acc[mt][qi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, wq[qi][j].z, acc[mt][qi], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int qi = 0; qi < 3; qi++) acc[mt][qi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, wq[qi][j].w, acc[mt][qi], 0, 0, 0);
            }
            // q and k carry ch^(-1/4) = 0.5 (ch = 16, exact): rows of qh; v goes to vT[channel ln][pixels 4 lq .. 4 lq + 3 of the tile]
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    qh[(row0 + 16 * mt + 4 * lq + r) * A16_QLD + ln] = (acc[mt][0][r] + bias[0]) * 0.5f;
                    qh[(row0 + 16 * mt + 4 * lq + r) * A16_QLD + AB_CH + ln] = (acc[mt][1][r] + bias[1]) * 0.5f;
                }
                *reinterpret_cast<float4 *>(vT + ln * A16_VLD + row0 + 16 * mt + 4 * lq) =
                    make_float4(acc[mt][2][0] + bias[2], acc[mt][2][1] + bias[2], acc[mt][2][2] + bias[2], acc[mt][2][3] + bias[2]);
            }
        }
        // the head's slice of the proj weights (fragment j = h of each output tile), requested ahead of the attention
        float4 wpj[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wpj[nt] = reinterpret_cast<const float4 *>(p.wproj)[((int64_t)nt * 4 + h) * 64 + lane];
        __syncthreads();
        A16_MARK(1);
        // ---- (2) attention of head h: query tiles wave, wave + 8; a_h -> ah
#pragma unroll 1
        for (int u = wave; u < MT; u += 8) {
            const int t0 = 16 * u;
            const float *kb = qh + AB_CH;
            const float4 qv = *reinterpret_cast<const float4 *>(qh + (t0 + ln) * A16_QLD + 4 * lq);
            floatx4 acc[MT];
#pragma unroll
            for (int jg = 0; jg < MT; jg += 4) {      // four key tiles at a time: four independent accumulator chains
                float4 kv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc[jg + j] = floatx4{0.f, 0.f, 0.f, 0.f};
                    kv[j] = *reinterpret_cast<const float4 *>(kb + (16 * (jg + j) + ln) * A16_QLD + 4 * lq);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) acc[jg + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[j].x, qv.x, acc[jg + j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; j++) acc[jg + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[j].y, qv.y, acc[jg + j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; j++) acc[jg + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[j].z, qv.z, acc[jg + j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; j++) acc[jg + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[j].w, qv.w, acc[jg + j], 0, 0, 0);
            }
            // acc[j][r] = S[query t0 + ln][key 16 j + 4 lq + r]: softmax over the keys = registers (j, r) and the four lane groups
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < MT; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) mx = fmaxf(mx, acc[j][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < MT; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    acc[j][r] = __expf(acc[j][r] - mx);
                    sum += acc[j][r];
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float rsum = 1.0f / sum;
            // P V: key tile j contributes to chain j & 3 (four independent accumulators, summed pairwise at the end); the B operand of the
            // four MFMAs of a key tile = v of keys 16 j + 4 lq .. + 3, channel ln = one 16-byte word of vT
            floatx4 o4[4];
#pragma unroll
            for (int c = 0; c < 4; c++) o4[c] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jg = 0; jg < MT; jg += 4) {
                float4 vv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) vv[j] = *reinterpret_cast<const float4 *>(vT + ln * A16_VLD + 16 * (jg + j) + 4 * lq);
#pragma unroll
                for (int j = 0; j < 4; j++) o4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[jg + j][0] * rsum, vv[j].x, o4[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; j++) o4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[jg + j][1] * rsum, vv[j].y, o4[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; j++) o4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[jg + j][2] * rsum, vv[j].z, o4[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; j++) o4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[jg + j][3] * rsum, vv[j].w, o4[j], 0, 0, 0);
            }
            // D: row = query 4 lq + r of the tile, column = channel ln of the head
#pragma unroll
            for (int r = 0; r < 4; r++) ah[(t0 + 4 * lq + r) * A16_ALD + ln] = (o4[0][r] + o4[1][r]) + (o4[2][r] + o4[3][r]);
        }
        __syncthreads();
        A16_MARK(2);
        // ---- (3) out += a_h Wproj[:, 16 h ..]: one K = 16 fragment per (pixel tile, output tile)
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const float4 a = *reinterpret_cast<const float4 *>(ah + (row0 + 16 * mt + ln) * A16_ALD + 4 * lq);
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                pacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wpj[nt].x, pacc[mt][nt], 0, 0, 0);
                pacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wpj[nt].y, pacc[mt][nt], 0, 0, 0);
                pacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wpj[nt].z, pacc[mt][nt], 0, 0, 0);
                pacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wpj[nt].w, pacc[mt][nt], 0, 0, 0);
            }
        }
        A16_MARK(3);
        // (the next head's (1) overwrites qh -- every wave is past (2); its (2) overwrites ah behind the barrier that follows (1))
    }
    // ---- epilogue: out = x + (proj + bias); x is re-read (L2: this workgroup loaded it at entry)
    const float *xg = p.x + (int64_t)b * T * AB_C;
    float *og = p.out + (int64_t)b * T * AB_C;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pix = row0 + 16 * mt + 4 * lq + r, co = 16 * nt + ln;
                const float v = xg[pix * AB_C + co] + (pacc[mt][nt][r] + pr_bp[nt]);
                pacc[mt][nt][r] = v;
                s[nt] += v;
                og[pix * AB_C + co] = v;
            }
    if (p.stats_out) {   // per-image (mean, centred sum of squares) per channel: the eight waves' partials meet in LDS, fixed order
        __syncthreads();     // (every wave is done with ah / qh)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            s[nt] += __shfl_xor(s[nt], 16);
            s[nt] += __shfl_xor(s[nt], 32);
            if (lq == 0) red[wave * AB_C + 16 * nt + ln] = s[nt];
        }
        __syncthreads();
        float mean[4], m2[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; w++) t += red[w * AB_C + 16 * nt + ln];
            mean[nt] = t * (1.0f / (float)T);
            m2[nt] = 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float d = pacc[mt][nt][r] - mean[nt];
                    m2[nt] = fmaf(d, d, m2[nt]);
                }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            m2[nt] += __shfl_xor(m2[nt], 16);
            m2[nt] += __shfl_xor(m2[nt], 32);
            if (lq == 0) red[(8 + wave) * AB_C + 16 * nt + ln] = m2[nt];
        }
        __syncthreads();
        if (tid < AB_C) {
            float t = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                t += red[w * AB_C + tid];
                q += red[(8 + w) * AB_C + tid];
            }
            p.stats_out[(int64_t)b * AB_C + tid] = make_float2(t * (1.0f / (float)T), q);
        }
    }
    A16_MARK(4);
#ifdef DLPM_PHASE_TIMING
    if (p.phase && tid == 0) {
#pragma unroll
        for (int i = 0; i < 5; i++) atomicAdd(p.phase + i, _sec[i]);
        atomicAdd(p.phase + 5, 1ull);
    }
#endif
}

// OIHW (taps = ks * ks) -> Wf[cout / 16][fragment = tap * Cin / 16 + j][lane = lk * 16 + li][e]:
//   W[cout = 16 w + li][cin = 16 j + 4 lk + e][tap]
__global__ void k_relayout_weight_small(const float *oihw, float *dst, int Cout, int Cin, int taps) {
    const int64_t total = (int64_t)Cout * Cin * taps;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const int JC = Cin >> 4, F = taps * JC;
    const int64_t rest = i >> 8;
    const int f = (int)(rest % F), w = (int)(rest / F);
    const int tap = f / JC, j = f - tap * JC;
    const int cin = 16 * j + 4 * (lane >> 4) + e, cout = 16 * w + (lane & 15);
    dst[i] = oihw[((int64_t)cout * Cin + cin) * taps + tap];
}

size_t res_small_lds_bytes(int HS, int Cin) {
    const int HW = HS * HS, HP = (HS + 2) * (HS + 2), LDI = Cin + 4, MT = HW / 16;
    return (size_t)(HW * LDI + HP * LDI + HP * RS_LDO + 4 * MT * 256 + 2 * Cin) * sizeof(float);
}

}  // namespace

bool small_blocks_enabled() {   // DLPM_NO_FUSED_BLOCKS=1: every block through the tiled per-layer kernels (A/B runs)
    static int off = -1;
    if (off < 0) { const char *e = getenv("DLPM_NO_FUSED_BLOCKS"); off = (e && e[0] == '1') ? 1 : 0; }
    return !off;
}

bool small_weight_ok(int Cout, int Cin, int ks) {
    return (ks == 1 || ks == 3) && ((Cout == RS_CO && (Cin == 64 || Cin == 128)) || (Cout == 3 * RS_CO && Cin == 64 && ks == 1));
}

int64_t small_weight_floats(int Cout, int Cin, int ks) { return (int64_t)Cout * Cin * ks * ks; }

int relayout_weight_small(const float *oihw_dev, float *dst_dev, int Cout, int Cin, int ks, hipStream_t st) {
    const int64_t n = (int64_t)Cout * Cin * ks * ks;
    k_relayout_weight_small<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin, ks * ks);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

bool res_small_ok(const ResSmallLaunch &r) {
    if (!small_blocks_enabled() || !r.w1f || !r.w2f) return false;
    const int Cin = r.C0 + r.C1;
    if (r.H != r.W || (r.H != 8 && r.H != 4)) return false;
    if ((Cin != 64 && Cin != 128) || (r.C0 & 3) || (r.C1 & 3)) return false;
    if ((Cin == 64) == (r.wsf != nullptr)) return false;      // 64 -> 64: identity skip; 128 -> 64: the 1x1 skip convolution
    return true;
}

bool attn_small_ok(const AttnSmallLaunch &a) {
    return small_blocks_enabled() && a.wqkv && a.wproj && a.C == AB_C && a.heads == AB_HEADS && a.H == a.W && (a.H == 8 || a.H == 4);
}

// 16x16 images, round 6: the whole block in one launch, one head at a time (k_attnblock16).  DLPM_ATTN16=0 (A/B runs) restores round 4's
// GroupNorm + qkv launch followed by attention.hip's kernel and the proj GEMM.
bool attn16_ok(const AttnSmallLaunch &a) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("DLPM_ATTN16"); on = e ? atoi(e) : 1; }
    return on && small_blocks_enabled() && a.wqkv && a.wproj && a.C == AB_C && a.heads == AB_HEADS && a.H == 16 && a.W == 16;
}

int launch_attnblock16(const AttnSmallLaunch &a, hipStream_t st) {
    if (!attn16_ok(a)) {
        set_error("launch_attnblock16: unsupported block shape");
        return DLPM_ERR_UNSUPPORTED;
    }
    const int T = A16_T;
    const double fl = 2.0 * a.B * T * (4.0 * AB_C * AB_C + 2.0 * T * AB_C);
    ProfScope ps("attnblock16:H16", fl, 4.0 * (2.0 * a.B * T * AB_C + 4.0 * AB_C * AB_C), st);
    const size_t lds = (size_t)(T * AB_LD + T * A16_QLD + AB_CH * A16_VLD + T * A16_ALD + 2 * AB_C) * sizeof(float);
    int e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_attnblock16), 160 * 1024);
    if (e != DLPM_OK) return e;
#ifdef DLPM_PHASE_TIMING
    const_cast<AttnSmallLaunch &>(a).phase = phase_buffer();
#endif
    k_attnblock16<<<(unsigned)a.B, RS_NT, lds, st>>>(a);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

// 16x16 images: GroupNorm + qkv in one launch per image (a.out = qkv [B][256][192]); attention and proj_out stay separate launches
bool gnqkv_small_ok(const AttnSmallLaunch &a) {
    return small_blocks_enabled() && a.wqkv && a.C == AB_C && a.H == 16 && a.W == 16;
}

int launch_gnqkv_small(const AttnSmallLaunch &a, hipStream_t st) {
    if (!gnqkv_small_ok(a)) {
        set_error("launch_gnqkv_small: unsupported block shape");
        return DLPM_ERR_UNSUPPORTED;
    }
    const int T = a.H * a.W;
    ProfScope ps("gn_qkv_small:H16", 2.0 * a.B * T * 3.0 * AB_C * AB_C, 4.0 * (4.0 * a.B * T * AB_C + 3.0 * AB_C * AB_C), st);
    const size_t lds = (size_t)(T * AB_LD + 2 * AB_C) * sizeof(float);
    int e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_attnblock_small<16, true>), 160 * 1024);
    if (e != DLPM_OK) return e;
    k_attnblock_small<16, true><<<(unsigned)a.B, RS_NT, lds, st>>>(a);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int launch_attnblock_small(const AttnSmallLaunch &a, hipStream_t st) {
    if (!attn_small_ok(a)) {
        set_error("launch_attnblock_small: unsupported block shape");
        return DLPM_ERR_UNSUPPORTED;
    }
    const int T = a.H * a.W;
    const double fl = 2.0 * a.B * T * (4.0 * AB_C * AB_C + 2.0 * T * AB_C);
    ProfScope ps(a.H == 8 ? "attnblock_small:H8" : "attnblock_small:H4", fl, 4.0 * (2.0 * a.B * T * AB_C + 4.0 * AB_C * AB_C), st);
    const size_t lds = (size_t)(2 * T * AB_LD + T * AB_QLD + 2 * AB_C) * sizeof(float);
    if (a.H == 8) {
        int e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_attnblock_small<8, false>), 160 * 1024);
        if (e != DLPM_OK) return e;
        k_attnblock_small<8, false><<<(unsigned)a.B, RS_NT, lds, st>>>(a);
    } else {
        int e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_attnblock_small<4, false>), 160 * 1024);
        if (e != DLPM_OK) return e;
        k_attnblock_small<4, false><<<(unsigned)a.B, RS_NT, lds, st>>>(a);
    }
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int launch_resblock_small(const ResSmallLaunch &r, hipStream_t st) {
    if (!res_small_ok(r)) {
        set_error("launch_resblock_small: unsupported block shape");
        return DLPM_ERR_UNSUPPORTED;
    }
    const int Cin = r.C0 + r.C1, HW = r.H * r.W;
    const double fl = 2.0 * r.B * HW * RS_CO * (9.0 * Cin + 9.0 * RS_CO + (r.wsf ? Cin : 0));
    const double by = 4.0 * ((double)r.B * HW * (Cin + RS_CO) + RS_CO * (9.0 * Cin + 9.0 * RS_CO + (r.wsf ? Cin : 0)));
    ProfScope ps(r.H == 8 ? "resblock_small:H8" : "resblock_small:H4", fl, by, st);
    const size_t lds = res_small_lds_bytes(r.H, Cin);
    if (r.H == 8) {
        int e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_resblock_small<8>), 160 * 1024);
        if (e != DLPM_OK) return e;
        k_resblock_small<8><<<(unsigned)r.B, RS_NT, lds, st>>>(r);
    } else {
        int e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_resblock_small<4>), 160 * 1024);
        if (e != DLPM_OK) return e;
        k_resblock_small<4><<<(unsigned)r.B, RS_NT, lds, st>>>(r);
    }
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
