// conv_wino.hip -- 3x3 stride-1 convolution as Winograd F(2x2, 3x3) on the gfx950 fp32 MFMA.
//
// Replaces the same F.conv2d call sites as k_conv3x3_halo_ws (dlpm/models/unet.py:143,157 and the Upsample conv
// :64-75) with 16 multiplies per 2x2 output tile and (cin, cout) pair instead of 36:
//     Y = A^T [ sum_c (G g_c G^T) . (B^T d_c B) ] A          (Lavin & Gray, arXiv:1509.09308, F(2x2,3x3))
// In GEMM form: for each of the 16 transform positions p,  M_p[tile, cout] = sum_cin V_p[tile, cin] U_p[cin, cout].
//
// Since conv_wino4.hip (F(4x4,3x3)) took over the layers at 8x8 pixels and above with Cout % 128 == 0, this kernel
// serves what that one does not: 4x4-pixel tensors and Cout = 64 (MNIST / toy nets).  Two earlier generations of this
// kernel (4 waves with all 16 positions per wave; the same software-pipelined inside the wave) are described in
// DESIGN.md section 3 with their measurements and were removed from the source once k_conv3x3_wino_q had replaced them.
//
// Per 8-channel phase of the input:
//   1. the raw halo patch is loaded to registers ahead of time and stored to LDS with the fused GroupNorm affine + SiLU
//      applied once per element (zero padding = zeros AFTER activation); virtual concat and nearest-x2 upsampling are
//      address arithmetic, as in the halo kernel;
//   2. threads (tile, channel quad, row of V) compute V = B^T d B into LDS;
//   3. the waves run the 16 position GEMMs on v_mfma_f32_32x32x2_f32: A fragments by ds_read_b128 from V, B fragments
//      (U, pre-transformed at finalize and stored in fragment order) streamed from L2 into a register ring.
// Numerics: fp32 throughout; F(2x2,3x3)'s transforms only add and halve, the measured deviation from the reference
// through the whole CIFAR UNet is 3e-6 (tolerance 1e-4).
#include <cstdlib>

#include "conv.h"

namespace dlpm {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int WN = 64;        // output channels a layer must be a multiple of
constexpr int WKC = 16;       // input channels a layer must be a multiple of
constexpr int RAW_MAXPIX = 576;   // halo pixels per phase of a 64-tile block (half of it for 32 tiles)
constexpr int WGRP = 32;      // weight fragment groups (256 floats) per 16 input channels and 32 output channels

// ---------------------------------------------------------------------------------------------
// k_conv3x3_wino_q: eight waves per workgroup, two per SIMD.
// Measured on gfx950 (tools/mb/mfma_valu_overlap.hip): a wave's own VALU instructions do NOT run in the shadow of its
// MFMAs (68 cycles per MFMA alone, 79 with four v_fma behind each one) -- only ANOTHER wave of the SIMD can use the
// vector ALU while the matrix pipe is busy.  So the 256-KB accumulator block of the workgroup is split over 8 waves:
// wave (ph, wm, wn) holds the 8 positions of transform rows a = 2ph, 2ph+1 for 32 tiles x 32 channels (128 AGPRs, two
// waves per SIMD), and the staging / transform work of one wave overlaps the MFMAs of its SIMD neighbour.  The output
// transform Y = A^T M A splits by rows: each wave reduces its 8 accumulators to 4 partial 2x2-output tiles in registers,
// the ph = 1 waves pass theirs through LDS, the ph = 0 waves add them (fixed order: deterministic).
// Phases are software-pipelined: double-buffered raw / V, one barrier per 8-channel chunk.
// ---------------------------------------------------------------------------------------------
constexpr int QRING = 8;      // weight prefetch ring of the 8-wave kernel (groups of 4 MFMAs); must divide the 8 groups of a phase

// MT = tiles per workgroup: 64 (x 64 output channels) or 32 (x 128 channels).  The accumulator block is 256 KB either
// way; the 32-tile shape stages a smaller halo patch per MFMA (180-288 instead of 324-576 pixels, for twice the
// channels) and transforms half as many tiles, i.e. ~40 % less VALU work per MFMA -- which matters because VALU time
// is added to MFMA time on this chip -- at the price of streaming every weight fragment for one MFMA tile only.
// NW = waves per workgroup: 8 (one workgroup per CU, 256-KB accumulator block) or 4 (MT = 32 only: 32 tiles x 64
// channels, 128-KB block, TWO workgroups per CU, so that one's prologue / epilogue / barriers overlap the other's MFMAs).
// KC = input channels per phase: 8, or 16 (MT = 32, NW = 8 only: the smaller V / raw tiles leave room for it): half as many
// barriers and phase start-ups per MFMA.
template <bool UPS, int MT, int NW = 8, int ABL = 0, int KC = 8>   // ABL (DLPM_WINO_ABLATIONS builds): 1 no S, 2 no X, 4 no raw loads, 8 no barrier, 16 no weight loads, 32 no MFMA
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) k_conv3x3_wino_q(ConvLaunch p, int bh, int bw, int nimg) {
    static_assert(NW == 8 || (NW == 4 && MT == 32), "wave layout");
    static_assert(KC == 8 || (KC == 16 && MT == 32 && NW == 8), "chunk");
    constexpr int NT = NW * 64;                      // threads
    constexpr int NQ = NW == 8 ? 4096 / MT : 64;     // output channels per workgroup
    constexpr int RAWPIX = MT == 64 ? RAW_MAXPIX : RAW_MAXPIX / 2;
    constexpr int NQD = KC / 4;                      // channel quads per pixel and chunk
    constexpr int QNIT = (RAWPIX * NQD + NT - 1) / NT; // staging items per thread
    constexpr int PVLD = KC + 4, PRLD = KC + 4;      // padded V / raw rows: conflict-free ds_read_b128 over consecutive tiles
    constexpr int CFS = 16 * 2 * KC;                 // floats per coefficient slot: [16 images][2][KC]
    constexpr int NJQ = KC / 8;                      // float4 fragments per position and lane
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *V = wsm;                                  // [2][16][MT][PVLD]
    float *raw = wsm + 2 * 16 * MT * PVLD;           // [2][RAWPIX][PRLD]
    float *Cf = raw + 2 * RAWPIX * PRLD;             // [2 slots][16 images][2][8]

    DLPM_PHASE_DECL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int ph = NW == 8 ? wave >> 2 : wave >> 1;
    const int wn = MT == 64 ? (wave & 1) : (NW == 8 ? (wave & 3) : (wave & 1)), wm = MT == 64 ? ((wave >> 1) & 1) : 0;
    const int W = p.Wout, H = p.Hout, TW = W >> 1, TH = H >> 1;
    const int Ws = UPS ? (W >> 1) : W, Hs = UPS ? (H >> 1) : H;
    const int Cin = p.C0 + p.C1;
    int nch = Cin / KC, kb = 0;
    unsigned bid = blockIdx.x, nblk = gridDim.x;      // (unsigned, like the grid built-ins: the prologue's divisions stay what they were)
    if (p.ksplit > 1) {
        // split-K (round 6): grid copy ks walks chunks [kb, kb + nch) and writes its partial outputs behind those of the copies before
        // it; such a launch carries no bias / residual / statistics (launch_splitk_reduce adds them)
        nblk = gridDim.x / p.ksplit;
        const int ks = (int)(blockIdx.x / nblk);
        bid = blockIdx.x - (unsigned)ks * nblk;
        nch /= p.ksplit;
        kb = ks * nch;
        p.out += (int64_t)ks * p.B * H * W * p.Cout;
    }
    // n-tile-major grid: all workgroups in flight stream the SAME half of the Winograd-domain weights (2.1 MB for a
    // 256 x 256 layer: fits the 4-MB L2 of an XCD; both halves together do not)
    const int ntn = p.Cout / NQ;
    const int nmb = nblk / ntn;
    const int mb = bid % nmb, n0 = (bid / nmb) * NQ;
    int img0, ty0, tx0, blk_in_img = 0;
    if (nimg == 1) {
        const int bpr = TW / bw, bpi = (TH / bh) * bpr;
        img0 = mb / bpi;
        blk_in_img = mb - img0 * bpi;
        ty0 = (blk_in_img / bpr) * bh;
        tx0 = (blk_in_img % bpr) * bw;
    } else {
        img0 = mb * nimg;
        ty0 = tx0 = 0;
    }
    const int RH = UPS ? bh + 2 : 2 * bh + 2, RW = UPS ? bw + 2 : 2 * bw + 2;
    const int oy = UPS ? ty0 - 1 : 2 * ty0 - 1, ox = UPS ? tx0 - 1 : 2 * tx0 - 1;
    const int rpi = RH * RW, npix = nimg * rpi;

    // ---- raw staging: item = (pixel, channel quad of the chunk)
    const int squad = tid & (NQD - 1);
    int off[QNIT], cfo[QNIT];
#pragma unroll
    for (int it = 0; it < QNIT; it++) {
        const int pix = it * (NT / NQD) + (tid / NQD);
        const int img = min(pix / rpi, nimg - 1), r = pix - img * rpi;
        const int ry = r / RW, rx = r - ry * RW;
        const int iy = oy + ry, ix = ox + rx;
        const bool pad = iy < 0 || iy >= Hs || ix < 0 || ix >= Ws || (img0 + img) >= p.B;
        off[it] = pix >= npix ? -2 : (pad ? -1 : (((img0 + img) * Hs + iy) * Ws + ix));
        cfo[it] = img * 2 * KC + squad * 4;
    }
    const bool has_coef = p.coefA != nullptr;
    const int cf_img = tid / (2 * NQD), cf_isb = (tid / NQD) & 1;
    const bool cf_mine = has_coef && tid < nimg * 2 * NQD;
    const float *cf_base = has_coef ? ((cf_isb ? p.coefB : p.coefA) + (int64_t)min(img0 + cf_img, p.B - 1) * Cin + kb * KC + squad * 4) : nullptr;
    float4 xr[QNIT], cfr = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_raw = [&](int chunk) {
        const int c = (kb + chunk) * KC + squad * 4;
        const bool first = c < p.C0;
        const float *sb = first ? p.src0 + c : p.src1 + (c - p.C0);
        const int ld = first ? p.C0 : p.C1;
#pragma unroll
        for (int it = 0; it < QNIT; it++) xr[it] = *reinterpret_cast<const float4 *>(sb + (int64_t)max(off[it], 0) * ld);
    };
    auto load_coef = [&](int chunk) {
        if (cf_mine) cfr = *reinterpret_cast<const float4 *>(cf_base + chunk * KC);
    };
    auto store_coef = [&](int slot) {
        if (cf_mine) *reinterpret_cast<float4 *>(Cf + slot * CFS + cf_img * 2 * KC + cf_isb * KC + squad * 4) = cfr;
    };
    auto store_raw_item = [&](int slot, int it) {
        float *rb = raw + slot * RAWPIX * PRLD;
        {
            if (off[it] == -2) return;
            float4 x = xr[it];
            if (has_coef) {
                const float4 ca = *reinterpret_cast<const float4 *>(Cf + slot * CFS + cfo[it]);
                const float4 cb = *reinterpret_cast<const float4 *>(Cf + slot * CFS + cfo[it] + KC);
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
            if (off[it] < 0) x = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding applies AFTER the activation
            *reinterpret_cast<float4 *>(rb + (it * (NT / NQD) + (tid / NQD)) * PRLD + squad * 4) = x;
        }
    };
    auto store_raw = [&](int slot) {
#pragma unroll
        for (int it = 0; it < QNIT; it++) store_raw_item(slot, it);
    };

    // ---- input transform: thread = (tile, quad) over lanes x one row r of V (wave-uniform): V[r][.] = (B^T d)[r] B
    //   row 0: d0 - d2   row 1: d1 + d2   row 2: d2 - d1   row 3: d1 - d3
    const int trow = wave & 3;
    const bool xform_mine = MT * NQD * 4 >= NT || wave < 4;   // MT tiles x NQD quads x 4 rows items (256: waves 0..3 only)
    const int rA = trow == 0 ? 0 : trow == 2 ? 2 : 1, rB = trow == 0 ? 2 : trow == 1 ? 2 : trow == 2 ? 1 : 3;
    const float tsg = trow == 1 ? 1.f : -1.f;
    int roA, roB, coloff[4], vofs;
    {
        const int pair = MT * NQD > 64 ? (wave >> 2) * 64 + lane : lane;
        const int tile = pair / NQD, tquad = pair & (NQD - 1);
        const int timg = tile / (bh * bw), r = tile - timg * (bh * bw);
        const int ty = r / bw, tx = r - ty * bw;
#pragma unroll
        for (int k = 0; k < 4; k++) coloff[k] = (UPS ? tx + ((k + 1) >> 1) : 2 * tx + k) * PRLD;
        const int rrA = UPS ? ty + ((rA + 1) >> 1) : 2 * ty + rA, rrB = UPS ? ty + ((rB + 1) >> 1) : 2 * ty + rB;
        roA = (timg * rpi + rrA * RW) * PRLD + tquad * 4;
        roB = (timg * rpi + rrB * RW) * PRLD + tquad * 4;
        vofs = (trow * 4) * MT * PVLD + tile * PVLD + tquad * 4;
    }
    auto transform = [&](int slot) {
        if (!xform_mine) return;
        const float *rb = raw + slot * RAWPIX * PRLD;
        float *vb = V + slot * 16 * MT * PVLD + vofs;
        float4 t[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float4 a = *reinterpret_cast<const float4 *>(rb + roA + coloff[c]);
            const float4 b = *reinterpret_cast<const float4 *>(rb + roB + coloff[c]);
            t[c] = make_float4(fmaf(tsg, b.x, a.x), fmaf(tsg, b.y, a.y), fmaf(tsg, b.z, a.z), fmaf(tsg, b.w, a.w));
        }
        *reinterpret_cast<float4 *>(vb + 0 * MT * PVLD) = make_float4(t[0].x - t[2].x, t[0].y - t[2].y, t[0].z - t[2].z, t[0].w - t[2].w);
        *reinterpret_cast<float4 *>(vb + 1 * MT * PVLD) = make_float4(t[1].x + t[2].x, t[1].y + t[2].y, t[1].z + t[2].z, t[1].w + t[2].w);
        *reinterpret_cast<float4 *>(vb + 2 * MT * PVLD) = make_float4(t[2].x - t[1].x, t[2].y - t[1].y, t[2].z - t[1].z, t[2].w - t[1].w);
        *reinterpret_cast<float4 *>(vb + 3 * MT * PVLD) = make_float4(t[1].x - t[3].x, t[1].y - t[3].y, t[1].z - t[3].z, t[1].w - t[3].w);
    };

    // ---- weight stream of this wave: Wf[nb][ph][chunk][8 positions][lane][4]
    const float4 *__restrict__ wbase = reinterpret_cast<const float4 *>(p.w_wino) + lane;
    int64_t woff = ((int64_t)(((n0 >> 5) + wn) * 2 + ph) * (Cin / KC) + kb) * 8 * NJQ * 64;
    constexpr int AHEAD = QRING - 1;
    float4 bq[QRING];
    const float *asrc = V + (ph * 8) * MT * PVLD + (wm * 32 + l31) * PVLD + kh * (KC / 2);

    floatx16 acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;

    // ---- prologue: S(0), S(1), X(0), G(2) and the coefficient slots.  One workgroup per CU: nothing hides these
    // latencies, so everything the first two chunks need is requested at once (one exposed round trip instead of three).
    const int last = nch - 1;
    float4 xr1[QNIT], cfr1 = make_float4(0.f, 0.f, 0.f, 0.f), cfr2 = cfr1;
    load_raw(0);
    {
        const int c = (kb + min(1, last)) * KC + squad * 4;
        const bool first = c < p.C0;
        const float *sb = first ? p.src0 + c : p.src1 + (c - p.C0);
        const int ld = first ? p.C0 : p.C1;
#pragma unroll
        for (int it = 0; it < QNIT; it++) xr1[it] = *reinterpret_cast<const float4 *>(sb + (int64_t)max(off[it], 0) * ld);
    }
    load_coef(0);
    if (cf_mine) {
        cfr1 = *reinterpret_cast<const float4 *>(cf_base + min(1, last) * KC);
        cfr2 = *reinterpret_cast<const float4 *>(cf_base + min(2, last) * KC);
    }
#pragma unroll
    for (int a = 0; a < AHEAD; a++) bq[a] = wbase[woff + a * 64];
    store_coef(0);
    cfr = cfr1;
    store_coef(1);
    __syncthreads();
    store_raw(0);
#pragma unroll
    for (int it = 0; it < QNIT; it++) xr[it] = xr1[it];
    store_raw(1);
    load_raw(min(2, last));
    __syncthreads();
    transform(0);
    cfr = cfr2;
    store_coef(0);
    __syncthreads();
    DLPM_PHASE(p, 8);

    for (int chunk = 0; chunk < nch; chunk++) {
        const int cur = chunk & 1, nxt = cur ^ 1;
        const float *ab = asrc + cur * 16 * MT * PVLD;
        load_coef(min(chunk + 3, last));
        // Half of the MFMAs, the side work, the other half.  (Measured, tools/mb/mfma_valu_2waves.hip: on gfx950 VALU
        // instructions do not overlap MFMAs of the SAME SIMD even from another wave -- 3460 cycles for 2048 cycles of MFMA
        // in one wave next to 1233 cycles of v_fma in the other -- so the side work's VALU time is simply added; what the
        // second wave does hide is latency: LDS, global loads, barriers.  Running the two waves of a SIMD in opposite
        // order measured 4 % slower.)
#pragma unroll
        for (int g = 0; g < 8 * NJQ; g++) {
            const int q = g / NJQ, jq = g % NJQ;
            // the side work is spread over the phase (one piece behind each of the first positions): bunched in the
            // middle of the phase it measured 2.5 % slower (both waves of a SIMD reach it together and the matrix pipe idles)
            // placement sweep (DLPM_BUILD_DEFS="WQ_S0=.. WQ_X=.."): staging from an odd position on is 3-6 % faster than
            // from an even one (the compiler pairs the MFMAs of positions 2k, 2k+1); X position does not matter
#ifndef WQ_S0
#define WQ_S0 3
#endif
#ifndef WQ_X
#define WQ_X 6
#endif
            if (!(ABL & 1) && g % NJQ == 0 && g / NJQ >= WQ_S0 && g / NJQ < WQ_S0 + QNIT)
                store_raw_item(cur, g / NJQ - WQ_S0);                   // S(chunk+2): raw[cur] was read by X(chunk), a barrier ago
            if (g == (WQ_S0 + QNIT) * NJQ && !(ABL & 4)) load_raw(min(chunk + 3, last));   // G(chunk+3)
            if (g == WQ_X * NJQ && !(ABL & 2)) transform(nxt);          // X(chunk+1): raw[nxt] -> V[nxt]
            if (!(ABL & 16)) bq[(g + AHEAD) % QRING] = wbase[woff + AHEAD * 64];
            woff += 64;
            const float4 af = *reinterpret_cast<const float4 *>(ab + q * MT * PVLD + jq * 4);
            const float4 b = bq[g % QRING];
            if (!(ABL & 32)) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, b.x, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, b.y, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, b.z, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, b.w, acc[q], 0, 0, 0);
            }
        }
        store_coef(nxt);
        if (!(ABL & 8)) __syncthreads();
    }
    if (ABL & 8) __syncthreads();
    DLPM_PHASE(p, 9);

    // ---- epilogue addressing + residual prefetch (8 rows per thread)
    constexpr int C4N = NQ / 4, NRG = NT / C4N;    // float4 columns per row, row groups
    constexpr int NPASS = 4 * MT / NRG;             // rows per thread (8)
    const int c4 = tid % C4N, rg = tid / C4N;
    const int n = n0 + c4 * 4;
    const int R1 = p.Cout - p.R0;
    int64_t mrow[NPASS];
    float4 resq[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; pass++) {
        const int row = pass * NRG + rg;
        const int tile = row >> 2, i = (row >> 1) & 1, j = row & 1;
        const int timg = tile / (bh * bw), r = tile - timg * (bh * bw);
        const int ty = r / bw, tx = r - ty * bw;
        const bool ok = img0 + timg < p.B;
        const int64_t m = ((int64_t)min(img0 + timg, p.B - 1) * H + 2 * (ty0 + ty) + i) * W + 2 * (tx0 + tx) + j;
        mrow[pass] = ok ? m : -1;
        if (p.res0)
            resq[pass] = (n < p.R0) ? *reinterpret_cast<const float4 *>(p.res0 + m * p.R0 + n)
                                    : *reinterpret_cast<const float4 *>(p.res1 + m * R1 + (n - p.R0));
    }
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias = *reinterpret_cast<const float4 *>(p.bias + n);

    // ---- partial output transform in registers: this wave holds M[a][b] for a = 2ph, 2ph+1 (acc[(a - 2ph)*4 + b])
    //   s0[a] = M[a][0] + M[a][1] + M[a][2],  s1[a] = M[a][1] - M[a][2] - M[a][3]
    //   ph 0: P[0][j] = sj[0] + sj[1], P[1][j] = sj[1]        ph 1: P[0][j] = sj[2], P[1][j] = -sj[2] - sj[3]
    floatx16 P[4];   // index i*2 + j
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float s0a = acc[0][r] + acc[1][r] + acc[2][r], s1a = acc[1][r] - acc[2][r] - acc[3][r];
        const float s0b = acc[4][r] + acc[5][r] + acc[6][r], s1b = acc[5][r] - acc[6][r] - acc[7][r];
        if (ph == 0) {
            P[0][r] = s0a + s0b; P[1][r] = s1a + s1b; P[2][r] = s0b; P[3][r] = s1b;
        } else {
            P[0][r] = s0a; P[1][r] = s1a; P[2][r] = -s0a - s0b; P[3][r] = -s1a - s1b;
        }
    }
    // exchange: ph = 1 waves -> LDS [wm][wn][ij][r][lane]; ph = 0 waves add (Y = P0 + P1, fixed order)
    float *xch = wsm;                                   // NW/2 waves x 4 x 16 x 64 floats (64 or 32 KB)
    constexpr int ELD = NQ + 4;
    float *img = wsm + (NW / 2) * 4 * 16 * 64;          // row image [4 MT][ELD] behind it (<= 69.6 KB)
    if (ph == 1) {
        float *dst = xch + ((wave & (NW / 2 - 1)) * 64) * 64 + lane;
#pragma unroll
        for (int ij = 0; ij < 4; ij++)
#pragma unroll
            for (int r = 0; r < 16; r++) dst[(ij * 16 + r) * 64] = P[ij][r];
    }
    __syncthreads();
    if (ph == 0) {
        const float *src = xch + ((wave & (NW / 2 - 1)) * 64) * 64 + lane;
#pragma unroll
        for (int ij = 0; ij < 4; ij++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int tile = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                img[(tile * 4 + ij) * ELD + wn * 32 + l31] = P[ij][r] + src[(ij * 16 + r) * 64];
            }
    }
    __syncthreads();
    const bool do_stats = p.stats_out != nullptr && nimg == 1;
    float4 K = make_float4(0.f, 0.f, 0.f, 0.f), s1 = K, s2 = K;
    int cnt = 0;
#pragma unroll
    for (int pass = 0; pass < NPASS; pass++) {
        const int row = pass * NRG + rg;
        const int64_t m = mrow[pass];
        if (m < 0) continue;
        float4 v = *reinterpret_cast<const float4 *>(img + row * ELD + c4 * 4);
        v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
        if (p.res0) {
            const float4 q = resq[pass];
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (do_stats) {
            if (cnt == 0) K = v;
            float d;
            d = v.x - K.x; s1.x += d; s2.x = fmaf(d, d, s2.x);
            d = v.y - K.y; s1.y += d; s2.y = fmaf(d, d, s2.y);
            d = v.z - K.z; s1.z += d; s2.z = fmaf(d, d, s2.z);
            d = v.w - K.w; s1.w += d; s2.w = fmaf(d, d, s2.w);
            cnt++;
        }
        *reinterpret_cast<float4 *>(p.out + m * p.Cout + n) = v;
    }
    if (do_stats) {
        __syncthreads();
        float2 *part = reinterpret_cast<float2 *>(wsm);
        const float fc = (float)(cnt > 0 ? cnt : 1);
        const float mx = s1.x / fc, my = s1.y / fc, mz = s1.z / fc, mw = s1.w / fc;
        part[rg * NQ + c4 * 4 + 0] = make_float2(K.x + mx, fmaxf(s2.x - s1.x * mx, 0.f));
        part[rg * NQ + c4 * 4 + 1] = make_float2(K.y + my, fmaxf(s2.y - s1.y * my, 0.f));
        part[rg * NQ + c4 * 4 + 2] = make_float2(K.z + mz, fmaxf(s2.z - s1.z * mz, 0.f));
        part[rg * NQ + c4 * 4 + 3] = make_float2(K.w + mw, fmaxf(s2.w - s1.w * mw, 0.f));
        __syncthreads();
        if (tid < NQ) {
            const float npart = (float)NPASS;   // rows behind each partial
            float mean = part[tid].x, M2 = part[tid].y, na = npart;
            for (int g = 1; g < NRG; g++) {
                const float2 q = part[g * NQ + tid];
                const float d = q.x - mean, N = na + npart;
                mean += d * (npart / N);
                M2 += q.y + d * d * (na * npart / N);
                na = N;
            }
            const int nt = (H * W) / (4 * MT);
            p.stats_out[((int64_t)img0 * nt + blk_in_img) * p.Cout + n0 + tid] = make_float2(mean, M2);
        }
    }
    DLPM_PHASE(p, 10);
#ifdef DLPM_PHASE_TIMING
    if (p.phase && tid == 0) atomicAdd(p.phase + 11, 1ull);
#endif
}

// OIHW (3x3) -> U = G g G^T in k_conv3x3_wino_q's fragment order  Wf[nb][ph][chunk8][pos8][lane][4]:
// lane = h*32 + n holds U_pos[cin = chunk*8 + h*4 + e][cout = nb*32 + n], pos = ph*8 + pos8.
__global__ void k_relayout_weight_wino_q(const float *oihw, float *dst, int Cout, int Cin, int kc) {
    // kc = 8: Wf[nb][ph][chunk][pos8][lane][4], cin = chunk*8 + h*4 + e
    // kc = 16: Wf[nb][ph][chunk][pos8][jq][lane][4], cin = chunk*16 + h*8 + jq*4 + e
    const int nbk = Cout / 32, nch = Cin / kc, njq = kc / 8;
    const int64_t total = (int64_t)nbk * 2 * nch * 8 * njq * 64 * 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    int64_t r = i >> 8;
    const int jq = (int)(r % njq); r /= njq;
    const int pos8 = (int)(r & 7); r >>= 3;
    const int chunk = (int)(r % nch); r /= nch;
    const int ph = (int)(r & 1);
    const int nb = (int)(r >> 1);
    const int pos = ph * 8 + pos8;
    const int h = lane >> 5, nn = lane & 31;
    const int cin = chunk * kc + h * (kc / 2) + jq * 4 + e, cout = nb * 32 + nn;
    const float *g = oihw + ((int64_t)cout * Cin + cin) * 9;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int a = pos >> 2, b = pos & 3;
    float u = 0.f;
    for (int ii = 0; ii < 3; ii++) {
        float row = 0.f;
        for (int jj = 0; jj < 3; jj++) row += g[ii * 3 + jj] * G[b][jj];
        u += G[a][ii] * row;
    }
    dst[i] = u;
}

}  // namespace

static bool wino_disabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLPM_NO_WINO"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

int wino_waves(const ConvLaunch &c);
// input channels per phase of the 8-wave kernel for a layer with Cout output channels: 8; DLPM_WINO_KC=16 selects 16 for
// the 32-tile x 128-channel shape (half the barriers per MFMA -- measured 2 % SLOWER: 61.3 vs 60.1 ms/step).  It fixes the
// Winograd-domain weight layout, so it depends on Cout only
static int wino_kc_for(int Cout) {
    static int pref = -1, nw = -1, mtp = -1;
    if (pref < 0) {
        const char *e = getenv("DLPM_WINO_KC"); pref = e ? atoi(e) : 8;
        const char *f = getenv("DLPM_WINO_NW"); nw = f ? atoi(f) : 8;
        const char *g = getenv("DLPM_WINO_MT"); mtp = g ? atoi(g) : 32;
    }
    return (pref == 16 && nw != 4 && mtp == 32 && Cout % 128 == 0) ? 16 : 8;
}
// Small launches of 64-channel-multiple layers (the MNIST-width nets): when a 64-tile block would hold SEVERAL whole
// images (8x8 / 4x4-pixel tensors) and the grid of 64-tile x 64-channel blocks would leave most CUs empty, the 4-wave
// 32-tile x 64-channel shape is used instead: twice the workgroups, two per CU.  Per (tile, channel) the arithmetic is the
// same in both shapes (same position split, same fixed-order exchange), and multi-image blocks emit no GroupNorm
// statistics in either, so the choice may follow the batch of the call without changing a bit
// (test_conv_winograd_f2_small_launch_shape_is_bit_identical).
static bool wino_small_launch(const ConvLaunch &c) {
    if (c.Cout % 128 == 0) return false;                       // those layers already run 32-tile blocks
    const int tpi = (c.Hout / 2) * (c.Wout / 2);               // tiles per image
    if (tpi >= 64 || 64 % tpi != 0) return false;              // only blocks of whole small images
    const int64_t wgs64 = ceil_div((int64_t)c.B, 64 / tpi) * (c.Cout / 64);
    return wgs64 < 256;
}

// tiles per workgroup: 32 (x 128 output channels) for the 8-wave kernel when Cout allows it, else 64 (x 64 channels)
int wino_tiles(const ConvLaunch &c) {
    static int pref = -1;
    if (pref < 0) { const char *e = getenv("DLPM_WINO_MT"); pref = e ? atoi(e) : 32; }
    if (wino_waves(c) == 4) return 32;
    return (pref == 32 && c.Cout % 128 == 0) ? 32 : 64;
}

// waves per workgroup of the 8-wave kernel family: DLPM_WINO_NW=4 selects the two-workgroups-per-CU shape (32 x 64 blocks)
int wino_waves(const ConvLaunch &c) {
    static int pref = -1;
    if (pref < 0) { const char *e = getenv("DLPM_WINO_NW"); pref = e ? atoi(e) : 8; }
    return (pref == 4 || wino_small_launch(c)) ? 4 : 8;
}

bool wino_geometry(const ConvLaunch &c, int *bh, int *bw, int *nimg) {
    if (wino_disabled() || c.gen == DLPM_CONV_IGEMM || !c.w_wino || c.ks != 3 || c.stride != 1 || c.in_nchw || c.out_nchw || c.abl) return false;
    if ((c.Hout & 1) || (c.Wout & 1) || c.Cout % WN != 0 || (c.C0 + c.C1) % WKC != 0 || c.C0 % WKC != 0) return false;
    if ((c.R0 & 3) != 0) return false;
    if (c.ups && ((c.Hout & 3) || (c.Wout & 3))) return false;
    const int TH = c.Hout / 2, TW = c.Wout / 2;
    const int T = wino_tiles(c);
    int h, w, n;
    if (TH * TW >= T) {            // a block inside one image
        w = TW < 8 ? TW : 8;
        if (T % w != 0) return false;
        h = T / w;
        if (TW % w != 0 || TH % h != 0) return false;
        n = 1;
    } else {                       // several whole small images per block
        if (T % (TH * TW) != 0) return false;
        h = TH; w = TW; n = T / (TH * TW);
        if (n > 16) return false;
    }
    const int RH = c.ups ? h + 2 : 2 * h + 2, RW = c.ups ? w + 2 : 2 * w + 2;
    if (c.ups && ((h & 1) || (w & 1))) return false;   // source-resolution halo needs even block origins
    if (n * RH * RW > (T == 64 ? RAW_MAXPIX : RAW_MAXPIX / 2)) return false;
    *bh = h; *bw = w; *nimg = n;
    return true;
}

int launch_conv_wino(const ConvLaunch &c, hipStream_t st) {
    int bh, bw, nimg;
    if (!wino_geometry(c, &bh, &bw, &nimg)) {
        set_error("launch_conv_wino: unsupported shape");
        return DLPM_ERR_UNSUPPORTED;
    }
#ifdef DLPM_PHASE_TIMING
    const_cast<ConvLaunch &>(c).phase = phase_buffer();
#endif
    using KFn = void (*)(ConvLaunch, int, int, int);
    const int mt = wino_tiles(c);
    const int nw = wino_waves(c);
    const int kc = wino_kc_for(c.Cout);
    KFn fn;
    if (mt == 32 && nw == 4) fn = c.ups ? &k_conv3x3_wino_q<true, 32, 4> : &k_conv3x3_wino_q<false, 32, 4>;
    else if (mt == 32 && kc == 16) fn = c.ups ? &k_conv3x3_wino_q<true, 32, 8, 0, 16> : &k_conv3x3_wino_q<false, 32, 8, 0, 16>;
    else if (mt == 32) fn = c.ups ? &k_conv3x3_wino_q<true, 32> : &k_conv3x3_wino_q<false, 32>;
    else fn = c.ups ? &k_conv3x3_wino_q<true, 64> : &k_conv3x3_wino_q<false, 64>;
#ifdef DLPM_WINO_ABLATIONS
    static int abl = -1;   // timing-only ablations (results are wrong when set)
    if (abl < 0) { const char *e = getenv("DLPM_WABL"); abl = e ? atoi(e) : 0; }
    if (!c.ups && mt == 32 && nw == 8 && kc == 8) {
        switch (abl) {
            case 1: fn = &k_conv3x3_wino_q<false, 32, 8, 1>; break;
            case 2: fn = &k_conv3x3_wino_q<false, 32, 8, 2>; break;
            case 3: fn = &k_conv3x3_wino_q<false, 32, 8, 3>; break;
            case 8: fn = &k_conv3x3_wino_q<false, 32, 8, 8>; break;
            case 16: fn = &k_conv3x3_wino_q<false, 32, 8, 16>; break;
            case 31: fn = &k_conv3x3_wino_q<false, 32, 8, 31>; break;
            case 32: fn = &k_conv3x3_wino_q<false, 32, 8, 32>; break;
            default: break;
        }
    }
#endif
    {
        int r = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), 160 * 1024);
        if (r != DLPM_OK) return r;
    }
    const int nq = nw == 8 ? 4096 / mt : 64;
    size_t shmem = (size_t)(2 * 16 * mt * (kc + 4) + 2 * (mt == 64 ? RAW_MAXPIX : RAW_MAXPIX / 2) * (kc + 4) + 2 * 16 * 2 * kc) * sizeof(float);
    const size_t epi = (size_t)((nw / 2) * 4 * 16 * 64 + 4 * mt * (nq + 4)) * sizeof(float);   // exchange + row image
    if (shmem < epi) shmem = epi;
    const int64_t tiles = (int64_t)c.B * (c.Hout / 2) * (c.Wout / 2);
    const int64_t mblocks = nimg == 1 ? tiles / mt : ceil_div(c.B, nimg);
    const int ks = c.ksplit > 1 ? c.ksplit : 1;
    if (ks > 1 && (c.bias || c.res0 || c.stats_out || ((c.C0 + c.C1) / kc) % ks != 0)) {
        set_error("launch_conv_wino: a split-K launch carries no bias / residual / statistics and divides its chunks evenly");
        return DLPM_ERR_ARG;
    }
    fn<<<(unsigned)(mblocks * (c.Cout / nq) * ks), nw * 64, shmem, st>>>(c, bh, bw, nimg);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int64_t wino_grid_at(const ConvLaunch &c, int64_t B) {
    int bh, bw, nimg;
    if (!wino_geometry(c, &bh, &bw, &nimg)) return 0;
    const int mt = wino_tiles(c), nw = wino_waves(c);
    const int nq = nw == 8 ? 4096 / mt : 64;
    const int64_t tiles = B * (c.Hout / 2) * (c.Wout / 2);
    const int64_t mblocks = nimg == 1 ? tiles / mt : ceil_div(B, (int64_t)nimg);
    return mblocks * (c.Cout / nq);
}

int wino_chunk_channels(const ConvLaunch &c) { return wino_kc_for(c.Cout); }

int64_t wino_weight_floats(int Cout, int Cin) {
    // + AHEAD groups of padding: the prefetch ring reads past the last group
    return ((int64_t)(Cout / 32) * (Cin / WKC) * WGRP + 8) * 256;   // 8 >= every kernel's ring depth - 1
}

int relayout_weight_wino(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st) {
    const int64_t n = (int64_t)(Cout / 32) * (Cin / WKC) * WGRP * 256;
    DLPM_HIP(hipMemsetAsync(dst_dev + n, 0, (size_t)8 * 256 * sizeof(float), st));
    k_relayout_weight_wino_q<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin, wino_kc_for(Cout));
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
