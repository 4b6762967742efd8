// conv_split.hip -- the 1x1 convolutions (qkv, proj, skip connections: conv1d(k=1) / 1x1 conv2d, unet.py:157,213,215)
// as an fp32 GEMM on the bf16 matrix pipe of gfx950.
//
// Every fp32 operand is cut EXACTLY into three bf16 planes, x = x0 + x1 + x2 (8 + 8 + 8 significand bits: x0 = the top
// half of the fp32 word, x1 = the top half of x - x0, x2 = x - x0 - x1 -- each subtraction is exact), and a product is
// the six partial products whose weight is at least 2^-16 of it,
//     a b  ~=  a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),
// each exact in fp32 (8 x 8 bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  What is dropped (a1 b2 + a2 b1 +
// a2 b2) is below 2^-23 |a b|, the size of one fp32 rounding of the product; tests/test_gpu_kernels.py measures the
// result against float64 next to the fp32 MFMA kernel's.  Six bf16 MFMAs cost 6/16 of the fp32 MFMA's pipe time
// (tools/mb/mfma_bf16.hip: 2.1 PFLOP/s executed from LDS-fed 64x64 wave tiles = 354 fp32-equivalent TFLOP/s, against
// the fp32 pipe's 157 peak), which turns these launches from matrix-pipe bound (95-115 TFLOP/s) into HBM bound.
//
// Tile: 128 pixels x 128 channels per workgroup, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles; K in steps of 32.
// LDS holds ONE stage (the next one waits in registers): per operand 3 planes x [k-step 2][k-half 2][row 128] x 16 B in
// MFMA fragment order, so a fragment load is one ds_read_b128 on consecutive lanes (conflict-free) and two workgroups
// fit a CU (48 KB each) -- one covers the other's prologue / epilogue.  Weights are split once, at plan time, into
// exactly this stage image (k_relayout_weight_split): staging them is a 16-byte-per-lane copy.  Activations are split
// on the way into LDS, after the fused GroupNorm affine (+ SiLU): ~6 VALU operations per element, half of which the
// bf16 MFMAs hide (same microbenchmark: VALU work co-issues with the bf16 pipe, unlike the fp32 MFMAs).
#include "conv.h"
#include "igemm_epilogue.h"

#ifndef SPLIT_ABL   // developer builds only (DLPM_BUILD_DEFS): 1 no re-load after stage 0, 2 no split arithmetic, 4 no MFMAs
#define SPLIT_ABL 0   // (results are wrong when set)
#endif

#ifdef DLPM_IGEMM_ABLATIONS   // developer builds only: DLPM_ABL bits at run time (results are wrong when set): 1 no output stores,
#define SABL(b) (p.abl & (b))  // 2 no activation re-loads, 4 no weight re-loads, 8 no MFMAs, 16 no activation staging, 32 no weight staging, 64 no B fragment reads
#else
#define SABL(b) 0
#endif

#ifndef SPLIT_WGS
#define SPLIT_WGS 2   // workgroups per CU the register budget is set for
#endif

namespace dlpm {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SKC = 32;                       // channels per stage
constexpr int SPLANE = 128 * SKC * 2;         // bytes of one plane of one operand stage (8 KB)
constexpr int SOPER = 3 * SPLANE;             // one operand stage (24 KB)
constexpr int SCHUNKS = SOPER / 16;           // 16-byte chunks per operand stage (1536)

// fp32 pair -> the packed bf16 pairs of the three planes.  __builtin_amdgcn_perm(hi, lo, 0x07060302) = {hi[31:16], lo[31:16]}.
__device__ __forceinline__ void split2(float lo, float hi, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
    const uint32_t ul = __float_as_uint(lo), uh = __float_as_uint(hi);
    p0 = __builtin_amdgcn_perm(uh, ul, 0x07060302u);
    const float rl = lo - __uint_as_float(ul & 0xffff0000u), rh = hi - __uint_as_float(uh & 0xffff0000u);
    const uint32_t vl = __float_as_uint(rl), vh = __float_as_uint(rh);
    p1 = __builtin_amdgcn_perm(vh, vl, 0x07060302u);
    const float sl = rl - __uint_as_float(vl & 0xffff0000u), sh = rh - __uint_as_float(vh & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(sh), __float_as_uint(sl), 0x07060302u);
}

// OIHW ([Cout][Cin][taps]) -> [tap][Cout/128][Cin/32][plane 3][k-step 2][k-half 2][row 128][8 bf16]
__global__ void k_relayout_weight_split(const float *w, uint4 *dst, int Cout, int Cin, int taps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (tap, n, group of 8 k)
    const int kg = Cin / 8;
    if (i >= (int64_t)taps * Cout * kg) return;
    const int tap = (int)(i / ((int64_t)Cout * kg));
    const int64_t j = i - (int64_t)tap * Cout * kg;
    const int n = (int)(j / kg), g = (int)(j - (int64_t)n * kg);
    const float *src = w + ((int64_t)n * Cin + g * 8) * taps + tap;
    uint32_t P[3][4];
#pragma unroll
    for (int e = 0; e < 4; e++) split2(src[(2 * e) * taps], src[(2 * e + 1) * taps], P[0][e], P[1][e], P[2][e]);
    const int nt = n >> 7, row = n & 127, kc = g >> 2, q = g & 3;   // q = k-step * 2 + k-half
    uint4 *tile = dst + (((int64_t)tap * (Cout >> 7) + nt) * (Cin / SKC) + kc) * SCHUNKS;
#pragma unroll
    for (int pl = 0; pl < 3; pl++) tile[pl * (SPLANE / 16) + q * 128 + row] = make_uint4(P[pl][0], P[pl][1], P[pl][2], P[pl][3]);
}

// Epilogue of k_conv1x1_split without fused statistics, straight from the accumulators: lane = channel (lane & 31), registers =
// rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5), so one store instruction writes 32 consecutive channels of 2 pixels = two whole
// 128-byte lines.  No LDS, no barrier, every wave busy; the residual's loads are issued before the first add.
// MASKED: the tile's last rows lie beyond M (ragged last tile).
// STATS (8 x 8 images, HW == 64: the 64 rows of a wave are exactly one image, 32 of them in each k-half lane): the fused GroupNorm
// statistics of the output, one (mean, M2) per image and channel -- per-lane shifted sums over 32 rows, merged with the partner lane.
template <bool MASKED, int RN, bool STATS = false>
__device__ __forceinline__ void split_store_from_registers(const ConvLaunch &p, floatx16 (&acc)[2][RN], int64_t m0, int nw0, int wm,
                                                     int l31, int kh, int mrem) {
    const int R1 = p.Cout - p.R0;
    const int rlim = mrem - 1 - (wm * 64 + 4 * kh);   // last valid row, counted from this lane's first row (negative: the lane has none)
    const int64_t row0 = m0 + wm * 64 + 4 * kh;
    // residual rows of a ragged tile are clamped to the tile's last valid row IN ABSOLUTE terms: a lane whose first row already
    // lies beyond M (rlim < 0) reads row mrem - 1 of the tile, never memory behind the tensor
    const int rres0 = MASKED ? min(wm * 64 + 4 * kh, mrem - 1) : 0;   // this lane's first residual row, counted from m0
#pragma unroll
    for (int j = 0; j < RN; j++) {
        const int n = nw0 + j * 32 + l31;
        const float bias = p.bias ? p.bias[n] : 0.f;
        float *op = p.out + row0 * p.Cout + n;
        float q[2][16];
        if (p.res0) {
            const bool r0 = n < p.R0;                    // uniform per (wave, j): R0 % 32 == 0 (gemm_split_ok)
            const int64_t rrow0 = MASKED ? m0 + rres0 : row0;
            const float *rp = r0 ? p.res0 + rrow0 * p.R0 + n : p.res1 + rrow0 * R1 + (n - p.R0);
            const int rs = r0 ? p.R0 : R1;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                    q[i][r] = rp[(MASKED ? max(min(row, rlim), 0) : row) * rs];
                }
        }
        float K = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] + bias;
                if (p.res0) v += q[i][r];
                if (STATS) {
                    if (i == 0 && r == 0) K = v;
                    const float dd = v - K;
                    s1 += dd;
                    s2 = fmaf(dd, dd, s2);
                }
                if (SABL(1) && v == v) continue;
                if (!MASKED || row <= rlim) op[(int64_t)row * p.Cout] = v;
            }
        if (STATS) {
            // this lane: 32 of the image's 64 rows; the k-half partner (lane ^ 32) has the other 32: Chan merge, k-half 0 first
            const float mean = K + s1 * (1.f / 32.f), M2 = fmaxf(s2 - s1 * s1 * (1.f / 32.f), 0.f);
            const float om = __shfl_xor(mean, 32), oM2 = __shfl_xor(M2, 32);
            const float lo_m = kh ? om : mean, hi_m = kh ? mean : om;
            const float dd = hi_m - lo_m;
            const int64_t img = (m0 + wm * 64) >> 6;
            if (kh == 0 && img < p.B) p.stats_out[img * p.Cout + n] = make_float2(lo_m + dd * 0.5f, (kh ? oM2 : M2) + (kh ? M2 : oM2) + dd * dd * 16.f);
        }
    }
}

// NW waves: 4 = 2 x 2 waves of 64 x 64 (2 workgroups per CU); 8 = 2 x 4 waves of 64 x 32 (2 workgroups = 4 waves per SIMD).
// (Also measured: 8 waves with two register stages -- 138 registers, one workgroup per CU -- 19 % slower.)
// DIST: stages of operands waiting in registers.  TAPS: 1 = 1x1 convolution, 9 = 3x3 (padding 1, stride 1 or 2) as an implicit
// GEMM over K = 9 Cin (channel chunk outer, tap inner: the taps of a chunk re-read the same input lines from L2).
// What bounds this kernel is the chip's power budget, not its structure: on all-zero operands the same launches run 27-31 %
// faster (profiles/r02/gemm_bf16x3_zero_operands_dvfs.txt: the bare MFMA + fragment-read + barrier loop then reaches 2.0
// PFLOP/s, 80 % of the bf16 pipe; on random data 1.45) -- the bf16 MFMAs at this rate with toggling operands pull the clock
// down.  Which is why every structural variant measured here landed within +-5 % of this one (all under profiles/r02/gemm_*):
// two 8-wave groups in one workgroup one barrier phase apart (ping-pong), 256-pixel tiles with two LDS stages and one
// barrier per stage, 4 waves at three workgroups per CU, two register stages at 128 VGPRs.
// NW = 16 (round 3, measured and NOT the default -- launch_conv_split): ONE workgroup of 2 x 8 waves per CU computes 128 pixels x 256
// channels -- the activation stage (loads, GroupNorm affine, SiLU, the three-plane split, LDS writes) is shared by twice the MFMAs,
// half as many prologues / epilogues per layer; LDS 24 KB of A + 48 KB of B per stage.  Same per-output accumulation order as the
// other shapes: same bits.
template <int NW, int DIST, int TAPS>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? SPLIT_WGS : 1) k_conv_split(ConvLaunch p, int nsamp, int xcd_map) {
    constexpr int NT = NW * 64, RN = NW == 4 ? 2 : 1, WN = NW == 16 ? 8 : 4 / RN;   // threads, MFMA column tiles per wave, waves across N
    constexpr int NB = NW == 16 ? 2 : 1, BNW = NB * 128;                              // 128-channel weight tiles per stage, tile width
    constexpr int RSTEP = NT / 8, NV = 128 / RSTEP, NWV = NB * SCHUNKS / NT;   // staging: rows per pass, passes, weight chunks per thread
    // [A stage 24 KB][B stage 24 KB] (the statistics epilogue's row image afterwards) [GroupNorm coefficients of the tile's samples]
    extern __shared__ __align__(16) unsigned char smem[];
    DLPM_PHASE_DECL;
#ifdef DLPM_PHASE_TIMING   // loop sub-phases accumulate in registers (one atomic per counter per workgroup: atomics inside the loop
    const long long _c0 = clock64(), _r0 = wall_clock64();   // shader cycles and 100-MHz ticks: their ratio is the clock the chip holds
    long long lp[4] = {0, 0, 0, 0};   // queue behind each other at L2 and distort the very waits being measured)
#define SPLIT_LP(i) do { if (p.phase && threadIdx.x == 0) { const long long _n = clock64(); lp[i] += _n - _pt; _pt = _n; } } while (0)
#else
#define SPLIT_LP(i)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int Cin = p.C0 + p.C1;
    const int ntile_n = p.Cout / BNW;
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each with its own L2.  The ntile_n channel tiles of one
    // pixel tile read the same activations: they go to ONE XCD, back to back (xcd_map: pixel tiles % 8 == 0).
    int mt_i, nt;
    if (xcd_map) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        mt_i = (slot / ntile_n) * 8 + xcd;
        nt = slot % ntile_n;
    } else {
        mt_i = blockIdx.x / ntile_n;
        nt = blockIdx.x % ntile_n;
    }
    const int64_t m0 = (int64_t)mt_i * BM;
    const int n0 = nt * BNW;
    const int HWo = p.Hout * p.Wout;

    // A staging: 8 consecutive lanes read one pixel's 32 channels (a whole 128-byte line), a wave instruction 8 pixels;
    // a thread owns channels 4q..4q+3 of pixels rb, rb + RSTEP, ...
    const int q = tid & 7, rb = tid >> 3;
    const int ksh = q >> 1;                       // k-step * 2 + k-half of this thread's channels; q & 1 = which 8 bytes of the chunk
    // a ragged last tile (B * HW not a multiple of 128: small batches of 8x8 / 4x4 tensors) re-reads its last valid pixel
    // for the rows beyond M and never stores them: the kernel a layer takes must not depend on the batch
    const int64_t M = (int64_t)p.B * HWo;
    const int mrem = (int)min((int64_t)BM, M - m0);
    const float *a0 = p.src0 + (TAPS == 1 ? m0 * p.C0 : 0) + 4 * q;
    const float *a1 = p.src1 ? p.src1 + (TAPS == 1 ? m0 * p.C1 : 0) + 4 * q - p.C0 : a0;
    int rcl[NV];        // TAPS 1: row of the tile (clamped);  TAPS 9: first input pixel of the sample
    int iy0[NV], ix0[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        rcl[v] = min(rb + RSTEP * v, mrem - 1);
        if (TAPS > 1) {
            const int64_t m = m0 + rcl[v];
            const int b = (int)(m / HWo), rem = (int)(m - (int64_t)b * HWo);
            const int oy = rem / p.Wout;
            iy0[v] = oy * p.stride - 1;
            ix0[v] = (rem - oy * p.Wout) * p.stride - 1;
            rcl[v] = b * p.Hin * p.Win;
        }
    }
    const int nkc = Cin / SKC;
    // weight image [tap][Cout / 128][Cin / 32][SCHUNKS]: a 256-wide tile stages two consecutive 128-channel tiles, the second one
    // nkc * SCHUNKS chunks behind the first (thread -> chunk v * NT + tid of the 2 * SCHUNKS; NT = 1024 < SCHUNKS: a thread's chunks
    // v = 0, 1 lie in the first tile or straddle, computed per chunk below)
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(p.w_split) + (int64_t)nt * NB * nkc * SCHUNKS + (NB == 1 ? tid : 0);
    const bool has_coef = p.coefA != nullptr;

    // stage s = (channel chunk kc, tap); `ok` collects which of this thread's rows read inside the image (zero padding)
    auto load_step = [&](float4 (&xa)[NV], u32x4 (&wb)[NWV], int &ok, int s) {
        const int kc = TAPS == 1 ? s : s / TAPS, tap = TAPS == 1 ? 0 : s - kc * TAPS;
        const int c0 = kc * SKC;
        const bool first = c0 < p.C0;             // uniform: C0 % 32 == 0
        const float *src = first ? a0 + c0 : a1 + c0;
        const int rs = first ? p.C0 : p.C1;
        if (!(SABL(2) && s >= 2)) {
            if (TAPS == 1) {
#pragma unroll
                for (int v = 0; v < NV; v++) xa[v] = *reinterpret_cast<const float4 *>(src + rcl[v] * rs);
            } else {
                const int ky = tap / 3, kx = tap - 3 * ky;
                ok = 0;
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const int iy = iy0[v] + ky, ix = ix0[v] + kx;
                    if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) ok |= 1 << v;
                    const int pix = rcl[v] + min(max(iy, 0), p.Hin - 1) * p.Win + min(max(ix, 0), p.Win - 1);
                    xa[v] = *reinterpret_cast<const float4 *>(src + (int64_t)pix * rs);   // unconditional: a load in a branch drains the queue
                }
            }
        }
        if (!(SABL(4) && s >= 2)) {
            const u32x4 *ws = wsrc + ((int64_t)tap * (p.Cout >> 7) * nkc + kc) * SCHUNKS;
#pragma unroll
            for (int v = 0; v < NWV; v++) {
                if (NB == 1) wb[v] = ws[v * NT];
                else {
                    const int c = v * NT + tid, half = c >= SCHUNKS ? 1 : 0;     // chunk of the 2 x SCHUNKS stage image
                    wb[v] = ws[(int64_t)half * nkc * SCHUNKS + (c - half * SCHUNKS)];
                }
            }
        }
    };
    // LDS chunk (16 B = 8 channels of one row of one plane) of (ksh, row): ksh * 128 + (row ^ 8 ksh) -- the XOR spreads the
    // 8-byte staging writes of a wave (4 ksh x 8 rows) over the banks and leaves a fragment read (32 consecutive rows) contiguous
    uint2 *As = reinterpret_cast<uint2 *>(smem);
    u32x4 *Bs = reinterpret_cast<u32x4 *>(smem + SOPER);
    const float *cf = reinterpret_cast<const float *>(smem + (1 + NB) * SOPER);
    int cfo[NV];                                  // this thread's rows' coefficient rows in the LDS table
#pragma unroll
    for (int v = 0; v < NV; v++) cfo[v] = (nsamp > 1 ? min(rb + RSTEP * v, mrem - 1) / HWo : 0) * Cin + 4 * q;   // rows beyond M: the last valid sample's (written) coefficients
    const int wofs = (ksh * 128 + (rb ^ (ksh * 8))) * 2 + (q & 1);
    auto store_step = [&](float4 (&xa)[NV], u32x4 (&wb)[NWV], int ok, int s) {
        const int c0 = (TAPS == 1 ? s : s / TAPS) * SKC;
#pragma unroll
        for (int v = 0; v < NV; v++) {
            if (SABL(16) && s >= 2) break;
            float4 x = xa[v];
            if (has_coef) {
                const float4 ca = *reinterpret_cast<const float4 *>(cf + cfo[v] + c0);
                const float4 cb = *reinterpret_cast<const float4 *>(cf + nsamp * Cin + cfo[v] + c0);
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
            if (TAPS > 1 && !(ok >> v & 1)) x = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding of the ACTIVATED tensor
            uint32_t P[3][2];
            if (SPLIT_ABL & 2) {
                for (int pl = 0; pl < 3; pl++) { P[pl][0] = __float_as_uint(x.x + x.y); P[pl][1] = __float_as_uint(x.z + x.w); }
            } else {
                split2(x.x, x.y, P[0][0], P[1][0], P[2][0]);
                split2(x.z, x.w, P[0][1], P[1][1], P[2][1]);
            }
#pragma unroll
            for (int pl = 0; pl < 3; pl++) As[pl * (SPLANE / 8) + wofs + v * (RSTEP * 2)] = make_uint2(P[pl][0], P[pl][1]);
        }
        if (!(SABL(32) && s >= 2)) {
#pragma unroll
            for (int v = 0; v < NWV; v++) Bs[v * NT + tid] = wb[v];
        }
    };

    floatx16 acc[2][RN];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < RN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const bf16x8 *af = reinterpret_cast<const bf16x8 *>(smem) + kh * 128 + wm * 64;
    const bf16x8 *bf = reinterpret_cast<const bf16x8 *>(smem + SOPER + (NW == 16 ? (wn >> 2) * SOPER : 0)) + kh * 128 +
                       (NW == 16 ? (wn & 3) * 32 : wn * (RN * 32)) + l31;
    const int ax0 = l31 ^ (kh * 8), ax1 = l31 ^ ((2 + kh) * 8);   // swizzled row of this lane for k-step 0 / 1
    auto mfma_step = [&]() {
        if (SABL(8)) return;
        if (!(SPLIT_ABL & 16)) __builtin_amdgcn_s_setprio(1);   // MFMA-issuing waves first: +4-8 % on the long launches (profiles/r02/gemm_bf16x3_priority.txt)
#pragma unroll
        for (int ks = 0; ks < ((SPLIT_ABL & 4) ? 0 : 2); ks++) {
            bf16x8 A[2][3], B[RN][3];
            const int ax = ks ? ax1 : ax0;
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
#pragma unroll
                for (int i = 0; i < 2; i++) A[i][pl] = af[pl * (SPLANE / 16) + ks * 256 + i * 32 + ax];
#pragma unroll
                for (int j = 0; j < RN; j++) B[j][pl] = SABL(64) ? A[j][pl] : bf[pl * (SPLANE / 16) + ks * 256 + j * 32];
            }
            // smallest terms first within a (tile, k-step)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < RN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][2], B[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][0], acc[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    const int nsteps = nkc * TAPS;
    int ok0 = 0, ok1 = 0;
    float4 xa0[NV], xa1[NV];
    u32x4 wb0[NWV], wb1[NWV];
    // GroupNorm coefficients of the tile's samples -> LDS [A | B][nsamp][Cin] (read by every stage).  Their loads go out
    // FIRST: loads retire in order, so the table's LDS writes wait for nothing but themselves.
    constexpr int CR = NT >= 512 ? 1 : 512 / NT;  // rounds of NT float4 that ride in registers (the first 512, or the first NT)
    f32x4 cq[2][CR] = {};
    const int n4 = has_coef ? nsamp * Cin / 4 : 0;   // float4 per array (<= 1024, gemm_split_ok)
    const int nv4 = has_coef ? (int)min((int64_t)n4, (p.B - m0 / HWo) * (int64_t)(Cin / 4)) : 0;   // the batch may end inside the tile
    if (has_coef) {
        const int64_t pb0 = m0 / HWo;
#pragma unroll
        for (int r = 0; r < CR; r++) {
            const int i = min(tid + r * NT, nv4 - 1);
            cq[0][r] = reinterpret_cast<const f32x4 *>(p.coefA + pb0 * Cin)[i];
            cq[1][r] = reinterpret_cast<const f32x4 *>(p.coefB + pb0 * Cin)[i];
        }
    }
    load_step(xa0, wb0, ok0, 0);
    if (DIST == 2 && nsteps > 1) load_step(xa1, wb1, ok1, 1);
    if (has_coef) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(smem + (1 + NB) * SOPER);
#pragma unroll
        for (int r = 0; r < CR; r++) {
            const int i = tid + r * NT;
            if (i < n4) { dst[i] = cq[0][r]; dst[n4 + i] = cq[1][r]; }
        }
        const int64_t pb0 = m0 / HWo;
        for (int i = tid + CR * NT; i < nv4; i += NT) {   // tables beyond the register rounds (8 samples x 512 channels): the slow way
            dst[i] = reinterpret_cast<const f32x4 *>(p.coefA + pb0 * Cin)[i];
            dst[n4 + i] = reinterpret_cast<const f32x4 *>(p.coefB + pb0 * Cin)[i];
        }
        __syncthreads();
    }
    DLPM_PHASE(p, 0);
    if constexpr (DIST == 2) {
        // two register sets: the loads of stage s + 2 are issued as soon as stage s has left its registers
        for (int s = 0; s < nsteps; s += 2) {
            store_step(xa0, wb0, ok0, s);
            SPLIT_LP(0);   // developer counters 4..7: stage (incl. the wait for its loads) / barrier / MFMAs / barrier
            __syncthreads();
            SPLIT_LP(1);
            if (s + 2 < nsteps && !(SPLIT_ABL & 1)) load_step(xa0, wb0, ok0, s + 2);
            mfma_step();
            SPLIT_LP(2);
            __syncthreads();   // the stage is dead: the next store_step may overwrite it
            SPLIT_LP(3);
            if (s + 1 < nsteps) {
                store_step(xa1, wb1, ok1, s + 1);
                SPLIT_LP(0);
                __syncthreads();
                SPLIT_LP(1);
                if (s + 3 < nsteps && !(SPLIT_ABL & 1)) load_step(xa1, wb1, ok1, s + 3);
                mfma_step();
                SPLIT_LP(2);
                __syncthreads();
                SPLIT_LP(3);
            }
        }
    } else {
        for (int s = 0; s < nsteps; s++) {
            if (!(SPLIT_ABL & 8) || s == 0) store_step(xa0, wb0, ok0, s);   // SPLIT_ABL 8: stage 0 only
            SPLIT_LP(0);
            __syncthreads();
            SPLIT_LP(1);
            if (s + 1 < nsteps && !(SPLIT_ABL & 1)) load_step(xa0, wb0, ok0, s + 1);   // in flight under the MFMAs below
            mfma_step();
            SPLIT_LP(2);
            __syncthreads();
            SPLIT_LP(3);
        }
    }
    if constexpr (NW == 4) {
        if (p.stats_out) {   // fused GroupNorm statistics of the output: the row epilogue through LDS carries them
            epilogue_rows_full<128, 2, 2, 2, 2>(p, acc, reinterpret_cast<float *>(smem), m0, n0, tid, wm, wn, l31, kh);
            DLPM_PHASE(p, 2);
            return;
        }
    }
    if (NW != 4 && p.stats_out) {   // 8x8 images (launch_conv_split): per-image statistics from the registers
        if (mrem == BM) split_store_from_registers<false, RN, true>(p, acc, m0, n0 + wn * (RN * 32), wm, l31, kh, BM);
        else split_store_from_registers<true, RN, true>(p, acc, m0, n0 + wn * (RN * 32), wm, l31, kh, mrem);
    } else if (mrem == BM) split_store_from_registers<false, RN>(p, acc, m0, n0 + wn * (RN * 32), wm, l31, kh, BM);
    else split_store_from_registers<true, RN>(p, acc, m0, n0 + wn * (RN * 32), wm, l31, kh, mrem);
    DLPM_PHASE(p, 2);
#ifdef DLPM_PHASE_TIMING
    if (p.phase && tid == 0) {
        atomicAdd(p.phase + 3, 1ull);
        for (int i = 0; i < 4; i++) atomicAdd(p.phase + 4 + i, (unsigned long long)lp[i]);
        atomicAdd(p.phase + 12, (unsigned long long)(clock64() - _c0));
        atomicAdd(p.phase + 13, (unsigned long long)(wall_clock64() - _r0));
    }
#endif
}

}  // namespace

int64_t split_weight_floats(int Cout, int Cin, int taps) { return ((int64_t)taps * Cout * Cin * 6 + 3) / 4; }   // 3 bf16 planes

int relayout_weight_split(const float *oihw_dev, void *dst_dev, int Cout, int Cin, int taps, hipStream_t st) {
    if (Cout % 128 != 0 || Cin % SKC != 0 || (taps != 1 && taps != 9)) {
        set_error("relayout_weight_split: Cout %d %% 128, Cin %d %% 32 or %d taps", Cout, Cin, taps);
        return DLPM_ERR_UNSUPPORTED;
    }
    const int64_t n = (int64_t)taps * Cout * (Cin / 8);
    k_relayout_weight_split<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(oihw_dev, reinterpret_cast<uint4 *>(dst_dev), Cout, Cin, taps);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

// 1x1 / stride 1 and 3x3 / padding 1 / stride 1 or 2 over NHWC, 128-channel tiles.  Nothing here looks at the batch.
bool conv_split_ok(const ConvLaunch &c) {
    const int HW = c.Hout * c.Wout;
    if (!c.w_split || c.gemm == DLPM_GEMM_F32 || c.ups || c.in_nchw || c.out_nchw) return false;
    if (c.ks == 1 ? c.stride != 1 : (c.ks != 3 || (c.stride != 1 && c.stride != 2))) return false;
    if (c.Hout != (c.Hin - 1) / c.stride + 1 || c.Wout != (c.Win - 1) / c.stride + 1) return false;
    return c.Cout % 128 == 0 && c.C0 % SKC == 0 && (c.C0 + c.C1) % SKC == 0 && (c.R0 & 31) == 0 &&
           (HW >= BM ? HW % BM == 0 : (BM % HW == 0 && (!c.coefA || (BM / HW) * (c.C0 + c.C1) * 8 <= 32 * 1024)));   // tiles hold whole samples
}

int launch_conv_split(const ConvLaunch &c, hipStream_t st) {
    const int64_t M = (int64_t)c.B * c.Hout * c.Wout;
    static_assert(2 * SOPER >= 64 * 132 * 4, "epilogue image must fit the stage buffers");
    const int HW = c.Hout * c.Wout;
    const int nsamp = HW >= BM ? 1 : BM / HW;     // samples a 128-pixel tile spans
    // DLPM_SPLIT_NW16=1 (A/B runs; same bits either way): 256-wide channel tiles, 16 waves, ONE workgroup per CU, wherever Cout is a
    // multiple of 256 -- the activation stage shared by twice the MFMAs.  Measured 8-40 % SLOWER per launch on every shape and +0.5 ms
    // on the whole step (profiles/r03/gemm_bf16x3_256wide_tiles.txt, bench_cifar_alternating_256wide.txt): what two co-resident
    // workgroups hide of each other's barriers, prologues and epilogues is worth more than the staging they repeat.  Off by default.
    static int nw16 = -1;
    if (nw16 < 0) { const char *e = getenv("DLPM_SPLIT_NW16"); nw16 = e ? atoi(e) : 0; }
    const bool row_stats = c.stats_out && HW % BM == 0;      // per 128-pixel tile, through the 4-wave shape's row epilogue
    const bool wide = nw16 && c.Cout % 256 == 0 && !row_stats;
    const int bnw = wide ? 256 : 128;
    const int lds = (wide ? 3 : 2) * SOPER + (c.coefA ? nsamp * (c.C0 + c.C1) * 8 : 0);
    const int64_t mt = ceil_div(M, BM);
    const unsigned grid = (unsigned)(mt * (c.Cout / bnw));
    const int xcd_map = (c.Cout > bnw && mt % 8 == 0) ? 1 : 0;
    if (c.stats_out && !row_stats && HW != 64) {   // one partial per in-image 128-pixel tile, or per whole 8x8 image
        set_error("launch_conv_split: fused statistics need whole 128-pixel tiles inside one image, or 8x8 images (HW %d)", HW);
        return DLPM_ERR_UNSUPPORTED;
    }
#define DLPM_SPLIT_LAUNCH(NW_, DIST_, TAPS_)                                                                              \
    do {                                                                                                                  \
        const int r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_conv_split<NW_, DIST_, TAPS_>), 3 * SOPER + 32 * 1024); \
        if (r != DLPM_OK) return r;                                                                                       \
        k_conv_split<NW_, DIST_, TAPS_><<<grid, NW_ * 64, lds, st>>>(c, nsamp, xcd_map);                                  \
    } while (0)
    // 8 waves (four per SIMD with two workgroups on a CU) measured 5 % faster than 4 waves with two register stages
    // (profiles/r02/gemm_1x1_bf16x3_variants.txt); the 4-wave shape carries the row epilogue with the fused statistics.
    // All shapes accumulate every output in the same order: which one runs does not change a bit of the result.
    if (c.ks == 1) {
        if (row_stats) DLPM_SPLIT_LAUNCH(4, 2, 1);
        else if (wide) DLPM_SPLIT_LAUNCH(16, 1, 1);
        else DLPM_SPLIT_LAUNCH(8, 1, 1);
    } else {
        if (row_stats) DLPM_SPLIT_LAUNCH(4, 2, 9);
        else if (wide) DLPM_SPLIT_LAUNCH(16, 1, 9);
        else DLPM_SPLIT_LAUNCH(8, 1, 9);
    }
#undef DLPM_SPLIT_LAUNCH
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
