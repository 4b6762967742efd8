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
#include <type_traits>

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
// Row swizzle of the A stage image: chunk (16 B = the 8 k slots of one row, one k-half, one plane) of (k-step h, k-half kh, row) sits at
// (2 h + kh) * 128 + (row ^ 4 kh).  A staging thread writes ONE dword of it (its channel pair); a ds_write_b32 banks modulo 32
// dwords in groups of 32 lanes = 4 rows x 2 kh x 4 pairs: the XOR puts the two k-halves on opposite halves of the 128-byte bank
// row -- conflict-free; a fragment read (one kh per 16-lane group) stays a permutation of 32 consecutive rows.  (Rounds 2-4 wrote
// 8 bytes per lane with the XOR on row bit 3, laid out for 64 banks: every ds_write banks modulo 32, the four k slots of a row
// shared one 16-byte slot, and the 4-way conflict on every staging write was the 37 % of LDS cycles that
// profiles/r04/split_gemm_clock_and_mfma_busy.txt counted.)
namespace dlpm {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
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

// OIHW ([Cout][Cin][taps]) -> [tap][Cout/128][Cin/32][plane 3][k-step 2][k-half 2][row 128][8 bf16], a 32-channel stage's k slots
// in STAGING order: the k index of an MFMA is free as long as both operands agree, and the activation side wants every thread of
// the staging pass (which holds 4 consecutive channels 4q .. 4q+3 of a pixel: 8 lanes read a whole 128-byte line) to contribute to
// BOTH k-steps of the stage -- so k-step h takes channels 4q + 2h, 4q + 2h + 1 of every q: slot (k-half q >> 2, bf16 pair q & 3).
// Channel of slot (h, kh, j): 16 kh + 4 (j >> 1) + 2 h + (j & 1).  (Rounds 2-4 had the channels in order, k-step = channel >> 4.)
__global__ void k_relayout_weight_split(const float *w, uint4 *dst, int Cout, int Cin, int taps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (tap, n, group of 8 k slots)
    const int kg = Cin / 8;
    if (i >= (int64_t)taps * Cout * kg) return;
    const int tap = (int)(i / ((int64_t)Cout * kg));
    const int64_t j = i - (int64_t)tap * Cout * kg;
    const int n = (int)(j / kg), g = (int)(j - (int64_t)n * kg);
    const int kc = g >> 2, q = g & 3, h = q >> 1, kh = q & 1;   // q = k-step * 2 + k-half
    const float *src = w + ((int64_t)n * Cin + kc * SKC + 16 * kh + 2 * h) * taps + tap;
    uint32_t P[3][4];
#pragma unroll
    for (int e = 0; e < 4; e++) split2(src[(4 * e) * taps], src[(4 * e + 1) * taps], P[0][e], P[1][e], P[2][e]);
    const int nt = n >> 7, row = n & 127;
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
    const int q = tid & 7, rb = tid >> 3;         // the thread's channel pair of k-step h: slot (k-half q >> 2, pair q & 3)
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
    uint32_t *As = reinterpret_cast<uint32_t *>(smem);
    u32x4 *Bs = reinterpret_cast<u32x4 *>(smem + SOPER);
    const float *cf = reinterpret_cast<const float *>(smem + (1 + NB) * SOPER);
    int cfo[NV];                                  // this thread's rows' coefficient rows in the LDS table
#pragma unroll
    for (int v = 0; v < NV; v++) cfo[v] = (nsamp > 1 ? min(rb + RSTEP * v, mrem - 1) / HWo : 0) * Cin + 4 * q;   // rows beyond M: the last valid sample's (written) coefficients
    const int wofs = ((q >> 2) * 128 + (rb ^ (q & 4))) * 4 + (q & 3);   // dword of k-step 0; k-step 1: + 256 chunks
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
            for (int pl = 0; pl < 3; pl++) {
                As[pl * (SPLANE / 4) + wofs + v * (RSTEP * 4)] = P[pl][0];
                As[pl * (SPLANE / 4) + wofs + v * (RSTEP * 4) + 1024] = P[pl][1];
            }
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
    const int ax0 = l31 ^ (kh * 4), ax1 = ax0;   // swizzled row of this lane
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


// ---- Round 5: the same GEMM with the stage pipeline inside the workgroup -------------------------------------------------------------
// k_conv_split (above) holds ONE 32-channel stage in LDS: store -> barrier -> MFMAs -> barrier, so within a workgroup nothing runs
// beside the MFMAs and nothing beside the staging pass; it lives off a second workgroup on the CU being in the other half of that
// cycle.  Counters of round 4: MFMA busy 0.48-0.60, 39 % of a workgroup's loop at its two barriers.
// Here a stage is ONE k-step (16 channels: 12 KB per operand) and LDS holds two of them per operand (48 KB: still two workgroups per
// CU), so iteration i = { barrier; read the fragments of k-step i; split + write k-step i + 1 into the other buffer; issue the global
// loads of the steps behind it; the 12 MFMAs of k-step i } -- one barrier per k-step, a wave that arrives there has already written its
// share of the next one, and the staging instructions of a wave sit in the same basic block as its MFMAs.
// Operands wait in registers two steps deep: activations per 32-channel pair of k-steps (two float4 per thread: whole 128-byte lines;
// the weight image's k-slot order lets every thread feed both k-steps), weights per k-step (three 8-byte loads per thread = the 12 KB
// image of the step, copied to LDS as it lies).  Per-output accumulation order = k_conv_split's: same bits.
__global__ void __launch_bounds__(512, 4) k_conv_split_pipe(ConvLaunch p, int nsamp, int xcd_map) {
    constexpr int KB = 12 * 1024;                 // one k-step of one operand: 3 planes x [k-half 2][row 128] x 16 B
    extern __shared__ __align__(16) unsigned char smem[];   // [A0][A1][B0][B1][GroupNorm coefficients of the tile's samples]
    DLPM_PHASE_DECL;
#ifdef DLPM_PHASE_TIMING
    const long long _c0 = clock64(), _r0 = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;      // 2 x 4 waves of 64 pixels x 32 channels
    const int Cin = p.C0 + p.C1;
    const int ntile_n = p.Cout >> 7;
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
    const int n0 = nt * 128;
    const int HWo = p.Hout * p.Wout;
    const int q = tid & 7, rb = tid >> 3;         // staging: channels 4q .. 4q+3 of pixels rb and rb + 64
    const int64_t M = (int64_t)p.B * HWo;
    const int mrem = (int)min((int64_t)BM, M - m0);
    // Addresses = a wave-uniform 64-bit base (SGPRs: a buffer resource at the tile's first row) + a scalar offset per stage + the thread's
    // 32-bit byte offset: the loop keeps no pointer in VGPRs (at 128 registers a spill costs twice here -- scratch reloads count on
    // vmcnt and drain the operand loads in flight).
    uint32_t rcl[2];      // row of the tile (clamped: a ragged last tile re-reads its last valid pixel)
#pragma unroll
    for (int v = 0; v < 2; v++) rcl[v] = min(rb + 64 * v, mrem - 1);
    const int nkc = Cin / SKC;
    const int npairs = nkc, nks = 2 * npairs;
    const bool has_coef = p.coefA != nullptr;
    // (offsets stay far below the 2-GB window: a tile is 128 rows of <= 2 KB; the weights of one 128-channel tile are Cin * 768 B)
    auto rsrc = [](const void *base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000); };
    const __amdgpu_buffer_rsrc_t r0 = rsrc(p.src0 + m0 * p.C0), r1 = rsrc(p.src1 ? p.src1 + m0 * p.C1 : p.src0 + m0 * p.C0);
    const __amdgpu_buffer_rsrc_t rw = rsrc(reinterpret_cast<const unsigned char *>(p.w_split) + (int64_t)nt * nkc * SOPER);
    const uint32_t q16 = 16 * q, t8 = 8 * tid;

    // activations of pair s (channels 32 s .. 32 s + 31) -> registers
    auto load_a = [&](f32x4 (&xa)[2], int s) {
        const int c0 = s * SKC;
        const bool first = c0 < p.C0;             // uniform: C0 % 32 == 0
        const __amdgpu_buffer_rsrc_t r = first ? r0 : r1;
        const int so = 4 * (first ? c0 : c0 - p.C0);
        const uint32_t rs4 = 4u * (uint32_t)(first ? p.C0 : p.C1);
#pragma unroll
        for (int v = 0; v < 2; v++) xa[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(rcl[v] * rs4 + q16), so, 0));
    };
    // weights of k-step i (pair i >> 1, half i & 1): plane pl of the step = 4 KB at plane * 8 KB + half * 4 KB of the pair's image
    auto load_b = [&](u32x2 (&wb)[3], int i) {
        const int so = (i >> 1) * SOPER + (i & 1) * (SPLANE / 2);
#pragma unroll
        for (int pl = 0; pl < 3; pl++) wb[pl] = __builtin_amdgcn_raw_buffer_load_b64(rw, (int)t8, so + pl * SPLANE, 0);
    };
    const float *cf = reinterpret_cast<const float *>(smem + 4 * KB);
    int cfo[2];
#pragma unroll
    for (int v = 0; v < 2; v++) cfo[v] = (nsamp > 1 ? min(rb + 64 * v, mrem - 1) / HWo : 0) * Cin + 4 * q;
    // GroupNorm affine and SiLU, once per loaded value (in place, before half 0 is split)
    auto activate = [&](f32x4 (&xa)[2], int s) {
        const int c0 = s * SKC;
#pragma unroll
        for (int v = 0; v < 2; v++) {
            f32x4 x = xa[v];
            if (has_coef) {
                const f32x4 ca = *reinterpret_cast<const f32x4 *>(cf + cfo[v] + c0);
                const f32x4 cb = *reinterpret_cast<const f32x4 *>(cf + nsamp * Cin + cfo[v] + c0);
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
            xa[v] = x;
        }
    };
    // half h of a pair -> buffer `buf`: dword (k-half q >> 2, row ^ 4 kh, pair q & 3) of each plane
    const int wofs = ((q >> 2) * 128 + (rb ^ (q & 4))) * 4 + (q & 3);
    auto stage_a = [&](const f32x4 (&xa)[2], int h, int buf) {
        uint32_t *As = reinterpret_cast<uint32_t *>(smem + buf * KB);
#pragma unroll
        for (int v = 0; v < 2; v++) {
            uint32_t P0, P1, P2;
            if (h == 0) split2(xa[v].x, xa[v].y, P0, P1, P2);
            else split2(xa[v].z, xa[v].w, P0, P1, P2);
            As[wofs + v * 256] = P0;
            As[1024 + wofs + v * 256] = P1;
            As[2048 + wofs + v * 256] = P2;
        }
    };
    auto stage_b = [&](const u32x2 (&wb)[3], int buf) {
        u32x2 *Bs = reinterpret_cast<u32x2 *>(smem + (2 + buf) * KB) + tid;
#pragma unroll
        for (int pl = 0; pl < 3; pl++) Bs[pl * 512] = wb[pl];
    };

    floatx16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int aofs = kh * 128 + wm * 64 + (l31 ^ (kh * 4)), bofs = kh * 128 + wn * 32 + l31;
    bf16x8 A[2][3], Bf[3];
    auto read_frags = [&](int buf) {
        const bf16x8 *af = reinterpret_cast<const bf16x8 *>(smem + buf * KB) + aofs;
        const bf16x8 *bf = reinterpret_cast<const bf16x8 *>(smem + (2 + buf) * KB) + bofs;
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            A[0][pl] = af[pl * 256];
            A[1][pl] = af[pl * 256 + 32];
            Bf[pl] = bf[pl * 256];
        }
    };
    auto mfmas = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 2; i++) {   // smallest terms first within a (tile, k-step)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][2], Bf[0], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], Bf[1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], Bf[2], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], Bf[0], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], Bf[1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], Bf[0], acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    f32x4 xa0[2], xa1[2];
    u32x2 wb0[3], wb1[3];
    // GroupNorm coefficients of the tile's samples -> LDS [A | B][nsamp][Cin]; their loads go out first (loads retire in order)
    f32x4 cq[2] = {};
    const int n4 = has_coef ? nsamp * Cin / 4 : 0;
    const int nv4 = has_coef ? (int)min((int64_t)n4, (p.B - m0 / HWo) * (int64_t)(Cin / 4)) : 0;
    if (has_coef) {
        const int64_t pb0 = m0 / HWo;
        const int i = min(tid, nv4 - 1);
        cq[0] = reinterpret_cast<const f32x4 *>(p.coefA + pb0 * Cin)[i];
        cq[1] = reinterpret_cast<const f32x4 *>(p.coefB + pb0 * Cin)[i];
    }
    // (every prologue load is unconditional -- a one-pair GEMM re-reads pair 0 / k-step 1: the loop's vmcnt counts are the minimum
    // over the paths into it)
    load_a(xa0, 0);
    load_b(wb0, 0);
    load_b(wb1, 1);
    load_a(xa1, min(1, npairs - 1));
    if (has_coef) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(smem + 4 * KB);
        if (tid < n4) { dst[tid] = cq[0]; dst[n4 + tid] = cq[1]; }
        const int64_t pb0 = m0 / HWo;
        for (int i = tid + 512; i < nv4; i += 512) {
            dst[i] = reinterpret_cast<const f32x4 *>(p.coefA + pb0 * Cin)[i];
            dst[n4 + i] = reinterpret_cast<const f32x4 *>(p.coefB + pb0 * Cin)[i];
        }
        __syncthreads();
    }
    activate(xa0, 0);
    stage_a(xa0, 0, 0);
    stage_b(wb0, 0);
    load_b(wb0, min(2, nks - 1));
    DLPM_PHASE(p, 0);
    // iteration i: k-step i from buffer i & 1; k-step i + 1 -> the other buffer.  Unrolled by 4 = two pairs: pair (i >> 1) & 1 lives
    // in xa0 / xa1, the weights of k-step i in wb0 / wb1 by parity; each register set is reloaded as soon as it has been staged.
    // (The steady-state trips carry no condition around a load: hipcc's vmcnt counts are exact only when every path into a wait has
    // issued the same loads -- with the tail's guards in the loop every staging pass waited for vmcnt(0).)
    auto four_steps = [&](int i, auto guard) {
        constexpr bool G = decltype(guard)::value;   // tail trips: what lies beyond the last k-step is neither loaded nor staged
        {   // k-step i (pair P = i / 2, half 0); stage half 1 of P (xa0), then xa0 <- pair P + 2
            __syncthreads();
            read_frags(0);
            stage_a(xa0, 1, 1);
            stage_b(wb1, 1);
            if (!G || i + 4 < nks) load_a(xa0, (i >> 1) + 2);
            if (!G || i + 3 < nks) load_b(wb1, i + 3);
            mfmas();
        }
        {   // k-step i + 1 (half 1 of P); stage half 0 of P + 1 (xa1)
            __syncthreads();
            read_frags(1);
            if (!G || i + 2 < nks) {
                activate(xa1, (i >> 1) + 1);
                stage_a(xa1, 0, 0);
                stage_b(wb0, 0);
                if (!G || i + 4 < nks) load_b(wb0, i + 4);
            }
            mfmas();
        }
        if (!G || i + 2 < nks) {
            {   // k-step i + 2 (half 0 of P + 1); stage its half 1, then xa1 <- pair P + 3
                __syncthreads();
                read_frags(0);
                stage_a(xa1, 1, 1);
                stage_b(wb1, 1);
                if (!G || i + 6 < nks) load_a(xa1, (i >> 1) + 3);
                if (!G || i + 5 < nks) load_b(wb1, i + 5);
                mfmas();
            }
            {   // k-step i + 3; stage half 0 of P + 2 (xa0)
                __syncthreads();
                read_frags(1);
                if (!G || i + 4 < nks) {
                    activate(xa0, (i >> 1) + 2);
                    stage_a(xa0, 0, 0);
                    stage_b(wb0, 0);
                    if (!G || i + 6 < nks) load_b(wb0, i + 6);
                }
                mfmas();
            }
        }
    };
    int i = 0;
    for (; i + 6 < nks; i += 4) four_steps(i, std::false_type());
    for (; i < nks; i += 4) four_steps(i, std::true_type());
    DLPM_PHASE(p, 1);
    floatx16 accs[2][1];
    accs[0][0] = acc[0];
    accs[1][0] = acc[1];
    if (p.stats_out) {   // 8x8 images (launch_conv_split): per-image statistics from the registers
        if (mrem == BM) split_store_from_registers<false, 1, true>(p, accs, m0, n0 + wn * 32, wm, l31, kh, BM);
        else split_store_from_registers<true, 1, true>(p, accs, m0, n0 + wn * 32, wm, l31, kh, mrem);
    } else if (mrem == BM) split_store_from_registers<false, 1>(p, accs, m0, n0 + wn * 32, wm, l31, kh, BM);
    else split_store_from_registers<true, 1>(p, accs, m0, n0 + wn * 32, wm, l31, kh, mrem);
    DLPM_PHASE(p, 2);
#ifdef DLPM_PHASE_TIMING
    if (p.phase && tid == 0) {
        atomicAdd(p.phase + 3, 1ull);
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
    // Round 5: the pipelined kernel (one k-step per LDS stage, two stages, one barrier per k-step) wherever the 8-wave shape ran;
    // DLPM_SPLIT_PIPE=0 restores k_conv_split<8, 1, *> for A/B runs (same bits).  The 4-wave shape carries the row epilogue with the
    // fused statistics.  All shapes accumulate every output in the same order: which one runs does not change a bit of the result.
    static int pipe = -1;
    if (pipe < 0) { const char *e = getenv("DLPM_SPLIT_PIPE"); pipe = e ? atoi(e) : 1; }
    // (the pipelined kernel's LDS limit is 80 KB = 48 KB of stages + the fused GroupNorm coefficients, nsamp x Cin x 8 bytes; launches
    //  beyond it -- Cin > 4096 with coefficients -- stay on k_conv_split<8, 1, 1>, whose 104-KB limit takes Cin up to 7168: ADVICE r05)
    const int lds_p = 48 * 1024 + (c.coefA ? nsamp * (c.C0 + c.C1) * 8 : 0);
    if (pipe && c.ks == 1 && !row_stats && !wide && lds_p <= 80 * 1024) {
        const int r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_conv_split_pipe), 80 * 1024);
        if (r != DLPM_OK) return r;
        k_conv_split_pipe<<<grid, 512, lds_p, st>>>(c, nsamp, xcd_map);
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
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
