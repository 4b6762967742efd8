// head_fused.hip -- the UNet's head convolution AND the sampler's reverse update as ONE pass over HBM (round 4).
//
// Replaces out = conv3x3(silu(GN(h))) (dlpm/models/unet.py:433-435, Cout = image channels) followed by
// x <- (x - c_eps eps) / gamma + c_noise z (dlpm/methods/dlpm.py:272-278, GenerativeLevyProcess.py:225-239).
//
// Round 3 split the head into a 1x1 GEMM onto its 9 Cout tap channels  P = act(h) W'  (every input line fetched once) and a 9-point
// gather over P that carried the update: the pair wrote and re-read the 113 MB of P that the algorithm does not have (0.195 ms,
// 0.37 of the HBM roof on the bytes the algorithm does have: the head's input once + 12 B per state element).  Here P never
// leaves the CU: one workgroup owns a band of TH output rows of one image (the WHOLE 32x32 image for the CIFAR shape: no halo
// rows to recompute), its 8 waves walk the band's rows as 32-pixel MFMA tiles
//     P[pixel][n = tap Cout + co] = sum_ci act(h)[pixel][ci] W'[ci][n]        (v_mfma_f32_32x32x2_f32, 27 of 32 columns live)
// and drop them into an LDS image [rows + 2][W][9 Cout]; after one barrier the gather  eps[p, co] = b + sum_tap P[p + off(tap)][tap, co]
// runs from LDS and applies the update to the NCHW state in place with the Philox counters of k_update_rows.  HBM traffic = the
// algorithmic bytes (+ the halo rows of a band when the image does not fit: 64x64).
//
// A wave's tile: 32 consecutive pixels of one row x 32 input channels per K chunk.  Global loads are line-coalesced (8 lanes x 16
// bytes = the 32 channels of a pixel), GroupNorm affine + SiLU are applied in registers, and a wave-private 4.6-KB LDS patch turns
// [pixel][channel] into the MFMA's A layout (lane = pixel, k slot = channel): the k index of an MFMA is free as long as A and B
// agree, so slot kh of MFMA (q, e) is channel 32 c + 8 q + 4 kh + e and one ds_read_b128 feeds four MFMAs.  W' sits in registers
// for the whole kernel in that fragment order (64 VGPRs at 128 channels).
//
// BF = true (the default pipe; DLPM_HEAD_F32=1 or the fp32 GEMM policy take the form above): the same GEMM on the bf16 matrix pipe with
// conv_split.hip's exact three-plane split -- act(h) is cut into three bf16 planes in registers right after the SiLU (and + subtract twice,
// v_perm packs), W' is cut once at finalize, six v_mfma_f32_32x32x16_bf16 per 16 channels replace eight fp32 MFMAs per 16 channels at a
// quarter of the cycles each (the fp32 form was SIMD-issue-bound: fp32 MFMAs 40 % + VALU 36 % of its cycles, and the fp32 MFMA shares
// the vector lanes; the bf16 one runs beside the VALU).  The transpose patch holds [plane][pixel][32 channels] bf16 with the 16-byte
// units of a pixel row XOR-swizzled by (pixel >> 1) & 3 (no padding: P + the patches fill the 160 KB).
#include "conv.h"
#include <type_traits>
#include "philox.h"

namespace dlpm {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HF_NT = 512;        // 8 waves
constexpr int HF_SLD = 36;        // floats per pixel row of a wave's transpose patch (32 channels + 4: conflict-free b128 reads)
constexpr int HF_MAXCH = 4;       // K chunks of 32 channels the register-resident weights cover (Cin <= 128)
constexpr int HF_STG_F32 = 32 * HF_SLD;   // dwords of a wave's transpose patch, fp32 form
constexpr int HF_STG_BF = 3 * 32 * 16;    // ... bf16 form: [plane][pixel][32 bf16]
constexpr float HF_LOG2E = 1.44269504088896340736f;

// fp32 pair -> the packed bf16 pairs of its three planes (conv_split.hip's split, exact: 8 + 8 + 8 significand bits)
__device__ __forceinline__ void hf_split2(float lo, float hi, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
    const uint32_t ul = __float_as_uint(lo), uh = __float_as_uint(hi);
    p0 = __builtin_amdgcn_perm(uh, ul, 0x07060302u);
    f32x2 r = f32x2{lo, hi} - f32x2{__uint_as_float(ul & 0xffff0000u), __uint_as_float(uh & 0xffff0000u)};
    const uint32_t vl = __float_as_uint(r.x), vh = __float_as_uint(r.y);
    p1 = __builtin_amdgcn_perm(vh, vl, 0x07060302u);
    f32x2 q = r - f32x2{__uint_as_float(vl & 0xffff0000u), __uint_as_float(vh & 0xffff0000u)};
    p2 = __builtin_amdgcn_perm(__float_as_uint(q.y), __float_as_uint(q.x), 0x07060302u);
}

// silu(x a + b) on a pair: silu_f's arithmetic (v * rcp(1 + exp2(-v log2e))), written on 2-vectors so that the multiplies and adds
// become v_pk_*_f32
__device__ __forceinline__ f32x2 hf_act2(f32x2 x, f32x2 a, f32x2 b) {
    const f32x2 v = __builtin_elementwise_fma(x, a, b);
    const f32x2 t = -v * HF_LOG2E;
    const f32x2 d = 1.0f + f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    return v * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

#define HF_LDS_EXCHANGE()                                             \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
        __builtin_amdgcn_wave_barrier();                              \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); \
    } while (0)

// rows of P a workgroup keeps: its TH output rows + the halo rows that lie inside the picture
__host__ __device__ __forceinline__ int hf_rows_kept(int H, int TH) { return H == TH ? TH : (H == 2 * TH ? TH + 1 : TH + 2); }

// workgroup barrier that orders LDS only: __syncthreads() is a workgroup-scope fence over ALL address spaces, i.e. s_waitcnt vmcnt(0) --
// every wave would sit out the write acknowledgements of its update stores and the landing of its prefetch at each barrier
#define HF_LDS_BARRIER()                                              \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
        __builtin_amdgcn_s_barrier();                                 \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
    } while (0)

struct HeadFusedArgs {
    const float *h;               // [B][H][W][C] NHWC
    const float *coefA, *coefB;   // [B][C] GroupNorm affine (SiLU follows)
    const float *wf;              // W' in fragment order: fp32 [C/32][4][64][4], behind it the bf16 planes [C/32][2][3][64][8]
    const float *bias;            // [Cout]
    float *out;                   // eps (plain forward): NCHW [B][Cout][H][W] or NHWC
    int out_nchw;
    int B, H, W, C, TH;
    HeadUpdate u;
#ifdef DLPM_PHASE_TIMING
    unsigned long long *phase;    // developer builds: 0 prologue, 1 tile loop, 2 barrier + gather/update, 3 workgroups, 12/13 clock, 16+w barrier wait
#endif
};

// One tile of a wave (expanded twice in the kernel: the tiles of a band that have a successor in it, and the band's last one).
// step(c, NEXT): chunk c's A fragments out of the patch, then its MFMAs next to the staging of the wave's next chunk (NEXT: 1 = chunk
// c + 1 of this tile, 2 = chunk 0 of the next tile, 0 = none: the band's last step).  Wave-private exchanges: LDS operations of one wave
// execute in order; the LDS-only fences keep the compiler from moving them (a plain wavefront fence also drains the GLOBAL loads in
// flight: the prefetch).  The sched_barrier at the end of a step: nothing of a later step moves up -- hipcc otherwise hoists the
// GroupNorm FMAs of ALL later chunks between this chunk's MFMAs and waits for loads issued one chunk ago.  ns = the tile after t
// (chunks 1.. are reloaded from it), ns2 = the one after that (chunk 0).  D layout of the 32x32 MFMA: register i holds row
// 8 (i / 4) + 4 kh + (i % 4) (pixel), column lm (tap channel).  (A macro, not a lambda: called twice, a lambda is not inlined before
// SROA and every register array of the kernel lands in scratch.)
#define HF_TILE(LASTV)                                                                                  \
    do {                                                                                                \
        constexpr bool last = LASTV;                                                                    \
            floatx16 acc; \
_Pragma("unroll") \
            for (int r = 0; r < 16; r++) acc[r] = 0.f; \
            const float *ns = !last ? tile_src(wg, t + 8) : (more ? tile_src(wgn, wave) : p.wf + 4 * lc); \
            const float *ns2 = t + 16 < ntile ? tile_src(wg, t + 16) \
                                              : (more ? tile_src(wgn, !last ? wave : wave + 8) : p.wf + 4 * lc); \
            auto step = [&](auto cc, auto nn) __attribute__((always_inline)) { \
                constexpr int c = decltype(cc)::value, NEXT = decltype(nn)::value; \
                if constexpr (BF) { \
                    bf16x8 A[2][3]; \
_Pragma("unroll") \
                    for (int j = 0; j < 2; j++) \
_Pragma("unroll") \
                        for (int pl = 0; pl < 3; pl++) \
                            A[j][pl] = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const uint32_t *>(stg) + pl * 512 + lm * 16 + \
                                                                         4 * ((2 * j + kh) ^ rd_sw)); \
                    HF_LDS_EXCHANGE(); \
                    if constexpr (NEXT == 1) { stage(c + 1); load_chunk(ns, c + 1); } \
                    if constexpr (NEXT == 2) { stage(0); load_chunk(ns2, 0); } \
_Pragma("unroll") \
                    for (int j = 0; j < 2; j++) { \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][2], bwb[c][j][0], acc, 0, 0, 0); \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][1], bwb[c][j][1], acc, 0, 0, 0); \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][0], bwb[c][j][2], acc, 0, 0, 0); \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][1], bwb[c][j][0], acc, 0, 0, 0); \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][0], bwb[c][j][1], acc, 0, 0, 0); \
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][0], bwb[c][j][0], acc, 0, 0, 0); \
                    } \
                } else { \
                    float4 a[4]; \
_Pragma("unroll") \
                    for (int q = 0; q < 4; q++) a[q] = *reinterpret_cast<const float4 *>(stg + lm * HF_SLD + 8 * q + 4 * kh); \
                    HF_LDS_EXCHANGE(); \
                    if constexpr (NEXT == 1) { stage(c + 1); load_chunk(ns, c + 1); } \
                    if constexpr (NEXT == 2) { stage(0); load_chunk(ns2, 0); } \
_Pragma("unroll") \
                    for (int q = 0; q < 4; q++) { \
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bw[c][q].x, acc, 0, 0, 0); \
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bw[c][q].y, acc, 0, 0, 0); \
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bw[c][q].z, acc, 0, 0, 0); \
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bw[c][q].w, acc, 0, 0, 0); \
                    } \
                } \
                HF_LDS_EXCHANGE(); \
                __builtin_amdgcn_sched_barrier(0); \
            }; \
            auto store_tile = [&]() __attribute__((always_inline)) { \
                if (lm < NV) { \
                    const int sr = t / tpr, x0 = (t - sr * tpr) * 32; \
                    float *dst = P + lm * PL + sr * W + x0 + 4 * kh; \
_Pragma("unroll") \
                    for (int j = 0; j < 4; j++) \
                        *reinterpret_cast<float4 *>(dst + 8 * j) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]); \
                } \
            }; \
            if constexpr (NCH > 1) step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); \
            if constexpr (NCH > 2) step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); \
            if constexpr (NCH > 3) step(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}); \
            step(std::integral_constant<int, NCH - 1>{}, std::integral_constant<int, last ? 0 : 2>{}); \
            store_tile(); \
    } while (0)

// NCH = C / 32 K chunks, a compile-time constant: with a run-time bound the chunk loop's loads sit inside (uniform) branches and hipcc
// drains them at every join (s_waitcnt vmcnt(0)).
//
// PERSISTENT: the grid is one workgroup per CU and a workgroup walks the bands wg = blockIdx.x, + gridDim.x, ...: W' is fetched once
// (all 256 CUs pulling the same 24 KB four times over was 10 k of a band's 75 k cycles), and the next band's coefficients and first
// tile are requested before the barrier, so HBM keeps streaming while the gather runs.
// PIPELINED: step (t, c) of a wave = read chunk c's A fragments from the patch, then MFMAs(c) next to stage(c + 1) -- SiLU + split of the
// NEXT chunk into the patch -- in one scheduling region: the bf16 MFMA runs beside the VALU, and a wave alone on its SIMD (the
// older wave of a SIMD finishes its tiles first) keeps both busy.
template <int COUT, int NCH, bool BF>
__global__ void __launch_bounds__(HF_NT, 1) k_head_fused(HeadFusedArgs p) {
    constexpr int NV = 9 * COUT;                       // live tap channels (27)
    constexpr int C = 32 * NCH;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.W, H = p.H, TH = p.TH;
    const int bands = H / TH, nwg = p.B * bands, G = gridDim.x;
    // P as NV planes [row s - s_lo][W] of pitch PL = rows W + 4 floats: a lane's four consecutive pixels are one 16-byte LDS access on
    // both sides (MFMA registers 4 j .. 4 j + 3 in; a gather item's pixel quad out), the + 4 spreads the 27 lanes of a store over the banks
    const int PL = hf_rows_kept(H, TH) * W + 4;
    float *P = sm;
    float *stg = sm + NV * PL + wave * (BF ? HF_STG_BF : HF_STG_F32);
    float *cf = sm + NV * PL + 8 * (BF ? HF_STG_BF : HF_STG_F32);   // [2][C] GroupNorm affine of the band's image (registers are short)
    const int tpr = W >> 5;                            // 32-pixel tiles per row
    const int64_t HW = (int64_t)H * W;
    const int lp = lane >> 3, lc = lane & 7;           // load role: pixel 8 i + lp, channels 4 lc .. 4 lc + 3 of the chunk
    const int lm = lane & 31, kh = lane >> 5;          // MFMA role: pixel lm, k slot kh
    // bf16 form, dword offsets into the wave's patch: a pixel row is 16 dwords = four 16-byte units (8 channels each), unit u of pixel
    // px sits at slot u ^ ((px >> 1) & 3) -- the 8 lanes of a b128 read phase (8 consecutive pixels, same u) then cover all 32 banks
    const int wr_off = lp * 16 + 4 * ((lc >> 1) ^ ((lp >> 1) & 3)) + 2 * (lc & 1);      // + plane 512 + i 128
    const int rd_sw = (lm >> 1) & 3;                                                      // unit 2 j + kh -> slot ^ rd_sw

    // ---- W' fragments: registers for the whole kernel
    float4 bw[BF ? 1 : NCH][4];
    bf16x8 bwb[BF ? NCH : 1][2][3];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if constexpr (BF) {
            const bf16x8 *wb = reinterpret_cast<const bf16x8 *>(p.wf + (int64_t)C * 32);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) bwb[c][j][pl] = wb[((c * 2 + j) * 3 + pl) * 64 + lane];
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) bw[c][q] = reinterpret_cast<const float4 *>(p.wf)[(c * 4 + q) * 64 + lane];
        }
    }

    // band geometry: image b, first output row y0; row slots of P are image rows y0 - 1 .. y0 + TH, of which only those inside the
    // picture are kept (s_lo .. s_hi - 1)
    auto band_y0 = [&](int wg) __attribute__((always_inline)) { return (wg - (wg / bands) * bands) * TH; };
    auto tile_src = [&](int wg, int t) __attribute__((always_inline)) {               // tile t of band wg: 32 pixels of one row, this lane's channel quad of chunk 0
        const int b = wg / bands, y0 = band_y0(wg), s = (y0 == 0 ? 1 : 0) + t / tpr, x0 = (t - (t / tpr) * tpr) * 32;
        return p.h + (((int64_t)b * H + (y0 - 1 + s)) * W + x0) * C + 4 * lc;
    };
    // Input chunks travel ONE TILE ahead: buffer c holds chunk c of the tile being worked on and is reloaded with chunk c of the
    // wave's next tile as soon as it has been staged (16 KB per wave in flight).
    float4 xb[NCH][4];
    auto load_chunk = [&](const float *src, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; i++) xb[c][i] = *reinterpret_cast<const float4 *>(src + 32 * c + (int64_t)(8 * i + lp) * C);
    };
    // the image's coefficients: threads 0 .. C / 2 - 1 fetch one float4 each (A then B) ...
    auto fetch_coefs = [&](int wg) __attribute__((always_inline)) {
        const int b = wg / bands;
        const float *src = tid < C / 4 ? p.coefA + (int64_t)b * C + 4 * tid : p.coefB + (int64_t)b * C + 4 * (tid - C / 4);
        return tid < C / 2 ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // ... and drop it into LDS once nobody reads the previous image's any more (after the band's first barrier)
    auto put_coefs = [&](float4 v) __attribute__((always_inline)) {
        if (tid < C / 2) *reinterpret_cast<float4 *>(cf + 4 * tid) = v;
    };
    // stage(c): GroupNorm affine + SiLU (+ the three-plane split) of buffer c into the wave's patch
    auto stage = [&](int c) __attribute__((always_inline)) {
        const float4 cA = *reinterpret_cast<const float4 *>(cf + 32 * c + 4 * lc), cB = *reinterpret_cast<const float4 *>(cf + C + 32 * c + 4 * lc);
        const f32x2 a01 = {cA.x, cA.y}, a23 = {cA.z, cA.w}, b01 = {cB.x, cB.y}, b23 = {cB.z, cB.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 xr = xb[c][i];
            const f32x2 v01 = hf_act2(f32x2{xr.x, xr.y}, a01, b01), v23 = hf_act2(f32x2{xr.z, xr.w}, a23, b23);
            if constexpr (BF) {
                uint32_t q0[2], q1[2], q2[2];
                hf_split2(v01.x, v01.y, q0[0], q1[0], q2[0]);
                hf_split2(v23.x, v23.y, q0[1], q1[1], q2[1]);
                uint32_t *dst = reinterpret_cast<uint32_t *>(stg) + wr_off + i * 128;
                *reinterpret_cast<uint2 *>(dst) = make_uint2(q0[0], q0[1]);
                *reinterpret_cast<uint2 *>(dst + 512) = make_uint2(q1[0], q1[1]);
                *reinterpret_cast<uint2 *>(dst + 1024) = make_uint2(q2[0], q2[1]);
            } else {
                *reinterpret_cast<float4 *>(stg + (8 * i + lp) * HF_SLD + 4 * lc) = make_float4(v01.x, v01.y, v23.x, v23.y);
            }
        }
    };

    const HeadUpdate &u = p.u;
    const int nq = W >> 2, per_co = TH * nq, nitem = COUT * per_co;
    const int64_t D = (int64_t)COUT * HW;
    int wg = blockIdx.x;
    put_coefs(fetch_coefs(wg));
    {
        const float *src = tile_src(wg, wave);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            load_chunk(src, c);
            // chunk order, as the loop reloads them (hipcc sorts these 16 loads by address otherwise, and the loop's static vmcnt
            // waits are the merge of all entry paths)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    HF_LDS_BARRIER();
    for (;;) {
#ifdef DLPM_PHASE_TIMING
        const long long _c0 = clock64(), _r0 = wall_clock64();
#endif
        const int b = wg / bands, y0 = band_y0(wg);
        const int s_lo = y0 == 0 ? 1 : 0, s_hi = (y0 + TH == H) ? TH + 1 : TH + 2;
        const int ntile = (s_hi - s_lo) * tpr;         // >= 8 (head_fused_ok): every wave has a tile
        const int wgn = wg + G;
        const bool more = wgn < nwg;

        // ---- phase A: the band's rows as 32-pixel tiles, round-robin over the waves
        stage(0);
        {   // (the tile after this one: in this band, else the first of the next band, else a harmless L2-resident address --
            //  the loads stay unconditional, so that every path into the loop has the same loads in flight)
            const float *n0 = wave + 8 < ntile ? tile_src(wg, wave + 8) : (more ? tile_src(wgn, wave) : p.wf + 4 * lc);
            load_chunk(n0, 0);
        }
        HF_LDS_EXCHANGE();
        __builtin_amdgcn_sched_barrier(0);
        // The band's last tile is its own copy of the code, behind the loop: as a branch inside one copy ("stage chunk 0 of the next
        // tile unless this is the last"), hipcc's waitcnt pass sees a path around those loads into the loop's back edge and counts every
        // wait of the loop for it (vmcnt(11) instead of (15): the read-ahead one chunk shorter)
        int t = wave;
        for (; t + 8 < ntile; t += 8) HF_TILE(false);
        HF_TILE(true);

        // ---- phase B: 9-point gather from LDS + the reverse update (k_head_gather's arithmetic, tap by tap in its order; taps outside
        // the picture add a literal zero where k_head_gather adds the zero its padded P holds).  An item = (channel, row, 4 pixels); at
        // most two per thread (head_fused_ok).  The state quads and the next band's coefficients are requested BEFORE the barrier: the
        // waves that finish their tiles first wait there anyway.
        int64_t e0k[2];
        int cok[2], rk[2], qk[2];
        float4 xk[2];
        float bvk[2];
        // (recomputed per band behind an opaque copy of the thread index: hoisted out of the band loop as invariants, the item
        //  coordinates live across the tile loop, i.e. in scratch, and every reload is an s_waitcnt vmcnt(0))
        int tb = tid;
        asm volatile("" : "+v"(tb));
        // the update's per-image scalars, requested here: three dependent loads (t -> g, c_eps, c_noise) otherwise sit behind the barrier
        int tt = 0;
        float g = 1.f, rg = 1.f, ce = 0.f, cn = 0.f;
        uint64_t seed = 0, gidx = 0;
        float *hr = nullptr;
        if (u.x) {
            tt = *u.t;
            g = u.g[tt];
            rg = 1.0f / g;
            ce = u.c_eps[(int64_t)tt * u.B + b];
            cn = u.c_noise[(int64_t)tt * u.B + b];
            seed = u.key ? u.key[0] : u.seed;
            gidx = (uint64_t)((u.key ? (int64_t)u.key[1] : u.sample_offset) + b);
            hr = u.hist_pp ? *u.hist_pp : nullptr;
            if (hr) hr += ((int64_t)(u.T - tt) * u.B + b) * D;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int it = tb + k * HF_NT;
            cok[k] = it / per_co;
            const int rq = it - cok[k] * per_co;
            rk[k] = rq / nq;
            qk[k] = rq - rk[k] * nq;
            e0k[k] = (int64_t)cok[k] * HW + (int64_t)(y0 + rk[k]) * W + 4 * qk[k];
            if (u.x && it < nitem) xk[k] = *reinterpret_cast<const float4 *>(u.x + (int64_t)b * D + e0k[k]);
            bvk[k] = (p.bias && it < nitem) ? p.bias[cok[k]] : 0.f;
        }
        const float4 ncf = fetch_coefs(more ? wgn : wg);
#ifdef DLPM_PHASE_TIMING
        const long long _c2 = clock64();
#endif
        HF_LDS_BARRIER();
#ifdef DLPM_PHASE_TIMING
        const long long _c3 = clock64();
#endif
        put_coefs(ncf);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int it = tb + k * HF_NT;
            if (it >= nitem) break;
            const int co = cok[k], r = rk[k], q = qk[k];
            const float bv = bvk[k];
            float acc[4] = {bv, bv, bv, bv};
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                const int s = r + ky;                                              // slot of image row y0 + r + ky - 1
                const int sc = min(max(s, s_lo), s_hi - 1) - s_lo;
                const bool inb = s >= s_lo && s < s_hi;
                float4 Q[3];                                                       // (row by row: registers are short here)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) Q[kx] = *reinterpret_cast<const float4 *>(P + ((ky * 3 + kx) * COUT + co) * PL + sc * W + 4 * q);
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    float4 v = Q[kx];
                    if (!inb) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kx == 0) {         // pixels 4 q - 1 .. 4 q + 2: the left neighbour's last value comes over the lanes (items of a row are adjacent lanes)
                        float l = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.w), 0x111, 0xf, 0xf, true));   // row_shr:1
                        if (q == 0) l = 0.f;
                        acc[0] += l; acc[1] += v.x; acc[2] += v.y; acc[3] += v.z;
                    } else if (kx == 1) {
                        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                    } else {               // pixels 4 q + 1 .. 4 q + 4
                        float rr = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0x101, 0xf, 0xf, true));  // row_shl:1
                        if (q == nq - 1) rr = 0.f;
                        acc[0] += v.y; acc[1] += v.z; acc[2] += v.w; acc[3] += rr;
                    }
                }
            }
            const int64_t pix = (int64_t)(y0 + r) * W + 4 * q;
            const int64_t e0 = e0k[k];
            if (u.x) {
                const float4 x = xk[k];
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
            } else if (p.out_nchw) {
                *reinterpret_cast<float4 *>(p.out + (int64_t)b * D + e0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else {
#pragma unroll
                for (int px = 0; px < 4; px++) p.out[((int64_t)b * HW + pix + px) * COUT + co] = acc[px];
            }
        }
#ifdef DLPM_PHASE_TIMING
        if (p.phase && lane == 0) {
            atomicAdd(p.phase + 16 + wave, (unsigned long long)(_c3 - _c2));
            if (wave == 0) {
                const long long _c4 = clock64();
                atomicAdd(p.phase + 1, (unsigned long long)(_c2 - _c0));
                atomicAdd(p.phase + 2, (unsigned long long)(_c4 - _c2));
                atomicAdd(p.phase + 3, 1ull);
                atomicAdd(p.phase + 12, (unsigned long long)(_c4 - _c0));
                atomicAdd(p.phase + 13, (unsigned long long)(wall_clock64() - _r0));
            }
        }
#endif
        if (!more) break;
        HF_LDS_BARRIER();                               // the gather has read P: the next band's tiles may land
        wg = wgn;
    }
}

// OIHW (3x3, Cout <= 3) -> W' fragments [C/32][4 q][lane][e]: lane (n = lane % 32, kh = lane / 32) holds W'[ci = 32 c + 8 q + 4 kh + e][n],
// W'[ci][n = tap Cout + co] = w[co][ci][tap], columns n >= 9 Cout zero
__global__ void k_relayout_weight_head_fused(const float *oihw, float *dst, int Cout, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * 32) return;
    const int e = i & 3, lane = (i >> 2) & 63, q = (i >> 8) & 3, c = i >> 10;
    const int n = lane & 31, kh = lane >> 5, ci = 32 * c + 8 * q + 4 * kh + e;
    const int tap = n / Cout, co = n - tap * Cout;
    dst[i] = n < 9 * Cout ? oihw[((int64_t)co * Cin + ci) * 9 + tap] : 0.f;
}

// ... -> the bf16 planes [C/32][2 j][3 planes][lane][8]: lane (n, g = lane / 32) holds W'[ci = 32 c + 16 j + 8 g + e][n], e < 8
__global__ void k_relayout_weight_head_fused_bf(const float *oihw, uint4 *dst, int Cout, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // one thread per (c, j, lane)
    if (i >= (Cin / 32) * 2 * 64) return;
    const int lane = i & 63, j = (i >> 6) & 1, c = i >> 7;
    const int n = lane & 31, g = lane >> 5, ci0 = 32 * c + 16 * j + 8 * g;
    const int tap = n / Cout, co = n - tap * Cout;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; e++) w[e] = n < 9 * Cout ? oihw[((int64_t)co * Cin + ci0 + e) * 9 + tap] : 0.f;
    uint32_t P[3][4];
#pragma unroll
    for (int e = 0; e < 4; e++) hf_split2(w[2 * e], w[2 * e + 1], P[0][e], P[1][e], P[2][e]);
#pragma unroll
    for (int pl = 0; pl < 3; pl++) dst[((c * 2 + j) * 3 + pl) * 64 + lane] = make_uint4(P[pl][0], P[pl][1], P[pl][2], P[pl][3]);
}

bool head_fused_bf(const ConvLaunch &c) {     // which matrix pipe: bf16 x 3 unless the fp32 GEMM policy (or DLPM_HEAD_F32=1) asks otherwise
    static int f32 = -1;
    if (f32 < 0) { const char *e = getenv("DLPM_HEAD_F32"); f32 = (e && e[0] == '1') ? 1 : 0; }
    return !f32 && c.gemm != DLPM_GEMM_F32;
}

size_t head_fused_lds_floats(const ConvLaunch &c, int th) {
    return (size_t)9 * c.Cout * (hf_rows_kept(c.Hout, th) * c.Wout + 4) + 8 * (head_fused_bf(c) ? HF_STG_BF : HF_STG_F32) + 2 * c.C0;
}

int head_fused_rows(const ConvLaunch &c) {   // output rows per workgroup: the whole image when its P image fits beside the transpose patches
    int th = c.Hout;
    while (th > 1 && head_fused_lds_floats(c, th) * sizeof(float) > 160 * 1024) th >>= 1;
    return th;
}

}  // namespace

bool head_fused_ok(const ConvLaunch &c) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("DLPM_NO_HEAD_FUSED"); off = (e && e[0] == '1') ? 1 : 0; }
    if (off || !c.w_hfused || c.ks != 3 || c.stride != 1 || c.ups || c.in_nchw || c.C1 != 0 || c.res0 || !c.coefA || !c.act_silu) return false;
    if (c.Cout < 1 || c.Cout > 3 || c.C0 % 32 != 0 || c.C0 > 32 * HF_MAXCH || c.Hin != c.Hout || c.Win != c.Wout) return false;
    if ((c.Wout & 31) || c.Wout > 64 || c.Hout < 4) return false;
    const int th = head_fused_rows(c);
    const int min_rows = c.Hout == th ? th : th + 1;                                   // rows of P an edge band keeps
    return th >= 4 && c.Hout % th == 0 && c.Cout * th * (c.Wout >> 2) <= 2 * HF_NT &&   // (two gather items per thread)
           min_rows * (c.Wout >> 5) >= 8;                                              // (every wave has a tile)
}

int64_t head_fused_weight_floats(int Cin) { return (int64_t)Cin * 32 + (int64_t)Cin * 48; }   // fp32 fragments + three bf16 planes

int relayout_weight_head_fused(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st) {
    k_relayout_weight_head_fused<<<(unsigned)ceil_div(Cin * 32, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin);
    DLPM_LAUNCH_CHECK();
    k_relayout_weight_head_fused_bf<<<(unsigned)ceil_div((Cin / 32) * 128, 128), 128, 0, st>>>(
        oihw_dev, reinterpret_cast<uint4 *>(dst_dev + (int64_t)Cin * 32), Cout, Cin);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int launch_conv_head_fused(const ConvLaunch &c, const HeadUpdate *hu, hipStream_t st) {
    HeadFusedArgs a{};
    a.h = c.src0; a.coefA = c.coefA; a.coefB = c.coefB; a.wf = c.w_hfused; a.bias = c.bias; a.out = c.out; a.out_nchw = c.out_nchw;
    a.B = c.B; a.H = c.Hout; a.W = c.Wout; a.C = c.C0; a.TH = head_fused_rows(c);
    if (hu) a.u = *hu;
#ifdef DLPM_PHASE_TIMING
    a.phase = phase_buffer();
#endif
    const int64_t M = (int64_t)c.B * c.Hout * c.Wout;
    // algorithmic bytes: the head's input once + the state read and written (or eps written)
    const double bytes = 4.0 * ((double)M * c.C0 + (double)M * c.Cout * (hu ? 2 + (hu->z ? 1 : 0) + (hu->eps_out ? 1 : 0) : 1));
    ProfScope ps(hu ? "head_fused+update" : "head_fused", 2.0 * M * c.Cout * 9.0 * c.C0, bytes, st);
    const size_t lds = head_fused_lds_floats(c, a.TH) * sizeof(float);
    const bool bf = head_fused_bf(c);
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        DLPM_HIP(hipGetDevice(&dev));
        DLPM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int nwg = c.B * (c.Hout / a.TH);
    const unsigned grid = (unsigned)(nwg < ncu ? nwg : ncu);     // persistent: one workgroup per CU walks the bands
#define DLPM_HF1(CO, NCH, BFV)                                                                            \
    do {                                                                                                  \
        int r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_head_fused<CO, NCH, BFV>), 160 * 1024); \
        if (r != DLPM_OK) return r;                                                                       \
        k_head_fused<CO, NCH, BFV><<<grid, HF_NT, lds, st>>>(a);                                          \
    } while (0)
#define DLPM_HF(CO, NCH)                                                                                  \
    do {                                                                                                  \
        if (bf) DLPM_HF1(CO, NCH, true);                                                                  \
        else DLPM_HF1(CO, NCH, false);                                                                    \
    } while (0)
#define DLPM_HFC(CO)                                                                                      \
    do {                                                                                                  \
        switch (c.C0 >> 5) {                                                                              \
            case 1: DLPM_HF(CO, 1); break;                                                                \
            case 2: DLPM_HF(CO, 2); break;                                                                \
            case 3: DLPM_HF(CO, 3); break;                                                                \
            default: DLPM_HF(CO, 4); break;                                                               \
        }                                                                                                 \
    } while (0)
    if (c.Cout == 1) DLPM_HFC(1);
    else if (c.Cout == 2) DLPM_HFC(2);
    else DLPM_HFC(3);
#undef DLPM_HFC
#undef DLPM_HF
#undef DLPM_HF1
#undef HF_TILE
#undef HF_LDS_BARRIER
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
