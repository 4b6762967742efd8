// head_fused.hip -- the UNet's head convolution AND the sampler's reverse update as ONE pass over HBM (round 4).
//
// Replaces out = conv3x3(silu(GN(h))) (dlpm/models/unet.py:433-435, Cout = image channels) followed by
// x <- (x - c_eps eps) / gamma + c_noise z (dlpm/methods/dlpm.py:272-278, GenerativeLevyProcess.py:225-239).
//
// Round 3 split the head into a 1x1 GEMM onto its 9 Cout tap channels  P = act(h) W'  (every input line fetched once) and a 9-point
// gather over P that carried the update: the pair wrote and re-read the 113 MB of P that the algorithm does not have (0.195 ms,
// 0.37 of the HBM roof on the bytes the algorithm does have: the head's input once + 12 B per state element).  Here P never
// leaves the CU.
//
// THE KERNEL'S SHAPE.  Persistent four-wave workgroups, two per CU (they drift apart: while one gathers, the other streams), each
// walking images b = blockIdx.x, + gridDim.x, ...  An image is cut into PHASES of 8 tiles (a tile = 32 pixels of one row x all
// channels; NR = 8 / (W / 32) rows), two tiles per wave:
//   tile steps:  the phase's rows of  P[n = tap Cout + co][row][x] = sum_ci act(h)[row][x][ci] W'[ci][n]  into an LDS RING of NR + 2 rows
//   barrier, gather: the 9-point sums + the reverse update for the rows whose three P rows are complete -- rows ph NR - 1 .. (ph + 1) NR - 2
//                    (the image's first and last phase take the border row as well); the ring keeps the two rows the next phase needs
//   barrier
// so no row of P is computed twice and P is 35 KB.
// NO TRANSPOSITION.  The k index of an MFMA is free as long as A and B agree, so the input is loaded in the MFMA's own operand layout:
// lane = (pixel lm, k-half kh) reads the 16 bytes of channels 8 j + 4 kh .. + 3 of its pixel (load j; a load instruction = 32 pixels x
// 32 contiguous bytes -- tools/mb/stream_pattern.hip: 5.5-5.9 TB/s, the same as whole lines), GroupNorm affine + SiLU run on the
// registers, and loads 2 m, 2 m + 1 ARE the lane's eight k slots of K-step m (16 channels).  (The earlier forms loaded line-coalesced
// and turned [pixel][channel] around through a wave-private LDS patch: 18 LDS operations and two dependent LDS round trips per 32
// channels in front of every MFMA chain.)  W' -- in that slot order -- and the GroupNorm coefficients sit in LDS (b128 reads, the same
// for every wave), which leaves the registers to the input: TWO tiles in flight per wave (32 KB; buffer s = the wave's s-th tile of
// a phase, each load reissued for the same tile of the NEXT phase as soon as its K-step has been staged).
// bf16 form (the default pipe; DLPM_HEAD_F32=1 or the fp32 GEMM policy take the fp32 MFMA): conv_split.hip's exact three-plane split --
// act(h) is cut into three bf16 planes in registers right after the SiLU (and + subtract twice, v_perm packs), W' is cut once at
// finalize, six v_mfma_f32_32x32x16_bf16 per K-step replace eight v_mfma_f32_32x32x2_f32 at a quarter of the cycles each, and the bf16
// MFMA runs BESIDE the VALU (the fp32 one shares its lanes): K-step m's MFMAs are issued next to the staging of K-step m + 1.
// All loads are unconditional (past the last image they read the L2-resident weight buffer) and the update's stores are hidden from
// the compiler (hf_store4), so every wait of the loop is an exact vmcnt count.
#include "conv.h"
#include "philox.h"

namespace dlpm {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t hf_u32x4 __attribute__((ext_vector_type(4)));

constexpr int HF_NT = 256;        // 4 waves, two workgroups per CU
constexpr int HF_NW = HF_NT / 64;
constexpr int HF_MAXCH = 4;       // Cin <= 128 (registers: two tiles of input per wave)
constexpr float HF_LOG2E = 1.44269504088896340736f;
#ifndef HF_ABLATE          // timing-only developer builds (wrong results): 1 no MFMAs, 2 no SiLU / split arithmetic, 4 no gather, 8 no reloads
#define HF_ABLATE 0
#endif

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

// the gather's item of a thread: (channel co, row y, pixel quad q) of the rows g_lo .. ; co >= Cout = none
struct HfItem { int co, y, q; };
__device__ __forceinline__ HfItem hf_item(int tid, int g_lo, int rows, int nq) {
    int tb = tid;
    asm volatile("" : "+v"(tb));     // (opaque: hoisted out of the image walk as invariants, the coordinates would live in scratch)
    const int per_co = rows * nq, co = tb / per_co, rq = tb - co * per_co, r = rq / nq;
    return HfItem{co, g_lo + r, rq - r * nq};
}

// A 16-byte global store the compiler does not see as one.  hipcc's waitcnt pass treats a store and the loads in flight as unordered
// (one counter on gfx9, no vscnt): the first wait for a LOAD after a store becomes s_waitcnt vmcnt(0) -- the head of every phase would
// sit out the write acknowledgements of the update and the landing of two tiles of read-ahead.  Hidden, the store still counts in the
// hardware's vmcnt, which only makes the loop's counted waits stricter: loads retire in order among themselves, so "at most N
// operations outstanding" still implies that a load with N younger loads behind it has landed, whatever the stores do.  (s_nop: the
// data registers of a store wider than 8 bytes may not be overwritten in the next cycle, and the hazard recognizer does not read asm.)
typedef float hf_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void hf_store4(float *ptr, float4 v) {
    const hf_f32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(ptr), "v"(d) : "memory");
}

// workgroup barrier that orders LDS only: __syncthreads() is a workgroup-scope fence over ALL address spaces, i.e. s_waitcnt vmcnt(0) --
// every wave would sit out the write acknowledgements of its update stores and the landing of its read-ahead at each barrier
#define HF_LDS_BARRIER()                                              \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
        __builtin_amdgcn_s_barrier();                                 \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
    } while (0)

struct HeadFusedArgs {
    const float *h;               // [B][H][W][C] NHWC
    const float *coefA, *coefB;   // [B][C] GroupNorm affine (SiLU follows)
    const float *wf;              // W' in slot order: fp32 [C/8][64][4], behind it the bf16 planes [C/16][3][64][8]
    const float *bias;            // [Cout]
    float *out;                   // eps (plain forward): NCHW [B][Cout][H][W] or NHWC
    int out_nchw;
    int B, H, W, C;
    HeadUpdate u;
#ifdef DLPM_PHASE_TIMING
    unsigned long long *phase;    // developer builds: 1 tile steps, 2 barrier + gather/update, 3 phases, 12/13 clock, 16+w barrier wait
#endif
};

// NCH = C / 32, a compile-time constant: everything below is straight-line code per phase
template <int COUT, int NCH, bool BF>
__global__ void __launch_bounds__(HF_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) k_head_fused(HeadFusedArgs p) {
    constexpr int NV = 9 * COUT;                       // live tap channels (27)
    constexpr int C = 32 * NCH, NK = C / 16, NL = C / 8;   // K-steps and loads per tile
    constexpr int WFL = BF ? C * 48 : C * 32;          // floats of W' this form reads
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.W, H = p.H, G = gridDim.x;
    const int tpr = W >> 5, NR = 8 / tpr, RR = NR + 2, NPH = H / NR;       // tiles per row, rows per phase, ring rows, phases per image
    // P as NV planes [ring row][W] of pitch PL = RR W + 4 floats: a lane's four consecutive pixels are one 16-byte LDS access on both
    // sides (MFMA registers 4 j .. 4 j + 3 in; a gather item's pixel quad out), the + 4 spreads the 27 lanes of a store over the banks
    const int PL = RR * W + 4;
    float *P = sm;
    const float *wl = sm + NV * PL;                    // W' (this form's slot order)
    float *cf = sm + NV * PL + WFL;                    // [2 slots][2][C] GroupNorm affine: slot = parity of the image's position in this workgroup's walk
    const int64_t HW = (int64_t)H * W;
    const int lm = lane & 31, kh = lane >> 5;          // pixel lm of the tile, k-half kh

    {   // W' -> LDS, once
        const float4 *src = reinterpret_cast<const float4 *>(p.wf + (BF ? (int64_t)C * 32 : 0));
        float4 *dst = reinterpret_cast<float4 *>(sm + NV * PL);
        for (int i = tid; i < WFL / 4; i += HF_NT) dst[i] = src[i];
    }
    // the coefficients of an image: threads 0 .. C / 2 - 1 fetch one float4 each (A then B) ...
    auto fetch_coefs = [&](int b) __attribute__((always_inline)) {
        const float *src = tid < C / 4 ? p.coefA + (int64_t)b * C + 4 * tid : p.coefB + (int64_t)b * C + 4 * (tid - C / 4);
        return tid < C / 2 ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // ... and drop it into a slot nobody reads any more
    auto put_coefs = [&](float4 v, int slot) __attribute__((always_inline)) {
        if (tid < C / 2) *reinterpret_cast<float4 *>(cf + slot * 2 * C + 4 * tid) = v;
    };
    // tile s (0 / 1) of phase ph of image b for this wave: this lane's pixel, + its k-half; past the last image a harmless L2-resident
    // address (the loads stay unconditional)
    auto tile_src = [&](int b, int ph, int s) __attribute__((always_inline)) {
        const int id = wave + HF_NW * s, y = ph * NR + id / tpr, x0 = (id - (id / tpr) * tpr) * 32;
        return b < p.B ? p.h + (((int64_t)b * H + y) * W + x0 + lm) * C + 4 * kh : p.wf + 4 * lane;
    };
    float4 xb[2][NL];
    auto load2 = [&](const float *src, int s, int m) __attribute__((always_inline)) {       // the two loads of K-step m
        if (!(HF_ABLATE & 8)) {
            xb[s][2 * m] = *reinterpret_cast<const float4 *>(src + 16 * m);
            xb[s][2 * m + 1] = *reinterpret_cast<const float4 *>(src + 16 * m + 8);
        }
    };
    // stage(s, m, slot): GroupNorm affine + SiLU of the lane's eight values of K-step m -> its A operand: three bf16 planes, or the
    // eight fp32 values
    struct AFrag { hf_u32x4 pl[3]; };
    auto stage = [&](int s, int m, const float *cfs) __attribute__((always_inline)) {
        AFrag a;
        uint32_t q[3][4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float4 xr = xb[s][2 * m + h];
            const float4 cA = *reinterpret_cast<const float4 *>(cfs + 16 * m + 8 * h + 4 * kh);
            const float4 cB = *reinterpret_cast<const float4 *>(cfs + C + 16 * m + 8 * h + 4 * kh);
            f32x2 v01, v23;
            if (HF_ABLATE & 2) { v01 = f32x2{xr.x, xr.y} + f32x2{cA.x, cA.y}; v23 = f32x2{xr.z, xr.w} + f32x2{cB.z, cB.w}; }
            else {
                v01 = hf_act2(f32x2{xr.x, xr.y}, f32x2{cA.x, cA.y}, f32x2{cB.x, cB.y});
                v23 = hf_act2(f32x2{xr.z, xr.w}, f32x2{cA.z, cA.w}, f32x2{cB.z, cB.w});
            }
            if constexpr (BF) {
                if (HF_ABLATE & 2) {
                    q[0][2 * h] = q[1][2 * h] = q[2][2 * h] = __float_as_uint(v01.x) ^ __float_as_uint(v01.y);
                    q[0][2 * h + 1] = q[1][2 * h + 1] = q[2][2 * h + 1] = __float_as_uint(v23.x) ^ __float_as_uint(v23.y);
                } else {
                    hf_split2(v01.x, v01.y, q[0][2 * h], q[1][2 * h], q[2][2 * h]);
                    hf_split2(v23.x, v23.y, q[0][2 * h + 1], q[1][2 * h + 1], q[2][2 * h + 1]);
                }
            } else {   // fp32 form: planes 0 / 1 carry the eight values themselves
                q[h][0] = __float_as_uint(v01.x); q[h][1] = __float_as_uint(v01.y);
                q[h][2] = __float_as_uint(v23.x); q[h][3] = __float_as_uint(v23.y);
                q[2][2 * h] = q[2][2 * h + 1] = 0;
            }
        }
#pragma unroll
        for (int pl = 0; pl < 3; pl++) a.pl[pl] = hf_u32x4{q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
        return a;
    };
    // W' fragments of K-step m out of LDS (bf16 form: read one step AHEAD, so that a step's MFMAs have their operands when it begins
    // and can be issued between the staging's VALU instructions)
    struct BFrag { bf16x8 pl[3]; };
    auto load_b = [&](int m) __attribute__((always_inline)) {
        BFrag r;
        if constexpr (BF) {
#pragma unroll
            for (int pl = 0; pl < 3; pl++) r.pl[pl] = reinterpret_cast<const bf16x8 *>(wl)[(m * 3 + pl) * 64 + lane];
        }
        return r;
    };
    // K-step m's MFMAs
    auto mfmas = [&](floatx16 &acc, const AFrag &a, const BFrag &bfr, int m) __attribute__((always_inline)) {
        if (HF_ABLATE & 1) { asm volatile("" :: "v"(a.pl[0]), "v"(a.pl[1]), "v"(a.pl[2])); return; }
        if constexpr (BF) {
            const bf16x8 A0 = __builtin_bit_cast(bf16x8, a.pl[0]), A1 = __builtin_bit_cast(bf16x8, a.pl[1]), A2 = __builtin_bit_cast(bf16x8, a.pl[2]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, bfr.pl[0], acc, 0, 0, 0);      // small terms first (conv_split.hip's order)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, bfr.pl[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, bfr.pl[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, bfr.pl[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, bfr.pl[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, bfr.pl[0], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float4 bq = reinterpret_cast<const float4 *>(wl)[(2 * m + h) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.pl[h][0]), bq.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.pl[h][1]), bq.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.pl[h][2]), bq.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.pl[h][3]), bq.w, acc, 0, 0, 0);
            }
        }
    };

#if defined(DLPM_PHASE_TIMING) && defined(HF_CLOCK_ONLY)     // the clock the chip holds in this kernel, at the cost of two atomics per workgroup
    const long long _k0 = clock64(), _w0 = wall_clock64();
#endif
    const HeadUpdate &u = p.u;
    const int nq = W >> 2;
    const int64_t D = (int64_t)COUT * HW;
    // the update's launch constants (t -> g is a DEPENDENT pair of loads: once, here)
    int tt = 0;
    float g = 1.f, rg = 1.f;
    uint64_t seed = 0;
    int64_t soff = 0;
    float *hist = nullptr;
    if (u.x) {
        tt = *u.t;
        g = u.g[tt];
        rg = 1.0f / g;
        seed = u.key ? u.key[0] : u.seed;
        soff = u.key ? (int64_t)u.key[1] : u.sample_offset;
        hist = u.hist_pp ? *u.hist_pp : nullptr;
    }
    // ---- the walk: position (image b, phase ph); n = the image's index in this workgroup's sequence (its coefficient slot is n & 1)
    int b = blockIdx.x, ph = 0, n = 0;
    put_coefs(fetch_coefs(b), 0);
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const float *src = tile_src(b, 0, s);
#pragma unroll
        for (int m = 0; m < NK; m++) {
            load2(src, s, m);
            __builtin_amdgcn_sched_barrier(0);         // (issue order = consumption order: the loop's vmcnt waits count on it)
        }
    }
    HF_LDS_BARRIER();
    AFrag acur = stage(0, 0, cf);
    BFrag bcur = load_b(0);
    load2(tile_src(b, 1, 0), 0, 0);                    // (NPH >= 2)
    __builtin_amdgcn_sched_barrier(0);
    for (;;) {
#if defined(DLPM_PHASE_TIMING) && !defined(HF_CLOCK_ONLY)
        const long long _c0 = clock64(), _r0 = wall_clock64();
#endif
        // the next two positions of the walk
        const bool lastph = ph + 1 == NPH;
        const int b1 = lastph ? b + G : b, ph1 = lastph ? 0 : ph + 1, n1 = lastph ? n + 1 : n;
        const bool lastph1 = ph1 + 1 == NPH;
        const int b2 = lastph1 ? b1 + G : b1, ph2 = lastph1 ? 0 : ph1 + 1;
        const float *cf0 = cf + (n & 1) * 2 * C, *cf1 = cf + (n1 & 1) * 2 * C;
        const float *nx0 = tile_src(b1, ph1, 0), *nx1 = tile_src(b1, ph1, 1), *nxx = tile_src(b2, ph2, 0);
        // the gather covers the rows whose P rows will be complete, g_lo .. g_hi - 1; an item = (channel, row, 4 pixels), one per thread
        const int g_lo = ph == 0 ? 0 : ph * NR - 1, g_hi = lastph ? H : (ph + 1) * NR - 1;
        // the next image's coefficients travel during its predecessor's first phase (its first K-step is staged in the LAST step of the
        // predecessor's last phase; NPH >= 2)
        const bool carry = ph == 0 && b + G < p.B;
        float4 xq = make_float4(0.f, 0.f, 0.f, 0.f), ncf = make_float4(0.f, 0.f, 0.f, 0.f);
        float bv = 0.f, ce = 0.f, cn = 0.f;

        // ---- tile steps.  Step (s, m): K-step m's MFMAs next to the staging of the NEXT K-step of the wave's sequence -- (s, m + 1), or
        // (1, 0), or K-step 0 of the next position's first tile -- whose two loads are then reissued for the same tile one position on.
        // The sched_barrier: nothing of a later step moves up (hipcc otherwise hoists the affine FMAs of ALL later K-steps and waits for
        // their loads here).
#pragma unroll
        for (int s = 0; s < 2; s++) {
            floatx16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
            for (int m = 0; m < NK; m++) {
                AFrag anext;
                const BFrag bnext = load_b(m + 1 < NK ? m + 1 : 0);
                if (m + 1 < NK) { anext = stage(s, m + 1, cf0); load2(s == 0 ? nx0 : nx1, s, m + 1); }
                else if (s == 0) { anext = stage(1, 0, cf0); load2(nx1, 1, 0); }
                else { anext = stage(0, 0, cf1); load2(nxx, 0, 0); }
                mfmas(acc, acur, bcur, m);
                acur = anext;
                bcur = bnext;
                if constexpr (BF) {
                    // the LDS reads first, then one MFMA and a share of the staging's VALU work, six times: left alone, hipcc issues the
                    // six MFMAs in a row BEHIND the staging (SQ_VALU_MFMA_COEXEC_CYCLES 7 % of the MFMA cycles)
                    __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);          // DS reads: W' of the next step, the coefficients
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);     // VALU
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // D layout of the 32x32 MFMA: register i holds row 8 (i / 4) + 4 kh + (i % 4) (pixel), column lm (tap channel)
            if (lm < NV) {
                const int id = wave + HF_NW * s, y = ph * NR + id / tpr, x0 = (id - (id / tpr) * tpr) * 32;
                float *dst = P + lm * PL + (y % RR) * W + x0 + 4 * kh;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<float4 *>(dst + 8 * j) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
            }
            if (s == 0) {
                // What the gather reads from global memory is requested HALFWAY through the phase: vmcnt retires in order, so a wait
                // for anything requested after the last tile's reloads would wait for all of them (an HBM latency per phase), whereas
                // behind these sit only the second tile's 16.  Unconditional (dummy addresses), so the loop's waits stay exact counts.
                const HfItem it = hf_item(tid, g_lo, g_hi - g_lo, nq);
                const bool item = it.co < COUT;
                const float *px = (u.x && item) ? u.x + (int64_t)b * D + (int64_t)it.co * HW + (int64_t)it.y * W + 4 * it.q : p.wf + 4 * lane;
                xq = *reinterpret_cast<const float4 *>(px);
                bv = *((p.bias && item) ? p.bias + it.co : p.wf);   // (no bias: the dummy load keeps the wait counts; its value is dropped below)
                ce = *(u.x ? u.c_eps + (int64_t)tt * u.B + b : p.wf);
                cn = *(u.x ? u.c_noise + (int64_t)tt * u.B + b : p.wf);
                if (carry) ncf = fetch_coefs(b + G);
            }
        }

        // ---- gather + update.  k_head_gather's arithmetic, tap by tap in its order; taps outside the picture add a literal zero where
        // k_head_gather adds the zero its padded P holds.
#if defined(DLPM_PHASE_TIMING) && !defined(HF_CLOCK_ONLY)
        const long long _c2 = clock64();
#endif
        HF_LDS_BARRIER();
#if defined(DLPM_PHASE_TIMING) && !defined(HF_CLOCK_ONLY)
        const long long _c3 = clock64();
#endif
        asm volatile("" :: "v"(xq.x), "v"(xq.y), "v"(xq.z), "v"(xq.w), "v"(bv), "v"(ce), "v"(cn));   // (every path consumes what it requested)
        if (carry) put_coefs(ncf, (n + 1) & 1);
        const uint64_t gidx = (uint64_t)(soff + b);
        float *hr = hist ? hist + ((int64_t)(u.T - tt) * u.B + b) * D : nullptr;
        const HfItem it = hf_item(tid, g_lo, g_hi - g_lo, nq);
        const int co = it.co, y = it.y, q = it.q;
        const bool item = co < COUT && !(HF_ABLATE & 4);
        const int64_t e0 = (int64_t)co * HW + (int64_t)y * W + 4 * q;
        if (item) {
            if (!p.bias) bv = 0.f;   // a launch without bias (valid for dlpm_conv2d_f32): the dummy load above read W'[0]
            float acc[4] = {bv, bv, bv, bv};
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                const int yy = y + ky - 1;
                const bool inb = yy >= 0 && yy < H;
                const int yc = min(max(yy, 0), H - 1);
                float4 Q[3];
#pragma unroll
                for (int kx = 0; kx < 3; kx++) Q[kx] = *reinterpret_cast<const float4 *>(P + ((ky * 3 + kx) * COUT + co) * PL + (yc % RR) * W + 4 * q);
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
            if (u.x) {
                float4 z;
                if (u.z) z = *reinterpret_cast<const float4 *>(u.z + (int64_t)b * D + e0);
                else z = (cn != 0.0f) ? philox_normal4(seed, gidx, (uint32_t)(e0 >> 2), kPurposeStepZ, (uint32_t)tt) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 o;
                o.x = fmaf(cn, z.x, div_by(xq.x - ce * acc[0], g, rg));
                o.y = fmaf(cn, z.y, div_by(xq.y - ce * acc[1], g, rg));
                o.z = fmaf(cn, z.z, div_by(xq.z - ce * acc[2], g, rg));
                o.w = fmaf(cn, z.w, div_by(xq.w - ce * acc[3], g, rg));
                hf_store4(u.x + (int64_t)b * D + e0, o);
                if (hr) hf_store4(hr + e0, o);
                if (u.eps_out) hf_store4(u.eps_out + (int64_t)b * D + e0, make_float4(acc[0], acc[1], acc[2], acc[3]));
            } else if (p.out_nchw) {
                hf_store4(p.out + (int64_t)b * D + e0, make_float4(acc[0], acc[1], acc[2], acc[3]));
            } else {
                const int64_t pix = (int64_t)y * W + 4 * q;
#pragma unroll
                for (int px = 0; px < 4; px++) p.out[((int64_t)b * HW + pix + px) * COUT + co] = acc[px];
            }
        }
#if defined(DLPM_PHASE_TIMING) && !defined(HF_CLOCK_ONLY)
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
        if (b1 >= p.B) break;
        HF_LDS_BARRIER();                              // the gather has read the ring: the next phase's tiles may land
        b = b1; ph = ph1; n = n1;
    }
#if defined(DLPM_PHASE_TIMING) && defined(HF_CLOCK_ONLY)
    if (p.phase && tid == 0) {
        atomicAdd(p.phase + 3, 1ull);
        atomicAdd(p.phase + 12, (unsigned long long)(clock64() - _k0));
        atomicAdd(p.phase + 13, (unsigned long long)(wall_clock64() - _w0));
    }
#endif
}

// OIHW (3x3, Cout <= 3) -> W' in the kernel's slot order, W'[ci][n = tap Cout + co] = w[co][ci][tap], columns n >= 9 Cout zero.
// fp32 form [C/8 j][lane][4]: lane (n = lane % 32, g = lane / 32) holds W'[ci = 8 j + 4 g + e][n], e < 4 -- what the lane (pixel, k-half g)
// of the activations reads with its load j
__global__ void k_relayout_weight_head_fused(const float *oihw, float *dst, int Cout, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * 32) return;
    const int e = i & 3, lane = (i >> 2) & 63, j = i >> 8;
    const int n = lane & 31, g = lane >> 5, ci = 8 * j + 4 * g + e;
    const int tap = n / Cout, co = n - tap * Cout;
    dst[i] = n < 9 * Cout ? oihw[((int64_t)co * Cin + ci) * 9 + tap] : 0.f;
}

// ... -> the bf16 planes [C/16 m][3 planes][lane][8]: lane (n, g) holds the k slots of K-step m, slot e < 4: ci = 16 m + 4 g + e,
// slot e >= 4: ci = 16 m + 8 + 4 g + (e - 4)  (loads 2 m and 2 m + 1 of the activations)
__global__ void k_relayout_weight_head_fused_bf(const float *oihw, uint4 *dst, int Cout, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // one thread per (m, lane)
    if (i >= (Cin / 16) * 64) return;
    const int lane = i & 63, m = i >> 6;
    const int n = lane & 31, g = lane >> 5;
    const int tap = n / Cout, co = n - tap * Cout;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int ci = 16 * m + (e < 4 ? 4 * g + e : 8 + 4 * g + (e - 4));
        w[e] = n < 9 * Cout ? oihw[((int64_t)co * Cin + ci) * 9 + tap] : 0.f;
    }
    uint32_t P[3][4];
#pragma unroll
    for (int e = 0; e < 4; e++) hf_split2(w[2 * e], w[2 * e + 1], P[0][e], P[1][e], P[2][e]);
#pragma unroll
    for (int pl = 0; pl < 3; pl++) dst[(m * 3 + pl) * 64 + lane] = make_uint4(P[pl][0], P[pl][1], P[pl][2], P[pl][3]);
}

bool head_fused_bf(const ConvLaunch &c) {     // which matrix pipe: bf16 x 3 unless the fp32 GEMM policy (or DLPM_HEAD_F32=1) asks otherwise
    static int f32 = -1;
    if (f32 < 0) { const char *e = getenv("DLPM_HEAD_F32"); f32 = (e && e[0] == '1') ? 1 : 0; }
    return !f32 && c.gemm != DLPM_GEMM_F32;
}

// LDS: the ring of P rows + W' + two coefficient slots
size_t head_fused_lds_floats(const ConvLaunch &c) {
    const int nr = 8 / (c.Wout >> 5);
    return (size_t)9 * c.Cout * ((nr + 2) * c.Wout + 4) + (size_t)c.C0 * (head_fused_bf(c) ? 48 : 32) + 4 * c.C0;
}

}  // namespace

bool head_fused_ok(const ConvLaunch &c) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("DLPM_NO_HEAD_FUSED"); off = (e && e[0] == '1') ? 1 : 0; }
    if (off || !c.w_hfused || c.ks != 3 || c.stride != 1 || c.ups || c.in_nchw || c.C1 != 0 || c.res0 || !c.coefA || !c.act_silu) return false;
    if (c.Cout < 1 || c.Cout > 3 || c.C0 % 32 != 0 || c.C0 > 32 * HF_MAXCH || c.Hin != c.Hout || c.Win != c.Wout) return false;
    if (c.Wout != 32 && c.Wout != 64) return false;
    const int nr = 8 / (c.Wout >> 5);                                                  // rows per phase (8 tiles)
    return c.Hout % nr == 0 && c.Hout / nr >= 2 && head_fused_lds_floats(c) * sizeof(float) <= 80 * 1024;   // two workgroups per CU
}

int64_t head_fused_weight_floats(int Cin) { return (int64_t)Cin * 32 + (int64_t)Cin * 48; }   // fp32 fragments + three bf16 planes

int relayout_weight_head_fused(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st) {
    k_relayout_weight_head_fused<<<(unsigned)ceil_div(Cin * 32, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin);
    DLPM_LAUNCH_CHECK();
    k_relayout_weight_head_fused_bf<<<(unsigned)ceil_div((Cin / 16) * 64, 128), 128, 0, st>>>(
        oihw_dev, reinterpret_cast<uint4 *>(dst_dev + (int64_t)Cin * 32), Cout, Cin);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int launch_conv_head_fused(const ConvLaunch &c, const HeadUpdate *hu, hipStream_t st) {
    HeadFusedArgs a{};
    a.h = c.src0; a.coefA = c.coefA; a.coefB = c.coefB; a.wf = c.w_hfused; a.bias = c.bias; a.out = c.out; a.out_nchw = c.out_nchw;
    a.B = c.B; a.H = c.Hout; a.W = c.Wout; a.C = c.C0;
    if (hu) a.u = *hu;
#ifdef DLPM_PHASE_TIMING
    a.phase = phase_buffer();
#endif
    const int64_t M = (int64_t)c.B * c.Hout * c.Wout;
    // algorithmic bytes: the head's input once + the state read and written (or eps written)
    const double bytes = 4.0 * ((double)M * c.C0 + (double)M * c.Cout * (hu ? 2 + (hu->z ? 1 : 0) + (hu->eps_out ? 1 : 0) : 1));
    ProfScope ps(hu ? "head_fused+update" : "head_fused", 2.0 * M * c.Cout * 9.0 * c.C0, bytes, st);
    const size_t lds = head_fused_lds_floats(c) * sizeof(float);
    const bool bf = head_fused_bf(c);
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        DLPM_HIP(hipGetDevice(&dev));
        DLPM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    static int wgs = 0;                                                  // (DLPM_HEAD_WGS=1: one workgroup per CU -- A/B runs)
    if (!wgs) { const char *e = getenv("DLPM_HEAD_WGS"); wgs = (e && e[0] == '1') ? 1 : 2; }
    const unsigned grid = (unsigned)(c.B < wgs * ncu ? c.B : wgs * ncu);   // persistent: two workgroups per CU walk the images
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
#undef HF_LDS_BARRIER
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
