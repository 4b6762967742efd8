// conv_wino4.hip -- 3x3 / stride-1 convolution as Winograd F(4x4,3x3) on the gfx950 fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Replaces the same F.conv2d calls as conv_wino.hip (dlpm/models/unet.py:64,96,143,157,168 via nn.py:25-35) for the
// layers whose output is a multiple of 4x4 pixels and 128 channels.  F(4x4,3x3) spends 36 multiplies per 16 outputs
// and channel pair (2.25 per output) where F(2x2,3x3) spends 4 and the direct form 9:
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        d: 6x6 input tile, g: 3x3 filter, Y: 4x4 output tile
//
// with Toom-Cook transform matrices for the interpolation points {0, +-a, +-b, inf}, (a, b) = (11/16, 3/2) (F4_PA, F4_PB below;
// Lavin & Gray's matrices are a = 1, b = 2): rows of B^T above transform(), rows of A^T in the epilogue, G in
// k_relayout_weight_wino4.
// The price is numerical: the products are accumulated over the input channels IN the Winograd domain, where the partial sums
// are larger than the result they cancel to, and the point set fixes by how much.  tools/err_wino4_points.py (round 4,
// profiles/r04/err_wino4_points.txt) emulates the kernel's arithmetic for the whole symmetric family -- every member costs
// the same VALU instructions in the loop, only the constants change -- and finds a flat optimum around a in [0.62, 0.71],
// b in [1.4, 1.6]: 2.2x lower rms and 3.5-4x lower max error than (1, 2) (8.3e-7 / 6e-6 against 1.8e-6 / 2.1-3.0e-5 of the fp64
// convolution at 128-256 channels; fp32 direct convolution: 5e-7).  All constants (a, b, their squares, cubes, a^2 + b^2,
// a^2 b^2) are exact in fp32.  F4_POINTS_LAVIN (DLPM_BUILD_DEFS) rebuilds the round-1..3 point set for A/B error measurements.
//
// Work split (one workgroup per CU, 8 waves, 16 tiles x 128 output channels):
//   * the 36 transform positions are 36 independent GEMMs  M_pos[tile][cout] = sum_cin V_pos[tile][cin] U_pos[cin][cout];
//     wave w owns output channels 16w..16w+15 for ALL positions: 36 accumulators of the 16x16x4 MFMA (144 registers),
//     so the output transform never leaves the wave's registers -- no cross-wave exchange as in k_conv3x3_wino_q;
//   * U (filters in the Winograd domain, fragment order) streams L2 -> registers through a ring, exactly once per wave;
//   * V = B^T d B is built in LDS per 8-channel phase by waves 0..2 (one row pair of B^T each, lanes = channel pair x tile)
//     from a raw halo patch that all waves stage with the fused GroupNorm affine + SiLU.  V is kept in A-FRAGMENT order,
//     V[position pair][k = channel pair][tile][2 positions x 2 channels]: one conflict-free ds_read_b128 per lane feeds the
//     four MFMAs of a position pair (the tile-major layout it replaces put 16 lanes 32 bytes apart: 4-way bank conflicts on
//     every A read, the LDS array busy 57 % of the kernel with two thirds of that conflict cycles -- SQ_LDS_BANK_CONFLICT);
//   * phases are software-pipelined like k_conv3x3_wino_q: S(c+2) raw stores, X(c+1) transform, G(c+3) global loads
//     ride between the MFMAs of phase c, one barrier per phase;
//   * epilogue: A^T M A per lane (36 -> 16 values), then bias / residual / fused GroupNorm statistics / stores straight
//     from the registers (a lane owns 64 outputs of one channel): no LDS, no barrier.
#include <cstdlib>
#include <type_traits>

#include "conv.h"
#include "gn_stats.h"

namespace dlpm {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int F4_TILES = 16;      // 4x4-output tiles per workgroup (the M of the 16x16x4 MFMA)
constexpr int F4_NQ = 128;        // output channels per workgroup of the main shape (8 waves x 16)
constexpr int F4_KC = 8;          // input channels per phase (2 MFMA k-steps)
constexpr int F4_RAWPIX = 576;    // halo pixels per phase: one 18x18 patch, 4 x 10x10 or 16 x 6x6 (whole small images)
constexpr int F4_PRLD = F4_KC + 4;
constexpr int F4_NT = 512;
constexpr int F4_QNIT = (F4_RAWPIX * 2 + F4_NT - 1) / F4_NT;   // staging items (pixel, channel quad) per thread for a full 576-pixel patch: 3
                                                               // (only 16 one-tile images fill it; every other block shape is <= 512 pixels: QN = 2)
constexpr int F4_RAWBUF = F4_RAWPIX * F4_PRLD + 16;   // floats per raw buffer, row skew included
// Round 5 -- narrow layers (Cout a multiple of 32 but not of 128: the 32- and 64-channel levels of the MNIST-sized nets).  The work
// split is the wave's: 16 output channels x 36 positions x 16 tiles, so a 16-tile block of a 64- / 32-channel n-tile has MFMA work for
// NW = 4 / 2 waves only -- and the SAME staging and input transform as a 128-channel one.  Letting those few waves carry the side
// work between their MFMAs (the 8-wave kernel's scheme, where it is 1/8 of a wave's instructions) made a phase 6-15 k cycles for
// 2.3 k cycles of MFMAs: one wave per SIMD, nothing hides a wait.  So the input transform gets waves of its own: NW MFMA waves (weight
// ring, A fragments, MFMAs, their share of the staging between the MFMAs as in the 8-wave kernel, epilogue) + HELPER waves that stage
// their share and run the transform -- 3 (one row pair of B^T each) beside 4 MFMA waves, 2 (row pairs 0 + 1 | 2) beside 2, so that two
// of the 32-channel workgroups fit a CU (4 waves x 255 registers).  (Helpers that also did ALL the staging were the long pole: 6.2 k
// cycles per phase against 2.3 k of MFMAs, profiles/r05/.)  One barrier per phase as before; same V / raw / weight-fragment layouts, same
// arithmetic per output as the 8-wave shape; raw buffer for halo patches of <= 400 pixels (an 18x18 block patch, or four 10x10 images).
constexpr int F4_RAWPIX_N = 400;
constexpr int F4_RAWBUF_N = F4_RAWPIX_N * F4_PRLD + 16;
constexpr int f4_nq_of(int cout) { return cout % 128 == 0 ? 128 : cout % 64 == 0 ? 64 : 32; }
constexpr int F4_CFS = 16 * 2 * F4_KC;   // floats per GroupNorm-coefficient slot: [16 images][A | B][8]
constexpr int F4_VBUF = 36 * F4_TILES * F4_KC;   // floats per V buffer: [18 position pairs][4 channel pairs][16 tiles][4]
#ifndef F4_RING_DEPTH
#define F4_RING_DEPTH 6
#endif
constexpr int F4_RING = F4_RING_DEPTH;   // weight fragments (float4 = 2 positions x 2 k-steps) in flight; divides the 18 of a phase
constexpr int F4_PAD = 8;         // float4 fragments of zero padding behind the weights (ring read-ahead of the last phase)

#ifndef F4_AAHEAD
#define F4_AAHEAD 1   // A-fragment read-ahead (position pairs); 2 spills three registers inside the phase loop
#endif
#ifndef F4_PRIO
#define F4_PRIO 1     // s_setprio around the MFMA groups (0: off)
#endif
#ifndef F4_S0
#define F4_S0 6    // position pair behind which the staging stores start (round 4, QN = 2: 6 is 0.8-1.2 % faster than 4, 5, 7..10 -- four alternating
                   // repetitions, profiles/r04/conv_layers_sidework_placement_qn2.txt; same bits)
#endif
#ifndef F4_EPI_T
#define F4_EPI_T 0  // 1: accumulators transposed (lane = tile, 4 consecutive channels): 16-byte epilogue accesses, 32 instead of 128 memory
                    // instructions per lane -- bit-identical outputs, measured 0.3-0.7 % SLOWER over the eight layer shapes
                    // (profiles/r03/conv_layers_transposed_epilogue.txt: the same 512 cache-line accesses per wave either way);
                    // 0 (the build): lane = channel, 4 tiles
#endif
#ifndef F4N_ABL
#define F4N_ABL 0   // developer builds (DLPM_BUILD_DEFS), narrow shapes' epilogue, timing only (results are wrong): 1 no stores, 2 no residual loads, 4 no statistics
#endif
#ifndef F4_RES_AHEAD
#define F4_RES_AHEAD 0   // 1: the 8-wave shape, too, requests all four tiles' residual values before the first output transform (the narrow
                         // shapes always do): 64 more live registers in the epilogue, where the weight ring / A fragments / staging items are dead
#endif
#ifndef F4_SKIP_TAIL
#define F4_SKIP_TAIL 0   // 1: skip the look-ahead staging / transform of the last phases (they feed chunks that do not exist): 2 of 16 phases'
                         // side work on the K = 128 layers -- measured 1.3-2.5 % SLOWER (the two uniform branches cost the phase its schedule:
                         // 256 VGPRs; profiles/r03/conv_layers_skip_tail_phases.txt, +0.45 ms on the whole step in alternating runs)
#endif
#ifndef F4_X
#define F4_X 13    // position pair behind which waves 0..2 run the input transform (placement sweep, DLPM_BUILD_DEFS="F4_S0=..
                   // F4_X=..": X = 13 is 3-8 % faster than 3, 9, 11, 12, 14..17 for every S0; S0 = 3, 5, 7 are within 0.5 %)
#endif

// Interpolation points {0, +-a, +-b, inf}.  Row j of B^T holds the coefficients of prod_{l != j} (x - p_l) (the inf row: of the
// product over all finite points), A^T[i][j] = p_j^i, G[j][k] = p_j^k / prod_{l != j} (p_j - p_l):
// one line of B^T d (or of T B): the six transform rows from six samples x0..x5
//   r0 = a^2 b^2 x0 - (a^2 + b^2) x2 + x4              r5 = a^2 b^2 x1 - (a^2 + b^2) x3 + x5
//   r1 = (x4 - b^2 x2) + a (x3 - b^2 x1)               r2 = (x4 - b^2 x2) - a (x3 - b^2 x1)
//   r3 = (x4 - a^2 x2) + b (x3 - a^2 x1)               r4 = (x4 - a^2 x2) - b (x3 - a^2 x1)
// and of A^T m:  y0 = m0 + (m1 + m2) + (m3 + m4)         y1 = a (m1 - m2) + b (m3 - m4)
//                y2 = a^2 (m1 + m2) + b^2 (m3 + m4)      y3 = a^3 (m1 - m2) + b^3 (m3 - m4) + m5
#ifdef F4_POINTS_LAVIN
#define F4_PA 1.0
#define F4_PB 2.0
#else
#define F4_PA 0.6875
#define F4_PB 1.5
#endif
constexpr float PA = (float)F4_PA, PB = (float)F4_PB, PA2 = (float)(F4_PA * F4_PA), PB2 = (float)(F4_PB * F4_PB);
constexpr float PA3 = (float)(F4_PA * F4_PA * F4_PA), PB3 = (float)(F4_PB * F4_PB * F4_PB);
constexpr float PS2 = (float)(F4_PA * F4_PA + F4_PB * F4_PB), PP2 = (float)(F4_PA * F4_PA * F4_PB * F4_PB);
__device__ __forceinline__ float2 f2fma(float a, float2 x, float2 y) { return make_float2(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y)); }
__device__ __forceinline__ float2 f2add(float2 x, float2 y) { return make_float2(x.x + y.x, x.y + y.y); }
__device__ __forceinline__ float2 f2sub(float2 x, float2 y) { return make_float2(x.x - y.x, x.y - y.y); }

// ABL (DLPM_WINO_ABLATIONS builds, DLPM_WABL): timing-only ablations, results are wrong: 1 no staging stores, 2 no transform,
// 4 no raw loads, 8 no barrier, 16 no weight loads, 32 no MFMA, 64 no A-fragment reads
// QN: staging items per thread (2 for halo patches of <= 512 pixels -- every block shape but 16 whole 4x4 images; round 4: the third
// item's geometry registers and its dead branch per phase pushed the main instantiation into a scratch reload inside the loop)
// SPEC (round 4, VERDICT r03 next #4b): 1 / 2 = the 32x32 128 -> 128 ResBlock convolution without / with a residual -- 7 launches and
// 7.2 of the 30.7 ms this kernel takes per CIFAR step, its worst shape (16 phases: prologue + epilogue are 19 % of a workgroup's life).
// Channel counts (row pitches, the phase count, the single input / residual source) and the presence of the fused GroupNorm +
// SiLU / bias / residual become compile-time constants, so the generic address arithmetic of the staging and the epilogue folds
// away.  (The image and block geometry stay run-time values: as constants they let hipcc hoist per-lane addresses out of the
// phase loop -- 8 to 12 spilled registers, with reloads inside the loop.)
// VS = 1 (round 4; measured neutral, off -- see wino4_vsplit): waves = 2 POSITION HALVES (transform rows 0-2 / 3-5) x 4 channel quarters of 32 instead of 8
// channel eighths for all 36 positions: a wave still owns 36 accumulator tiles (18 positions x 2 channel tiles) and streams the same
// weight bytes, but reads only ITS half of V -- every V fragment is read by 4 waves instead of 8 (the held-clock ablations put 12 % of
// the clock on that LDS traffic).  Y = A^T M A splits by rows of A^T: each wave of a pair turns its 18 positions into 16 partial outputs
// per (tile, channel), hands the partials of the channel tile its PARTNER finalises over through LDS (one barrier pair, 128 KB: the
// loop's buffers are dead by then) and runs the unchanged register epilogue on the other.
// PERS = 1 (round 6, the 8-wave shape): PERSISTENT workgroups -- the grid is one workgroup per CU and each walks the tiles wgid, wgid + gridDim.x,
// ... of the n-tile-major order (all workgroups in flight still stream the same slab of weights); the tile body is unchanged.  What it
// saves is the turnover between the 16 workgroups a CU runs per launch of the CIFAR net (teardown, dispatch, LDS allocation, kernel-argument
// loads: ~4 us per round by the launch time against the sum of the workgroups' lives, profiles/r06/persistent_wino4/).
template <bool UPS, int ABL = 0, int QN = F4_QNIT, int SPEC = 0, int VS = 0, int NW = 8, int PERS = 0>
__global__ void __launch_bounds__((NW + (NW == 8 ? 0 : NW == 4 ? 3 : 2)) * 64, NW == 2 ? 2 : 1) k_conv3x3_wino4(ConvLaunch p_in, int bh_in, int bw_in, int nimg_in) {
    static_assert(PERS == 0 || NW == 8, "persistent workgroups: the 8-wave shape");
    constexpr bool HELP = NW != 8;                                        // MFMA waves 0 .. NW-1 + helper waves NW .. NW+NH-1
    constexpr int NH = NW == 8 ? 0 : NW == 4 ? 3 : 2;
    constexpr int F4_NT = (NW + NH) * 64, F4_NQ = NW * 16;                // threads, channels (shadow the main shape's constants)
    // Transposed accumulators (lane = tile, 4 consecutive channels: 16-byte epilogue accesses, 32 instead of 128 memory instructions per
    // lane): F4_EPI_T above.
    constexpr bool EPI_T = F4_EPI_T != 0;   // (measured on the narrow shapes too, with the residual requested ahead: 64 float4 of residual + Y beside the
                                            //  accumulators spill 48-63 registers and the epilogue gets LONGER, 28 -> 35 k cycles: profiles/r05/)
    constexpr int F4_RAWBUF = NW == 8 ? dlpm::F4_RAWBUF : F4_RAWBUF_N;
    static_assert(NW == 8 || (VS == 0 && SPEC == 0), "wave-split and specialised forms exist for the 128-channel shape only");
    ConvLaunch p = p_in;
    int bh = bh_in, bw = bw_in, nimg = nimg_in;
    if (SPEC) {
        p.C0 = 128; p.C1 = 0; p.Cout = 128; p.R0 = 128; p.act_silu = 1;
        p.src1 = nullptr; p.res1 = nullptr;
        if (SPEC == 1) p.res0 = nullptr;
        nimg = 1;
        __builtin_assume(p.coefA != nullptr);
        __builtin_assume(p.coefB != nullptr);
        __builtin_assume(p.bias != nullptr);
        if (SPEC == 2) __builtin_assume(p.res0 != nullptr);
    }
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *V = wsm;                               // [2][18][4][16][4]
    float *raw = wsm + 2 * F4_VBUF;               // [2][F4_RAWPIX][F4_PRLD] (+ skew)
    float *Cf = raw + 2 * F4_RAWBUF;              // [2][16][2][8]

    DLPM_PHASE_DECL;
#ifdef DLPM_PHASE_TIMING
    const long long _c0 = clock64(), _r0 = wall_clock64();   // shader cycles and 100-MHz ticks: their ratio is the clock the chip holds
#endif
#ifdef DLPM_PHASE_TIMING
    long long _wait = 0;      // cycles this wave spends at the phase barrier (developer builds: slots 16 + wave)
#endif
    unsigned wgid = blockIdx.x;
    const unsigned ngrid = PERS ? (unsigned)p.pers_total : gridDim.x;
    do {      // (one trip unless PERS)
    int tid_ = threadIdx.x;
    // (persistent form: the thread index is opaque per tile -- otherwise every lane-dependent constant of the body is hoisted out of the
    //  tile loop and lives in scratch memory across the K loop: 79 spilled registers)
    if (PERS) asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const bool mfma_wave = !HELP || wave < NW;                 // wave-uniform
    const int li = lane & 15, lk = lane >> 4;
    const int W = p.Wout, H = p.Hout, TW = W >> 2, TH = H >> 2;
    const int Ws = UPS ? (W >> 1) : W, Hs = UPS ? (H >> 1) : H;
    const int Cin = p.C0 + p.C1;
    int nch = Cin / F4_KC, kb = 0;
    unsigned bid = wgid, nblk = ngrid;      // (unsigned, like the grid built-ins: the prologue's divisions stay what they were)
    if constexpr (HELP) {
        // split-K (round 6, narrow shapes only: the 8-wave instantiations compile without it): grid copy ks walks chunks [kb, kb + nch) and
        // writes its partial outputs behind those of the copies before it (the launch carries no bias / residual / statistics)
        if (p.ksplit > 1) {
            nblk = ngrid / p.ksplit;
            const int ks = (int)(wgid / nblk);
            bid = wgid - (unsigned)ks * nblk;
            nch /= p.ksplit;
            kb = ks * nch;
            p.out += (int64_t)ks * p.B * H * W * p.Cout;
        }
    }
    // n-tile-major grid: all workgroups in flight stream the same 128-channel slab of the Winograd-domain weights
    const int ntn = p.Cout / F4_NQ;
    const int nmb = nblk / ntn;
    const int mb = bid % nmb, n0 = (bid / nmb) * F4_NQ;
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
    // halo patch in SOURCE pixels (nearest-x2 upsampling folded into the addressing): output rows 4 ty0 - 1 .. 4 (ty0 + bh)
    const int RH = UPS ? 2 * bh + 2 : 4 * bh + 2, RW = UPS ? 2 * bw + 2 : 4 * bw + 2;
    const int oy = UPS ? 2 * ty0 - 1 : 4 * ty0 - 1, ox = UPS ? 2 * tx0 - 1 : 4 * tx0 - 1;
    const int rpi = RH * RW, npix = nimg * rpi;

    // ---- raw staging: item = (pixel, channel quad of the phase)
    // (the second item of a thread runs over the threads in REVERSE order: a 324-pixel patch has 68 pixels beyond the first
    //  256, and waves 0..2, which also run the input transform, are the last to get one of those)
    const int squad = tid & 1;
    auto pix_of = [&](int it) { return it == 1 ? (F4_NT / 2) + ((F4_NT - 1 - tid) >> 1) : it * (F4_NT / 2) + (tid >> 1); };
    // Row skew of the raw patch (one image per block, no upsampling): patch row ry starts 4 (ry >> 2) floats late.  The
    // transform's lanes are 16 tiles x 2 channel pairs per LDS pass; without the skew the four tile rows of a block start
    // 864 ty floats apart = the same 16-bank group for ty and ty + 2 (2-way to 4-way conflicts on its 84 reads per phase).
    const bool skewed = !UPS && nimg == 1 && bh <= 4;   // (18 patch rows: the skew stays inside the 16 floats of slack)
    int off[QN], lo[QN];   // lo: LDS float offset of the item (bits 0..15) | its image's coefficient offset (bits 16..)
#pragma unroll
    for (int it = 0; it < QN; it++) {
        const int pix = pix_of(it);
        const int img = min(pix / rpi, nimg - 1), r = pix - img * rpi;
        const int ry = r / RW, rx = r - ry * RW;
        const int iy = oy + ry, ix = ox + rx;
        const bool pad = iy < 0 || iy >= Hs || ix < 0 || ix >= Ws || (img0 + img) >= p.B;
        off[it] = pix >= npix ? -2 : (pad ? -1 : (((img0 + img) * Hs + iy) * Ws + ix));
        lo[it] = (pix * F4_PRLD + squad * 4 + (skewed ? 4 * (ry >> 2) : 0)) | ((img * 2 * F4_KC + squad * 4) << 16);
    }
    const bool has_coef = p.coefA != nullptr;
    const int cf_img = tid >> 2, cf_isb = (tid >> 1) & 1;
    const bool cf_mine = has_coef && tid < nimg * 4;
    const float *cf_base = has_coef ? ((cf_isb ? p.coefB : p.coefA) + (int64_t)min(img0 + cf_img, p.B - 1) * Cin + kb * F4_KC + squad * 4) : nullptr;
    float4 xr[QN], cfr = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_raw_into = [&](float4 (&dst)[QN], int chunk) {
        const int c = (kb + chunk) * F4_KC + squad * 4;
        const bool first = c < p.C0;
        const float *sb = first ? p.src0 + c : p.src1 + (c - p.C0);
        const int ld = first ? p.C0 : p.C1;
#pragma unroll
        for (int it = 0; it < QN; it++) dst[it] = *reinterpret_cast<const float4 *>(sb + (int64_t)max(off[it], 0) * ld);
    };
    auto load_coef = [&](int chunk) {
        if (cf_mine) cfr = *reinterpret_cast<const float4 *>(cf_base + chunk * F4_KC);
    };
    auto store_coef = [&](int slot) {
        if (cf_mine) *reinterpret_cast<float4 *>(Cf + slot * F4_CFS + cf_img * 2 * F4_KC + cf_isb * F4_KC + squad * 4) = cfr;
    };
    auto store_raw_item = [&](int slot, int it) {
        if (off[it] == -2) return;
        float *rb = raw + slot * F4_RAWBUF;
        float4 x = xr[it];
        if (has_coef) {
            const float4 ca = *reinterpret_cast<const float4 *>(Cf + slot * F4_CFS + (lo[it] >> 16));
            const float4 cb = *reinterpret_cast<const float4 *>(Cf + slot * F4_CFS + (lo[it] >> 16) + F4_KC);
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
        *reinterpret_cast<float4 *>(rb + (lo[it] & 0xffff)) = x;
    };
    auto store_raw = [&](int slot) {
#pragma unroll
        for (int it = 0; it < QN; it++) store_raw_item(slot, it);
    };

    // ---- input transform V = B^T d B: waves 0..2 take the row pairs (0,5), (1,2), (3,4) of B^T; lane = (channel pair, tile),
    // tile fastest: the 16 lanes of an LDS pass write 256 contiguous bytes of V
    int rbase, vofs;
    {
        const int tile = lane & 15, pair = lane >> 4;
        const int timg = tile / (bh * bw), r = tile - timg * (bh * bw);
        const int ty = r / bw, tx = r - ty * bw;
        rbase = (timg * rpi + (UPS ? 2 : 4) * ty * RW + (UPS ? 2 : 4) * tx) * F4_PRLD + pair * 2 + (skewed ? 4 * ty : 0);
        vofs = (pair * F4_TILES + tile) * 4;
    }
    auto transform_rp = [&](int slot, int rp) {
        const float *rb = raw + slot * F4_RAWBUF + rbase;
        float *vb = V + slot * F4_VBUF + vofs;
        // d(i, c): sample row i, column c of this tile's 6x6 patch (upsampled: source row (i + 1) >> 1 of the 4x4 source patch);
        // rows 4 and 5 of the patch lie in the next skew group
        auto d = [&](int i, int c) {
            const int ri = UPS ? (i + 1) >> 1 : i, ci = UPS ? (c + 1) >> 1 : c;
            return *reinterpret_cast<const float2 *>(rb + (ri * RW + ci) * F4_PRLD + ((skewed && i >= 4) ? 4 : 0));
        };
        float2 Ta[6], Tb[6];
        int a0, a1;
        if (rp == 0) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                Ta[c] = f2fma(PP2, d(0, c), f2fma(-PS2, d(2, c), d(4, c)));
                Tb[c] = f2fma(PP2, d(1, c), f2fma(-PS2, d(3, c), d(5, c)));
            }
            a0 = 0; a1 = 5;
        } else if (rp == 1) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const float2 e = f2fma(-PB2, d(2, c), d(4, c)), o = f2fma(-PB2, d(1, c), d(3, c));
                Ta[c] = f2fma(PA, o, e);
                Tb[c] = f2fma(-PA, o, e);
            }
            a0 = 1; a1 = 2;
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const float2 e = f2fma(-PA2, d(2, c), d(4, c)), o = f2fma(-PA2, d(1, c), d(3, c));
                Ta[c] = f2fma(PB, o, e);
                Tb[c] = f2fma(-PB, o, e);
            }
            a0 = 3; a1 = 4;
        }
        // positions (a, 0..5) of row a = position pairs 3 a .. 3 a + 2: one float4 {pos 2 pp: 2 channels, pos 2 pp + 1: 2 channels} each
        auto row_out = [&](const float2 (&T)[6], int a) {
            float *vr = vb + a * 3 * (4 * F4_TILES * 4);
            const float2 e1 = f2fma(-PB2, T[2], T[4]), o1 = f2fma(-PB2, T[1], T[3]);
            const float2 e2 = f2fma(-PA2, T[2], T[4]), o2 = f2fma(-PA2, T[1], T[3]);
            const float2 v0 = f2fma(PP2, T[0], f2fma(-PS2, T[2], T[4])), v1 = f2fma(PA, o1, e1), v2 = f2fma(-PA, o1, e1);
            const float2 v3 = f2fma(PB, o2, e2), v4 = f2fma(-PB, o2, e2), v5 = f2fma(PP2, T[1], f2fma(-PS2, T[3], T[5]));
            *reinterpret_cast<float4 *>(vr + 0 * (4 * F4_TILES * 4)) = make_float4(v0.x, v0.y, v1.x, v1.y);
            *reinterpret_cast<float4 *>(vr + 1 * (4 * F4_TILES * 4)) = make_float4(v2.x, v2.y, v3.x, v3.y);
            *reinterpret_cast<float4 *>(vr + 2 * (4 * F4_TILES * 4)) = make_float4(v4.x, v4.y, v5.x, v5.y);
        };
        row_out(Ta, a0);
        row_out(Tb, a1);
    };
    auto transform = [&](int slot) {
        if (HELP) {
            if (mfma_wave) return;
            if (NH == 3) {
                transform_rp(slot, wave - NW);
            } else if (wave == NW) {
                transform_rp(slot, 0);
                transform_rp(slot, 1);
            } else {
                transform_rp(slot, 2);
            }
        } else if (wave < 3) {
            transform_rp(slot, wave);
        }
    };

    // ---- weight stream of this wave: Wf[ntile][wave][phase][18 position pairs][lane][4], contiguous per wave
    // (wave-uniform pointer + lane: the per-fragment advance is scalar arithmetic)
    const float4 *__restrict__ wp = reinterpret_cast<const float4 *>(p.w_wino4) +
                                    ((int64_t)((n0 / F4_NQ) * NW + __builtin_amdgcn_readfirstlane(wave)) * (Cin / F4_KC) + kb) * 18 * 64;
    constexpr int AHEAD = F4_RING - 1;
    float4 bq[F4_RING];
    // A fragments: lane (li = tile, lk) reads channels 2 lk, 2 lk + 1 of the phase (= k index lk of the two k-steps) for both
    // positions of a pair: 16 bytes, consecutive lanes at consecutive addresses
    const float *asrc = V + lane * 4;

    floatx4 acc[36];
#pragma unroll
    for (int q = 0; q < 36; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[q][r] = 0.f;

    // ---- prologue: S(0), S(1), X(0), G(2) and the coefficient slots; everything the first two phases need is
    // requested at once (one exposed round trip)
    const int last = nch - 1;
    float4 xr1[QN], cfr1 = make_float4(0.f, 0.f, 0.f, 0.f), cfr2 = cfr1;
    load_raw_into(xr, 0);
    load_raw_into(xr1, min(1, last));
    load_coef(0);
    if (cf_mine) {
        cfr1 = *reinterpret_cast<const float4 *>(cf_base + min(1, last) * F4_KC);
        cfr2 = *reinterpret_cast<const float4 *>(cf_base + min(2, last) * F4_KC);
    }
    if (mfma_wave) {
#pragma unroll
        for (int a = 0; a < AHEAD; a++) bq[a] = wp[a * 64 + lane];
    }
    store_coef(0);
    cfr = cfr1;
    store_coef(1);
    __syncthreads();
    store_raw(0);
#pragma unroll
    for (int it = 0; it < QN; it++) xr[it] = xr1[it];
    store_raw(1);
    load_raw_into(xr, min(2, last));
    __syncthreads();
    transform(0);
    cfr = cfr2;
    store_coef(0);
    __syncthreads();
    DLPM_PHASE(p, 8);

#pragma unroll 1
    for (int chunk = 0; chunk < nch; chunk++) {
        const int cur = chunk & 1, nxt = cur ^ 1;
        const float *ab = asrc + cur * F4_VBUF;
        load_coef(min(chunk + 3, last));
        float4 aq[F4_AAHEAD + 1];
#pragma unroll
        for (int a = 0; a < F4_AAHEAD; a++)
            if (!VS) aq[a] = (ABL & 64) ? make_float4(1.f, 2.f, 1.f, 2.f) : *reinterpret_cast<const float4 *>(ab + a * (4 * F4_TILES * 4));
        if (ABL & 64) aq[F4_AAHEAD] = make_float4(1.f, 2.f, 1.f, 2.f);
        if (HELP && !mfma_wave) {
            // helper wave: its share of S(chunk+2) raw stores and G(chunk+3) global loads, X(chunk+1) its row pair(s) of the transform
#pragma unroll
            for (int it = 0; it < QN; it++) store_raw_item(cur, it);
            load_raw_into(xr, min(chunk + 3, last));
            transform(nxt);
        } else if (VS) {
            // slot s = (position pair pp = s / 2 of this wave's nine, channel tile nt = s % 2): 18 slots, one weight fragment and four
            // MFMAs each, side work keyed on the slot exactly as on the position pair of the VS = 0 loop
            const float *abh = ab + (wave >> 2) * 9 * (4 * F4_TILES * 4);
            aq[0] = *reinterpret_cast<const float4 *>(abh);
#pragma unroll
            for (int sl = 0; sl < 18; sl++) {
                const int pp = sl >> 1, nt = sl & 1;
                if (sl >= F4_S0 && sl < F4_S0 + QN) store_raw_item(cur, sl - F4_S0);
                if (sl == F4_S0 + QN) load_raw_into(xr, min(chunk + 3, last));
                if (sl == F4_X) transform(nxt);
                bq[(sl + AHEAD) % F4_RING] = wp[AHEAD * 64 + lane];
                wp += 64;
                if (nt == 0 && pp + 1 < 9) aq[(pp + 1) & 1] = *reinterpret_cast<const float4 *>(abh + (pp + 1) * (4 * F4_TILES * 4));
                const float4 aa = aq[pp & 1];
                const float2 a0 = make_float2(aa.x, aa.y), a1 = make_float2(aa.z, aa.w);
                const float4 b = bq[sl % F4_RING];
                if (F4_PRIO) __builtin_amdgcn_s_setprio(0);
                acc[4 * pp + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc[4 * pp + nt], 0, 0, 0);
                acc[4 * pp + 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.z, acc[4 * pp + 2 + nt], 0, 0, 0);
                acc[4 * pp + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc[4 * pp + nt], 0, 0, 0);
                acc[4 * pp + 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.w, acc[4 * pp + 2 + nt], 0, 0, 0);
                if (F4_PRIO) __builtin_amdgcn_s_setprio(1);
            }
        } else {
#pragma unroll
        for (int pp = 0; pp < 18; pp++) {
#if F4_SKIP_TAIL
            // the pipeline's look-ahead work of the LAST phases feeds chunks that do not exist: S(chunk+2) in the last two phases and
            // X(chunk+1) in the last one are skipped (one-sided wave-uniform branches; the loads stay unconditional)
            if (!(ABL & 1) && pp >= F4_S0 && pp < F4_S0 + QN && chunk + 2 < nch) store_raw_item(cur, pp - F4_S0);
            if (!(ABL & 4) && pp == F4_S0 + QN) load_raw_into(xr, min(chunk + 3, last));
            if (!(ABL & 2) && pp == F4_X && chunk + 1 < nch) transform(nxt);
#else
            if (!(ABL & 1) && pp >= F4_S0 && pp < F4_S0 + QN) store_raw_item(cur, pp - F4_S0);   // S(chunk+2): raw[cur] was read by X(chunk), a barrier ago
            if (!(ABL & 4) && pp == F4_S0 + QN) load_raw_into(xr, min(chunk + 3, last));        // G(chunk+3)
            if (!HELP && !(ABL & 2) && pp == F4_X) transform(nxt);                                    // X(chunk+1): raw[nxt] -> V[nxt] (narrow shapes: the helper waves')
#endif
            if (!(ABL & 16)) bq[(pp + AHEAD) % F4_RING] = wp[AHEAD * 64 + lane];
            wp += 64;
            // A fragments are read F4_AAHEAD position pairs ahead (left to the compiler each ds_read sat directly in
            // front of its MFMAs with an s_waitcnt lgkmcnt(0) between them)
            if (!(ABL & 64) && pp + F4_AAHEAD < 18)
                aq[(pp + F4_AAHEAD) % (F4_AAHEAD + 1)] = *reinterpret_cast<const float4 *>(ab + (pp + F4_AAHEAD) * (4 * F4_TILES * 4));
            const float4 aa = aq[pp % (F4_AAHEAD + 1)];
            const float2 a0 = make_float2(aa.x, aa.y), a1 = make_float2(aa.z, aa.w);
            const float4 b = bq[pp % F4_RING];
            // wave priority: everything that is not an MFMA (operand fetch, staging, transform) issues at priority 1, the
            // MFMAs at 0 -- when both waves of a SIMD are ready, the one with side work goes first and the other's MFMAs
            // fill the pipe behind it.  Measured 45.6 -> 43.9 ms/step (the opposite assignment: 44.4)
            if (F4_PRIO) __builtin_amdgcn_s_setprio(0);
            if (!(ABL & 32)) {
            if constexpr (EPI_T) {
            // operands swapped: D rows = the wave's 16 output channels, D columns = the 16 tiles, i.e. a lane (li = tile, lk)
            // ends up with FOUR CONSECUTIVE CHANNELS (4 lk + r) of one tile -- same products, same order, transposed registers
            acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a0.x, acc[2 * pp], 0, 0, 0);
            acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, a1.x, acc[2 * pp + 1], 0, 0, 0);
            acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a0.y, acc[2 * pp], 0, 0, 0);
            acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, a1.y, acc[2 * pp + 1], 0, 0, 0);
            } else {
            acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc[2 * pp], 0, 0, 0);
            acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.z, acc[2 * pp + 1], 0, 0, 0);
            acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc[2 * pp], 0, 0, 0);
            acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.w, acc[2 * pp + 1], 0, 0, 0);
            }
            } else {
                acc[2 * pp][0] += a0.x * b.x + a0.y * b.y;   // keep the operands alive
                acc[2 * pp + 1][0] += a1.x * b.z + a1.y * b.w;
            }
            if (F4_PRIO) __builtin_amdgcn_s_setprio(1);
        }
        }
        store_coef(nxt);
#ifdef DLPM_PHASE_TIMING
        const long long _w0 = clock64();
#endif
        if (!(ABL & 8)) __syncthreads();
#ifdef DLPM_PHASE_TIMING
        _wait += clock64() - _w0;
#endif
    }
    if (ABL & 8) __syncthreads();
    if (F4_PRIO) __builtin_amdgcn_s_setprio(0);
    DLPM_PHASE(p, 9);
    if (HELP && !mfma_wave) return;   // (no barrier behind the loop: the epilogue runs out of the MFMA waves' registers)
#if defined(DLPM_PHASE_TIMING) && !defined(DLPM_PHASE_DEFER)
    if (p.phase && lane == 0) atomicAdd(p.phase + 16 + wave, (unsigned long long)_wait);
#endif

    if constexpr (EPI_T) {
        // ---- epilogue from registers: no LDS, no barrier.  With the MFMA operands swapped a lane (li = tile, lk) holds
        // M_pos[tile li][channels 16 wave + 4 lk .. + 3] in acc[pos][0..3]: the output transform runs once per channel, and
        // bias / residual / stores move FOUR consecutive channels of a pixel per instruction (16 bytes per lane, the four lk
        // lanes of a tile = one 64-byte segment): 16 loads + 16 stores per lane where the lane = channel layout of round 2
        // issued 64 + 64 four-byte ones (the epilogue is bound by the number of memory instructions the CU's one
        // texture-address unit takes from 8 waves, not by bytes).  Fused GroupNorm statistics: per-lane shifted sums over
        // the tile's 16 pixels, then the 16 tiles merged by four butterfly steps in a fixed order.
        const int64_t pix0 = ((int64_t)img0 * H + 4 * ty0) * W + 4 * tx0;      // wave-uniform
        const int ch = n0 + 16 * wave + 4 * lk;
        float4 bias_v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bias_v = *reinterpret_cast<const float4 *>(p.bias + ch);
        const bool has_res = p.res0 != nullptr;
        const bool res_first = __builtin_amdgcn_readfirstlane(n0 + 16 * wave) < p.R0;   // R0 % 16 == 0 (wino4_geometry)
        const float *res_u = has_res ? (res_first ? p.res0 : p.res1 - p.R0) : nullptr;
        const int res_ld = res_first ? p.R0 : p.Cout - p.R0;
        const int lbw = 31 - __builtin_clz(bw), lbhw = 31 - __builtin_clz(bh * bw);      // block shapes are powers of two
        const int timg = li >> lbhw, ty = (li & (bh * bw - 1)) >> lbw, tx = li & (bw - 1);
        const int tpix = (timg * H + 4 * ty) * W + 4 * tx;
        const bool ok = img0 + timg < p.B;
        float *__restrict__ out_t = p.out + (pix0 + tpix) * p.Cout + ch;
        const float *__restrict__ res_t = has_res ? res_u + (pix0 + tpix) * res_ld + ch : nullptr;
        const bool do_stats = p.stats_out != nullptr && nimg == 1;
        float Y[4][16];   // [channel][pixel 4 i + j]
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float Z[4][6];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const float m0 = acc[0 * 6 + b][r], m1 = acc[1 * 6 + b][r], m2 = acc[2 * 6 + b][r];
                const float m3 = acc[3 * 6 + b][r], m4 = acc[4 * 6 + b][r], m5 = acc[5 * 6 + b][r];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                Z[0][b] = m0 + s12 + s34;
                Z[1][b] = fmaf(PB, d34, PA * d12);
                Z[2][b] = fmaf(PB2, s34, PA2 * s12);
                Z[3][b] = fmaf(PB3, d34, PA3 * d12) + m5;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float s12 = Z[i][1] + Z[i][2], d12 = Z[i][1] - Z[i][2], s34 = Z[i][3] + Z[i][4], d34 = Z[i][3] - Z[i][4];
                Y[r][i * 4 + 0] = Z[i][0] + s12 + s34;
                Y[r][i * 4 + 1] = fmaf(PB, d34, PA * d12);
                Y[r][i * 4 + 2] = fmaf(PB2, s34, PA2 * s12);
                Y[r][i * 4 + 3] = fmaf(PB3, d34, PA3 * d12) + Z[i][5];
            }
        }
        float4 K = make_float4(0.f, 0.f, 0.f, 0.f), s1 = K, s2 = K;
        if (ok) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float4 rs[4];
                if (has_res) {
#pragma unroll
                    for (int j = 0; j < 4; j++) rs[j] = *reinterpret_cast<const float4 *>(res_t + (i * W + j) * res_ld);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float4 v = make_float4(Y[0][i * 4 + j] + bias_v.x, Y[1][i * 4 + j] + bias_v.y, Y[2][i * 4 + j] + bias_v.z,
                                           Y[3][i * 4 + j] + bias_v.w);
                    if (has_res) { v.x += rs[j].x; v.y += rs[j].y; v.z += rs[j].z; v.w += rs[j].w; }
                    if (do_stats) {
                        if (i == 0 && j == 0) K = v;
                        float d;
                        d = v.x - K.x; s1.x += d; s2.x = fmaf(d, d, s2.x);
                        d = v.y - K.y; s1.y += d; s2.y = fmaf(d, d, s2.y);
                        d = v.z - K.z; s1.z += d; s2.z = fmaf(d, d, s2.z);
                        d = v.w - K.w; s1.w += d; s2.w = fmaf(d, d, s2.w);
                    }
                    *reinterpret_cast<float4 *>(out_t + (i * W + j) * p.Cout) = v;
                }
            }
        }
        if (do_stats) {
            float mean[4] = {K.x + s1.x * (1.f / 16.f), K.y + s1.y * (1.f / 16.f), K.z + s1.z * (1.f / 16.f), K.w + s1.w * (1.f / 16.f)};
            float M2[4] = {fmaxf(s2.x - s1.x * s1.x * (1.f / 16.f), 0.f), fmaxf(s2.y - s1.y * s1.y * (1.f / 16.f), 0.f),
                           fmaxf(s2.z - s1.z * s1.z * (1.f / 16.f), 0.f), fmaxf(s2.w - s1.w * s1.w * (1.f / 16.f), 0.f)};
            float na = 16.f;
#pragma unroll
            for (int sft = 1; sft <= 8; sft <<= 1) {      // the 16 tiles (lane bits 0..3), lower lane first
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float om = __shfl_xor(mean[c], sft), oM2 = __shfl_xor(M2[c], sft);
                    const float lo_m = (lane & sft) ? om : mean[c], hi_m = (lane & sft) ? mean[c] : om;
                    const float lo_M = (lane & sft) ? oM2 : M2[c], hi_M = (lane & sft) ? M2[c] : oM2;
                    const float dd = hi_m - lo_m;
                    mean[c] = lo_m + dd * 0.5f;
                    M2[c] = lo_M + hi_M + dd * dd * (na * 0.5f);
                }
                na *= 2.f;
            }
            if (li == 0) {
                float2 *so = p.stats_out + ((int64_t)img0 * ((H * W) / 256) + blk_in_img) * p.Cout + ch;
#pragma unroll
                for (int c = 0; c < 4; c++) so[c] = make_float2(mean[c], M2[c]);
            }
        }
    } else {
        // ---- epilogue from registers: no LDS, no barrier.  A lane holds 64 outputs of ONE channel (4 tiles x 16 pixels:
        // M_pos[tile 4 lk + r][channel 16 wave + li] in acc[pos][r]), so the output transform, bias, residual, the fused
        // GroupNorm statistics (per-lane shifted sums, then the four 16-lane groups merged in a fixed order) and the stores
        // all happen where the values are; a 16-lane group writes 64 contiguous bytes of an NHWC row.  (The round-1
        // epilogue staged the block through a 135-KB LDS row image for whole-row stores: 3 % slower once the loop's own LDS
        // traffic had been cut, profiles/r02/conv_layers_register_epilogue*.txt.)
        const int64_t pix0 = ((int64_t)img0 * H + 4 * ty0) * W + 4 * tx0;      // wave-uniform
        // VS: wave (q = wave & 3, ph = wave >> 2) finalises channel tile ph of its quarter: channels 32 q + 16 ph ..
        const int vph = wave >> 2, cw = VS ? 32 * (wave & 3) + 16 * vph : 16 * wave;
        const int ch = n0 + cw + li;
        const float bias_v = p.bias ? p.bias[ch] : 0.f;
        const bool has_res = p.res0 != nullptr;
        const bool res_first = __builtin_amdgcn_readfirstlane(n0 + cw) < p.R0;   // R0 % 16 == 0 (wino4_geometry)
        // VS: the 16 partial outputs of tile r and channel tile nt from this wave's three transform rows (ph = 0: the points 0, +a, -a;
        // ph = 1: +b, -b, inf), then the partner's channel tile goes to LDS: xb[writer wave][r][k][lane]
        float *xb = wsm;
        // (nt arrives as a compile-time constant: a run-time index into acc[] would move the accumulators to scratch memory)
        auto partial = [&](int r, auto nt_c, float (&y)[16]) {
            constexpr int nt = decltype(nt_c)::value;
            float Z[4][6];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const float ma = acc[(0 * 6 + b) * 2 + nt][r], mb = acc[(1 * 6 + b) * 2 + nt][r], mc = acc[(2 * 6 + b) * 2 + nt][r];
                if (vph == 0) {
                    const float sm = mb + mc, df = mb - mc;
                    Z[0][b] = ma + sm; Z[1][b] = PA * df; Z[2][b] = PA2 * sm; Z[3][b] = PA3 * df;
                } else {
                    const float sm = ma + mb, df = ma - mb;
                    Z[0][b] = sm; Z[1][b] = PB * df; Z[2][b] = PB2 * sm; Z[3][b] = fmaf(PB3, df, mc);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float s12 = Z[i][1] + Z[i][2], d12 = Z[i][1] - Z[i][2], s34 = Z[i][3] + Z[i][4], d34 = Z[i][3] - Z[i][4];
                y[i * 4 + 0] = Z[i][0] + s12 + s34;
                y[i * 4 + 1] = fmaf(PB, d34, PA * d12);
                y[i * 4 + 2] = fmaf(PB2, s34, PA2 * s12);
                y[i * 4 + 3] = fmaf(PB3, d34, PA3 * d12) + Z[i][5];
            }
        };
        if (VS) {
            // (the loop's last phase barrier is behind every wave: V / raw are dead, the exchange buffer may overwrite them)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float yo[16];
                if (vph == 0) partial(r, std::integral_constant<int, 1>(), yo);
                else partial(r, std::integral_constant<int, 0>(), yo);
#pragma unroll
                for (int k = 0; k < 16; k++) xb[((wave * 4 + r) * 16 + k) * 64 + lane] = yo[k];
            }
            __syncthreads();
        }
        const float *res_u = has_res ? (res_first ? p.res0 : p.res1 - p.R0) : nullptr;
        const int res_ld = res_first ? p.R0 : p.Cout - p.R0;
        const int lbw = 31 - __builtin_clz(bw), lbhw = 31 - __builtin_clz(bh * bw);      // block shapes are powers of two
        float *__restrict__ out_blk = p.out + pix0 * p.Cout;
        const float *__restrict__ res_blk = has_res ? res_u + pix0 * res_ld : nullptr;
        // statistics: one partial per 256-pixel block inside an image (nimg == 1), or -- four whole 8x8 images per block, four tiles
        // each -- one per IMAGE: a lane's 64 outputs (tiles 4 lk .. 4 lk + 3) are then exactly image lk's pixels of its channel
        const bool img_stats = p.stats_out != nullptr && nimg == 4 && bh * bw == 4;
        const bool do_stats = ((p.stats_out != nullptr && nimg == 1) || img_stats) && !(HELP && (F4N_ABL & 4));
        float K = 0.f, s1 = 0.f, s2 = 0.f;
        constexpr bool RES_AHEAD = HELP || F4_RES_AHEAD != 0;
        float rs_all[RES_AHEAD ? 4 : 1][16] = {};   // (read unconditionally below, used only under has_res)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int tile = 4 * lk + r;
            const int timg = tile >> lbhw, ty = (tile & (bh * bw - 1)) >> lbw, tx = tile & (bw - 1);
            const int tpix = (timg * H + 4 * ty) * W + 4 * tx;
            const bool ok = img0 + timg < p.B;
            float rs[16];
            // addresses = wave-uniform base (block origin + pixel (i, j) of the tile) + a 32-bit per-lane byte offset
            const uint32_t bo_o = (uint32_t)(tpix * p.Cout + ch) * 4u, bo_r = (uint32_t)(tpix * res_ld + ch) * 4u;
            if (RES_AHEAD) {
                // narrow shapes (one or two waves per SIMD, nothing else hides a wait): ALL four tiles' residual values are requested
                // before the first output transform -- the registers of the weight ring and the A fragments are free by now
                if (r == 0 && has_res && !(HELP && (F4N_ABL & 2))) {
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const int t2 = 4 * lk + rr;
                        const int tp2 = ((t2 >> lbhw) * H + 4 * ((t2 & (bh * bw - 1)) >> lbw)) * W + 4 * (t2 & (bw - 1));
                        const uint32_t b2 = (uint32_t)(tp2 * res_ld + ch) * 4u;
                        const bool ok2 = img0 + (t2 >> lbhw) < p.B;
#pragma unroll
                        for (int k = 0; k < 16; k++)
                            rs_all[rr][k] = ok2 ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(res_blk + ((k >> 2) * W + (k & 3)) * res_ld) + b2) : 0.f;
                    }
                }
#pragma unroll
                for (int k = 0; k < 16; k++) rs[k] = rs_all[r][k];
            } else if (has_res && ok) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        rs[i * 4 + j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(res_blk + (i * W + j) * res_ld) + bo_r);
            }
            float Yt[16];
            if (VS) {
                if (vph == 0) partial(r, std::integral_constant<int, 0>(), Yt);
                else partial(r, std::integral_constant<int, 1>(), Yt);
#pragma unroll
                for (int k = 0; k < 16; k++) Yt[k] += xb[(((wave ^ 4) * 4 + r) * 16 + k) * 64 + lane];
            } else {
                float Z[4][6];
#pragma unroll
                for (int b = 0; b < 6; b++) {
                    const float m0 = acc[0 * 6 + b][r], m1 = acc[1 * 6 + b][r], m2 = acc[2 * 6 + b][r];
                    const float m3 = acc[3 * 6 + b][r], m4 = acc[4 * 6 + b][r], m5 = acc[5 * 6 + b][r];
                    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                    Z[0][b] = m0 + s12 + s34;
                    Z[1][b] = fmaf(PB, d34, PA * d12);
                    Z[2][b] = fmaf(PB2, s34, PA2 * s12);
                    Z[3][b] = fmaf(PB3, d34, PA3 * d12) + m5;
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float s12 = Z[i][1] + Z[i][2], d12 = Z[i][1] - Z[i][2], s34 = Z[i][3] + Z[i][4], d34 = Z[i][3] - Z[i][4];
                    Yt[i * 4 + 0] = Z[i][0] + s12 + s34;
                    Yt[i * 4 + 1] = fmaf(PB, d34, PA * d12);
                    Yt[i * 4 + 2] = fmaf(PB2, s34, PA2 * s12);
                    Yt[i * 4 + 3] = fmaf(PB3, d34, PA3 * d12) + Z[i][5];
                }
            }
            if (!ok) continue;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float *y = Yt + 4 * i;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float v = y[j] + bias_v;
                    if (has_res) v += rs[i * 4 + j];
                    if (do_stats) {
                        if (r == 0 && i == 0 && j == 0) K = v;
                        const float dd = v - K;
                        s1 += dd;
                        s2 = fmaf(dd, dd, s2);
                    }
                    if (HELP && (F4N_ABL & 1)) { if (v == 123.456f) *reinterpret_cast<float *>(reinterpret_cast<char *>(out_blk) + bo_o) = v; continue; }
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(out_blk + (i * W + j) * p.Cout) + bo_o) = v;
                }
            }
        }
        if (img_stats) {
            if (img0 + lk < p.B)
                p.stats_out[(int64_t)(img0 + lk) * p.Cout + ch] = make_float2(K + s1 * (1.f / 64.f), fmaxf(s2 - s1 * s1 * (1.f / 64.f), 0.f));
        } else if (do_stats) {
            float mean = K + s1 * (1.f / 64.f), M2 = fmaxf(s2 - s1 * s1 * (1.f / 64.f), 0.f), na = 64.f;
#pragma unroll
            for (int sft = 16; sft <= 32; sft <<= 1) {
                const float om = __shfl_xor(mean, sft), oM2 = __shfl_xor(M2, sft);
                const float lo_m = (lane & sft) ? om : mean, hi_m = (lane & sft) ? mean : om;
                const float lo_M = (lane & sft) ? oM2 : M2, hi_M = (lane & sft) ? M2 : oM2;
                const float dd = hi_m - lo_m;
                mean = lo_m + dd * 0.5f;
                M2 = lo_M + hi_M + dd * dd * (na * 0.5f);
                na *= 2.f;
            }
            if (lk == 0) p.stats_out[((int64_t)img0 * ((H * W) / 256) + blk_in_img) * p.Cout + ch] = make_float2(mean, M2);
        }
    }
    } while (PERS && (wgid += gridDim.x) < ngrid);
    DLPM_PHASE(p, 10);
    DLPM_PHASE_FLUSH(p, 8);
#ifdef DLPM_PHASE_TIMING
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#endif
#ifdef DLPM_PHASE_DEFER
    if (p.phase && lane == 0) atomicAdd(p.phase + 16 + wave, (unsigned long long)_wait);
#endif
#ifdef DLPM_PHASE_TIMING
    if (p.phase && tid == 0) {
        atomicAdd(p.phase + 11, 1ull);
        atomicAdd(p.phase + 12, (unsigned long long)(clock64() - _c0));
        atomicAdd(p.phase + 13, (unsigned long long)(wall_clock64() - _r0));
    }
#endif
}

// =====================================================================================================================================
// Round 6 -- the WHOLE-IMAGE shape: 32 output channels on 32x32 images (the first level of the MNIST-sized nets: 10 of the step's 3x3
// launches, 0.68 of its 2.37 ms).  The 2 + 2-wave shape above gives such a layer 1024 workgroups of 16 tiles whose phases are mostly NOT
// matrix work: per 8-channel phase a workgroup has 2.3 k cycles of MFMAs on each of two waves, and as much VALU again in staging
// (GroupNorm affine + SiLU on 324 halo pixels) and the input transform, all on critical paths of one wave per SIMD; measured with the
// deferred counters (profiles/r06/epilogue_lockstep/): 47.5 k cycles per workgroup alone on a CU, 83 k with its twin, 62 us per launch.
// On gfx950 the fp32 MFMA and the VALU do not overlap on a SIMD anyway (tools/mb/mfma_valu_2waves), so nothing is lost by giving the
// side work to EVERY wave and running the phases one after the other:
//   * one workgroup = one image = 64 tiles (four 16-tile M-blocks = the image's quadrants) x 32 channels, 8 waves = 4 M-blocks x 2 channel
//     halves: every wave is an MFMA wave (36 accumulator tiles), and two waves per SIMD hide each other's waits;
//   * per 8-channel chunk:  transform(c): raw -> V, 24 (row group, M-block) units over the 8 waves (rows {1,2}+{0} | {3,4}+{5});
//     barrier;  MFMAs(c) with the global loads of chunk c+1 in flight, then activate + store raw(c+1);  barrier.  V and raw are single
//     buffers (73.7 + 55.7 KB); the barriers order LDS only, so the read-ahead stays in flight across them;
//   * the halo is the image's own border: 1156 staged pixels per 1024 outputs (the 16-tile blocks stage 1296), every element activated once;
//   * SAME arithmetic per output as the 2 + 2-wave shape -- same staged values, same transform expressions, same k order per accumulator,
//     same register epilogue and the same four 256-pixel statistics partials per image -- so the results are bit-identical to it
//     (tests/test_gpu_kernels.py::test_conv_winograd_f4_whole_image_is_bit_identical) and the weights are its fragment stream
//     Wf[ntile][wave < 2][phase][18][lane][4], here read by the four M-block waves of a channel half.
constexpr int FI_TILES = 64, FI_RW = 34, FI_NPIX = FI_RW * FI_RW, FI_NT = 512;
constexpr int FI_QN = (FI_NPIX * 2 + FI_NT - 1) / FI_NT;      // staging items (pixel, channel quad) per thread: 5
constexpr int FI_RAWBUF = FI_NPIX * F4_PRLD + 48;             // + the row skew (4 floats per four patch rows: <= 32)
constexpr int FI_VPP = 4 * FI_TILES * 4;                      // floats per position pair: [4 channel pairs][64 tiles][4]
constexpr int FI_VBUF = 18 * FI_VPP;

// workgroup barrier that orders LDS only (__syncthreads() fences every address space: s_waitcnt vmcnt(0) would sit out the read-ahead)
#define F4_LDS_BARRIER()                                                  \
    do {                                                                  \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   \
        __builtin_amdgcn_s_barrier();                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   \
    } while (0)

__global__ void __launch_bounds__(FI_NT, 1) k_conv3x3_wino4_img(ConvLaunch p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *V = wsm;                      // [18 position pairs][4 channel pairs][64 tiles][2 pos x 2 ch]
    float *raw = V + FI_VBUF;            // [34 x 34][F4_PRLD] (+ skew)
    float *Cf = raw + FI_RAWBUF;         // [A | B][Cin] GroupNorm affine of this image
    DLPM_PHASE_DECL;
#ifdef DLPM_PHASE_TIMING
    const long long _c0 = clock64(), _r0 = wall_clock64();
#endif
    constexpr int H = 32, W = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int mb = wave >> 1, nw = wave & 1;        // MFMA role: M-block (image quadrant, row-major) and channel half
    const int img = blockIdx.x;
    const int Cin = p.C0 + p.C1, nch = Cin / F4_KC, last = nch - 1;
    const bool has_coef = p.coefA != nullptr;

    // ---- raw staging: item = (halo pixel, channel quad of the chunk), 5 per thread
    // One register per item: LDS float offset (bits 0..13) | source pixel inside the image (bits 14..23) | padding (bit 24) | no item (bit 25)
    const int squad = tid & 1;
    int itm[FI_QN];
#pragma unroll
    for (int it = 0; it < FI_QN; it++) {
        const int pix = it * (FI_NT / 2) + (tid >> 1);
        const int ry = pix / FI_RW, rx = pix - ry * FI_RW;
        const int iy = ry - 1, ix = rx - 1;
        const bool pad = iy < 0 || iy >= H || ix < 0 || ix >= W;
        const int lo = pix * F4_PRLD + squad * 4 + 4 * (ry >> 2);      // patch row ry starts 4 (ry >> 2) floats late (bank spread of the transform's reads)
        itm[it] = pix >= FI_NPIX ? (1 << 25) : (lo | (pad ? (1 << 24) : ((iy * W + ix) << 14)));
    }
    float4 xr[FI_QN];
    auto load_raw = [&](int chunk) {
        const int c = chunk * F4_KC + squad * 4;
        const bool first = c < p.C0;
        const int ld = first ? p.C0 : p.C1;
        const float *sb = (first ? p.src0 + c : p.src1 + (c - p.C0)) + (int64_t)img * (H * W) * ld;
#pragma unroll
        for (int it = 0; it < FI_QN; it++) xr[it] = *reinterpret_cast<const float4 *>(sb + ((itm[it] >> 14) & 1023) * ld);   // (padding / no item: pixel 0)
    };
    auto store_raw = [&](int chunk) {
        const int c = chunk * F4_KC + squad * 4;
        float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_coef) {
            ca = *reinterpret_cast<const float4 *>(Cf + c);
            cb = *reinterpret_cast<const float4 *>(Cf + Cin + c);
        }
#pragma unroll
        for (int it = 0; it < FI_QN; it++) {
            int iv = itm[it];
            asm volatile("" : "+v"(iv));     // (opaque: hoisted out of the chunk loop, the five LDS addresses and padding masks cost registers the loop does not have)
            if (iv & (1 << 25)) continue;
            float4 x = xr[it];
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
            if (iv & (1 << 24)) x = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding applies AFTER the activation
            *reinterpret_cast<float4 *>(raw + (iv & 16383)) = x;
        }
    };

    // ---- input transform V = B^T d B: wave (tg = wave & 3, th = wave >> 2) takes M-block tg, transform rows {1, 2} + {0} (th = 0) or
    // {3, 4} + {5} (th = 1); lane = (channel pair, tile), tile fastest
    int rbase, vofs;
    {
        const int tg = wave & 3, tile = lane & 15, pair = lane >> 4;
        const int ty = 4 * (tg >> 1) + (tile >> 2), tx = 4 * (tg & 1) + (tile & 3);
        rbase = (4 * ty * FI_RW + 4 * tx) * F4_PRLD + pair * 2 + 4 * ty;
        vofs = (pair * FI_TILES + 16 * tg + tile) * 4;
    }
    const int th = wave >> 2;
    auto transform = [&]() {
        const float *rb = raw + rbase;
        float *vb = V + vofs;
        auto d = [&](int i, int c) {      // sample row i, column c of this tile's 6x6 patch; rows 4 and 5 lie in the next skew group
            return *reinterpret_cast<const float2 *>(rb + (i * FI_RW + c) * F4_PRLD + (i >= 4 ? 4 : 0));
        };
        auto row_out = [&](const float2 (&T)[6], int a) {
            float *vr = vb + a * 3 * FI_VPP;
            const float2 e1 = f2fma(-PB2, T[2], T[4]), o1 = f2fma(-PB2, T[1], T[3]);
            const float2 e2 = f2fma(-PA2, T[2], T[4]), o2 = f2fma(-PA2, T[1], T[3]);
            const float2 v0 = f2fma(PP2, T[0], f2fma(-PS2, T[2], T[4])), v1 = f2fma(PA, o1, e1), v2 = f2fma(-PA, o1, e1);
            const float2 v3 = f2fma(PB, o2, e2), v4 = f2fma(-PB, o2, e2), v5 = f2fma(PP2, T[1], f2fma(-PS2, T[3], T[5]));
            *reinterpret_cast<float4 *>(vr + 0 * FI_VPP) = make_float4(v0.x, v0.y, v1.x, v1.y);
            *reinterpret_cast<float4 *>(vr + 1 * FI_VPP) = make_float4(v2.x, v2.y, v3.x, v3.y);
            *reinterpret_cast<float4 *>(vr + 2 * FI_VPP) = make_float4(v4.x, v4.y, v5.x, v5.y);
        };
        float2 Ta[6], Tb[6];
        if (th == 0) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const float2 e = f2fma(-PB2, d(2, c), d(4, c)), o = f2fma(-PB2, d(1, c), d(3, c));
                Ta[c] = f2fma(PA, o, e);
                Tb[c] = f2fma(-PA, o, e);
            }
            row_out(Ta, 1);
            row_out(Tb, 2);
#pragma unroll
            for (int c = 0; c < 6; c++) Ta[c] = f2fma(PP2, d(0, c), f2fma(-PS2, d(2, c), d(4, c)));
            row_out(Ta, 0);
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const float2 e = f2fma(-PA2, d(2, c), d(4, c)), o = f2fma(-PA2, d(1, c), d(3, c));
                Ta[c] = f2fma(PB, o, e);
                Tb[c] = f2fma(-PB, o, e);
            }
            row_out(Ta, 3);
            row_out(Tb, 4);
#pragma unroll
            for (int c = 0; c < 6; c++) Tb[c] = f2fma(PP2, d(1, c), f2fma(-PS2, d(3, c), d(5, c)));
            row_out(Tb, 5);
        }
    };

    // ---- weight stream of this wave's channel half (the 2 + 2-wave shape's Wf[ntile 0][wave nw][phase][18][lane][4]) and its A fragments
    const float4 *__restrict__ wp = reinterpret_cast<const float4 *>(p.w_wino4) + (int64_t)nw * nch * 18 * 64;
    constexpr int AHEAD = F4_RING - 1;
    float4 bq[F4_RING];
    const float *asrc = V + (lk * FI_TILES + 16 * mb + li) * 4;
    floatx4 acc[36];
#pragma unroll
    for (int q = 0; q < 36; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[q][r] = 0.f;

    // ---- prologue: coefficients -> LDS, chunk 0 staged
    load_raw(0);
    if (has_coef) {
        for (int i = tid; i < 2 * Cin; i += FI_NT) Cf[i] = i < Cin ? p.coefA[(int64_t)img * Cin + i] : p.coefB[(int64_t)img * Cin + (i - Cin)];
    }
    F4_LDS_BARRIER();
    store_raw(0);
    F4_LDS_BARRIER();
    DLPM_PHASE(p, 8);

#pragma unroll 1
    for (int chunk = 0; chunk < nch; chunk++) {
        transform();                                   // raw(chunk) -> V
        // (the weight ring is refilled per chunk, BEHIND the transform: kept across it, its 20 registers beside the 36 accumulators
        //  and the transform's patch rows were scratch memory; the fragments are L2-resident and land during the barrier)
#pragma unroll
        for (int a = 0; a < AHEAD; a++) bq[a] = wp[a * 64 + lane];
        load_raw(min(chunk + 1, last));                // in flight across the barrier and the MFMAs
        F4_LDS_BARRIER();                              // V complete, raw free
        float4 aq[2];
        aq[0] = *reinterpret_cast<const float4 *>(asrc);
#pragma unroll
        for (int pp = 0; pp < 18; pp++) {
            if (pp + AHEAD < 18) bq[(pp + AHEAD) % F4_RING] = wp[(pp + AHEAD) * 64 + lane];
            if (pp + 1 < 18) aq[(pp + 1) & 1] = *reinterpret_cast<const float4 *>(asrc + (pp + 1) * FI_VPP);
            const float4 aa = aq[pp & 1];
            const float4 b = bq[pp % F4_RING];
            acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.x, b.x, acc[2 * pp], 0, 0, 0);
            acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.z, b.z, acc[2 * pp + 1], 0, 0, 0);
            acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.y, b.y, acc[2 * pp], 0, 0, 0);
            acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.w, b.w, acc[2 * pp + 1], 0, 0, 0);
        }
        wp += 18 * 64;
        if (chunk < last) store_raw(chunk + 1);
        F4_LDS_BARRIER();                              // raw(chunk + 1) complete, V free
    }
    DLPM_PHASE(p, 9);

    // ---- epilogue from registers (the 2 + 2-wave shape's: lane = channel, 4 tiles x 16 pixels; all four tiles' residual values are
    // requested before the first output transform; statistics = one partial per 256-pixel quadrant)
    {
        const int ty0 = 4 * (mb >> 1), tx0 = 4 * (mb & 1);
        const int64_t pix0 = ((int64_t)img * H + 4 * ty0) * W + 4 * tx0;      // wave-uniform
        const int ch = 16 * nw + li;
        const float bias_v = p.bias ? p.bias[ch] : 0.f;
        const bool has_res = p.res0 != nullptr;
        const bool res_first = 16 * nw < p.R0;   // R0 % 16 == 0 (wino4_geometry)
        const float *res_u = has_res ? (res_first ? p.res0 : p.res1 - p.R0) : nullptr;
        const int res_ld = res_first ? p.R0 : p.Cout - p.R0;
        float *__restrict__ out_blk = p.out + pix0 * p.Cout;
        const float *__restrict__ res_blk = has_res ? res_u + pix0 * res_ld : nullptr;
        const bool do_stats = p.stats_out != nullptr;
        float K = 0.f, s1 = 0.f, s2 = 0.f;
        float rs_all[4][16] = {};
        if (has_res) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int t2 = 4 * lk + rr;
                const int tp2 = (4 * (t2 >> 2)) * W + 4 * (t2 & 3);
                const uint32_t b2 = (uint32_t)(tp2 * res_ld + ch) * 4u;
#pragma unroll
                for (int k = 0; k < 16; k++)
                    rs_all[rr][k] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(res_blk + ((k >> 2) * W + (k & 3)) * res_ld) + b2);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int tile = 4 * lk + r;
            const int tpix = (4 * (tile >> 2)) * W + 4 * (tile & 3);
            const uint32_t bo_o = (uint32_t)(tpix * p.Cout + ch) * 4u;
            float Yt[16];
            float Z[4][6];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const float m0 = acc[0 * 6 + b][r], m1 = acc[1 * 6 + b][r], m2 = acc[2 * 6 + b][r];
                const float m3 = acc[3 * 6 + b][r], m4 = acc[4 * 6 + b][r], m5 = acc[5 * 6 + b][r];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                Z[0][b] = m0 + s12 + s34;
                Z[1][b] = fmaf(PB, d34, PA * d12);
                Z[2][b] = fmaf(PB2, s34, PA2 * s12);
                Z[3][b] = fmaf(PB3, d34, PA3 * d12) + m5;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float s12 = Z[i][1] + Z[i][2], d12 = Z[i][1] - Z[i][2], s34 = Z[i][3] + Z[i][4], d34 = Z[i][3] - Z[i][4];
                Yt[i * 4 + 0] = Z[i][0] + s12 + s34;
                Yt[i * 4 + 1] = fmaf(PB, d34, PA * d12);
                Yt[i * 4 + 2] = fmaf(PB2, s34, PA2 * s12);
                Yt[i * 4 + 3] = fmaf(PB3, d34, PA3 * d12) + Z[i][5];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float v = Yt[i * 4 + j] + bias_v;
                    if (has_res) v += rs_all[r][i * 4 + j];
                    if (do_stats) {
                        if (r == 0 && i == 0 && j == 0) K = v;
                        const float dd = v - K;
                        s1 += dd;
                        s2 = fmaf(dd, dd, s2);
                    }
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(out_blk + (i * W + j) * p.Cout) + bo_o) = v;
                }
            }
        }
        if (do_stats) {
            float mean = K + s1 * (1.f / 64.f), M2 = fmaxf(s2 - s1 * s1 * (1.f / 64.f), 0.f), na = 64.f;
#pragma unroll
            for (int sft = 16; sft <= 32; sft <<= 1) {
                const float om = __shfl_xor(mean, sft), oM2 = __shfl_xor(M2, sft);
                const float lo_m = (lane & sft) ? om : mean, hi_m = (lane & sft) ? mean : om;
                const float lo_M = (lane & sft) ? oM2 : M2, hi_M = (lane & sft) ? M2 : oM2;
                const float dd = hi_m - lo_m;
                mean = lo_m + dd * 0.5f;
                M2 = lo_M + hi_M + dd * dd * (na * 0.5f);
                na *= 2.f;
            }
            if (lk == 0) p.stats_out[((int64_t)img * 4 + mb) * p.Cout + ch] = make_float2(mean, M2);
        }
    }
    DLPM_PHASE(p, 10);
    DLPM_PHASE_FLUSH(p, 8);
#ifdef DLPM_PHASE_TIMING
    if (p.phase && tid == 0) {
        atomicAdd(p.phase + 11, 1ull);
        atomicAdd(p.phase + 12, (unsigned long long)(clock64() - _c0));
        atomicAdd(p.phase + 13, (unsigned long long)(wall_clock64() - _r0));
    }
#endif
}

// =====================================================================================================================================
// Round 6 -- a WHOLE ResBlock (dlpm/models/unet.py:105-196, use_scale_shift_norm) with 32 output channels on 32x32 images in ONE launch:
//     h = conv3x3(silu(GN1(x)));   a = silu(GN2(h) (1 + scale) + shift);   out = conv3x3(a) + skip
// Two passes of k_conv3x3_wino4_img's chunk loop in one workgroup per image.  Between them the workgroup -- which holds the whole image --
// does what the launches in between did: the four waves of a channel half leave their 256-pixel (mean, M2) partials in LDS, the
// coefficient arithmetic of k_gn_coeffs_stats runs on them (gn_stats.h: the same function), every lane activates its 64 values of h in
// registers and writes them ONCE, activated, to a scratch image that the second pass stages like any input (no coefficients, no SiLU
// there: a plain copy with zero padding).  That image is 128 KB per workgroup and is read back by the CU that wrote it: an L2 round
// trip.  GroupNorm-1's coefficients come from the producers' statistics inside the prologue (same function) when they exist.
// Every value is computed by the expressions of the separate launches (conv -> k_gn_coeffs_stats -> conv), so the block's output
// and its statistics partials are bit-identical to that path (tests/test_gpu_kernels.py::test_resblock_whole_image_*).
__global__ void __launch_bounds__(FI_NT, 1) k_resblock_wino4_img(ResImgLaunch p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *V = wsm;                      // [18 position pairs][4 channel pairs][64 tiles][2 pos x 2 ch]
    float *raw = V + FI_VBUF;            // [34 x 34][F4_PRLD] (+ skew)
    float *Cf = raw + FI_RAWBUF;         // [A | B][Cin <= 128]
    float *gsh = Cf + 256;               // GroupNorm scratch: 2 C + 2 G floats
    float2 *part = reinterpret_cast<float2 *>(gsh + 512);   // [4 quadrants][32]: conv1's statistics partials
    int *itmL = reinterpret_cast<int *>(gsh + 768);         // [5][512] staging items (in registers they were scratch memory beside 36 accumulators)
    DLPM_PHASE_DECL;
    constexpr int H = 32, W = 32, CO = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int mb = wave >> 1, nw = wave & 1;
    const int img = blockIdx.x;
    const int ch = 16 * nw + li;

    const int squad = tid & 1;
#pragma unroll
    for (int it = 0; it < FI_QN; it++) {
        const int pix = it * (FI_NT / 2) + (tid >> 1);
        const int ry = pix / FI_RW, rx = pix - ry * FI_RW;
        const int iy = ry - 1, ix = rx - 1;
        const bool pad = iy < 0 || iy >= H || ix < 0 || ix >= W;
        const int lo = pix * F4_PRLD + squad * 4 + 4 * (ry >> 2);
        itmL[it * FI_NT + tid] = pix >= FI_NPIX ? (1 << 25) : (lo | (pad ? (1 << 24) : ((iy * W + ix) << 14)));   // (read back by this thread only)
    }
    // ---- the source of the pass in flight (pass 1: x0 | x1 with GroupNorm-1 + SiLU; pass 2: the activated scratch image, plain)
    const float *s0 = p.x0, *s1 = p.x1;
    int c0 = p.C0, c1 = p.C1;
    float4 xr[FI_QN];
    auto load_raw = [&](int chunk) {
        const int c = chunk * F4_KC + squad * 4;
        const bool first = c < c0;
        const int ld = first ? c0 : c1;
        const float *sb = (first ? s0 + c : s1 + (c - c0)) + (int64_t)img * (H * W) * ld;
#pragma unroll
        for (int it = 0; it < FI_QN; it++) xr[it] = *reinterpret_cast<const float4 *>(sb + ((itmL[it * FI_NT + tid] >> 14) & 1023) * ld);
    };
    auto store_raw = [&](int chunk, auto act_c) {
        constexpr bool ACT = decltype(act_c)::value;      // pass 1: x A + B, SiLU; pass 2: plain copy
        const int c = chunk * F4_KC + squad * 4;
        float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ACT) {
            ca = *reinterpret_cast<const float4 *>(Cf + c);
            cb = *reinterpret_cast<const float4 *>(Cf + 128 + c);
        }
#pragma unroll
        for (int it = 0; it < FI_QN; it++) {
            const int iv = itmL[it * FI_NT + tid];
            if (iv & (1 << 25)) continue;
            float4 x = xr[it];
            if (ACT) {
                x.x = silu_f(fmaf(x.x, ca.x, cb.x));
                x.y = silu_f(fmaf(x.y, ca.y, cb.y));
                x.z = silu_f(fmaf(x.z, ca.z, cb.z));
                x.w = silu_f(fmaf(x.w, ca.w, cb.w));
            }
            if (iv & (1 << 24)) x = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(raw + (iv & 16383)) = x;
        }
    };
    int rbase, vofs;
    {
        const int tg = wave & 3, tile = lane & 15, pair = lane >> 4;
        const int ty = 4 * (tg >> 1) + (tile >> 2), tx = 4 * (tg & 1) + (tile & 3);
        rbase = (4 * ty * FI_RW + 4 * tx) * F4_PRLD + pair * 2 + 4 * ty;
        vofs = (pair * FI_TILES + 16 * tg + tile) * 4;
    }
    const int th = wave >> 2;
    auto transform = [&]() {
        const float *rb = raw + rbase;
        float *vb = V + vofs;
        auto d = [&](int i, int c) {
            return *reinterpret_cast<const float2 *>(rb + (i * FI_RW + c) * F4_PRLD + (i >= 4 ? 4 : 0));
        };
        auto row_out = [&](const float2 (&T)[6], int a) {
            float *vr = vb + a * 3 * FI_VPP;
            const float2 e1 = f2fma(-PB2, T[2], T[4]), o1 = f2fma(-PB2, T[1], T[3]);
            const float2 e2 = f2fma(-PA2, T[2], T[4]), o2 = f2fma(-PA2, T[1], T[3]);
            const float2 v0 = f2fma(PP2, T[0], f2fma(-PS2, T[2], T[4])), v1 = f2fma(PA, o1, e1), v2 = f2fma(-PA, o1, e1);
            const float2 v3 = f2fma(PB, o2, e2), v4 = f2fma(-PB, o2, e2), v5 = f2fma(PP2, T[1], f2fma(-PS2, T[3], T[5]));
            *reinterpret_cast<float4 *>(vr + 0 * FI_VPP) = make_float4(v0.x, v0.y, v1.x, v1.y);
            *reinterpret_cast<float4 *>(vr + 1 * FI_VPP) = make_float4(v2.x, v2.y, v3.x, v3.y);
            *reinterpret_cast<float4 *>(vr + 2 * FI_VPP) = make_float4(v4.x, v4.y, v5.x, v5.y);
        };
        float2 Ta[6], Tb[6];
        if (th == 0) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const float2 e = f2fma(-PB2, d(2, c), d(4, c)), o = f2fma(-PB2, d(1, c), d(3, c));
                Ta[c] = f2fma(PA, o, e);
                Tb[c] = f2fma(-PA, o, e);
            }
            row_out(Ta, 1);
            row_out(Tb, 2);
#pragma unroll
            for (int c = 0; c < 6; c++) Ta[c] = f2fma(PP2, d(0, c), f2fma(-PS2, d(2, c), d(4, c)));
            row_out(Ta, 0);
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const float2 e = f2fma(-PA2, d(2, c), d(4, c)), o = f2fma(-PA2, d(1, c), d(3, c));
                Ta[c] = f2fma(PB, o, e);
                Tb[c] = f2fma(-PB, o, e);
            }
            row_out(Ta, 3);
            row_out(Tb, 4);
#pragma unroll
            for (int c = 0; c < 6; c++) Tb[c] = f2fma(PP2, d(1, c), f2fma(-PS2, d(3, c), d(5, c)));
            row_out(Tb, 5);
        }
    };
    constexpr int AHEAD = F4_RING - 1;
    const float *asrc = V + (lk * FI_TILES + 16 * mb + li) * 4;
    floatx4 acc[36];
    // one convolution: chunk 0 is in xr on entry (load_raw(0) issued by the caller); acc is zeroed here
    auto conv_pass = [&](const float *wfrag, int nch, auto act_c) {
        const int last = nch - 1;
        const float4 *__restrict__ wp = reinterpret_cast<const float4 *>(wfrag) + (int64_t)nw * nch * 18 * 64;
        float4 bq[F4_RING];
#pragma unroll
        for (int q = 0; q < 36; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[q][r] = 0.f;
        store_raw(0, act_c);
        F4_LDS_BARRIER();
#pragma unroll 1
        for (int chunk = 0; chunk < nch; chunk++) {
            transform();
#pragma unroll
            for (int a = 0; a < AHEAD; a++) bq[a] = wp[a * 64 + lane];
            load_raw(min(chunk + 1, last));
            F4_LDS_BARRIER();
            float4 aq[2];
            aq[0] = *reinterpret_cast<const float4 *>(asrc);
#pragma unroll
            for (int pp = 0; pp < 18; pp++) {
                if (pp + AHEAD < 18) bq[(pp + AHEAD) % F4_RING] = wp[(pp + AHEAD) * 64 + lane];
                if (pp + 1 < 18) aq[(pp + 1) & 1] = *reinterpret_cast<const float4 *>(asrc + (pp + 1) * FI_VPP);
                const float4 aa = aq[pp & 1];
                const float4 b = bq[pp % F4_RING];
                acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.x, b.x, acc[2 * pp], 0, 0, 0);
                acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.z, b.z, acc[2 * pp + 1], 0, 0, 0);
                acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.y, b.y, acc[2 * pp], 0, 0, 0);
                acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.w, b.w, acc[2 * pp + 1], 0, 0, 0);
            }
            wp += 18 * 64;
            if (chunk < last) store_raw(chunk + 1, act_c);
            F4_LDS_BARRIER();
        }
    };
    // Y = A^T M A of tile r of this lane (the 2 + 2-wave shape's expressions)
    auto out_tile = [&](int r, float (&Yt)[16]) {
        float Z[4][6];
#pragma unroll
        for (int b = 0; b < 6; b++) {
            const float m0 = acc[0 * 6 + b][r], m1 = acc[1 * 6 + b][r], m2 = acc[2 * 6 + b][r];
            const float m3 = acc[3 * 6 + b][r], m4 = acc[4 * 6 + b][r], m5 = acc[5 * 6 + b][r];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            Z[0][b] = m0 + s12 + s34;
            Z[1][b] = fmaf(PB, d34, PA * d12);
            Z[2][b] = fmaf(PB2, s34, PA2 * s12);
            Z[3][b] = fmaf(PB3, d34, PA3 * d12) + m5;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float s12 = Z[i][1] + Z[i][2], d12 = Z[i][1] - Z[i][2], s34 = Z[i][3] + Z[i][4], d34 = Z[i][3] - Z[i][4];
            Yt[i * 4 + 0] = Z[i][0] + s12 + s34;
            Yt[i * 4 + 1] = fmaf(PB, d34, PA * d12);
            Yt[i * 4 + 2] = fmaf(PB2, s34, PA2 * s12);
            Yt[i * 4 + 3] = fmaf(PB3, d34, PA3 * d12) + Z[i][5];
        }
    };
    // the quadrant's statistics partial from the per-lane shifted sums (the convolution kernels' epilogue): lanes lk == 0 hold it
    auto quadrant_stats = [&](float K, float s1, float s2, float &mean, float &M2) {
        mean = K + s1 * (1.f / 64.f);
        M2 = fmaxf(s2 - s1 * s1 * (1.f / 64.f), 0.f);
        float na = 64.f;
#pragma unroll
        for (int sft = 16; sft <= 32; sft <<= 1) {
            const float om = __shfl_xor(mean, sft), oM2 = __shfl_xor(M2, sft);
            const float lo_m = (lane & sft) ? om : mean, hi_m = (lane & sft) ? mean : om;
            const float lo_M = (lane & sft) ? oM2 : M2, hi_M = (lane & sft) ? M2 : oM2;
            const float dd = hi_m - lo_m;
            mean = lo_m + dd * 0.5f;
            M2 = lo_M + hi_M + dd * dd * (na * 0.5f);
            na *= 2.f;
        }
    };
    const int ty0 = 4 * (mb >> 1), tx0 = 4 * (mb & 1);
    const int64_t pix0 = ((int64_t)img * H + 4 * ty0) * W + 4 * tx0;      // wave-uniform: this quadrant's first pixel

    // ================= pass 1: h = conv1(silu(GN1(x)))
    const int Cin = p.C0 + p.C1;
    load_raw(0);
    if (p.st0) {
        gn_coeffs_from_stats_image(p.st0 + (int64_t)img * p.nt0 * p.C0, p.st1 ? p.st1 + (int64_t)img * p.nt1 * p.C1 : nullptr, p.C0, p.C1, p.nt0,
                                   p.nt1, H * W, Cin < 32 ? Cin : 32, p.gn1_w, p.gn1_b, nullptr, gsh, Cf, Cf + 128, 1e-5f, tid, FI_NT,
                                   [] { F4_LDS_BARRIER(); });
    } else {
        for (int i = tid; i < 2 * Cin; i += FI_NT) {
            if (i < Cin) Cf[i] = p.coefA1[(int64_t)img * Cin + i];
            else Cf[128 + (i - Cin)] = p.coefB1[(int64_t)img * Cin + (i - Cin)];
        }
    }
    F4_LDS_BARRIER();
    conv_pass(p.w1, Cin / F4_KC, std::integral_constant<bool, true>());
    DLPM_PHASE(p, 8);

    // ================= between: h + bias (registers), its statistics, GroupNorm-2 with scale-shift, a = silu(..) -> scratch image
    {
        int chv = ch;
        asm volatile("" : "+v"(chv));      // (opaque: the section's addresses must not be formed -- and kept -- in front of the chunk loop)
        const float bias_v = p.b1[chv];
        float hv[4][16];
        float K = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float Yt[16];
            out_tile(r, Yt);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float v = Yt[k] + bias_v;
                if (r == 0 && k == 0) K = v;
                const float dd = v - K;
                s1 += dd;
                s2 = fmaf(dd, dd, s2);
                hv[r][k] = v;
            }
        }
        float mean, M2;
        quadrant_stats(K, s1, s2, mean, M2);
        if (lk == 0) part[mb * CO + chv] = make_float2(mean, M2);
        F4_LDS_BARRIER();
        gn_coeffs_from_stats_image(part, nullptr, CO, 0, 4, 1, H * W, 32, p.gn2_w, p.gn2_b, p.emb + (int64_t)img * p.emb_stride + p.emb_off, gsh, Cf,
                                   Cf + 128, 1e-5f, tid, FI_NT, [] { F4_LDS_BARRIER(); });
        F4_LDS_BARRIER();
        const float a2 = Cf[chv], b2 = Cf[128 + chv];
        float *hb = p.hbuf + pix0 * CO;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int tile = 4 * lk + r;
            const int tpix = (4 * (tile >> 2)) * W + 4 * (tile & 3);
#pragma unroll
            for (int k = 0; k < 16; k++) hb[(tpix + (k >> 2) * W + (k & 3)) * CO + chv] = silu_f(fmaf(hv[r][k], a2, b2));
        }
    }
    // the scratch image is read back by other waves of THIS workgroup only: __syncthreads() is a workgroup-scope release / acquire over global
    // memory as well (s_waitcnt vmcnt(0) + barrier; the CU's L1 is write-through and shared by the workgroup).  An agent-scope
    // __threadfence() here writes the XCD's whole dirty L2 back, with 256 workgroups doing so at once: 190 us per launch instead of 70.
    __syncthreads();

    // ================= pass 2: out = conv2(a) + bias + skip
    s0 = p.hbuf; s1 = nullptr; c0 = CO; c1 = 0;
    load_raw(0);
    conv_pass(p.w2, CO / F4_KC, std::integral_constant<bool, false>());
    DLPM_PHASE(p, 9);
    {
        int chv = ch, lkv = lk;
        asm volatile("" : "+v"(chv), "+v"(lkv));      // (opaque: else the 64 pixel offsets of the section above stay live -- in scratch -- across pass 2)
        const float bias_v = p.b2[chv];
        float *__restrict__ out_blk = p.out + pix0 * CO;
        const float *__restrict__ res_blk = p.res + pix0 * CO;
        const bool do_stats = p.stats_out != nullptr;
        float K = 0.f, s1 = 0.f, s2 = 0.f;
        // the residual of tile r + 1 is requested before tile r's output transform (all four at once, as the convolution kernels
        // do, is 64 registers beside the 36 accumulators: scratch memory here)
        float rs[2][16];
        auto load_res = [&](int rr, float (&dst)[16]) {
            const int t2 = 4 * lkv + rr;
            const int tp2 = (4 * (t2 >> 2)) * W + 4 * (t2 & 3);
#pragma unroll
            for (int k = 0; k < 16; k++) dst[k] = res_blk[(tp2 + (k >> 2) * W + (k & 3)) * CO + chv];
        };
        load_res(0, rs[0]);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int tile = 4 * lkv + r;
            const int tpix = (4 * (tile >> 2)) * W + 4 * (tile & 3);
            if (r + 1 < 4) load_res(r + 1, rs[(r + 1) & 1]);
            float Yt[16];
            out_tile(r, Yt);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                float v = Yt[k] + bias_v;
                v += rs[r & 1][k];
                if (do_stats) {
                    if (r == 0 && k == 0) K = v;
                    const float dd = v - K;
                    s1 += dd;
                    s2 = fmaf(dd, dd, s2);
                }
                out_blk[(tpix + (k >> 2) * W + (k & 3)) * CO + chv] = v;
            }
        }
        if (do_stats) {
            float mean, M2;
            quadrant_stats(K, s1, s2, mean, M2);
            if (lkv == 0) p.stats_out[((int64_t)img * 4 + mb) * CO + chv] = make_float2(mean, M2);
        }
    }
    DLPM_PHASE(p, 10);
    DLPM_PHASE_FLUSH(p, 8);
#ifdef DLPM_PHASE_TIMING
    if (p.phase && tid == 0) atomicAdd(p.phase + 11, 1ull);
#endif
}

// =====================================================================================================================================
// Round 6 -- the same for 64 output channels on 16x16 images (the MNIST-sized nets' second level: five ResBlocks per step, each two
// 34-us launches of the 4 + 3-wave shape + two coefficient launches).  An image is ONE 16-tile M-block, so the matrix work has four
// 16-channel roles; the eight waves are those four x two K HALVES: wave (nw, kh) runs the MFMAs of the 8-channel phases 2 c + kh of
// every 16-channel chunk c on a full set of 36 accumulators, and at the end of a pass the halves meet through LDS -- each wave hands its
// partner the two tiles the partner finalises and keeps the other two, so the output transform, the GroupNorm statistics and the
// epilogue run on all eight waves.  Per chunk: waves 0..3 transform (two row groups x two channel-pair groups), barrier, every wave
// its 72 MFMAs with the next chunk's loads in flight, staging, barrier.  The intermediate h never leaves the CU: activated, it goes
// to an LDS image [256][68] from which pass 2 stages its chunks (LDS -> LDS, zero padding applied there).  The sums of the two K halves
// are added once per output, so this kernel rounds differently from the 4 + 3-wave launches it replaces (5e-6 of the reference's
// ResBlock, tests/golden/f14_blocks16.npz; bits independent of the batch: an image is a workgroup).
constexpr int F6_TILES = 16, F6_RW = 18, F6_NPIX = F6_RW * F6_RW, F6_KC = 16, F6_PRLD = F6_KC + 4;
constexpr int F6_QN = (F6_NPIX * 4 + FI_NT - 1) / FI_NT;        // staging items (pixel, channel quad of the chunk) per thread: 3
constexpr int F6_RAWBUF = F6_NPIX * F6_PRLD + 32;
constexpr int F6_VPP = 8 * F6_TILES * 4;                         // floats per position pair: [8 channel pairs][16 tiles][4]
constexpr int F6_VBUF = 18 * F6_VPP;
constexpr int F6_HLD = 64 + 4, F6_HBUF = 256 * F6_HLD;
constexpr int F6_XCH = 8 * 36 * 64;                              // floats of the K-half exchange buffer (one round: 18 accumulator tiles x 2 components per wave)
static_assert(F6_XCH <= F6_VBUF + F6_RAWBUF + F6_HBUF, "the exchange buffer aliases V, raw and the (then dead) intermediate image");

__global__ void __launch_bounds__(FI_NT, 1) k_resblock_wino4_img16(ResImgLaunch p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *V = wsm;                      // [18 position pairs][8 channel pairs][16 tiles][2 pos x 2 ch]
    float *raw = V + F6_VBUF;            // [18 x 18][F6_PRLD] (+ skew)
    float *hL = raw + F6_RAWBUF;         // [256][F6_HLD] silu(GN2(h) (1 + scale) + shift)
    float *Cf = hL + F6_HBUF;            // [A | B][Cin <= 128]
    float *gsh = Cf + 256;               // GroupNorm scratch (2 C + 2 G floats)
    float2 *part = reinterpret_cast<float2 *>(gsh + 512);   // [2 K halves][64]: statistics partials (128 pixels each)
    int *itmL = reinterpret_cast<int *>(gsh + 768);         // [3][512] staging items
    float *xb = wsm;                     // exchange buffer (aliases V / raw / hL between the passes' loops and their register sections)
    DLPM_PHASE_DECL;
    constexpr int H = 16, W = 16, CO = 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int nw = wave & 3, kh = wave >> 2;         // MFMA role: 16-channel output tile, K half
    const int img = blockIdx.x;
    const int ch = 16 * nw + li;

    // staging item e = it * 512 + tid: (halo pixel e >> 2, channel quad e & 3 of the 16-channel chunk)
#pragma unroll
    for (int it = 0; it < F6_QN; it++) {
        const int e = it * FI_NT + tid, pix = e >> 2, quad = e & 3;
        const int ry = pix / F6_RW, rx = pix - ry * F6_RW;
        const int iy = ry - 1, ix = rx - 1;
        const bool pad = iy < 0 || iy >= H || ix < 0 || ix >= W;
        const int lo = pix * F6_PRLD + quad * 4 + 4 * (ry >> 2);
        itmL[it * FI_NT + tid] = pix >= F6_NPIX ? (1 << 25) : (lo | (pad ? (1 << 24) : ((iy * W + ix) << 14)));
    }
    const int quad = tid & 3;
    const float *s0 = p.x0, *s1 = p.x1;
    int c0 = p.C0, c1 = p.C1;
    float4 xr[F6_QN];
    auto load_raw = [&](int chunk) __attribute__((always_inline)) {                 // pass 1: global -> registers
        const int c = chunk * F6_KC + quad * 4;
        const bool first = c < c0;
        const int ld = first ? c0 : c1;
        const float *sb = (first ? s0 + c : s1 + (c - c0)) + (int64_t)img * (H * W) * ld;
#pragma unroll
        for (int it = 0; it < F6_QN; it++) xr[it] = *reinterpret_cast<const float4 *>(sb + ((itmL[it * FI_NT + tid] >> 14) & 255) * ld);
    };
    auto store_raw = [&](int chunk, auto first_pass) __attribute__((always_inline)) {
        constexpr bool P1 = decltype(first_pass)::value;   // pass 1: x A + B, SiLU from registers; pass 2: copy from the LDS image
        const int c = chunk * F6_KC + quad * 4;
        float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P1) {
            ca = *reinterpret_cast<const float4 *>(Cf + c);
            cb = *reinterpret_cast<const float4 *>(Cf + 128 + c);
        }
#pragma unroll
        for (int it = 0; it < F6_QN; it++) {
            const int iv = itmL[it * FI_NT + tid];
            if (iv & (1 << 25)) continue;
            float4 x;
            if (P1) {
                x = xr[it];
                x.x = silu_f(fmaf(x.x, ca.x, cb.x));
                x.y = silu_f(fmaf(x.y, ca.y, cb.y));
                x.z = silu_f(fmaf(x.z, ca.z, cb.z));
                x.w = silu_f(fmaf(x.w, ca.w, cb.w));
            } else {
                x = *reinterpret_cast<const float4 *>(hL + ((iv >> 14) & 255) * F6_HLD + c);
            }
            if (iv & (1 << 24)) x = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(raw + (iv & 16383)) = x;
        }
    };
    // input transform: wave w < 4 = (row group th = w & 1, channel-pair group cp = w >> 1); lane = (pair of the group, tile)
    int rbase, vofs;
    {
        const int cp = (wave >> 1) & 1, tile = lane & 15, pair = 4 * cp + (lane >> 4);
        const int ty = tile >> 2, tx = tile & 3;
        rbase = (4 * ty * F6_RW + 4 * tx) * F6_PRLD + pair * 2 + 4 * ty;
        vofs = (pair * F6_TILES + tile) * 4;
    }
    const int th = wave & 1;
    auto transform = [&]() __attribute__((always_inline)) {
        const float *rb = raw + rbase;
        float *vb = V + vofs;
        auto d = [&](int i, int c) {
            return *reinterpret_cast<const float2 *>(rb + (i * F6_RW + c) * F6_PRLD + (i >= 4 ? 4 : 0));
        };
        auto row_out = [&](const float2 (&T)[6], int a) {
            float *vr = vb + a * 3 * F6_VPP;
            const float2 e1 = f2fma(-PB2, T[2], T[4]), o1 = f2fma(-PB2, T[1], T[3]);
            const float2 e2 = f2fma(-PA2, T[2], T[4]), o2 = f2fma(-PA2, T[1], T[3]);
            const float2 v0 = f2fma(PP2, T[0], f2fma(-PS2, T[2], T[4])), v1 = f2fma(PA, o1, e1), v2 = f2fma(-PA, o1, e1);
            const float2 v3 = f2fma(PB, o2, e2), v4 = f2fma(-PB, o2, e2), v5 = f2fma(PP2, T[1], f2fma(-PS2, T[3], T[5]));
            *reinterpret_cast<float4 *>(vr + 0 * F6_VPP) = make_float4(v0.x, v0.y, v1.x, v1.y);
            *reinterpret_cast<float4 *>(vr + 1 * F6_VPP) = make_float4(v2.x, v2.y, v3.x, v3.y);
            *reinterpret_cast<float4 *>(vr + 2 * F6_VPP) = make_float4(v4.x, v4.y, v5.x, v5.y);
        };
        float2 Ta[6], Tb[6];
        if (th == 0) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const float2 e = f2fma(-PB2, d(2, c), d(4, c)), o = f2fma(-PB2, d(1, c), d(3, c));
                Ta[c] = f2fma(PA, o, e);
                Tb[c] = f2fma(-PA, o, e);
            }
            row_out(Ta, 1);
            row_out(Tb, 2);
#pragma unroll
            for (int c = 0; c < 6; c++) Ta[c] = f2fma(PP2, d(0, c), f2fma(-PS2, d(2, c), d(4, c)));
            row_out(Ta, 0);
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const float2 e = f2fma(-PA2, d(2, c), d(4, c)), o = f2fma(-PA2, d(1, c), d(3, c));
                Ta[c] = f2fma(PB, o, e);
                Tb[c] = f2fma(-PB, o, e);
            }
            row_out(Ta, 3);
            row_out(Tb, 4);
#pragma unroll
            for (int c = 0; c < 6; c++) Tb[c] = f2fma(PP2, d(1, c), f2fma(-PS2, d(3, c), d(5, c)));
            row_out(Tb, 5);
        }
    };
    constexpr int AHEAD = F4_RING - 1;
    const float *asrc = V + ((4 * kh + lk) * F6_TILES + li) * 4;     // this K half's channel pairs of the chunk
    floatx4 acc[36];
    // one convolution over nch16 chunks of 16 channels; pass 1 has chunk 0 in xr on entry
    auto conv_pass = [&](const float *wfrag, int nch16, auto first_pass) __attribute__((always_inline)) {
        constexpr bool P1 = decltype(first_pass)::value;
        const int last = nch16 - 1;
        // Wf[ntile 0][wave nw][8-channel phase][18][lane][4]: this wave's phases are 2 c + kh
        const float4 *__restrict__ wp = reinterpret_cast<const float4 *>(wfrag) + ((int64_t)nw * (2 * nch16) + kh) * 18 * 64;
        float4 bq[F4_RING];
#pragma unroll
        for (int q = 0; q < 36; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[q][r] = 0.f;
        store_raw(0, first_pass);
        F4_LDS_BARRIER();
#pragma unroll 1
        for (int chunk = 0; chunk < nch16; chunk++) {
            if (wave < 4) transform();
#pragma unroll
            for (int a = 0; a < AHEAD; a++) bq[a] = wp[a * 64 + lane];
            if (P1) load_raw(min(chunk + 1, last));
            F4_LDS_BARRIER();
            float4 aq[2];
            aq[0] = *reinterpret_cast<const float4 *>(asrc);
#pragma unroll
            for (int pp = 0; pp < 18; pp++) {
                if (pp + AHEAD < 18) bq[(pp + AHEAD) % F4_RING] = wp[(pp + AHEAD) * 64 + lane];
                if (pp + 1 < 18) aq[(pp + 1) & 1] = *reinterpret_cast<const float4 *>(asrc + (pp + 1) * F6_VPP);
                const float4 aa = aq[pp & 1];
                const float4 b = bq[pp % F4_RING];
                acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.x, b.x, acc[2 * pp], 0, 0, 0);
                acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.z, b.z, acc[2 * pp + 1], 0, 0, 0);
                acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.y, b.y, acc[2 * pp], 0, 0, 0);
                acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.w, b.w, acc[2 * pp + 1], 0, 0, 0);
            }
            wp += 2 * 18 * 64;
            if (chunk < last) store_raw(chunk + 1, first_pass);
            F4_LDS_BARRIER();
        }
    };
    // the two K halves meet: wave (nw, kh) keeps tiles r = 2 kh, 2 kh + 1 of its lanes and receives the partner's sums for them (two
    // rounds of 18 accumulator tiles through xb); afterwards acc[q][RA], acc[q][RA + 1] hold the complete sums
    auto exchange = [&](auto ra_c) __attribute__((always_inline)) {
        constexpr int RA = decltype(ra_c)::value, RO = 2 - RA;     // mine, the partner's
        float2 *xb2 = reinterpret_cast<float2 *>(xb);      // [wave][18][lane] pairs of adjacent accumulator components: 8-byte LDS accesses
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
#pragma unroll
            for (int q = 0; q < 18; q++) xb2[(wave * 18 + q) * 64 + lane] = make_float2(acc[18 * hh + q][RO], acc[18 * hh + q][RO + 1]);
            F4_LDS_BARRIER();
#pragma unroll
            for (int q0 = 0; q0 < 18; q0 += 6) {      // six at a time: all 18 in flight at once are 36 more live registers beside the accumulators
                float2 t[6];
#pragma unroll
                for (int q = 0; q < 6; q++) t[q] = xb2[((wave ^ 4) * 18 + q0 + q) * 64 + lane];
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    acc[18 * hh + q0 + q][RA] += t[q].x;
                    acc[18 * hh + q0 + q][RA + 1] += t[q].y;
                }
                asm volatile("" ::: "memory");
            }
            F4_LDS_BARRIER();
        }
    };
    auto out_tile = [&](auto r_c, float (&Yt)[16]) __attribute__((always_inline)) {
        constexpr int r = decltype(r_c)::value;
        float Z[4][6];
#pragma unroll
        for (int b = 0; b < 6; b++) {
            const float m0 = acc[0 * 6 + b][r], m1 = acc[1 * 6 + b][r], m2 = acc[2 * 6 + b][r];
            const float m3 = acc[3 * 6 + b][r], m4 = acc[4 * 6 + b][r], m5 = acc[5 * 6 + b][r];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            Z[0][b] = m0 + s12 + s34;
            Z[1][b] = fmaf(PB, d34, PA * d12);
            Z[2][b] = fmaf(PB2, s34, PA2 * s12);
            Z[3][b] = fmaf(PB3, d34, PA3 * d12) + m5;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float s12 = Z[i][1] + Z[i][2], d12 = Z[i][1] - Z[i][2], s34 = Z[i][3] + Z[i][4], d34 = Z[i][3] - Z[i][4];
            Yt[i * 4 + 0] = Z[i][0] + s12 + s34;
            Yt[i * 4 + 1] = fmaf(PB, d34, PA * d12);
            Yt[i * 4 + 2] = fmaf(PB2, s34, PA2 * s12);
            Yt[i * 4 + 3] = fmaf(PB3, d34, PA3 * d12) + Z[i][5];
        }
    };
    // (mean, M2) of this wave's 128 pixels of its channel from the per-lane shifted sums over 32 values; every lane ends up with it
    auto half_stats = [&](float K, float s1, float s2, float &mean, float &M2) __attribute__((always_inline)) {
        mean = K + s1 * (1.f / 32.f);
        M2 = fmaxf(s2 - s1 * s1 * (1.f / 32.f), 0.f);
        float na = 32.f;
#pragma unroll
        for (int sft = 16; sft <= 32; sft <<= 1) {
            const float om = __shfl_xor(mean, sft), oM2 = __shfl_xor(M2, sft);
            const float lo_m = (lane & sft) ? om : mean, hi_m = (lane & sft) ? mean : om;
            const float lo_M = (lane & sft) ? oM2 : M2, hi_M = (lane & sft) ? M2 : oM2;
            const float dd = hi_m - lo_m;
            mean = lo_m + dd * 0.5f;
            M2 = lo_M + hi_M + dd * dd * (na * 0.5f);
            na *= 2.f;
        }
    };

    // ================= pass 1: h = conv1(silu(GN1(x)))
    const int Cin = p.C0 + p.C1;
    load_raw(0);
    if (p.st0) {
        gn_coeffs_from_stats_image(p.st0 + (int64_t)img * p.nt0 * p.C0, p.st1 ? p.st1 + (int64_t)img * p.nt1 * p.C1 : nullptr, p.C0, p.C1, p.nt0,
                                   p.nt1, H * W, Cin < 32 ? Cin : 32, p.gn1_w, p.gn1_b, nullptr, gsh, Cf, Cf + 128, 1e-5f, tid, FI_NT,
                                   [] { F4_LDS_BARRIER(); });
    } else {
        for (int i = tid; i < 2 * Cin; i += FI_NT) {
            if (i < Cin) Cf[i] = p.coefA1[(int64_t)img * Cin + i];
            else Cf[128 + (i - Cin)] = p.coefB1[(int64_t)img * Cin + (i - Cin)];
        }
    }
    F4_LDS_BARRIER();
    conv_pass(p.w1, Cin / F6_KC, std::integral_constant<bool, true>());
    DLPM_PHASE(p, 8);

    // ================= between: the halves meet; h + bias, statistics, GroupNorm-2 with scale-shift, a = silu(..) -> LDS image
    auto between = [&](auto ra_c) __attribute__((always_inline)) {
        constexpr int RA = decltype(ra_c)::value;
        exchange(ra_c);
        int chv = ch, lkv = lk;
        asm volatile("" : "+v"(chv), "+v"(lkv));
        const float bias_v = p.b1[chv];
        float hv[2][16];
        float K = 0.f, s1 = 0.f, s2 = 0.f;
        {
            float Yt[16];
            out_tile(std::integral_constant<int, RA>(), Yt);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float v = Yt[k] + bias_v;
                if (k == 0) K = v;
                const float dd = v - K;
                s1 += dd;
                s2 = fmaf(dd, dd, s2);
                hv[0][k] = v;
            }
            out_tile(std::integral_constant<int, RA + 1>(), Yt);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float v = Yt[k] + bias_v;
                const float dd = v - K;
                s1 += dd;
                s2 = fmaf(dd, dd, s2);
                hv[1][k] = v;
            }
        }
        float mean, M2;
        half_stats(K, s1, s2, mean, M2);
        if (lkv == 0) part[kh * CO + chv] = make_float2(mean, M2);
        F4_LDS_BARRIER();
        gn_coeffs_from_stats_image(part, nullptr, CO, 0, 2, 1, H * W, 32, p.gn2_w, p.gn2_b, p.emb + (int64_t)img * p.emb_stride + p.emb_off, gsh, Cf,
                                   Cf + 128, 1e-5f, tid, FI_NT, [] { F4_LDS_BARRIER(); });
        F4_LDS_BARRIER();
        const float a2 = Cf[chv], b2 = Cf[128 + chv];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int tile = 4 * lkv + RA + rr;
            const int tpix = (4 * (tile >> 2)) * W + 4 * (tile & 3);
#pragma unroll
            for (int k = 0; k < 16; k++) hL[(tpix + (k >> 2) * W + (k & 3)) * F6_HLD + chv] = silu_f(fmaf(hv[rr][k], a2, b2));
        }
    };
    if (kh == 0) between(std::integral_constant<int, 0>());
    else between(std::integral_constant<int, 2>());
    F4_LDS_BARRIER();

    // ================= pass 2: out = conv2(a) + bias + skip
    conv_pass(p.w2, CO / F6_KC, std::integral_constant<bool, false>());
    DLPM_PHASE(p, 9);
    auto epilogue = [&](auto ra_c) __attribute__((always_inline)) {
        constexpr int RA = decltype(ra_c)::value;
        exchange(ra_c);
        int chv = ch, lkv = lk;
        asm volatile("" : "+v"(chv), "+v"(lkv));
        const float bias_v = p.b2[chv];
        const int64_t pix0 = (int64_t)img * H * W;
        float *__restrict__ out_blk = p.out + pix0 * CO;
        const float *__restrict__ res_blk = p.res + pix0 * CO;
        const bool do_stats = p.stats_out != nullptr;
        float K = 0.f, s1 = 0.f, s2 = 0.f;
        auto one = [&](auto r_c, int rr) __attribute__((always_inline)) {
            const int tile = 4 * lkv + RA + rr;
            const int tpix = (4 * (tile >> 2)) * W + 4 * (tile & 3);
            float rs[16];
#pragma unroll
            for (int k = 0; k < 16; k++) rs[k] = res_blk[(tpix + (k >> 2) * W + (k & 3)) * CO + chv];
            float Yt[16];
            out_tile(r_c, Yt);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float v = (Yt[k] + bias_v) + rs[k];
                if (rr == 0 && k == 0) K = v;
                const float dd = v - K;
                s1 += dd;
                s2 = fmaf(dd, dd, s2);
                out_blk[(tpix + (k >> 2) * W + (k & 3)) * CO + chv] = v;
            }
        };
        one(std::integral_constant<int, RA>(), 0);
        one(std::integral_constant<int, RA + 1>(), 1);
        if (do_stats) {     // per-image statistics: the two halves' 128-pixel partials merged in a fixed order (Chan's update)
            float mean, M2;
            half_stats(K, s1, s2, mean, M2);
            if (lkv == 0) part[kh * CO + chv] = make_float2(mean, M2);
            F4_LDS_BARRIER();
            if (kh == 0 && lkv == 0) {
                const float2 a = part[chv], b = part[CO + chv];
                const float dd = b.x - a.x;
                p.stats_out[(int64_t)img * CO + chv] = make_float2(a.x + dd * 0.5f, a.y + b.y + dd * dd * 64.f);
            }
        }
    };
    if (kh == 0) epilogue(std::integral_constant<int, 0>());
    else epilogue(std::integral_constant<int, 2>());
    DLPM_PHASE(p, 10);
    DLPM_PHASE_FLUSH(p, 8);
#ifdef DLPM_PHASE_TIMING
    if (p.phase && tid == 0) atomicAdd(p.phase + 11, 1ull);
#endif
}

// OIHW (3x3) -> U = G g G^T (6x6 per filter) in the kernel's fragment order Wf[ntile][wave][phase][18][lane][4]:
// lane = lk*16 + li holds, for position pair pp and e = 0..3, U_pos[cin = phase*8 + 2 lk + (e & 1)][cout = ntile*128 + wave*16 + li]
// with pos = 2 pp + (e >> 1).  Computed in double, rounded once.
// vs = 1 (the VS kernel): Wf[ntile][wave = 4 ph + q][phase][slot = 2 pp + nt][lane][4]: lane holds, for e = 0..3,
// U_pos[cin = phase*8 + 2 lk + (e & 1)][cout = ntile*128 + 32 q + 16 nt + li] with pos = 18 ph + 2 pp + (e >> 1)
// nw = waves per workgroup (8 / 4 / 2 for 128- / 64- / 32-channel n-tiles): Wf[ntile][wave < nw][phase][18][lane][4]
__global__ void k_relayout_weight_wino4(const float *oihw, float *dst, int Cout, int Cin, int vs, int nw) {
    const int nch = Cin / F4_KC;
    const int64_t total = (int64_t)Cout * Cin * 36;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    int64_t r = i >> 8;
    const int pp = (int)(r % 18); r /= 18;
    const int chunk = (int)(r % nch); r /= nch;
    const int wave = (int)(r % nw);
    const int nt = (int)(r / nw);
    const int lk = lane >> 4, li = lane & 15;
    const int pos = vs ? 18 * (wave >> 2) + 2 * (pp >> 1) + (e >> 1) : 2 * pp + (e >> 1);
    const int cin = chunk * F4_KC + 2 * lk + (e & 1);
    const int cout = vs ? nt * F4_NQ + 32 * (wave & 3) + 16 * (pp & 1) + li : nt * (16 * nw) + wave * 16 + li;
    const float *g = oihw + ((int64_t)cout * Cin + cin) * 9;
    const double a = F4_PA, b = F4_PB, a2 = a * a, b2 = b * b;
    const double n0 = a2 * b2, na = 2. * a2 * (a2 - b2), nb = 2. * b2 * (b2 - a2);    // prod_{l != j} (p_j - p_l) for p_j = 0, +-a, +-b
    const double G[6][3] = {{1. / n0, 0., 0.},         {1. / na, a / na, a2 / na}, {1. / na, -a / na, a2 / na},
                            {1. / nb, b / nb, b2 / nb}, {1. / nb, -b / nb, b2 / nb}, {0., 0., 1.}};
    const int ra = pos / 6, rb = pos % 6;
    double u = 0.;
    for (int ii = 0; ii < 3; ii++) {
        double row = 0.;
        for (int jj = 0; jj < 3; jj++) row += (double)g[ii * 3 + jj] * G[rb][jj];
        u += G[ra][ii] * row;
    }
    dst[i] = (float)u;
}

int f4_mode() {   // DLPM_WINO_F4=0: keep every 3x3 layer on the F(2x2,3x3) kernel (default: F(4x4) where the shape qualifies)
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLPM_WINO_F4"); v = e ? atoi(e) : 1; }
    return v;
}

}  // namespace

bool wino4_enabled() { return f4_mode() != 0; }
// DLPM_WINO_VS=1: waves = 2 position halves x 4 channel quarters (every V fragment read by 4 waves instead of 8).  Built in round 4
// because the held-clock ablations blamed 12 % of the clock on the V reads; with REAL operands halving that LDS traffic is worth
// nothing: 12.25 / 12.31 / 12.29 ms against 12.20 / 12.25 / 12.27 over the eight layer shapes in three alternating pairs (long-K layers
// -0.4 %, K = 128 layers +3 %: the LDS hand-over of the split output transform), profiles/r04/conv_layers_vsplit.txt -- what the
// ablation had measured was the MFMA's own data activity (it replaces the A fragments by constants), not the LDS.  Off; same
// results to rounding (the output transform sums two partial transforms).  Read once per process: the weight fragment order built
// at finalize and the instantiation launched must agree.
bool wino4_vsplit() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLPM_WINO_VS"); v = e ? atoi(e) : 0; }
    return v != 0;
}
bool wino4_image_stats(int cout) { (void)cout; return F4_EPI_T == 0; }

// DLPM_WINO4_NARROW=0 (A/B runs): the 64- / 32-channel n-tile shapes off -- those layers take F(2x2) / the implicit GEMM as in rounds 1-4
static bool narrow_enabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLPM_WINO4_NARROW"); v = e ? atoi(e) : 1; }
    return v != 0;
}

// geometry of a launch on n-tiles of nq channels (the narrow shapes' raw buffer holds halo patches of <= 400 pixels)
static bool wino4_geometry_nq(const ConvLaunch &c, int nq, int *bh, int *bw, int *nimg) {
    if (nq != F4_NQ && !narrow_enabled()) return false;
    if (!c.w_wino4 || c.ks != 3 || c.stride != 1 || c.in_nchw || c.out_nchw || c.abl) return false;
    if ((c.Hout & 3) || (c.Wout & 3) || c.Cout % 32 != 0 || (c.C0 + c.C1) % F4_KC != 0 || c.C0 % F4_KC != 0) return false;
    const int rawpix = nq == F4_NQ ? F4_RAWPIX : F4_RAWPIX_N;
    if ((c.R0 & 15) != 0) return false;   // a wave's 16 output channels stay on one side of a residual concat
    const int TH = c.Hout / 4, TW = c.Wout / 4;
    int h, w, n;
    if (TH * TW >= F4_TILES) {       // a block inside one image
        w = TW < 4 ? TW : 4;
        if (F4_TILES % w != 0) return false;
        h = F4_TILES / w;
        if (TW % w != 0 || TH % h != 0) return false;
        n = 1;
    } else {                         // several whole small images per block
        if (F4_TILES % (TH * TW) != 0) return false;
        h = TH; w = TW; n = F4_TILES / (TH * TW);
    }
    const int RH = c.ups ? 2 * h + 2 : 4 * h + 2, RW = c.ups ? 2 * w + 2 : 4 * w + 2;
    if (n * RH * RW > rawpix) return false;
    *bh = h; *bw = w; *nimg = n;
    return true;
}

// Round 6 -- the n-tile width of a 128-channel-multiple layer under a DECLARED batch.  At dispatch_B <= 256 the
// 16x16 / 8x8 levels of the CIFAR net launch 32-128 workgroups of 128 channels on 256 CUs; the 64- / 32-channel n-tile shapes (4 + 3 and
// 2 + 2 waves) give 2x / 4x as many from the same fragment arithmetic (same bits: a wave's share is 16 channels x 36 positions x 16 tiles
// in every shape).  The widest n-tile whose grid fills the chip at the declared batch wins; if none does, the narrowest available.  A
// function of the layer and the declaration only -- never of the batch a launch carries.  DLPM_WINO4_NQ=128 switches it off (A/B runs).
static int wino4_nq_for(const ConvLaunch &c) {
    const int base = f4_nq_of(c.Cout);
    if (base != F4_NQ || c.dispatch_B <= 0 || (!c.w_wino4_n64 && !c.w_wino4_n32)) return base;
    static int floor_nq = -1;
    if (floor_nq < 0) { const char *e = getenv("DLPM_WINO4_NQ"); floor_nq = e ? atoi(e) : 32; }
    int best = F4_NQ;
    for (int nq = F4_NQ; nq >= 32 && nq >= floor_nq; nq >>= 1) {
        if (nq == 64 && !c.w_wino4_n64) continue;
        if (nq == 32 && !c.w_wino4_n32) continue;
        int h, w, n;
        if (!wino4_geometry_nq(c, nq, &h, &w, &n)) continue;
        best = nq;
        const int64_t tiles = c.dispatch_B * (c.Hout / 4) * (c.Wout / 4);
        const int64_t mblocks = n == 1 ? tiles / F4_TILES : ceil_div(c.dispatch_B, (int64_t)n);
        if (mblocks * (c.Cout / nq) >= 256) break;
    }
    return best;
}

bool wino4_geometry(const ConvLaunch &c, int *bh, int *bw, int *nimg) { return wino4_geometry_nq(c, wino4_nq_for(c), bh, bw, nimg); }
int wino4_launch_nq(const ConvLaunch &c) { return wino4_nq_for(c); }

// dispatch policy (a function of the layer and of ConvLaunch::gen / dispatch_B, never of the batch in this launch:
// the generations round differently and a sample must not depend on how its batch was sharded or chunked):
//   DLPM_CONV_F4     F(4x4) wherever the geometry qualifies;
//   DLPM_CONV_F2 / DLPM_CONV_IGEMM   never;
//   DLPM_CONV_AUTO   F(4x4) wherever it applies, except -- when the F(2x2) weights are there too --
//     * blocks of 16 one-tile images (4x4-pixel tensors): 576 halo pixels staged per 256 outputs, measured 1.55 vs 1.00 ms/step;
//     * with a caller-declared dispatch batch: launches whose grid would leave CUs empty AT THAT BATCH (fewer than 256
//       workgroups of 256 pixels x 128 channels, e.g. 8x8 tensors at batch 256): the F(2x2) kernel's 128-pixel blocks give
//       twice as many workgroups.
bool wino4_preferred(const ConvLaunch &c, int *bh, int *bw, int *nimg) {
    if (c.gen == DLPM_CONV_F2 || c.gen == DLPM_CONV_IGEMM) return false;
    if (!wino4_geometry(c, bh, bw, nimg)) return false;
    if (!c.w_wino || c.gen == DLPM_CONV_F4) return true;
    if (*nimg > 4) return false;
    if (c.dispatch_B <= 0) return true;
    const int64_t tiles = c.dispatch_B * (c.Hout / 4) * (c.Wout / 4);
    const int64_t mblocks = *nimg == 1 ? tiles / F4_TILES : ceil_div(c.dispatch_B, (int64_t)*nimg);
    return mblocks * (c.Cout / wino4_nq_for(c)) >= 256;
}

// DLPM_WINO4_IMG=0 (A/B runs): the 32-channel 32x32 layers on the 2 + 2-wave 16-tile shape as in round 5 (same bits either way).  A
// function of the layer's geometry only, like every other kernel choice.
static bool wino4_whole_image(const ConvLaunch &c) {
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLPM_WINO4_IMG"); v = e ? atoi(e) : 1; }
    return v != 0 && c.Cout == 32 && !c.ups && c.Hout == 32 && c.Wout == 32 && (c.C0 + c.C1) <= 1024;
}

static int launch_conv_wino4_nq(const ConvLaunch &c, int nq, int bh, int bw, int nimg, hipStream_t st);

int launch_conv_wino4(const ConvLaunch &c, hipStream_t st) {
    int bh, bw, nimg;
    if (!wino4_geometry(c, &bh, &bw, &nimg)) {
        set_error("launch_conv_wino4: unsupported shape");
        return DLPM_ERR_UNSUPPORTED;
    }
#ifdef DLPM_PHASE_TIMING
    const_cast<ConvLaunch &>(c).phase = phase_buffer();
#endif
    const int nq = wino4_nq_for(c);
    if (nq != f4_nq_of(c.Cout)) {     // a 128-multiple layer on narrow n-tiles: the matching fragment stream
        ConvLaunch cn = c;
        cn.w_wino4 = nq == 64 ? c.w_wino4_n64 : c.w_wino4_n32;
        cn.w_wino4_n64 = cn.w_wino4_n32 = nullptr;
        cn.dispatch_B = 0;
        return launch_conv_wino4_nq(cn, nq, bh, bw, nimg, st);
    }
    return launch_conv_wino4_nq(c, nq, bh, bw, nimg, st);
}

static int launch_conv_wino4_nq(const ConvLaunch &c, int nq, int bh, int bw, int nimg, hipStream_t st) {
    using KFn = void (*)(ConvLaunch, int, int, int);
    const int64_t tiles = (int64_t)c.B * (c.Hout / 4) * (c.Wout / 4);
    const int64_t mblocks = nimg == 1 ? tiles / F4_TILES : ceil_div(c.B, nimg);
    if (wino4_whole_image(c)) {   // 32 channels on 32x32 images: one workgroup per image, 8 MFMA waves (round 6)
        const int r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_conv3x3_wino4_img), 160 * 1024);
        if (r != DLPM_OK) return r;
        const size_t lds = (size_t)(FI_VBUF + FI_RAWBUF + 2 * (c.C0 + c.C1)) * sizeof(float);
        k_conv3x3_wino4_img<<<(unsigned)c.B, FI_NT, lds, st>>>(c);
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
    if (nq != F4_NQ) {   // narrow layers: 4 MFMA + 3 helper waves (448 threads: 2 staging items cover 400 halo pixels), or 2 + 2 (256 threads: 4 items)
        KFn fn = nq == 64 ? (c.ups ? &k_conv3x3_wino4<true, 0, 2, 0, 0, 4> : &k_conv3x3_wino4<false, 0, 2, 0, 0, 4>)
                          : (c.ups ? &k_conv3x3_wino4<true, 0, 4, 0, 0, 2> : &k_conv3x3_wino4<false, 0, 4, 0, 0, 2>);
        const int r = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), 160 * 1024);
        if (r != DLPM_OK) return r;
        const size_t lds = (size_t)(2 * F4_VBUF + 2 * F4_RAWBUF_N + 2 * F4_CFS) * sizeof(float);
        const int ks = c.ksplit > 1 ? c.ksplit : 1;
        if (ks > 1 && (c.bias || c.res0 || c.stats_out || ((c.C0 + c.C1) / F4_KC) % ks != 0)) {
            set_error("launch_conv_wino4: a split-K launch carries no bias / residual / statistics and divides its chunks evenly");
            return DLPM_ERR_ARG;
        }
        fn<<<(unsigned)(mblocks * (c.Cout / nq) * ks), (nq == 64 ? 7 : 4) * 64, lds, st>>>(c, bh, bw, nimg);
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
    if (c.ksplit > 1) {
        set_error("launch_conv_wino4: split-K is a narrow-shape launch");
        return DLPM_ERR_ARG;
    }
    const int RHp = c.ups ? 2 * bh + 2 : 4 * bh + 2, RWp = c.ups ? 2 * bw + 2 : 4 * bw + 2;
    const bool small = nimg * RHp * RWp <= F4_NT;      // two staging items per thread cover the patch
    KFn fn = c.ups ? (small ? &k_conv3x3_wino4<true, 0, 2> : &k_conv3x3_wino4<true>) : (small ? &k_conv3x3_wino4<false, 0, 2> : &k_conv3x3_wino4<false>);
    if (wino4_vsplit()) fn = c.ups ? (small ? &k_conv3x3_wino4<true, 0, 2, 0, 1> : &k_conv3x3_wino4<true, 0, F4_QNIT, 0, 1>)
                                   : (small ? &k_conv3x3_wino4<false, 0, 2, 0, 1> : &k_conv3x3_wino4<false, 0, F4_QNIT, 0, 1>);
    static int spec_on = -1;    // DLPM_WINO_SPEC=1: the channel-specialised instantiations (same bits; measured NEUTRAL, 1.0764-1.0823 vs 1.0786-1.0850 ms
                                // on the H32 128 -> 128 layer in three alternating runs, profiles/r04/conv_layers_specialised_h32_c128.txt: off)
    if (spec_on < 0) { const char *e = getenv("DLPM_WINO_SPEC"); spec_on = e ? atoi(e) : 0; }
    if (spec_on && !wino4_vsplit() && !c.ups && c.Hout == 32 && c.Wout == 32 && c.C0 == 128 && c.C1 == 0 && c.Cout == 128 && c.coefA && c.coefB && c.bias &&
        c.act_silu && bh == 4 && bw == 4 && nimg == 1 && !c.res1 && (!c.res0 || c.R0 == 128))
        fn = c.res0 ? &k_conv3x3_wino4<false, 0, 2, 2> : &k_conv3x3_wino4<false, 0, 2, 1>;

#ifdef DLPM_WINO_ABLATIONS
    static int abl = -1;
    if (abl < 0) { const char *e = getenv("DLPM_WABL"); abl = e ? atoi(e) : 0; }
    if (!c.ups && small) {       // (the measured shapes: halo patches of <= 512 pixels, the QN = 2 instantiation)
        switch (abl) {
            case 1: fn = &k_conv3x3_wino4<false, 1, 2>; break;
            case 2: fn = &k_conv3x3_wino4<false, 2, 2>; break;
            case 3: fn = &k_conv3x3_wino4<false, 3, 2>; break;
            case 7: fn = &k_conv3x3_wino4<false, 7, 2>; break;
            case 8: fn = &k_conv3x3_wino4<false, 8, 2>; break;
            case 16: fn = &k_conv3x3_wino4<false, 16, 2>; break;
            case 32: fn = &k_conv3x3_wino4<false, 32, 2>; break;
            case 64: fn = &k_conv3x3_wino4<false, 64, 2>; break;
            case 80: fn = &k_conv3x3_wino4<false, 80, 2>; break;
            case 87: fn = &k_conv3x3_wino4<false, 87, 2>; break;
            default: break;
        }
    }
#endif
    {
        int r = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), 160 * 1024);
        if (r != DLPM_OK) return r;
    }
    size_t loop_b = (size_t)(2 * F4_VBUF + 2 * F4_RAWBUF + 2 * F4_CFS) * sizeof(float);
    if (wino4_vsplit() && loop_b < (size_t)8 * 4 * 16 * 64 * sizeof(float)) loop_b = (size_t)8 * 4 * 16 * 64 * sizeof(float);   // the epilogue's exchange buffer
    const int64_t total = mblocks * (c.Cout / F4_NQ);
    // DLPM_WINO4_PERSIST=1 (round 6 experiment): one workgroup per CU walking total / CUs tiles (launches of more than two rounds, plain shapes only)
    static int pers = -1, ncu = 0;
    if (pers < 0) {
        const char *e = getenv("DLPM_WINO4_PERSIST"); pers = e ? atoi(e) : 0;
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    }
    if (pers && ncu > 0 && total > 2 * ncu && !wino4_vsplit() && fn == (c.ups ? (small ? (KFn)&k_conv3x3_wino4<true, 0, 2> : (KFn)&k_conv3x3_wino4<true>)
                                                                            : (small ? (KFn)&k_conv3x3_wino4<false, 0, 2> : (KFn)&k_conv3x3_wino4<false>))) {
        KFn pf = c.ups ? (small ? &k_conv3x3_wino4<true, 0, 2, 0, 0, 8, 1> : &k_conv3x3_wino4<true, 0, F4_QNIT, 0, 0, 8, 1>)
                       : (small ? &k_conv3x3_wino4<false, 0, 2, 0, 0, 8, 1> : &k_conv3x3_wino4<false, 0, F4_QNIT, 0, 0, 8, 1>);
        const int r = ensure_dynamic_lds(reinterpret_cast<const void *>(pf), 160 * 1024);
        if (r != DLPM_OK) return r;
        ConvLaunch cp = c;
        cp.pers_total = (int)total;
        pf<<<(unsigned)ncu, F4_NT, loop_b, st>>>(cp, bh, bw, nimg);
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
    fn<<<(unsigned)total, F4_NT, loop_b, st>>>(c, bh, bw, nimg);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

// DLPM_RES_IMG=0 (A/B runs): the 32-channel 32x32 ResBlocks as separate launches (conv, GroupNorm coefficients, conv) as before -- same bits
bool res_img_ok(const ResImgLaunch &r) {
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLPM_RES_IMG"); v = e ? atoi(e) : 1; }
    const int Cin = r.C0 + r.C1;
    const bool common = v != 0 && wino4_enabled() && r.w1 && r.w2 && r.res && Cin <= 128 && Cin >= 32 &&
                        (r.st0 ? (r.C1 == 0 || r.st1 != nullptr) : (r.coefA1 && r.coefB1));
    if (r.H == 16) {     // 64 output channels on 16x16 images (k_resblock_wino4_img16): 16-channel chunks that stay inside one concat source
        static int v16 = -1;
        if (v16 < 0) { const char *e = getenv("DLPM_RES_IMG16"); v16 = e ? atoi(e) : 1; }
        return common && v16 != 0 && Cin % F6_KC == 0 && r.C0 % F6_KC == 0;
    }
    return common && r.H == 32 && r.hbuf && Cin % F4_KC == 0 && r.C0 % F4_KC == 0;
}

int launch_resblock_img(const ResImgLaunch &r, hipStream_t st) {
    if (!res_img_ok(r)) {
        set_error("launch_resblock_img: unsupported block");
        return DLPM_ERR_UNSUPPORTED;
    }
    const int Cin = r.C0 + r.C1;
#ifdef DLPM_PHASE_TIMING
    const_cast<ResImgLaunch &>(r).phase = phase_buffer();
#endif
    if (r.H == 16) {
        const double M16 = (double)r.B * 256;
        ProfScope ps16("resblock_img:H16", 2.0 * M16 * 64 * 9.0 * (Cin + 64), 4.0 * (M16 * (Cin + 64 + 64) + 64 * 9.0 * (Cin + 64)), st);
        const int e16 = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_resblock_wino4_img16), 160 * 1024);
        if (e16 != DLPM_OK) return e16;
        const size_t lds16 = (size_t)(F6_VBUF + F6_RAWBUF + F6_HBUF + 256 + 512 + 256 + F6_QN * FI_NT) * sizeof(float);
        k_resblock_wino4_img16<<<(unsigned)r.B, FI_NT, lds16, st>>>(r);
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
    const double M = (double)r.B * 1024;
    ProfScope ps("resblock_img:H32", 2.0 * M * 32 * 9.0 * (Cin + 32), 4.0 * (M * (Cin + 32 + 32) + 32 * 9.0 * (Cin + 32)), st);
    const int e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_resblock_wino4_img), 160 * 1024);
    if (e != DLPM_OK) return e;
    const size_t lds = (size_t)(FI_VBUF + FI_RAWBUF + 256 + 512 + 256 + FI_QN * FI_NT) * sizeof(float);
    k_resblock_wino4_img<<<(unsigned)r.B, FI_NT, lds, st>>>(r);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int64_t wino4_weight_floats(int Cout, int Cin) { return (int64_t)Cout * Cin * 36 + F4_PAD * 256; }

int relayout_weight_wino4(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st, int nq_in) {
    const int64_t n = (int64_t)Cout * Cin * 36;
    DLPM_HIP(hipMemsetAsync(dst_dev + n, 0, (size_t)F4_PAD * 256 * sizeof(float), st));
    const int nq = nq_in > 0 ? nq_in : f4_nq_of(Cout);
    k_relayout_weight_wino4<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin, (nq == F4_NQ && wino4_vsplit()) ? 1 : 0, nq / 16);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
