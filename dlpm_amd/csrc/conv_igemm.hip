// conv_igemm.hip -- conv3x3 / conv1x1 / Linear as an implicit GEMM on the gfx950 fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32, bitwise a k-ordered fmaf chain).
//
// Replaces F.conv2d (3x3 s1/s2, 1x1), conv1d(k=1) and F.linear on the UNet path:
// dlpm/models/unet.py:64,96,143,157,168,213,215,336-338 -- 99.8 % of the path's FLOPs.
//
// GEMM view:  Out[m, n] = sum_{tap, c} Act(In[pix(m) + tap, c]) * W[tap][n][c]
//   m = output pixel (b, oy, ox) flattened over NHWC, n = output channel, K = taps * Cin.
// Workgroup tile 128 (pixels) x BN (channels), 4 waves (one per SIMD), K consumed in steps of
// one tap x 32 channels.  Per step the A tile (128 x 32) is gathered from global memory with
// the tap's spatial shift -- zero padding, stride 2, nearest-x2 upsampling and the two-pointer
// "virtual concat" of the skip connection are all just address arithmetic there -- and the fused
// GroupNorm affine + SiLU is applied on the way into LDS, so the normalised tensor never exists
// in HBM.  Loop order is channel-chunk outer / tap inner: the 9 taps of a chunk re-read the same
// input lines from L2.  LDS rows are padded to 36 floats so that the ds_read_b128 fragment loads
// (lane -> pixel row, 4 consecutive k) are bank-conflict free.  The k order inside a 32-channel
// step is permuted identically for A and B (lane half h reads channels 8j+4h..8j+4h+3), which
// lets every fragment load be one 16-byte LDS read feeding 4 MFMAs.
#include <cstdlib>

#include <algorithm>

#include "conv.h"
#include "igemm_epilogue.h"

namespace dlpm {
namespace {

constexpr int KC = 32;      // channels per K step
constexpr int LDS_LD = 36;  // padded row length (floats)

// Epilogue through LDS: the MFMA accumulator layout (lane = output channel, registers = rows) would
// store 4 bytes per lane to 2 rows per instruction and read the residual the same way; instead each
// wave row-block drops its accumulators into an LDS image [rows][BN+4] and all 256 threads stream
// it out as 16-byte row segments (bias + residual added on the way), i.e. whole 128..512-byte NHWC
// rows per instruction.  Measured on the CIFAR net: residual read + store cost 5 ms of 110 ms/step
// in the per-lane form.
template <int BN, int WAVES_M, int WAVES_N, int RM, int RN>
__device__ __forceinline__ void epilogue_rows(const ConvLaunch &p, floatx16 (&acc)[RM][RN], float *lds, int64_t m0, int n0,
                                              int64_t M, int tid, int wm, int wn, int l31, int kh) {
    constexpr int LD = BN + 4, PR = RM * 32, C4 = BN / 4, RG = 256 / C4;  // RG row groups share a column quad
    const int R1 = p.Cout - p.R0;
    // fused GroupNorm statistics of the output (tile inside one image only): every thread keeps the
    // same 4 channels across the whole epilogue, so it carries shifted sums for them in registers
    const bool do_stats = p.stats_out != nullptr;
    float4 K = make_float4(0.f, 0.f, 0.f, 0.f), s1 = K, s2 = K;
    int cnt = 0;
    for (int ph = 0; ph < WAVES_M; ph++) {
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
        for (int idx = tid; idx < PR * C4; idx += 256) {
            const int row = idx / C4, c4 = idx - row * C4;
            const int n = n0 + c4 * 4;
            const int64_t m = m0 + ph * PR + row;
            if (m < M && n < p.Cout) {
                float4 v = *reinterpret_cast<const float4 *>(lds + row * LD + c4 * 4);
                if (p.bias) {
                    const float4 b = *reinterpret_cast<const float4 *>(p.bias + n);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                if (p.res0 && !(p.abl & 2)) {
                    const float4 q = (n < p.R0) ? *reinterpret_cast<const float4 *>(p.res0 + m * p.R0 + n)
                                                : *reinterpret_cast<const float4 *>(p.res1 + m * R1 + (n - p.R0));
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                if (do_stats) {
                    if (cnt == 0) K = v;  // pivot = this thread's first value per channel
                    float d;
                    d = v.x - K.x; s1.x += d; s2.x = fmaf(d, d, s2.x);
                    d = v.y - K.y; s1.y += d; s2.y = fmaf(d, d, s2.y);
                    d = v.z - K.z; s1.z += d; s2.z = fmaf(d, d, s2.z);
                    d = v.w - K.w; s1.w += d; s2.w = fmaf(d, d, s2.w);
                    cnt++;
                }
                if ((p.abl & 1) && v.x == v.x) continue;  // timing ablation only
                *reinterpret_cast<float4 *>(p.out + m * p.Cout + n) = v;
            }
        }
        __syncthreads();
    }
    if (do_stats) {
        // per-thread (mean, M2) over its cnt rows -> LDS [RG][BN] -> threads < BN merge the RG
        // partials of one channel with Chan's update and write the tile's pair
        float2 *part = reinterpret_cast<float2 *>(lds);  // the row image is dead now
        const int c4 = tid % C4, rg = tid / C4;
        const float fc = (float)(cnt > 0 ? cnt : 1);
        const float mx = s1.x / fc, my = s1.y / fc, mz = s1.z / fc, mw = s1.w / fc;
        part[rg * BN + c4 * 4 + 0] = make_float2(K.x + mx, fmaxf(s2.x - s1.x * mx, 0.f));
        part[rg * BN + c4 * 4 + 1] = make_float2(K.y + my, fmaxf(s2.y - s1.y * my, 0.f));
        part[rg * BN + c4 * 4 + 2] = make_float2(K.z + mz, fmaxf(s2.z - s1.z * mz, 0.f));
        part[rg * BN + c4 * 4 + 3] = make_float2(K.w + mw, fmaxf(s2.w - s1.w * mw, 0.f));
        __syncthreads();
        if (tid < BN && n0 + tid < p.Cout) {
            const float npart = (float)(BM / RG);  // rows behind each partial (tiles are full here)
            float mean = part[tid].x, M2 = part[tid].y, na = npart;
            for (int g = 1; g < RG; g++) {
                const float2 q = part[g * BN + tid];
                const float d = q.x - mean, N = na + npart;
                mean += d * (npart / N);
                M2 += q.y + d * d * (na * npart / N);
                na = N;
            }
            p.stats_out[(m0 / BM) * p.Cout + n0 + tid] = make_float2(mean, M2);
        }
    }
}

// MODE 0: any shape.  MODE 1: full tiles (M % 128 == 0, Cout % BN == 0, NHWC out): epilogue_rows_full.
// MODE 2: MODE 1 and a 1x1 / stride 1 / NHWC convolution with C0 % 32 == 0 (the skip, qkv and proj convolutions):
// the A gather is a row pointer that advances 32 channels per step (no tap / padding / clamp arithmetic in the loop).
// (A persistent variant of MODE 2 -- one software pipeline over (tile, K step), the next tile's operands in LDS before
// the epilogue -- measured 10 % SLOWER: with both workgroups of a CU always inside the MFMA loop the two waves of a
// SIMD run at 78 % of the pipe (LDS / VMEM / VALU issue shares the port), whereas here a workgroup has the pipe to
// itself while its neighbour is in its prologue or epilogue.)
template <int BN, int WAVES_M, int WAVES_N, int RM, int RN, int MODE = 0>
__global__ void __launch_bounds__(256) k_conv_igemm(ConvLaunch p) {
    static_assert(WAVES_M * WAVES_N == 4 && WAVES_M * RM * 32 == BM && WAVES_N * RN * 32 == BN, "tile shape");
    constexpr int NBF = BN / 8;          // floats of the W tile staged per thread (BN*32/256)
    constexpr int NBV = NBF / 4;         // ... as float4s
    constexpr int TPR = KC / NBF;        // threads per W row
    constexpr int BUF = (BM + BN) * LDS_LD;
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    DLPM_PHASE_DECL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int Cin = p.C0 + p.C1;
    const int ntaps = p.ks * p.ks, pad = p.ks >> 1;
    const int HWo = p.Hout * p.Wout;
    const int64_t M = (int64_t)p.B * HWo;
    const int ntile_n = (p.Cout + BN - 1) / BN;
    const int64_t m0 = (int64_t)(blockIdx.x / ntile_n) * BM;
    const int n0 = (blockIdx.x % ntile_n) * BN;
    const int Hi = p.ups ? p.Hin * 2 : p.Hin, Wi = p.ups ? p.Win * 2 : p.Win;

    // ---- staging assignment: A row (pixel) and 16-channel segment of this thread
    const int ra = tid >> 1, sega = (tid & 1) * 16;
    const int64_t ma = m0 + ra;
    const bool row_ok = ma < M;
    int pb = 0, oy = 0, ox = 0;
    if (row_ok) {
        pb = (int)(ma / HWo);
        int rem = (int)(ma - (int64_t)pb * HWo);
        oy = rem / p.Wout;
        ox = rem - oy * p.Wout;
    }
    const int rb = tid / TPR, segb = (tid % TPR) * NBF;
    const bool wrow_ok = (n0 + rb) < p.Cout;

    float4 xa[4], ca[4], cb[4], wb[NBV];
    bool a_ok = false;
    const bool has_coef = p.coefA != nullptr;

    // All prefetch loads are UNCONDITIONAL (addresses clamped into range, zeros selected at store
    // time): a load inside a divergent branch makes hipcc copy its result at the join, which forces
    // s_waitcnt vmcnt(0) right there and serialises the prefetch with the MFMAs it should hide under.
    const int wrow = min(n0 + rb, p.Cout - 1);
    // MODE 2 row pointers: channel c of this thread's pixel is a0[c] for c < C0 and a1[c] beyond (a1 is pre-biased)
    const float *a0 = p.src0 + ma * p.C0 + sega;
    const float *a1 = p.src1 ? p.src1 + ma * p.C1 + sega - p.C0 : a0;
    const float *w0 = p.w + (int64_t)wrow * Cin + segb;
    const float *ka = has_coef ? p.coefA + (int64_t)pb * Cin + sega : nullptr;
    const float *kb = has_coef ? p.coefB + (int64_t)pb * Cin + sega : nullptr;
    auto load_step = [&](int s) {
        if constexpr (MODE == 2) {
            const int c0 = s * KC;
            a_ok = true;
            const float *src = (c0 < p.C0) ? a0 + c0 : a1 + c0;   // uniform: C0 % 32 == 0
#pragma unroll
            for (int v = 0; v < 4; v++) xa[v] = reinterpret_cast<const float4 *>(src)[v];
            if (has_coef) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    ca[v] = reinterpret_cast<const float4 *>(ka + c0)[v];
                    cb[v] = reinterpret_cast<const float4 *>(kb + c0)[v];
                }
            }
#pragma unroll
            for (int v = 0; v < NBV; v++) wb[v] = reinterpret_cast<const float4 *>(w0 + c0)[v];
            return;
        }
        const int chunk = s / ntaps, tap = s - chunk * ntaps;
        const int c0 = chunk * KC;
        const int ky = tap / p.ks, kx = tap - ky * p.ks;
        const int iy = oy * p.stride + ky - pad, ix = ox * p.stride + kx - pad;
        a_ok = row_ok && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
        const int c = c0 + sega;
        const int cy = min(max(iy, 0), Hi - 1), cx = min(max(ix, 0), Wi - 1);
        const int sy = p.ups ? (cy >> 1) : cy, sx = p.ups ? (cx >> 1) : cx;
        const int64_t pix = ((int64_t)pb * p.Hin + sy) * p.Win + sx;
        const float *src = (c < p.C0) ? p.src0 + pix * p.C0 + c : p.src1 + pix * p.C1 + (c - p.C0);
#pragma unroll
        for (int v = 0; v < 4; v++) xa[v] = reinterpret_cast<const float4 *>(src)[v];
        if (has_coef) {  // wave-uniform
            const float *pa = p.coefA + (int64_t)pb * Cin + c, *pbq = p.coefB + (int64_t)pb * Cin + c;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                ca[v] = reinterpret_cast<const float4 *>(pa)[v];
                cb[v] = reinterpret_cast<const float4 *>(pbq)[v];
            }
        }
        const float *wp = p.w + ((int64_t)tap * p.Cout + wrow) * Cin + c0 + segb;
#pragma unroll
        for (int v = 0; v < NBV; v++) wb[v] = reinterpret_cast<const float4 *>(wp)[v];
    };

    auto store_step = [&](int buf) {
        float *As = smem + buf * BUF, *Bs = As + BM * LDS_LD;
        float4 *da = reinterpret_cast<float4 *>(As + ra * LDS_LD + sega);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            float4 x = xa[v];
            if (has_coef) {
                x.x = fmaf(x.x, ca[v].x, cb[v].x);
                x.y = fmaf(x.y, ca[v].y, cb[v].y);
                x.z = fmaf(x.z, ca[v].z, cb[v].z);
                x.w = fmaf(x.w, ca[v].w, cb[v].w);
            }
            if (p.act_silu) {
                x.x = silu_f(x.x);
                x.y = silu_f(x.y);
                x.z = silu_f(x.z);
                x.w = silu_f(x.w);
            }
            da[v] = a_ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 *db = reinterpret_cast<float4 *>(Bs + rb * LDS_LD + segb);
#pragma unroll
        for (int v = 0; v < NBV; v++) db[v] = wrow_ok ? wb[v] : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    floatx16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nsteps = (Cin / KC) * ntaps;
    load_step(0);
    store_step(0);
    __syncthreads();
    DLPM_PHASE(p, 0);

    for (int s = 0; s < nsteps; s++) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_step(s + 1);  // global loads in flight under the MFMAs below

        const float *As = smem + buf * BUF, *Bs = As + BM * LDS_LD;
        const float *ap = As + (wm * RM * 32 + l31) * LDS_LD + kh * 4;
        const float *bp = Bs + (wn * RN * 32 + l31) * LDS_LD + kh * 4;
#pragma unroll
        for (int kk = 0; kk < KC / 8; kk++) {
            float4 af[RM], bf[RN];
#pragma unroll
            for (int i = 0; i < RM; i++) af[i] = *reinterpret_cast<const float4 *>(ap + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < RN; j++) bf[j] = *reinterpret_cast<const float4 *>(bp + j * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int i = 0; i < RM; i++)
#pragma unroll
                for (int j = 0; j < RN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (s + 1 < nsteps) store_step(buf ^ 1);  // buf^1 was last read in step s-1 (barrier since)
        __syncthreads();
    }

    DLPM_PHASE(p, 1);
    if constexpr (MODE >= 1) {
        epilogue_rows_full<BN, WAVES_M, WAVES_N, RM, RN>(p, acc, smem, m0, n0, tid, wm, wn, l31, kh);
        DLPM_PHASE(p, 2);
#ifdef DLPM_PHASE_TIMING
        if (p.phase && tid == 0) atomicAdd(p.phase + 3, 1ull);
#endif
        return;
    }
    if (!p.out_nchw && (p.Cout & 3) == 0 && (p.R0 & 3) == 0) {
        epilogue_rows<BN, WAVES_M, WAVES_N, RM, RN>(p, acc, smem, m0, n0, M, tid, wm, wn, l31, kh);
        DLPM_PHASE(p, 2);
#ifdef DLPM_PHASE_TIMING
        if (p.phase && tid == 0) atomicAdd(p.phase + 3, 1ull);
#endif
        return;
    }
    // ---- scalar epilogue (NCHW head, odd channel counts): C/D layout of the 32x32 MFMA:
    //      col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int R1 = p.Cout - p.R0;
#pragma unroll
    for (int j = 0; j < RN; j++) {
        const int n = n0 + (wn * RN + j) * 32 + l31;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (wm * RM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int64_t m = m0 + row;
                if (m >= M) continue;
                float v = acc[i][j][r] + bias;
                if (p.res0) v += (n < p.R0) ? p.res0[m * p.R0 + n] : p.res1[m * R1 + (n - p.R0)];
                if (p.out_nchw) {  // boundary tensor (the head's eps): (b, n, oy, ox)
                    const int64_t bb = m / HWo;
                    p.out[(bb * p.Cout + n) * HWo + (m - bb * HWo)] = v;
                } else {
                    p.out[m * p.Cout + n] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 convolution with a HALO tile: the (th+2) x (W+2) x 32-channel input patch of a
// 128-pixel tile (th full image rows) is staged into LDS ONCE per channel chunk -- GroupNorm
// affine + SiLU applied once per element instead of once per tap -- and the 9 taps read it at
// shifted LDS addresses; only the 16-KB weight tile streams per tap (double-buffered).  Compared
// with the shifted-GEMM kernel above this removes 8/9 of the global loads, address arithmetic and
// activation math from the MFMA loop.
// ---------------------------------------------------------------------------------------------
constexpr int HALO_NIT = 9;  // float4 items per thread per chunk, upper bound ((th+2)(W+2) <= 288)

template <int BN, int WAVES_M, int WAVES_N, int RM, int RN>
__global__ void __launch_bounds__(256) k_conv3x3_halo(ConvLaunch p, int th, int nimg) {
    static_assert(WAVES_M * WAVES_N == 4 && WAVES_M * RM * 32 == BM && WAVES_N * RN * 32 == BN, "tile shape");
    constexpr int NBF = BN / 8, NBV = NBF / 4, TPR = KC / NBF;
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    const int W = p.Wout, H = p.Hout, Wp = W + 2;
    const int hpi = (th + 2) * Wp;                // halo pixels per image
    const int hp = nimg * hpi;                    // halo pixels of the tile (nimg > 1: whole small images)
    float *Ah = hsm;
    float *Bsb = hsm + ((hp + 3) & ~3) * LDS_LD;   // two weight buffers of BN x LDS_LD
    float *Cf = Bsb + 2 * BN * LDS_LD;             // fused-norm coefficients of this chunk: [nimg][A(32) | B(32)]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int Cin = p.C0 + p.C1;
    const int HWo = H * W;
    const int ntile_n = (p.Cout + BN - 1) / BN;
    const int64_t M = (int64_t)p.B * HWo;
    const int64_t m0 = (int64_t)(blockIdx.x / ntile_n) * BM;
    const int n0 = (blockIdx.x % ntile_n) * BN;
    const int pb = (int)(m0 / HWo);               // first image of the tile
    const int y0 = (nimg > 1) ? 0 : (int)((m0 - (int64_t)pb * HWo) / W);

    // ---- halo staging: this thread owns channels c4..c4+3 of halo pixels it*32 + (tid >> 3)
    const int c4 = (tid & 7) * 4;
    const int nit = (hp * 8 + 255) / 256;
    int off[HALO_NIT];  // source pixel index, -1: zero padding, -2: beyond the halo
#pragma unroll
    for (int it = 0; it < HALO_NIT; it++) {
        const int hpix = it * 32 + (tid >> 3);
        const int img = hpix / hpi, hr = hpix - img * hpi;
        const int hy = hr / Wp, hx = hr - hy * Wp;
        const int iy = y0 + hy - 1, ix = hx - 1;
        const bool pad = iy < 0 || iy >= H || ix < 0 || ix >= W || (pb + img) >= p.B;
        off[it] = (it >= nit || hpix >= hp) ? -2 : (pad ? -1 : (((pb + img) * H + iy) * W + ix));
    }
    float4 xh[HALO_NIT], cfr;
    const bool has_coef = p.coefA != nullptr;
    // coefficient staging: thread t < nimg*16 carries one float4 of A (t & 8 == 0) or B of image t >> 4
    const int cf_img = tid >> 4, cf_isb = (tid >> 3) & 1;
    const bool cf_mine = has_coef && tid < nimg * 16 && (pb + cf_img) < p.B;
    const float *cf_base = has_coef ? ((cf_isb ? p.coefB : p.coefA) + (int64_t)min(pb + cf_img, p.B - 1) * Cin) : nullptr;

    auto load_halo = [&](int chunk) {
        const int c = chunk * KC + c4;
        const bool first = c < p.C0;
        const float *sb = first ? p.src0 + c : p.src1 + (c - p.C0);
        const int ld = first ? p.C0 : p.C1;
        // unconditional loads (see k_conv_igemm): padding / out-of-halo items read pixel offc = 0
#pragma unroll
        for (int it = 0; it < HALO_NIT; it++)
            if (it < nit) xh[it] = *reinterpret_cast<const float4 *>(sb + (int64_t)max(off[it], 0) * ld);
        if (has_coef) cfr = *reinterpret_cast<const float4 *>(cf_base + c);
    };
    auto store_coef = [&]() {
        if (cf_mine) *reinterpret_cast<float4 *>(Cf + cf_img * 64 + cf_isb * 32 + c4) = cfr;
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int it = 0; it < HALO_NIT; it++) {
            if (off[it] == -2) continue;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (off[it] >= 0) {
                x = xh[it];
                if (has_coef) {
                    const int img = (nimg > 1) ? (it * 32 + (tid >> 3)) / hpi : 0;
                    const float4 ca = *reinterpret_cast<const float4 *>(Cf + img * 64 + c4);
                    const float4 cb = *reinterpret_cast<const float4 *>(Cf + img * 64 + 32 + c4);
                    x.x = fmaf(x.x, ca.x, cb.x);
                    x.y = fmaf(x.y, ca.y, cb.y);
                    x.z = fmaf(x.z, ca.z, cb.z);
                    x.w = fmaf(x.w, ca.w, cb.w);
                }
                if (p.act_silu && !(p.abl & 4)) {
                    x.x = silu_f(x.x);
                    x.y = silu_f(x.y);
                    x.z = silu_f(x.z);
                    x.w = silu_f(x.w);
                }
            }
            *reinterpret_cast<float4 *>(Ah + (it * 32 + (tid >> 3)) * LDS_LD + c4) = x;
        }
    };

    // ---- weight tile staging (per tap, double-buffered)
    const int rb = tid / TPR, segb = (tid % TPR) * NBF;
    const bool wrow_ok = (n0 + rb) < p.Cout;
    float4 wb[NBV];
    const int wrow = min(n0 + rb, p.Cout - 1);
    auto load_w = [&](int chunk, int tap) {
        const float *wp = p.w + ((int64_t)tap * p.Cout + wrow) * Cin + chunk * KC + segb;
#pragma unroll
        for (int v = 0; v < NBV; v++) wb[v] = reinterpret_cast<const float4 *>(wp)[v];
    };
    auto store_w = [&](int buf) {
        float4 *db = reinterpret_cast<float4 *>(Bsb + buf * BN * LDS_LD + rb * LDS_LD + segb);
#pragma unroll
        for (int v = 0; v < NBV; v++) db[v] = wrow_ok ? wb[v] : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    // ---- A fragment base addresses: output pixel -> halo coordinates of tap (0, 0)
    int abase[RM];
    const int hwt = th * W;  // output pixels per image inside the tile
#pragma unroll
    for (int i = 0; i < RM; i++) {
        const int mloc = (wm * RM + i) * 32 + l31;
        const int img = mloc / hwt, ml = mloc - img * hwt;
        const int yy = ml / W, xx = ml - yy * W;
        abase[i] = (img * hpi + yy * Wp + xx) * LDS_LD + kh * 4;
    }
    const float *bpw = Bsb + (wn * RN * 32 + l31) * LDS_LD + kh * 4;

    floatx16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nch = Cin / KC;
    load_halo(0);
    load_w(0, 0);
    store_coef();
    store_w(0);
    __syncthreads();
    store_halo();
    __syncthreads();

    int s = 0;
    for (int chunk = 0; chunk < ((p.abl & 64) ? 0 : nch); chunk++) {
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++, s++) {
            const int buf = s & 1;
            const bool last_tap = tap == 8, more = (chunk + 1 < nch);
            // one UNCONDITIONAL weight prefetch per tap (harmlessly redundant on the very last step):
            // two call sites would make hipcc merge their results with copies behind a vmcnt(0)
            if (!(p.abl & 32)) load_w(last_tap ? min(chunk + 1, nch - 1) : chunk, last_tap ? 0 : tap + 1);
            if (last_tap && more) load_halo(chunk + 1);

            const int ky = tap / 3, kx = tap - ky * 3;
            const int toff = (ky * Wp + kx) * LDS_LD;
            const float *bp = bpw + buf * BN * LDS_LD;
#pragma unroll
            for (int kk = 0; kk < KC / 8; kk++) {
                float4 af[RM], bf[RN];
#pragma unroll
                for (int i = 0; i < RM; i++) af[i] = *reinterpret_cast<const float4 *>(Ah + abase[i] + toff + kk * 8);
#pragma unroll
                for (int j = 0; j < RN; j++) bf[j] = *reinterpret_cast<const float4 *>(bp + j * 32 * LDS_LD + kk * 8);
#pragma unroll
                for (int i = 0; i < RM; i++)
#pragma unroll
                    for (int j = 0; j < RN; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                    }
            }
            if ((!last_tap || more) && !(p.abl & 8)) store_w(buf ^ 1);
            if (last_tap && more) {
                store_coef();     // (the previous chunk's coefficients were consumed before its first tap)
                __syncthreads();  // every wave has finished reading this chunk's halo; coefficients visible
                store_halo();
            }
            if (!(p.abl & 16)) __syncthreads();
        }
    }

    if (!p.out_nchw && (p.Cout & 3) == 0 && (p.R0 & 3) == 0) {
        epilogue_rows<BN, WAVES_M, WAVES_N, RM, RN>(p, acc, hsm, m0, n0, M, tid, wm, wn, l31, kh);
        return;
    }
    const int R1 = p.Cout - p.R0;
#pragma unroll
    for (int j = 0; j < RN; j++) {
        const int n = n0 + (wn * RN + j) * 32 + l31;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (wm * RM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int64_t m = m0 + row;
                if (m >= M) continue;
                float v = acc[i][j][r] + bias;
                if (p.res0) v += (n < p.R0) ? p.res0[m * p.R0 + n] : p.res1[m * R1 + (n - p.R0)];
                if (p.out_nchw) {
                    const int64_t bb = m / HWo;
                    p.out[(bb * p.Cout + n) * HWo + (m - bb * HWo)] = v;
                } else {
                    p.out[m * p.Cout + n] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv3x3_halo_ws: the halo kernel with the WEIGHTS STREAMED STRAIGHT INTO REGISTERS.
// The weight tile is identical for every workgroup and every wave needs only its own 64 output
// channels of it, so instead of staging it through LDS (global load -> VGPR -> ds_write -> barrier
// -> ds_read, once per tap) the weights are stored pre-arranged in MFMA B-fragment order
//   Wf[n-block of 32][chunk][tap][k-group][lane][4]
// -- the 36 fragment groups of a chunk, and the chunks after one another, form one linear stream of
// 1-KB pieces per n-block -- and each wave reads its stream with fully coalesced 1-KB loads,
// prefetching one group (1024 MFMA cycles) ahead in a small register ring.  LDS then holds only
// the activation halo, and the per-tap workgroup barrier disappears: waves synchronise twice per
// 32-channel chunk instead of nine times.  Measured motivation (DLPM_ABL ablations, MFMA
// micro-benchmark tools/mb/mfma_loop.hip): the LDS weight staging + per-tap barrier cost ~8 % of the
// kernel.
// ---------------------------------------------------------------------------------------------
// TAPS = 9: the 3x3 convolution.  TAPS = 1: the same kernel as a plain GEMM over pixels for 1x1 convolutions / conv1d /
// Linear (no halo, 4 fragment groups per chunk) -- weights from registers instead of through LDS.
template <int BN, int WAVES_M, int WAVES_N, int RM, int RN, int RING, bool UPS, int TAPS = 9>
__global__ void __launch_bounds__(256, 2) k_conv3x3_halo_ws(ConvLaunch p, int th, int nimg) {
    // UPS: the conv runs on the nearest-x2 upsampled input (Upsample, unet.py:73-75).  The halo tile then holds
    // the SOURCE-resolution patch ((th/2+2) x (W/2+2) pixels) and each lane's tap address is
    // row_offset[ky] + col_offset[kx], which depend on the parity of its output pixel.
    static_assert(WAVES_M * WAVES_N == 4 && WAVES_M * RM * 32 == BM && WAVES_N * RN * 32 == BN, "tile shape");
    static_assert(TAPS == 9 || (TAPS == 1 && !UPS), "taps");
    constexpr int NG = TAPS * 4;  // fragment groups per chunk: taps x 4 k-groups
    constexpr int PADW = TAPS == 9 ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    const int W = p.Wout, H = p.Hout;
    const int Ws = UPS ? (W >> 1) : W, Hs = UPS ? (H >> 1) : H;   // source (staged) resolution
    const int Wp = Ws + 2 * PADW;
    const int hpi = ((UPS ? (th >> 1) : th) + 2 * PADW) * Wp;
    const int hp = nimg * hpi;
    float *Ah = hsm;
    float *Cf = hsm + ((hp + 3) & ~3) * LDS_LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int Cin = p.C0 + p.C1;
    const int HWo = H * W;
    const int ntile_n = (p.Cout + BN - 1) / BN;
    const int64_t M = (int64_t)p.B * HWo;
    const int64_t m0 = (int64_t)(blockIdx.x / ntile_n) * BM;
    const int n0 = (blockIdx.x % ntile_n) * BN;
    const int pb = (int)(m0 / HWo);
    const int y0 = (nimg > 1) ? 0 : (int)((m0 - (int64_t)pb * HWo) / W);
    const int nch = Cin / KC;

    // ---- halo staging (as k_conv3x3_halo)
    const int c4 = (tid & 7) * 4;
    const int nit = (hp * 8 + 255) / 256;
    int off[HALO_NIT];
#pragma unroll
    for (int it = 0; it < HALO_NIT; it++) {
        const int hpix = it * 32 + (tid >> 3);
        const int img = hpix / hpi, hr = hpix - img * hpi;
        const int hy = hr / Wp, hx = hr - hy * Wp;
        const int iy = (UPS ? (y0 >> 1) : y0) + hy - PADW, ix = hx - PADW;
        const bool pad = iy < 0 || iy >= Hs || ix < 0 || ix >= Ws || (pb + img) >= p.B;
        off[it] = (it >= nit || hpix >= hp) ? -2 : (pad ? -1 : (((pb + img) * Hs + iy) * Ws + ix));
    }
    float4 xh[HALO_NIT], cfr;
    const bool has_coef = p.coefA != nullptr;
    const int cf_img = tid >> 4, cf_isb = (tid >> 3) & 1;
    const bool cf_mine = has_coef && tid < nimg * 16 && (pb + cf_img) < p.B;
    const float *cf_base = has_coef ? ((cf_isb ? p.coefB : p.coefA) + (int64_t)min(pb + cf_img, p.B - 1) * Cin) : nullptr;
    auto load_halo = [&](int chunk) {
        const int c = chunk * KC + c4;
        const bool first = c < p.C0;
        const float *sb = first ? p.src0 + c : p.src1 + (c - p.C0);
        const int ld = first ? p.C0 : p.C1;
#pragma unroll
        for (int it = 0; it < HALO_NIT; it++)
            if (it < nit) xh[it] = *reinterpret_cast<const float4 *>(sb + (int64_t)max(off[it], 0) * ld);
        if (has_coef) cfr = *reinterpret_cast<const float4 *>(cf_base + c);
    };
    auto store_coef = [&]() {
        if (cf_mine) *reinterpret_cast<float4 *>(Cf + cf_img * 64 + cf_isb * 32 + c4) = cfr;
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int it = 0; it < HALO_NIT; it++) {
            if (off[it] == -2) continue;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (off[it] >= 0) {
                x = xh[it];
                if (has_coef) {
                    const int img = (nimg > 1) ? (it * 32 + (tid >> 3)) / hpi : 0;
                    const float4 ca = *reinterpret_cast<const float4 *>(Cf + img * 64 + c4);
                    const float4 cb = *reinterpret_cast<const float4 *>(Cf + img * 64 + 32 + c4);
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
            *reinterpret_cast<float4 *>(Ah + (it * 32 + (tid >> 3)) * LDS_LD + c4) = x;
        }
    };

    // ---- weight stream of this wave: RN linear streams of float4-per-lane fragments
    // (one base pointer + integer offsets: an array of advancing pointers degrades to flat_load, whose
    //  out-of-order completion forces vmcnt(0)/lgkmcnt(0) waits and kills the prefetch ring)
    const float4 *__restrict__ wbase = reinterpret_cast<const float4 *>(p.w_frag) + lane;
    int64_t woff[RN];
#pragma unroll
    for (int j = 0; j < RN; j++) woff[j] = (int64_t)((n0 >> 5) + wn * RN + j) * nch * NG * 64;
    constexpr int AHEAD = RING - 1;
    float4 bq[RING][RN];  // ring: group g lives in slot g % RING

    int abase[RM], rowoff[RM][3], coloff[RM][3];
    const int hwt = th * W;
#pragma unroll
    for (int i = 0; i < RM; i++) {
        const int mloc = (wm * RM + i) * 32 + l31;
        const int img = mloc / hwt, ml = mloc - img * hwt;
        const int yy = ml / W, xx = ml - yy * W;
        abase[i] = (img * hpi + yy * Wp + xx) * LDS_LD + kh * 4;
        if (UPS) {
#pragma unroll
            for (int d = 0; d < 3; d++) {   // source halo coordinates of tap row/column d (arithmetic shift: -1 -> -1)
                rowoff[i][d] = (img * hpi + (((yy + d - 1) >> 1) + 1) * Wp) * LDS_LD + kh * 4;
                coloff[i][d] = (((xx + d - 1) >> 1) + 1) * LDS_LD;
            }
        }
    }

    floatx16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    load_halo(0);
#pragma unroll
    for (int j = 0; j < RN; j++) {
#pragma unroll
        for (int a = 0; a < AHEAD; a++) bq[a][j] = wbase[woff[j] + a * 64];
    }
    store_coef();
    __syncthreads();
    store_halo();
    __syncthreads();

    for (int chunk = 0; chunk < nch; chunk++) {
        const bool more = chunk + 1 < nch;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int tap = g >> 2, kk = g & 3;
            // prefetch fragment group g + 2 of the linear stream (runs on into the next chunk; the
            // array is padded by two groups so the very last prefetches stay in bounds)
#pragma unroll
            for (int j = 0; j < RN; j++) {
                // (advancing offset + one small immediate: with static offsets up to 36 KB the compiler keeps a
                //  dozen 64-bit base registers alive, and this kernel is at the register cap)
                bq[(g + AHEAD) % RING][j] = wbase[woff[j] + AHEAD * 64];
                woff[j] += 64;
            }
            if (g == (TAPS == 9 ? 24 : 0) && more) load_halo(chunk + 1);  // 3x3: at tap 6, two taps of MFMAs cover its latency
            // pin the prefetch HERE: left alone, the scheduler sinks each load to just before its first
            // use (two groups later) to save registers and then waits for it with vmcnt(0)
            __builtin_amdgcn_sched_barrier(0);

            const int toff = ((tap / 3) * Wp + (tap % 3)) * LDS_LD + kk * 8;
            float4 af[RM];
#pragma unroll
            for (int i = 0; i < RM; i++)
                af[i] = *reinterpret_cast<const float4 *>(Ah + (UPS ? rowoff[i][tap / 3] + coloff[i][tap % 3] + kk * 8 : abase[i] + toff));
#pragma unroll
            for (int i = 0; i < RM; i++)
#pragma unroll
                for (int j = 0; j < RN; j++) {
                    const float4 b = bq[g % RING][j];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, b.x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, b.y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, b.z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, b.w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            store_coef();
            __syncthreads();  // every wave has finished reading this chunk's halo; coefficients visible
            store_halo();
            __syncthreads();
        }
    }
    __syncthreads();  // the epilogue reuses the halo LDS

    if (!p.out_nchw && (p.Cout & 3) == 0 && (p.R0 & 3) == 0) {
        epilogue_rows<BN, WAVES_M, WAVES_N, RM, RN>(p, acc, hsm, m0, n0, M, tid, wm, wn, l31, kh);
        return;
    }
    const int R1 = p.Cout - p.R0;
#pragma unroll
    for (int j = 0; j < RN; j++) {
        const int n = n0 + (wn * RN + j) * 32 + l31;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (wm * RM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int64_t m = m0 + row;
                if (m >= M) continue;
                float v = acc[i][j][r] + bias;
                if (p.res0) v += (n < p.R0) ? p.res0[m * p.R0 + n] : p.res1[m * R1 + (n - p.R0)];
                if (p.out_nchw) {
                    const int64_t bb = m / HWo;
                    p.out[(bb * p.Cout + n) * HWo + (m - bb * HWo)] = v;
                } else {
                    p.out[m * p.Cout + n] = v;
                }
            }
        }
    }
}

// OIHW (3x3) -> Wf[nb][chunk][tap][kk][lane][4]:  lane = kh*32 + l31 holds
// W[cout = nb*32 + l31][cin = chunk*32 + kk*8 + kh*4 + e][tap], zero beyond Cout.
__global__ void k_relayout_weight_frag(const float *oihw, float *dst, int Cout, int Cin, int taps) {
    const int nbk = (Cout + 31) / 32, nch = Cin / 32;
    const int64_t n = (int64_t)nbk * nch * taps * 4 * 64 * 4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    int64_t r = i >> 8;
    const int kk = (int)(r & 3);
    r >>= 2;
    const int tap = (int)(r % taps);
    r /= taps;
    const int chunk = (int)(r % nch);
    const int nb = (int)(r / nch);
    const int co = nb * 32 + (lane & 31), ci = chunk * 32 + kk * 8 + (lane >> 5) * 4 + e;
    dst[i] = (co < Cout) ? oihw[((int64_t)co * Cin + ci) * taps + tap] : 0.f;
}

// stem: Cin = image channels read from the caller's NCHW state, 3x3 stride 1, writes NHWC.
// One thread = one pixel x 4 output channels.  CIN is a template parameter so that the 9*CIN taps unroll
// into predicated (branch-free) loads issued together; the weight rows are coalesced float4 loads that
// the 32 threads of a pixel share, the inputs are broadcast across them.
template <int CIN>
__global__ void __launch_bounds__(256) k_conv_stem(ConvLaunch p) {
    const int Cq = p.Cout / 4;
    const int HWo = p.Hout * p.Wout;
    const int64_t total = (int64_t)p.B * HWo * Cq;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int nq = (int)(i % Cq);
    const int64_t m = i / Cq;
    const int b = (int)(m / HWo);
    const int rem = (int)(m - (int64_t)b * HWo);
    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
    float v[9 * CIN];
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            const bool ok = iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            const int cy = min(max(iy, 0), p.Hin - 1), cx = min(max(ix, 0), p.Win - 1);
#pragma unroll
            for (int c = 0; c < CIN; c++) {
                const float x = p.src0[(((int64_t)b * CIN + c) * p.Hin + cy) * p.Win + cx];
                v[(ky * 3 + kx) * CIN + c] = ok ? x : 0.f;
            }
        }
    float4 acc = reinterpret_cast<const float4 *>(p.bias)[nq];
#pragma unroll
    for (int k = 0; k < 9 * CIN; k++) {
        const float4 w = reinterpret_cast<const float4 *>(p.w + (int64_t)k * p.Cout)[nq];
        acc.x = fmaf(v[k], w.x, acc.x);
        acc.y = fmaf(v[k], w.y, acc.y);
        acc.z = fmaf(v[k], w.z, acc.z);
        acc.w = fmaf(v[k], w.w, acc.w);
    }
    reinterpret_cast<float4 *>(p.out + m * p.Cout)[nq] = acc;
}

// Stem with the weights held in registers: a thread owns 4 output channels (9*CIN float4 of weights + bias, loaded once)
// and walks over pixels; the 9*CIN inputs of a pixel are the same address for all threads of that pixel (broadcast).
// The first version re-read its 27 weight float4 per output: 0.53 ms for [1024,3,32,32] -> 128 channels, where the
// 537-MB output write alone is 0.1 ms.
template <int CIN>
__global__ void __launch_bounds__(256) k_conv_stem_regw(ConvLaunch p, int ppb) {
    const int Cq = p.Cout >> 2;                 // channel quads: a power of two <= 256
    const int pl_n = 256 / Cq;                  // pixels in flight per block
    const int nq = threadIdx.x & (Cq - 1), pl = threadIdx.x / Cq;
    const int HWo = p.Hout * p.Wout;
    const int64_t M = (int64_t)p.B * HWo;
    float4 w[9 * CIN];
#pragma unroll
    for (int k = 0; k < 9 * CIN; k++) w[k] = reinterpret_cast<const float4 *>(p.w + (int64_t)k * p.Cout)[nq];
    const float4 bias = reinterpret_cast<const float4 *>(p.bias)[nq];
    const int64_t m_lo = (int64_t)blockIdx.x * ppb;
    // fused GroupNorm statistics of the output (launcher: only when an image is a whole number of blocks): every thread
    // keeps shifted sums for its 4 channels over its ppb / pl_n pixels
    const bool do_stats = p.stats_out != nullptr;
    float4 K = make_float4(0.f, 0.f, 0.f, 0.f), s1 = K, s2 = K;
    for (int it = 0; it < ppb / pl_n; it++) {
        const int64_t m = m_lo + it * pl_n + pl;
        if (m >= M) break;
        const int b = (int)(m / HWo);
        const int rem = (int)(m - (int64_t)b * HWo);
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        float v[9 * CIN];
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int iy = oy + ky - 1, ix = ox + kx - 1;
                const bool ok = iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                const int cy = min(max(iy, 0), p.Hin - 1), cx = min(max(ix, 0), p.Win - 1);
#pragma unroll
                for (int c = 0; c < CIN; c++) {
                    const float x = p.src0[(((int64_t)b * CIN + c) * p.Hin + cy) * p.Win + cx];
                    v[(ky * 3 + kx) * CIN + c] = ok ? x : 0.f;
                }
            }
        float4 acc = bias;
#pragma unroll
        for (int k = 0; k < 9 * CIN; k++) {
            acc.x = fmaf(v[k], w[k].x, acc.x);
            acc.y = fmaf(v[k], w[k].y, acc.y);
            acc.z = fmaf(v[k], w[k].z, acc.z);
            acc.w = fmaf(v[k], w[k].w, acc.w);
        }
        reinterpret_cast<float4 *>(p.out + m * p.Cout)[nq] = acc;
        if (do_stats) {
            if (it == 0) K = acc;  // pivot = this thread's first value per channel
            float d;
            d = acc.x - K.x; s1.x += d; s2.x = fmaf(d, d, s2.x);
            d = acc.y - K.y; s1.y += d; s2.y = fmaf(d, d, s2.y);
            d = acc.z - K.z; s1.z += d; s2.z = fmaf(d, d, s2.z);
            d = acc.w - K.w; s1.w += d; s2.w = fmaf(d, d, s2.w);
        }
    }
    if (do_stats) {   // per-thread (mean, M2) -> LDS [pl_n][Cout] -> threads < Cout merge the pl_n partials (Chan)
        __shared__ float2 part[1024];
        const float fc = (float)(ppb / pl_n);
        const float mx = s1.x / fc, my = s1.y / fc, mz = s1.z / fc, mw = s1.w / fc;
        part[pl * p.Cout + nq * 4 + 0] = make_float2(K.x + mx, fmaxf(s2.x - s1.x * mx, 0.f));
        part[pl * p.Cout + nq * 4 + 1] = make_float2(K.y + my, fmaxf(s2.y - s1.y * my, 0.f));
        part[pl * p.Cout + nq * 4 + 2] = make_float2(K.z + mz, fmaxf(s2.z - s1.z * mz, 0.f));
        part[pl * p.Cout + nq * 4 + 3] = make_float2(K.w + mw, fmaxf(s2.w - s1.w * mw, 0.f));
        __syncthreads();
        if ((int)threadIdx.x < p.Cout) {
            float mean = part[threadIdx.x].x, M2 = part[threadIdx.x].y, na = fc;
            for (int g = 1; g < pl_n; g++) {
                const float2 q = part[g * p.Cout + threadIdx.x];
                const float d = q.x - mean, N = na + fc;
                mean += d * (fc / N);
                M2 += q.y + d * d * (na * fc / N);
                na = N;
            }
            p.stats_out[(int64_t)blockIdx.x * p.Cout + threadIdx.x] = make_float2(mean, M2);
        }
    }
}

// Stem, third generation: the block's input rows (+ halo, zero padding materialised) go through LDS once, channel-last
// [row][x][4], so a thread's 9 * CIN inputs are nine 16-byte LDS reads at constant offsets (broadcast: the Cout/4 threads of a
// pixel read the same address).  k_conv_stem_regw computed a clamped 64-bit global address per (tap, channel) and pixel:
// ~160 integer instructions next to 108 FMAs per thread and pixel, 0.36 ms for [1024,3,32,32] -> 128 channels where the
// 537-MB output write is 0.1 ms.  A block owns `ppb` = 1024 consecutive pixels = whole rows of ONE image (H W % 1024 == 0,
// W a power of two <= 64): weights in registers, output transform-free, fused GroupNorm statistics as before.
template <int CIN>
__global__ void __launch_bounds__(256) k_conv_stem_lds(ConvLaunch p, int ppb) {
    extern __shared__ __attribute__((aligned(16))) float4 patch[];   // [(rows + 2)][(W + 2)]
    const int Cq = p.Cout >> 2;                 // channel quads: a power of two <= 256
    const int pl_n = 256 / Cq;                  // pixels in flight per block
    const int nq = threadIdx.x & (Cq - 1), pl = threadIdx.x / Cq;
    const int W = p.Wout, H = p.Hout, HWo = H * W, W2 = W + 2;
    const int rows = ppb / W, bpi = HWo / ppb;
    const int b = blockIdx.x / bpi, y0 = (blockIdx.x % bpi) * rows;
    float4 w[9 * CIN];
#pragma unroll
    for (int k = 0; k < 9 * CIN; k++) w[k] = reinterpret_cast<const float4 *>(p.w + (int64_t)k * p.Cout)[nq];
    const float4 bias = reinterpret_cast<const float4 *>(p.bias)[nq];
    const float *xb = p.src0 + (int64_t)b * CIN * HWo;
    for (int idx = threadIdx.x; idx < (rows + 2) * W2; idx += 256) {
        const int hy = idx / W2, hx = idx - hy * W2;
        const int iy = y0 + hy - 1, ix = hx - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const float *s = xb + iy * W + ix;
            v.x = s[0];
            if (CIN > 1) v.y = s[HWo];
            if (CIN > 2) v.z = s[2 * HWo];
        }
        patch[idx] = v;
    }
    __syncthreads();
    const bool do_stats = p.stats_out != nullptr;
    float4 K = make_float4(0.f, 0.f, 0.f, 0.f), s1 = K, s2 = K;
    const int lw = 31 - __builtin_clz(W);       // W is a power of two
    float *ob = p.out + ((int64_t)b * HWo + (int64_t)y0 * W) * p.Cout;
    for (int it = 0; it < ppb / pl_n; it++) {
        const int lp = it * pl_n + pl;
        const int ly = lp >> lw, lx = lp & (W - 1);
        const float4 *src = patch + ly * W2 + lx;
        float4 acc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const float4 t = src[ky * W2 + kx];
                const float v[3] = {t.x, t.y, t.z};
#pragma unroll
                for (int c = 0; c < CIN; c++) {
                    const float4 ww = w[(ky * 3 + kx) * CIN + c];
                    acc.x = fmaf(v[c], ww.x, acc.x);
                    acc.y = fmaf(v[c], ww.y, acc.y);
                    acc.z = fmaf(v[c], ww.z, acc.z);
                    acc.w = fmaf(v[c], ww.w, acc.w);
                }
            }
        reinterpret_cast<float4 *>(ob + (int64_t)lp * p.Cout)[nq] = acc;
        if (do_stats) {
            if (it == 0) K = acc;  // pivot = this thread's first value per channel
            float d;
            d = acc.x - K.x; s1.x += d; s2.x = fmaf(d, d, s2.x);
            d = acc.y - K.y; s1.y += d; s2.y = fmaf(d, d, s2.y);
            d = acc.z - K.z; s1.z += d; s2.z = fmaf(d, d, s2.z);
            d = acc.w - K.w; s1.w += d; s2.w = fmaf(d, d, s2.w);
        }
    }
    if (do_stats) {   // per-thread (mean, M2) -> LDS [pl_n][Cout] -> threads < Cout merge the pl_n partials (Chan)
        __syncthreads();   // the patch is dead: its LDS carries the partials
        float2 *part = reinterpret_cast<float2 *>(patch);
        const float fc = (float)(ppb / pl_n);
        const float mx = s1.x / fc, my = s1.y / fc, mz = s1.z / fc, mw = s1.w / fc;
        part[pl * p.Cout + nq * 4 + 0] = make_float2(K.x + mx, fmaxf(s2.x - s1.x * mx, 0.f));
        part[pl * p.Cout + nq * 4 + 1] = make_float2(K.y + my, fmaxf(s2.y - s1.y * my, 0.f));
        part[pl * p.Cout + nq * 4 + 2] = make_float2(K.z + mz, fmaxf(s2.z - s1.z * mz, 0.f));
        part[pl * p.Cout + nq * 4 + 3] = make_float2(K.w + mw, fmaxf(s2.w - s1.w * mw, 0.f));
        __syncthreads();
        if ((int)threadIdx.x < p.Cout) {
            float mean = part[threadIdx.x].x, M2 = part[threadIdx.x].y, na = fc;
            for (int g = 1; g < pl_n; g++) {
                const float2 q = part[g * p.Cout + threadIdx.x];
                const float d = q.x - mean, N = na + fc;
                mean += d * (fc / N);
                M2 += q.y + d * d * (na * fc / N);
                na = N;
            }
            p.stats_out[(int64_t)blockIdx.x * p.Cout + threadIdx.x] = make_float2(mean, M2);
        }
    }
}

__global__ void k_relayout_weight(const float *oihw, float *dst, int Cout, int Cin, int ks, int for_igemm) {
    const int64_t n = (int64_t)Cout * Cin * ks * ks;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int taps = ks * ks;
    int tap = (int)(i % taps);
    int64_t r = i / taps;
    int ci = (int)(r % Cin);
    int co = (int)(r / Cin);
    int64_t d = for_igemm ? ((int64_t)tap * Cout + co) * Cin + ci : ((int64_t)tap * Cin + ci) * Cout + co;
    dst[d] = oihw[i];
}

}  // namespace

bool igemm_supported(const ConvLaunch &c) {
    const int Cin = c.C0 + c.C1;
    if (c.in_nchw) return false;
    if (Cin % KC != 0 || c.C0 % KC != 0) return false;
    if (c.Cout < 16 && !c.out_nchw) return false;
    if (c.ks != 1 && c.ks != 3) return false;
    return true;
}

static bool ws_disabled() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLPM_NO_WS"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

static bool halo_ok(const ConvLaunch &c, int *th, int *nimg) {
    static int disabled = -1;
    if (disabled < 0) { const char *e = getenv("DLPM_NO_HALO"); disabled = (e && e[0] == '1') ? 1 : 0; }
    if (disabled || c.ks != 3 || c.stride != 1) return false;
    if (c.ups && (!c.w_frag || ws_disabled() || (c.Wout & 1) || (c.Hout & 1))) return false;   // upsample: weight-streaming kernel only
    const int W = c.Wout, HW = c.Hout * c.Wout;
    if (W < 4 || W > 64 || BM % W != 0) return false;
    if (HW >= BM) {            // th full rows of one image
        if (HW % BM != 0) return false;
        *th = BM / W;
        *nimg = 1;
    } else {                   // several whole (small) images per tile
        if (BM % HW != 0) return false;
        *th = c.Hout;
        *nimg = BM / HW;
    }
    if (c.ups && ((*th & 1) || (c.C0 + c.C1) % KC != 0 || c.abl)) return false;
    return *nimg * (*th + 2) * (W + 2) * 8 <= HALO_NIT * 256;
}

template <int BN, int WAVES_M, int WAVES_N, int RM, int RN, int RING, bool UPS, int TAPS = 9>
static int launch_halo_ws_r(const ConvLaunch &c, int th, int nimg, int64_t grid, hipStream_t st) {
    constexpr int PADW = TAPS == 9 ? 1 : 0;
    const int hp = nimg * ((UPS ? th / 2 : th) + 2 * PADW) * ((UPS ? c.Wout / 2 : c.Wout) + 2 * PADW);
    size_t shmem = (size_t)((hp + 3) & ~3) * LDS_LD * sizeof(float) + (size_t)nimg * 64 * sizeof(float);
    const size_t epi = (size_t)(RM * 32) * (BN + 4) * sizeof(float);   // epilogue_rows' row image
    const size_t stats = (size_t)(256 / (BN / 4)) * BN * 2 * sizeof(float);
    if (shmem < epi) shmem = epi;
    if (shmem < stats) shmem = stats;
    {
        int r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_conv3x3_halo_ws<BN, WAVES_M, WAVES_N, RM, RN, RING, UPS, TAPS>), 64 * 1024);
        if (r != DLPM_OK) return r;
    }
    k_conv3x3_halo_ws<BN, WAVES_M, WAVES_N, RM, RN, RING, UPS, TAPS><<<(unsigned)grid, 256, shmem, st>>>(c, th, nimg);
    return DLPM_OK;
}

template <int BN, int WAVES_M, int WAVES_N, int RM, int RN>
static int launch_halo_ws(const ConvLaunch &c, int th, int nimg, int64_t grid, hipStream_t st) {
    // ring depth 2 = one fragment group (16 MFMAs = 1024 cycles) of prefetch distance; 3 measured equal
    if (c.ups) return launch_halo_ws_r<BN, WAVES_M, WAVES_N, RM, RN, 2, true>(c, th, nimg, grid, st);
    return launch_halo_ws_r<BN, WAVES_M, WAVES_N, RM, RN, 2, false>(c, th, nimg, grid, st);
}

// 1x1 / Linear through the weight-streaming kernel (TAPS = 1): geometry of the 128-pixel tile
static bool gemm_ws_ok(const ConvLaunch &c, int *th, int *nimg) {
    // opt-in (ConvLaunch::ws_gemm): measured on the CIFAR net it is 1-10 % SLOWER than k_conv_igemm for every 1x1 shape
    // (91 vs 101 TFLOP/s at H32, 128+128 -> 128): with one tap there are only 64 MFMAs per wave between two barriers,
    // and the LDS weight staging it removes was not what limits these launches
    if (!c.ws_gemm || ws_disabled() || !c.w_frag || c.ks != 1 || c.stride != 1 || c.ups || c.in_nchw || c.abl) return false;
    if ((c.C0 + c.C1) % KC != 0 || (c.C0 & 3) || c.Hin != c.Hout || c.Win != c.Wout) return false;
    const int W = c.Wout, HW = c.Hout * c.Wout;
    if (HW >= BM) {
        if (HW % BM != 0 || BM % W != 0) return false;
        *th = BM / W;
        *nimg = 1;
    } else {
        if (BM % HW != 0) return false;
        *th = c.Hout;
        *nimg = BM / HW;
        if (c.coefA && *nimg > 16) return false;   // 16 threads per image load the GroupNorm coefficients
    }
    return true;
}

template <int BN, int WAVES_M, int WAVES_N, int RM, int RN>
static int launch_gemm_ws(const ConvLaunch &c, int th, int nimg, int64_t grid, hipStream_t st) {
    return launch_halo_ws_r<BN, WAVES_M, WAVES_N, RM, RN, 2, false, 1>(c, th, nimg, grid, st);
}

template <int BN, int WAVES_M, int WAVES_N, int RM, int RN>
static int launch_halo(const ConvLaunch &c, int th, int nimg, int64_t grid, hipStream_t st) {
    const int hp = nimg * (th + 2) * (c.Wout + 2);
    const size_t shmem = (size_t)(((hp + 3) & ~3) + 2 * BN) * LDS_LD * sizeof(float) + (size_t)nimg * 64 * sizeof(float);
    {
        int r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_conv3x3_halo<BN, WAVES_M, WAVES_N, RM, RN>), 100 * 1024);
        if (r != DLPM_OK) return r;
    }
    k_conv3x3_halo<BN, WAVES_M, WAVES_N, RM, RN><<<(unsigned)grid, 256, shmem, st>>>(c, th, nimg);
    return DLPM_OK;
}

// the register-weight stem kernel emits one partial per 1024-pixel block
static bool stem_stats_ok(const ConvLaunch &c) {
    const int Cq = c.Cout / 4;
    return c.in_nchw && !c.out_nchw && c.ks == 3 && c.stride == 1 && !c.ups && c.C1 == 0 && (c.C0 == 3 || c.C0 == 1) && c.Cout % 4 == 0 &&
           Cq >= 1 && Cq <= 256 && (Cq & (Cq - 1)) == 0 && ((int64_t)c.Hout * c.Wout) % 1024 == 0;
}

int launch_conv_stem(const ConvLaunch &c, hipStream_t st) {
    const int64_t total = (int64_t)c.B * c.Hout * c.Wout * (c.Cout / 4);
    ProfScope ps("conv_stem", 2.0 * total * 4 * c.C0 * 9, 4.0 * ((double)c.B * c.Hin * c.Win * c.C0 + total * 4.0), st);
    const int Cq = c.Cout / 4;
    const bool regw = Cq >= 1 && Cq <= 256 && (Cq & (Cq - 1)) == 0;   // power-of-two channel quads
    const int64_t Mpix = (int64_t)c.B * c.Hout * c.Wout;
    const int ppb = 1024;                                              // pixels per block
    if (c.stats_out && !stem_stats_ok(c)) {
        set_error("launch_conv_stem: statistics requested for a shape that cannot emit them");
        return DLPM_ERR_UNSUPPORTED;
    }
    // whole rows of one image per block, W a power of two: the input patch goes through LDS (k_conv_stem_lds)
    static int nolds = -1;
    if (nolds < 0) { const char *e = getenv("DLPM_NO_STEM_LDS"); nolds = (e && e[0] == '1') ? 1 : 0; }
    const int W = c.Wout;
    const bool lds_ok = !nolds && regw && c.Hin == c.Hout && c.Win == c.Wout && (W & (W - 1)) == 0 && W >= 4 && W <= 64 && ppb % W == 0 &&
                        ((int64_t)c.Hout * W) % ppb == 0 && c.bias;
    if (lds_ok && (c.C0 == 3 || c.C0 == 1)) {
        // LDS: the (rows + 2) x (W + 2) float4 patch, or the [pixels in flight][Cout] float2 statistics partials, whichever is larger
        const size_t shmem = std::max((size_t)(ppb / W + 2) * (W + 2) * sizeof(float4), (size_t)(256 / Cq) * c.Cout * sizeof(float2));
        if (c.C0 == 3) k_conv_stem_lds<3><<<(unsigned)(Mpix / ppb), 256, shmem, st>>>(c, ppb);
        else k_conv_stem_lds<1><<<(unsigned)(Mpix / ppb), 256, shmem, st>>>(c, ppb);
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
    if (c.C0 == 3 && regw) k_conv_stem_regw<3><<<(unsigned)ceil_div(Mpix, ppb), 256, 0, st>>>(c, ppb);
    else if (c.C0 == 1 && regw) k_conv_stem_regw<1><<<(unsigned)ceil_div(Mpix, ppb), 256, 0, st>>>(c, ppb);
    else if (c.C0 == 3) k_conv_stem<3><<<(unsigned)ceil_div(total, 256), 256, 0, st>>>(c);
    else if (c.C0 == 1) k_conv_stem<1><<<(unsigned)ceil_div(total, 256), 256, 0, st>>>(c);
    else return launch_conv_direct(c, st);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int conv_stats_pixels(const ConvLaunch &c) {
    int a, b, n;
    if (conv_ksplit_for(c) > 1) return 0;      // a split-K launch emits partial outputs: no statistics (the consumer's GroupNorm reads the tensor)
    if (c.in_nchw) return stem_stats_ok(c) ? 1024 : 0;
    // same order as launch_conv_igemm's dispatch: a launch that carries split weights runs k_conv_split whatever else it carries
    // (k_conv_split: per 128-pixel tile through the 4-wave row epilogue, or per whole 8x8 image from the registers of the 8 / 16-wave shapes)
    if (conv_split_ok(c)) {
        const int64_t hw = (int64_t)c.Hout * c.Wout;
        return (c.R0 & 3) ? 0 : hw % BM == 0 ? BM : hw == 64 ? 64 : 0;
    }
    if (wino4_preferred(c, &a, &b, &n)) return n == 1 ? 256 : (n == 4 && a * b == 4 && wino4_image_stats(c.Cout)) ? 64 : 0;   // (four whole 8x8 images per block: one partial per image)
    if (wino_geometry(c, &a, &b, &n)) return n == 1 ? 4 * wino_tiles(c) : 0;
    if (c.out_nchw || (c.Cout & 3) || (c.R0 & 3)) return 0;
    return ((int64_t)c.Hout * c.Wout) % BM == 0 ? BM : 0;
}

int launch_conv_igemm(const ConvLaunch &c, hipStream_t st) {
    const int64_t M = (int64_t)c.B * c.Hout * c.Wout;
    const int64_t mt = ceil_div(M, BM);
    const double K = (double)(c.C0 + c.C1) * c.ks * c.ks;
    // algorithmic bytes: input once + weights once + output once (+ residual)
    const double bytes = 4.0 * ((double)c.B * c.Hin * c.Win * (c.C0 + c.C1) + K * c.Cout + (double)M * c.Cout * (c.res0 ? 2 : 1));
    char pname[96];
    int th = 0, nimg = 1;
    if (prof_enabled() && prof_detail())
        snprintf(pname, sizeof(pname), "conv%dx%d_igemm:H%d:Cin%d+%d:Cout%d:s%d:u%d:coef%d", c.ks, c.ks, c.Hout, c.C0, c.C1, c.Cout,
                 c.stride, c.ups, c.coefA ? 1 : 0);
    else
        snprintf(pname, sizeof(pname), "%s", c.ks == 3 ? (halo_ok(c, &th, &nimg) ? "conv3x3_halo" : "conv3x3_igemm") : "conv1x1_igemm");
    if (conv_split_ok(c)) {
        if (prof_enabled() && prof_detail())
            snprintf(pname, sizeof(pname), "conv%dx%d_bf16x3:H%d:Cin%d+%d:Cout%d:s%d:coef%d", c.ks, c.ks, c.Hout, c.C0, c.C1, c.Cout, c.stride,
                     c.coefA ? 1 : 0);
        else
            snprintf(pname, sizeof(pname), c.ks == 1 ? "conv1x1_bf16x3" : "conv3x3_bf16x3");
        // bytes as the kernel moves them: the weights are 6 B per element here
        ProfScope pss(pname, 2.0 * M * c.Cout * K, bytes + 2.0 * K * c.Cout, st);
#ifdef DLPM_PHASE_TIMING
        const_cast<ConvLaunch &>(c).phase = phase_buffer();
#endif
#ifdef DLPM_IGEMM_ABLATIONS
        { const char *e = getenv("DLPM_ABL"); if (e) const_cast<ConvLaunch &>(c).abl = atoi(e); }
#endif
        return launch_conv_split(c, st);
    }
    {
        int wb, ww, wi;
        if (wino4_preferred(c, &wb, &ww, &wi)) {
            if (prof_enabled() && prof_detail())
                snprintf(pname, sizeof(pname), "conv3x3_wino4:H%d:Cin%d+%d:Cout%d:u%d:coef%d", c.Hout, c.C0, c.C1, c.Cout, c.ups, c.coefA ? 1 : 0);
            else
                snprintf(pname, sizeof(pname), "conv3x3_wino4");
            ProfScope psw(pname, 2.0 * M * c.Cout * K, bytes, st);   // ALGORITHMIC flops (direct-conv count)
            return launch_conv_wino4(c, st);
        }
        if (wino_geometry(c, &wb, &ww, &wi)) {
            if (prof_enabled() && prof_detail())
                snprintf(pname, sizeof(pname), "conv3x3_wino:H%d:Cin%d+%d:Cout%d:u%d:coef%d", c.Hout, c.C0, c.C1, c.Cout, c.ups, c.coefA ? 1 : 0);
            else
                snprintf(pname, sizeof(pname), "conv3x3_wino");
            ProfScope psw(pname, 2.0 * M * c.Cout * K, bytes, st);   // ALGORITHMIC flops (direct-conv count)
            return launch_conv_wino(c, st);
        }
    }
    ProfScope ps(pname, 2.0 * M * c.Cout * K, bytes, st);
#ifdef DLPM_IGEMM_ABLATIONS   // developer builds only (DLPM_BUILD_DEFS): timing ablations, results are WRONG when set
    static int abl = -1;
    if (abl < 0) { const char *e = getenv("DLPM_ABL"); abl = e ? atoi(e) : 0; }
    if (abl) const_cast<ConvLaunch &>(c).abl = abl;
#endif
#ifdef DLPM_PHASE_TIMING
    const_cast<ConvLaunch &>(c).phase = phase_buffer();
#endif
    if (gemm_ws_ok(c, &th, &nimg)) {
        int r;
        if (c.Cout > 64) r = launch_gemm_ws<128, 2, 2, 2, 2>(c, th, nimg, mt * ceil_div(c.Cout, 128), st);
        else if (c.Cout > 32) r = launch_gemm_ws<64, 2, 2, 2, 1>(c, th, nimg, mt * ceil_div(c.Cout, 64), st);
        else r = launch_gemm_ws<32, 4, 1, 1, 1>(c, th, nimg, mt * ceil_div(c.Cout, 32), st);
        if (r != DLPM_OK) return r;
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
    if (halo_ok(c, &th, &nimg)) {
        int r;
        if (c.w_frag && !ws_disabled() && !c.abl && (c.C0 + c.C1) % KC == 0) {
            if (c.Cout > 64) r = launch_halo_ws<128, 2, 2, 2, 2>(c, th, nimg, mt * ceil_div(c.Cout, 128), st);
            else if (c.Cout > 32) r = launch_halo_ws<64, 2, 2, 2, 1>(c, th, nimg, mt * ceil_div(c.Cout, 64), st);
            else r = launch_halo_ws<32, 4, 1, 1, 1>(c, th, nimg, mt * ceil_div(c.Cout, 32), st);
            if (r != DLPM_OK) return r;
            DLPM_LAUNCH_CHECK();
            return DLPM_OK;
        }
        if (c.Cout > 64) r = launch_halo<128, 2, 2, 2, 2>(c, th, nimg, mt * ceil_div(c.Cout, 128), st);
        else if (c.Cout > 32) r = launch_halo<64, 2, 2, 2, 1>(c, th, nimg, mt * ceil_div(c.Cout, 64), st);
        else r = launch_halo<32, 4, 1, 1, 1>(c, th, nimg, mt * ceil_div(c.Cout, 32), st);
        if (r != DLPM_OK) return r;
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
    // full tiles take the lean epilogue (MODE 1); 1x1 NHWC convolutions also the pointer-walking gather (MODE 2)
    static int nofast = -1;
    if (nofast < 0) { const char *e = getenv("DLPM_NO_FAST_IGEMM"); nofast = (e && e[0] == '1') ? 1 : 0; }
    const int bn = c.Cout > 64 ? 128 : c.Cout > 32 ? 64 : 32;
    const bool full = !nofast && !c.abl && M % BM == 0 && c.Cout % bn == 0 && !c.out_nchw && (c.R0 & 3) == 0;
    const bool g1 = full && c.ks == 1 && c.stride == 1 && !c.ups && !c.in_nchw && c.C0 % KC == 0 && (c.C0 + c.C1) % KC == 0;
    const int mode = g1 ? 2 : full ? 1 : 0;
    const unsigned grid = (unsigned)(mt * ceil_div(c.Cout, bn));
#define DLPM_IGEMM_LAUNCH(BN_, WM_, WN_, RM_, RN_)                                                   \
    do {                                                                                             \
        if (mode == 2) k_conv_igemm<BN_, WM_, WN_, RM_, RN_, 2><<<grid, 256, 0, st>>>(c);            \
        else if (mode == 1) k_conv_igemm<BN_, WM_, WN_, RM_, RN_, 1><<<grid, 256, 0, st>>>(c);       \
        else k_conv_igemm<BN_, WM_, WN_, RM_, RN_, 0><<<grid, 256, 0, st>>>(c);                      \
    } while (0)
    if (bn == 128) DLPM_IGEMM_LAUNCH(128, 2, 2, 2, 2);
    else if (bn == 64) DLPM_IGEMM_LAUNCH(64, 2, 2, 2, 1);
    else DLPM_IGEMM_LAUNCH(32, 4, 1, 1, 1);
#undef DLPM_IGEMM_LAUNCH
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int64_t frag_weight_floats(int Cout, int Cin, int taps) {
    // + 2 groups of padding at the end: the 2-ahead prefetch of the last groups reads past the data
    return ((int64_t)((Cout + 31) / 32) * (Cin / 32) * taps * 4 + 2) * 256;
}

int relayout_weight_frag(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st, int taps) {
    const int64_t n = (int64_t)((Cout + 31) / 32) * (Cin / 32) * taps * 4 * 256;
    DLPM_HIP(hipMemsetAsync(dst_dev + n, 0, 2 * 256 * sizeof(float), st));
    k_relayout_weight_frag<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin, taps);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int relayout_weight(const float *oihw_dev, float *dst_dev, int Cout, int Cin, int ks, bool for_igemm, hipStream_t st) {
    const int64_t n = (int64_t)Cout * Cin * ks * ks;
    k_relayout_weight<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin, ks, for_igemm ? 1 : 0);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
