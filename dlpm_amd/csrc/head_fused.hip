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
#include "conv.h"
#include "philox.h"

namespace dlpm {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int HF_NT = 512;        // 8 waves
constexpr int HF_SLD = 36;        // floats per pixel row of a wave's transpose patch (32 channels + 4: conflict-free b128 reads)
constexpr int HF_MAXCH = 4;       // K chunks of 32 channels the register-resident weights cover (Cin <= 128)

#define HF_LDS_EXCHANGE()                                             \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
        __builtin_amdgcn_wave_barrier();                              \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); \
    } while (0)

struct HeadFusedArgs {
    const float *h;               // [B][H][W][C] NHWC
    const float *coefA, *coefB;   // [B][C] GroupNorm affine (SiLU follows)
    const float *wf;              // W' in fragment order [C/32][4][64][4]
    const float *bias;            // [Cout]
    float *out;                   // eps (plain forward): NCHW [B][Cout][H][W] or NHWC
    int out_nchw;
    int B, H, W, C, TH;
    HeadUpdate u;
};

// NCH = C / 32 K chunks, a compile-time constant: with a run-time bound the chunk loop's loads sit inside (uniform) branches and hipcc
// drains them at every join (s_waitcnt vmcnt(0)).
template <int COUT, int NCH>
__global__ void __launch_bounds__(HF_NT, 1) k_head_fused(HeadFusedArgs p) {
    constexpr int NV = 9 * COUT;                       // live tap channels (27), also the LDS pitch of P (odd: bank spread)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.W, H = p.H, TH = p.TH;
    constexpr int C = 32 * NCH;
    const int bands = H / TH, b = blockIdx.x / bands, y0 = (blockIdx.x - b * bands) * TH;
    const int R = TH + 2;                              // row slots of P: image rows y0 - 1 .. y0 + TH
    float *P = sm;                                     // [R][W][NV]
    float *stg = sm + ((R * W * NV + 3) & ~3) + wave * (32 * HF_SLD);
    const int tpr = W >> 5;                            // 32-pixel tiles per row
    const int64_t HW = (int64_t)H * W;

    // ---- W' fragments and this image's GroupNorm coefficients: registers for the whole kernel
    float4 bw[NCH][4], cA[NCH], cB[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
#pragma unroll
        for (int q = 0; q < 4; q++) bw[c][q] = reinterpret_cast<const float4 *>(p.wf)[(c * 4 + q) * 64 + lane];
        cA[c] = *reinterpret_cast<const float4 *>(p.coefA + (int64_t)b * C + 32 * c + 4 * (lane & 7));
        cB[c] = *reinterpret_cast<const float4 *>(p.coefB + (int64_t)b * C + 32 * c + 4 * (lane & 7));
    }
    // rows of the slot image that lie outside the picture are zero (the taps that would read them are simply absent)
    for (int s = 0; s < R; s++) {
        const int iy = y0 - 1 + s;
        if (iy < 0 || iy >= H)
            for (int i = tid; i < W * NV; i += HF_NT) P[s * W * NV + i] = 0.f;
    }

    // ---- phase A: the band's rows as 32-pixel tiles, round-robin over the waves
    const int s_lo = y0 == 0 ? 1 : 0, s_hi = (y0 + TH == H) ? R - 1 : R;      // slots with a real image row
    const int ntile = (s_hi - s_lo) * tpr;
    const int lp = lane >> 3, lc = lane & 7;           // load role: pixel 8 i + lp, channels 4 lc .. 4 lc + 3 of the chunk
    const int lm = lane & 31, kh = lane >> 5;          // MFMA role: pixel lm, k slot kh
    auto tile_src = [&](int t) {
        const int s = s_lo + t / tpr, x0 = (t - (t / tpr) * tpr) * 32;
        return p.h + (((int64_t)b * H + (y0 - 1 + s)) * W + x0) * C + 4 * lc;
    };
    // Input chunks travel ONE TILE ahead: buffer c holds chunk c of the tile being worked on and is reloaded with chunk c of the
    // wave's next tile as soon as it has been consumed (16 KB per wave, ~32 MB per chip in flight: what 5 TB/s x the loaded HBM
    // latency needs).  Latency is not what bounds the kernel, as it turned out -- one chunk ahead measured the same 0.17 ms:
    // MFMA 40 % + VALU 36 % of its cycles at 1.96 GHz (profiles/r04/head_fused/)
    float4 xb[NCH][4];
    if (wave < ntile) {
        const float *src = tile_src(wave);
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) xb[c][i] = *reinterpret_cast<const float4 *>(src + 32 * c + (int64_t)(8 * i + lp) * C);
    }
    for (int t = wave; t < ntile; t += 8) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        const float *nsrc = tile_src(t + 8 < ntile ? t + 8 : t);        // (the last tile reloads itself: unconditional loads)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 xr = xb[c][i];
                    float4 v;
                    v.x = silu_f(fmaf(xr.x, cA[c].x, cB[c].x));
                    v.y = silu_f(fmaf(xr.y, cA[c].y, cB[c].y));
                    v.z = silu_f(fmaf(xr.z, cA[c].z, cB[c].z));
                    v.w = silu_f(fmaf(xr.w, cA[c].w, cB[c].w));
                    *reinterpret_cast<float4 *>(stg + (8 * i + lp) * HF_SLD + 4 * lc) = v;
                }
#pragma unroll
                for (int i = 0; i < 4; i++) xb[c][i] = *reinterpret_cast<const float4 *>(nsrc + 32 * c + (int64_t)(8 * i + lp) * C);
                // wave-private exchange: LDS operations of one wave execute in order; the LDS-only fences keep the compiler from
                // moving them (a plain wavefront fence also drains the GLOBAL loads in flight: the prefetch)
                HF_LDS_EXCHANGE();
                float4 a[4];
#pragma unroll
                for (int q = 0; q < 4; q++) a[q] = *reinterpret_cast<const float4 *>(stg + lm * HF_SLD + 8 * q + 4 * kh);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bw[c][q].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bw[c][q].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bw[c][q].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bw[c][q].w, acc, 0, 0, 0);
                }
                HF_LDS_EXCHANGE();
            }
        }
        // D layout of the 32x32 MFMA: register i holds row 8 (i / 4) + 4 kh + (i % 4) (pixel), column lm (tap channel)
        if (lm < NV) {
            const int s = s_lo + t / tpr, x0 = (t - (t / tpr) * tpr) * 32;
            float *dst = P + ((int64_t)s * W + x0) * NV + lm;
#pragma unroll
            for (int i = 0; i < 16; i++) dst[(8 * (i >> 2) + 4 * kh + (i & 3)) * NV] = acc[i];
        }
    }
    __syncthreads();

    // ---- phase B: 9-point gather from LDS + the reverse update (k_head_gather's arithmetic, bit for bit)
    const int nq = W >> 2, per_co = TH * nq, nitem = COUT * per_co;
    const int64_t D = (int64_t)COUT * HW;
    for (int it = tid; it < nitem; it += HF_NT) {
        const int co = it / per_co, rq = it - co * per_co, r = rq / nq, q = rq - r * nq;
        const float bv = p.bias ? p.bias[co] : 0.f;
        float acc[4] = {bv, bv, bv, bv};
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const float *srow = P + ((int64_t)(r + ky) * W) * NV + (ky * 3 + kx) * COUT + co;
#pragma unroll
                for (int px = 0; px < 4; px++) {
                    const int ix = 4 * q + px + kx - 1;
                    if (ix >= 0 && ix < W) acc[px] += srow[ix * NV];
                }
            }
        const int64_t pix = (int64_t)(y0 + r) * W + 4 * q;
        const int64_t e0 = (int64_t)co * HW + pix;
        const HeadUpdate &u = p.u;
        if (u.x) {
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
        } else if (p.out_nchw) {
            *reinterpret_cast<float4 *>(p.out + (int64_t)b * D + e0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int px = 0; px < 4; px++) p.out[((int64_t)b * HW + pix + px) * COUT + co] = acc[px];
        }
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

int head_fused_rows(const ConvLaunch &c) {   // output rows per workgroup: the whole image when its P image fits beside the transpose patches
    int th = c.Hout;
    while (th > 1 && ((size_t)(th + 2) * c.Wout * 9 * c.Cout + 8 * 32 * HF_SLD + 8) * sizeof(float) > 160 * 1024) th >>= 1;
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
    return th >= 4 && c.Hout % th == 0;
}

int64_t head_fused_weight_floats(int Cin) { return (int64_t)Cin * 32; }

int relayout_weight_head_fused(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st) {
    k_relayout_weight_head_fused<<<(unsigned)ceil_div(Cin * 32, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int launch_conv_head_fused(const ConvLaunch &c, const HeadUpdate *hu, hipStream_t st) {
    HeadFusedArgs a{};
    a.h = c.src0; a.coefA = c.coefA; a.coefB = c.coefB; a.wf = c.w_hfused; a.bias = c.bias; a.out = c.out; a.out_nchw = c.out_nchw;
    a.B = c.B; a.H = c.Hout; a.W = c.Wout; a.C = c.C0; a.TH = head_fused_rows(c);
    if (hu) a.u = *hu;
    const int64_t M = (int64_t)c.B * c.Hout * c.Wout;
    // algorithmic bytes: the head's input once + the state read and written (or eps written)
    const double bytes = 4.0 * ((double)M * c.C0 + (double)M * c.Cout * (hu ? 2 + (hu->z ? 1 : 0) + (hu->eps_out ? 1 : 0) : 1));
    ProfScope ps(hu ? "head_fused+update" : "head_fused", 2.0 * M * c.Cout * 9.0 * c.C0, bytes, st);
    const size_t lds = ((size_t)(((a.TH + 2) * a.W * 9 * c.Cout + 3) & ~3) + 8 * 32 * HF_SLD) * sizeof(float);
    const unsigned grid = (unsigned)(c.B * (c.Hout / a.TH));
#define DLPM_HF(CO, NCH)                                                                                  \
    do {                                                                                                  \
        int r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_head_fused<CO, NCH>), 160 * 1024);   \
        if (r != DLPM_OK) return r;                                                                       \
        k_head_fused<CO, NCH><<<grid, HF_NT, lds, st>>>(a);                                               \
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
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
