// conv_direct.hip -- direct convolution for the shapes the MFMA tiling does not cover: the stem
// (Cin = image channels, reads the caller's NCHW state), the head (Cout = image channels, writes
// NCHW eps, fused GroupNorm affine + SiLU on its input) and any channel count that is not a
// multiple of 32.  These are ~0.1 % of the path's FLOPs (unet.py:347,435); the kernel is a plain
// one-thread-per-output FMA loop with channel-fastest (coalesced) weight and output accesses.
#include "conv.h"

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
// affine + SiLU on the input, NCHW output.  On the MFMA path it ran as a 32-wide N tile with 29 of 32 columns empty
// (0.70 ms for [1024,128,32,32] -> 3, the input read alone is 0.1 ms).  Here a workgroup owns a 256-pixel tile (whole
// rows of one image): per 32-channel chunk the activated halo goes through LDS once, every thread accumulates its
// pixel's <= 4 outputs with plain FMAs, and the weights -- the same for every lane -- are fetched with uniform
// addresses ([tap][cin][4], Cout padded to 4).
// ---------------------------------------------------------------------------------------------
constexpr int HEAD_LD = 36;

__global__ void __launch_bounds__(256) k_conv3x3_head(ConvLaunch p) {
    extern __shared__ __attribute__((aligned(16))) float hs[];
    const int W = p.Wout, H = p.Hout, TH = 256 / W, Wp = W + 2, hp = (TH + 2) * Wp;
    float *Cf = hs + hp * HEAD_LD;
    float *Ws = Cf + 64;   // this chunk's weights [tap][32][4]: uniform-address (broadcast) LDS reads in the FMA loop
                           // (from global the compiler issued 288 vector loads per chunk and thread, one per weight quad)
    const int tpi = H / TH;
    const int b = blockIdx.x / tpi, y0 = (blockIdx.x % tpi) * TH;
    const int tid = threadIdx.x, ty = tid / W, tx = tid - ty * W;
    const int Cin = p.C0, nch = Cin / 32;
    const float4 *__restrict__ w4 = reinterpret_cast<const float4 *>(p.w_small);
    float acc[4];
#pragma unroll
    for (int co = 0; co < 4; co++) acc[co] = (co < p.Cout && p.bias) ? p.bias[co] : 0.f;
    const bool has_coef = p.coefA != nullptr;
    for (int chunk = 0; chunk < nch; chunk++) {
        __syncthreads();   // the previous chunk's tile is no longer read
        if (has_coef && tid < 16) {
            const int isb = tid >> 3, q = tid & 7;
            *reinterpret_cast<float4 *>(Cf + isb * 32 + q * 4) =
                *reinterpret_cast<const float4 *>((isb ? p.coefB : p.coefA) + (int64_t)b * Cin + chunk * 32 + q * 4);
        }
        for (int idx = tid; idx < 9 * 32; idx += 256) {
            const int tap = idx >> 5, ci = idx & 31;
            *reinterpret_cast<float4 *>(Ws + idx * 4) = w4[(int64_t)tap * Cin + chunk * 32 + ci];
        }
        __syncthreads();
        for (int idx = tid; idx < hp * 8; idx += 256) {
            const int pix = idx >> 3, q = idx & 7;
            const int hy = pix / Wp, hx = pix - hy * Wp;
            const int iy = y0 + hy - 1, ix = hx - 1;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                x = *reinterpret_cast<const float4 *>(p.src0 + (((int64_t)b * H + iy) * W + ix) * Cin + chunk * 32 + q * 4);
                if (has_coef) {
                    const float4 ca = *reinterpret_cast<const float4 *>(Cf + q * 4);
                    const float4 cb = *reinterpret_cast<const float4 *>(Cf + 32 + q * 4);
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
            *reinterpret_cast<float4 *>(hs + pix * HEAD_LD + q * 4) = x;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const float *row = hs + ((ty + tap / 3) * Wp + tx + tap % 3) * HEAD_LD;
            const float4 *wt = reinterpret_cast<const float4 *>(Ws) + tap * 32;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float4 a = *reinterpret_cast<const float4 *>(row + q * 4);
                const float4 w0 = wt[q * 4 + 0], w1 = wt[q * 4 + 1], w2 = wt[q * 4 + 2], w3 = wt[q * 4 + 3];
                acc[0] = fmaf(a.x, w0.x, acc[0]); acc[1] = fmaf(a.x, w0.y, acc[1]); acc[2] = fmaf(a.x, w0.z, acc[2]); acc[3] = fmaf(a.x, w0.w, acc[3]);
                acc[0] = fmaf(a.y, w1.x, acc[0]); acc[1] = fmaf(a.y, w1.y, acc[1]); acc[2] = fmaf(a.y, w1.z, acc[2]); acc[3] = fmaf(a.y, w1.w, acc[3]);
                acc[0] = fmaf(a.z, w2.x, acc[0]); acc[1] = fmaf(a.z, w2.y, acc[1]); acc[2] = fmaf(a.z, w2.z, acc[2]); acc[3] = fmaf(a.z, w2.w, acc[3]);
                acc[0] = fmaf(a.w, w3.x, acc[0]); acc[1] = fmaf(a.w, w3.y, acc[1]); acc[2] = fmaf(a.w, w3.z, acc[2]); acc[3] = fmaf(a.w, w3.w, acc[3]);
            }
        }
    }
    const int64_t HW = (int64_t)H * W;
    const int64_t pix = (int64_t)(y0 + ty) * W + tx;
#pragma unroll
    for (int co = 0; co < 4; co++)
        if (co < p.Cout) {
            if (p.out_nchw) p.out[((int64_t)b * p.Cout + co) * HW + pix] = acc[co];
            else p.out[((int64_t)b * HW + pix) * p.Cout + co] = acc[co];
        }
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

bool head_conv_ok(const ConvLaunch &c) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("DLPM_NO_HEADK"); off = (e && e[0] == '1') ? 1 : 0; }
    if (off || !c.w_small || c.ks != 3 || c.stride != 1 || c.ups || c.in_nchw || c.C1 != 0 || c.res0) return false;
    if (c.Cout < 1 || c.Cout > 4 || c.C0 % 32 != 0 || c.Hin != c.Hout || c.Win != c.Wout) return false;
    const int W = c.Wout, H = c.Hout;
    if (W < 8 || W > 64 || 256 % W != 0) return false;
    const int TH = 256 / W;
    return H % TH == 0;
}

int launch_conv_head(const ConvLaunch &c, hipStream_t st) {
    const int W = c.Wout, H = c.Hout, TH = 256 / W;
    const int hp = (TH + 2) * (W + 2);
    const size_t shmem = (size_t)(hp * HEAD_LD + 64 + 9 * 32 * 4) * sizeof(float);
    const int64_t M = (int64_t)c.B * H * W;
    ProfScope ps("conv3x3_head", 2.0 * M * c.Cout * 9.0 * c.C0, 4.0 * ((double)M * c.C0 + (double)M * c.Cout), st);
    {
        int r = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_conv3x3_head), 64 * 1024);
        if (r != DLPM_OK) return r;
    }
    k_conv3x3_head<<<(unsigned)(c.B * (H / TH)), 256, shmem, st>>>(c);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int relayout_weight_head(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st) {
    k_relayout_weight_head<<<(unsigned)ceil_div(9 * Cin * 4, 256), 256, 0, st>>>(oihw_dev, dst_dev, Cout, Cin);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
