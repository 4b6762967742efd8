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
