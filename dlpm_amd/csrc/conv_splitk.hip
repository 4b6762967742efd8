// conv_splitk.hip -- the reduction behind a split-K convolution launch (round 6).
//
// At a small DECLARED batch the 8x8 / 4x4 levels of a UNet launch 16-128 workgroups that each walk a K loop of 32-64 serial phases
// (the CIFAR net at B = 64: 14 F(2x2) launches of ~100 us on 16 workgroups).  conv_ksplit_for cuts that loop over 2 / 4 / 8 grid copies
// (ConvLaunch::ksplit: the narrow F(4x4) shapes and the F(2x2) kernel); copy s leaves the partial OUTPUTS of its channel range -- the
// Winograd output transform is linear, so partial sums may be transformed before they are added -- in part[s], and this kernel forms
//     out = bias + (((part[0] + part[1]) + part[2]) + ...) (+ residual)
// in a fixed order: deterministic, a function of the layer and the declared batch only.  Replaces nothing in the reference (the
// convolution is F.conv2d, dlpm/models/unet.py:143,157 via nn.py:25-35); it is the tail of those launches.
#include "conv.h"

namespace dlpm {
namespace {

__global__ void __launch_bounds__(256) k_splitk_reduce(const float *__restrict__ part, int S, int64_t n4, int64_t stride4, int Cout,
                                                       const float *__restrict__ bias, const float *__restrict__ res0,
                                                       const float *__restrict__ res1, int R0, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 *p4 = reinterpret_cast<const float4 *>(part);
    float4 a = p4[i];
    for (int s = 1; s < S; s++) {
        const float4 b = p4[i + (int64_t)s * stride4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const int c4 = Cout >> 2;
    const int64_t pix = i / c4;
    const int c = (int)(i - pix * c4) * 4;
    if (bias) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + c);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (res0) {
        const float4 r = c < R0 ? *reinterpret_cast<const float4 *>(res0 + pix * R0 + c)
                                : *reinterpret_cast<const float4 *>(res1 + pix * (Cout - R0) + (c - R0));
        a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    }
    reinterpret_cast<float4 *>(out)[i] = a;
}

}  // namespace

int launch_splitk_reduce(const float *part, int S, int64_t npix, int Cout, const float *bias, const float *res0, const float *res1, int R0,
                         float *out, hipStream_t st) {
    DLPM_CHECK_ARG(part && out && S >= 2 && npix > 0 && (Cout & 3) == 0 && (R0 & 3) == 0, "launch_splitk_reduce: bad argument");
    const int64_t n4 = npix * (Cout >> 2);
    ProfScope ps("splitk_reduce", 0.0, 4.0 * (double)npix * Cout * (S + 1 + (res0 ? 1 : 0)), st);
    k_splitk_reduce<<<(unsigned)ceil_div(n4, 256), 256, 0, st>>>(part, S, n4, n4, Cout, bias, res0, res1, R0 ? R0 : Cout, out);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

int conv_ksplit_for(const ConvLaunch &c) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("DLPM_KSPLIT"); on = e ? atoi(e) : 1; }
    if (!on || c.gen != DLPM_CONV_AUTO || c.dispatch_B <= 0 || c.ks != 3 || c.stride != 1 || c.in_nchw || c.out_nchw) return 1;
    if ((c.Cout & 3) || conv_split_ok(c)) return 1;
    int a, b, n, nchunks;
    int64_t grid;
    if (wino4_preferred(c, &a, &b, &n)) {
        const int nq = wino4_launch_nq(c);
        if (nq == 128 || (c.Cout == 32 && c.Hout == 32 && c.Wout == 32)) return 1;      // the 8-wave and whole-image shapes carry no split
        const int64_t tiles = c.dispatch_B * (c.Hout / 4) * (c.Wout / 4);
        grid = (n == 1 ? tiles / 16 : ceil_div(c.dispatch_B, (int64_t)n)) * (c.Cout / nq);
        nchunks = (c.C0 + c.C1) / 8;
    } else if (wino_geometry(c, &a, &b, &n)) {
        grid = wino_grid_at(c, c.dispatch_B);
        nchunks = (c.C0 + c.C1) / wino_chunk_channels(c);
    } else {
        return 1;
    }
    if (grid <= 0) return 1;
    int S = 1;
    while (S < 8 && grid * S * 2 <= 256 && nchunks % (S * 2) == 0 && nchunks / (S * 2) >= 4) S *= 2;
    return S;
}

}  // namespace dlpm
