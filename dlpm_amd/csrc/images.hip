// images.hip -- [0,1] fp32 NCHW samples -> 8-bit RGB rows, the last device step before the PNG dump.
//
// The reference saves every generated image with torchvision.utils.save_image (one call per sample,
// bem/evaluate/EvaluationManager.py:188-190).  For a single [C,H,W] tensor that function (torchvision, not
// vendored in the reference and not pinned by it) replicates a 1-channel image to 3 channels (make_grid) and
// quantises with  mul(255).add_(0.5).clamp_(0,255).to(uint8)  before handing HWC bytes to the PNG writer.
// Here the same arithmetic (fp32, one rounding per op, truncating conversion) runs on the GPU so that only
// 3 bytes per pixel cross PCIe instead of 4*C.
#include "common.h"

namespace dlpm {
namespace {

__device__ __forceinline__ unsigned char quant(float v) {
    float q = __fadd_rn(__fmul_rn(v, 255.0f), 0.5f);
    q = fminf(fmaxf(q, 0.0f), 255.0f);
    return (unsigned char)(int)q;   // .to(uint8) truncates
}

// one thread per pixel: reads C planes (coalesced along the pixel index), writes 3 bytes
template <int C>
__global__ void k_images_to_rgb8(const float *__restrict__ x, unsigned char *__restrict__ out, int64_t npix, int HW) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    int64_t b = i / HW;
    int p = (int)(i - b * HW);
    const float *src = x + b * (int64_t)C * HW + p;
    unsigned char r = quant(src[0]);
    unsigned char g = C == 3 ? quant(src[HW]) : r;
    unsigned char bl = C == 3 ? quant(src[2 * (int64_t)HW]) : r;
    unsigned char *o = out + i * 3;
    o[0] = r;
    o[1] = g;
    o[2] = bl;
}

}  // namespace
}  // namespace dlpm

using namespace dlpm;

extern "C" int dlpm_images_to_rgb8(const float *x_dev, uint8_t *out_dev, int64_t B, int32_t C, int32_t H, int32_t W,
                                   dlpm_stream_t stream) {
    DLPM_CHECK_ARG(x_dev && out_dev && B > 0 && H > 0 && W > 0, "dlpm_images_to_rgb8: bad argument");
    DLPM_CHECK_ARG(C == 1 || C == 3, "dlpm_images_to_rgb8: %d channels (save_image writes 1- or 3-channel images)", C);
    int64_t npix = B * H * W;
    ProfScope ps("images_to_rgb8", 0.0, (double)npix * (4.0 * C + 3.0), as_stream(stream));
    unsigned grid = (unsigned)ceil_div(npix, 256);
    if (C == 3)
        k_images_to_rgb8<3><<<grid, 256, 0, as_stream(stream)>>>(x_dev, out_dev, npix, H * W);
    else
        k_images_to_rgb8<1><<<grid, 256, 0, as_stream(stream)>>>(x_dev, out_dev, npix, H * W);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}
