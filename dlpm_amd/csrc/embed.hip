// embed.hip -- sinusoidal timestep embedding, [cos | sin] halves (dlpm/models/nn.py:103-121).
// t arrives already divided by T (GenerativeLevyProcess._scale_timesteps, :92-96).
#include "conv.h"

namespace dlpm {
namespace {

__global__ void k_timestep_embedding(const float *__restrict__ t, float *__restrict__ emb, int64_t B, int dim) {
    const int half = dim / 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int64_t b = i / dim;
    const int j = (int)(i - b * dim);
    float v = 0.f;  // odd dim: trailing zero column
    if (j < 2 * half) {
        const int k = (j < half) ? j : j - half;
        // freqs = exp(-ln(10000) * arange(half) / half), every op rounded to fp32 as torch does
        const float f = expf(__fdiv_rn(__fmul_rn(-9.210340371976184f, (float)k), (float)half));
        const float ang = __fmul_rn(t[b], f);
        v = (j < half) ? cosf(ang) : sinf(ang);
    }
    emb[i] = v;
}

}  // namespace

int launch_timestep_embedding(const float *t, float *emb, int64_t B, int dim, hipStream_t st) {
    k_timestep_embedding<<<(unsigned)ceil_div(B * dim, 256), 256, 0, st>>>(t, emb, B, dim);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

}  // namespace dlpm
