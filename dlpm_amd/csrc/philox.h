// philox.h -- Philox4x32-10 (Salmon et al., SC'11) as a stateless device function.
// Counter = (global sample index lo, hi, element-quad index or table row, purpose | step << 8),
// key = the user seed: every normal is a pure function of (seed, global sample, step, element),
// so a batch sharded over N GPUs produces bit-identical samples for every N.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dlpm {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint64_t key) {
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32->64 multiply each (v_mad_u64_u32): integer multiplies are quarter rate on CDNA
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        c = make_uint4((uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// four N(0,1) draws: two Box-Muller pairs from 4 x 24-bit uniforms in (0,1)
__device__ __forceinline__ float4 philox_normal4(uint64_t seed, uint64_t gidx, uint32_t quad, uint32_t purpose,
                                                 uint32_t step) {
    uint4 r = philox4x32_10(make_uint4((uint32_t)gidx, (uint32_t)(gidx >> 32), quad, purpose | (step << 8)), seed);
    const float k = 1.0f / 16777216.0f;
    float u0 = ((float)(r.x >> 8) + 0.5f) * k, u1 = ((float)(r.y >> 8) + 0.5f) * k;
    float u2 = ((float)(r.z >> 8) + 0.5f) * k, u3 = ((float)(r.w >> 8) + 0.5f) * k;
    // Box-Muller on the hardware transcendentals: v_log_f32 is log2, v_sin/v_cos take REVOLUTIONS
    // (sin(2 pi x)), so no range reduction and no 2*pi multiply are needed.
    const float k2 = -1.3862943611198906f;  // -2 ln 2
    float r0 = __builtin_amdgcn_sqrtf(k2 * __builtin_amdgcn_logf(u0)), r1 = __builtin_amdgcn_sqrtf(k2 * __builtin_amdgcn_logf(u2));
    return make_float4(r0 * __builtin_amdgcn_cosf(u1), r0 * __builtin_amdgcn_sinf(u1), r1 * __builtin_amdgcn_cosf(u3),
                       r1 * __builtin_amdgcn_sinf(u3));
}

// Philox 'purpose' of the per-step normals of the reverse update (the other purposes live in noise.hip)
constexpr int kPurposeStepZ = 4;

// a / g as the correctly rounded quotient refined from the reciprocal (one Newton step on the residual): the division of
// the reverse update (x - c_eps eps) / gamma_t, shared by the update kernels and the head convolution's fused epilogue
__device__ __forceinline__ float div_by(float a, float g, float rg) {
    const float q = a * rg;
    return fmaf(fmaf(-q, g, a), rg, q);
}

}  // namespace dlpm
