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
        uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
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
    float r0 = sqrtf(-2.0f * __logf(u0)), r1 = sqrtf(-2.0f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.28318530717958647692f * u1, &s0, &c0);
    __sincosf(6.28318530717958647692f * u3, &s1, &c1);
    return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

}  // namespace dlpm
