// Does a wave's own VALU work run in the shadow of its MFMAs?  One wave per SIMD (16 accumulator tiles = 256 AGPRs,
// as in k_conv3x3_wino_p); per MFMA: NV independent v_fma (or NT v_exp) pinned between MFMAs with sched_barrier.
//   mode 0: MFMA only   mode 1: VALU only   mode 2: both interleaved
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb/mfma_valu_overlap tools/mb/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV, int NT, bool DEP>
__global__ void __launch_bounds__(256, 1) k(float *out, int iters, float seed) {
    const int tid = threadIdx.x;
    floatx16 acc[16];
    for (int q = 0; q < 16; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i * 0.001f + tid * 1e-6f;
    float a = seed * 0.5f, b = seed * 0.25f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int qq = DEP ? q : (q * 4 + j) % 16;   // DEP: 4 dependent MFMAs in a row on one accumulator
                if (MODE != 1) acc[qq] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[qq], 0, 0, 0);
                if (MODE != 0) {
#pragma unroll
                    for (int n = 0; n < NV; n++) v[n % 8] = fmaf(v[n % 8], 1.0001f, 0.5f);
#pragma unroll
                    for (int n = 0; n < NT; n++) v[n % 8] = __builtin_amdgcn_exp2f(v[n % 8]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < 16; q++) for (int r = 0; r < 16; r++) s += acc[q][r];
    for (int i = 0; i < 8; i++) s += v[i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NV, int NT, bool DEP>
void run(const char *name) {
    float *out;
    const int grid = 256, iters = 2000;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    k<MODE, NV, NT, DEP><<<grid, 256>>>(out, 10, 0.3f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE, NV, NT, DEP><<<grid, 256>>>(out, iters, 0.3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 64);   // cycles per MFMA gap at 2.4 GHz
    printf("%-56s %8.3f ms  %6.1f cycles per gap\n", name, ms, cyc);
    hipFree(out);
}

int main() {
    run<0, 0, 0, true>("MFMA only, dependent chains of 4");
    run<0, 0, 0, false>("MFMA only, round-robin accumulators");
    run<1, 4, 0, true>("VALU only: 4 fma per gap");
    run<2, 4, 0, true>("MFMA + 4 fma per gap (dependent chains)");
    run<2, 4, 0, false>("MFMA + 4 fma per gap (round-robin)");
    run<1, 8, 0, true>("VALU only: 8 fma per gap");
    run<2, 8, 0, true>("MFMA + 8 fma per gap");
    run<1, 12, 0, true>("VALU only: 12 fma per gap");
    run<2, 12, 0, true>("MFMA + 12 fma per gap");
    run<1, 0, 2, true>("VALU only: 2 exp per gap");
    run<2, 0, 2, true>("MFMA + 2 exp per gap");
    run<1, 4, 2, true>("VALU only: 4 fma + 2 exp per gap");
    run<2, 4, 2, true>("MFMA + 4 fma + 2 exp per gap");
    return 0;
}
