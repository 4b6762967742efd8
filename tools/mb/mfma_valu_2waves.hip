// Two waves per SIMD (512 threads, 8 waves): which waves share a SIMD, and does one wave's VALU work overlap the other's
// MFMAs?  Each wave runs either an MFMA-only loop or a VALU-only loop, selected by a bit of its wave index.
//   sel = 0: all MFMA   sel = 1: all VALU   sel = 2: bit0 picks   sel = 3: bit1 picks   sel = 4: bit2 picks
// If waves {w, w+4} share a SIMD, sel = 4 puts one MFMA wave and one VALU wave on every SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb/mfma_valu_2waves tools/mb/mfma_valu_2waves.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512, 1) k(float *out, int iters, float seed, int sel) {
    const int tid = threadIdx.x, wave = tid >> 6;
    floatx16 acc[8];
    for (int q = 0; q < 8; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i * 0.001f + tid * 1e-6f;
    const float a = seed * 0.5f, b = seed * 0.25f;
    bool do_mfma = sel == 0 ? true : sel == 1 ? false : ((wave >> (sel - 2)) & 1) == 0;
    if (do_mfma) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
#pragma unroll
                for (int j = 0; j < 4; j++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int n = 0; n < 32 * 16; n++) v[n % 8] = fmaf(v[n % 8], 1.0001f, 0.5f);   // 16 fma per MFMA slot = 64 cycles
        }
    }
    float s = 0.f;
    for (int q = 0; q < 8; q++) for (int r = 0; r < 16; r++) s += acc[q][r];
    for (int i = 0; i < 8; i++) s += v[i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

int main() {
    float *out;
    const int grid = 256, iters = 2000;
    hipMalloc(&out, (size_t)grid * 512 * 4);
    const char *names[] = {"all 8 waves MFMA (32 MFMA / iter each)", "all 8 waves VALU (512 fma / iter each)",
                           "wave bit0: even MFMA, odd VALU", "wave bit1 picks", "wave bit2 picks (w, w+4 differ)"};
    for (int sel = 0; sel < 5; sel++) {
        k<<<grid, 512>>>(out, 10, 0.3f, sel);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<grid, 512>>>(out, iters, 0.3f, sel);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.3f ms   %7.1f cycles per iteration\n", names[sel], ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
