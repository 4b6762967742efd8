// Does dense VALU code take more cycles when every CU runs it at the same moment?  (Round 5: the narrow F(4x4) kernels' register epilogue
// takes 25.8 k cycles on a 256-workgroup launch and 5.8 k on a 16-workgroup one -- with its memory instructions removed.)
// One workgroup = 4 waves (one per SIMD).  Each wave: [optional MFMA phase of `mfma` instructions] -> straight-line burst of NI independent
// v_fma (UNROLLED: code size ~ 8 NI bytes, executed once) or the same count as a ROLLED loop -> cycles of the burst by clock64().
//   valu_burst: grid in {16, 64, 128, 256, 512}, MFMA phase on / off, unrolled / rolled
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb/valu_burst tools/mb/valu_burst.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NI, bool ROLLED>
__global__ void __launch_bounds__(256, 1) k(float *out, unsigned long long *cyc, int mfma, float seed) {
    const int tid = threadIdx.x;
    floatx4 acc[8];
    for (int q = 0; q < 8; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a = seed * 0.5f + tid * 1e-6f, b = seed * 0.25f;
    for (int it = 0; it < mfma; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = seed + i * 0.001f + tid * 1e-6f + acc[i & 7][i & 3];
    __syncthreads();
    const long long t0 = clock64();
    if (ROLLED) {
#pragma unroll 1
        for (int it = 0; it < NI / 64; it++) {
#pragma unroll
            for (int n = 0; n < 64; n++) v[n & 15] = fmaf(v[n & 15], 1.0001f, 0.5f);
        }
    } else {
#pragma unroll
        for (int n = 0; n < NI; n++) v[n & 15] = fmaf(v[n & 15], 1.0001f, 0.5f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i];
    asm volatile("" ::"v"(s));
    const long long t1 = clock64();
    if (tid == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int NI, bool ROLLED>
void run(const char *name, int mfma) {
    float *out;
    unsigned long long *cyc, h;
    hipMalloc(&out, (size_t)1024 * 256 * 4);
    hipMalloc(&cyc, 8);
    printf("%-40s", name);
    for (int grid : {16, 64, 128, 256, 512, 1024}) {
        hipMemset(cyc, 0, 8);
        k<NI, ROLLED><<<grid, 256>>>(out, cyc, mfma, 0.3f);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("  grid %4d: %7.0f", grid, (double)h / grid);
    }
    printf("   cycles per workgroup for %d v_fma per wave\n", NI);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int rep = 0; rep < 2; rep++) {
        run<2048, false>("unrolled 2048 v_fma, no MFMA phase", 0);
        run<2048, true>("rolled   2048 v_fma, no MFMA phase", 0);
        run<2048, false>("unrolled 2048 v_fma after 4000 MFMAs", 500);
        run<2048, true>("rolled   2048 v_fma after 4000 MFMAs", 500);
        run<8192, false>("unrolled 8192 v_fma after 4000 MFMAs", 500);
    }
    return 0;
}
