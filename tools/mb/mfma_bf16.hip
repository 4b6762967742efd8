// How fast is the bf16 matrix pipe of gfx950 next to the fp32 one, does VALU work run in its shadow (it does not for
// the fp32 MFMAs: profiles/r01/mb_mfma_valu_overlap.txt), and what does a 64x64-per-wave tile fed from LDS reach?
// Question behind it: an fp32 product as three bf16 planes per operand and six bf16 MFMAs (a0b0, a0b1, a1b0, a0b2,
// a1b1, a2b0; fp32 accumulate) costs 6/16 of the fp32 MFMA's pipe time IF the pipe runs at its nominal 16x.
//   kind 0: v_mfma_f32_32x32x2_f32      kind 1: v_mfma_f32_32x32x16_bf16      kind 2: v_mfma_f32_16x16x32_bf16
//   NV: independent v_fma between MFMAs;  LDSFEED: 2 A + 2 B fragments (ds_read_b128) per 2x2 MFMA tiles
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb/mfma_bf16 tools/mb/mfma_bf16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int KIND, int NV, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k_rate(float *out, int iters, float seed, unsigned long long *clk) {
    const int tid = threadIdx.x;
    floatx16 acc[4];
    floatx4 acc4[16];
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    for (int q = 0; q < 16; q++) for (int r = 0; r < 4; r++) acc4[q][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i * 0.001f + tid * 1e-6f;
    const float a = seed * 0.5f, b = seed * 0.25f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; i++) { ab[i] = (__bf16)(seed + i); bb[i] = (__bf16)(seed - i); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (KIND == 0) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j & 3], 0, 0, 0);
            if (KIND == 1) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[j & 3], 0, 0, 0);
            if (KIND == 2) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc4[j], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; n++) v[n % 8] = fmaf(v[n % 8], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    float s = 0.f;
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) s += acc[q][r];
    for (int q = 0; q < 16; q++) for (int r = 0; r < 4; r++) s += acc4[q][r];
    for (int i = 0; i < 8; i++) s += v[i];
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
}

// 4 waves, each a 64x64 tile as 2x2 tiles of 32x32x16 bf16; per k16 step 2 A + 2 B fragments (16 B per lane each) come
// from LDS.  PLANES = 1: plain bf16 GEMM inner loop; PLANES = 3: the six-product split (3 A planes + 3 B planes per
// tile pair feed 6 MFMAs per tile, i.e. 12 fragments per 24 MFMAs).
template <int PLANES>
__global__ void __launch_bounds__(256, 1) k_ldsfeed(float *out, int iters, float seed, unsigned long long *clk) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16x8 *lds = reinterpret_cast<bf16x8 *>(smem);
    for (int i = tid; i < 6 * 8 * 64; i += 256) {
        bf16x8 t;
        for (int e = 0; e < 8; e++) t[e] = (__bf16)(seed + (i & 7) + e);
        lds[i] = t;
    }
    __syncthreads();
    floatx16 acc[4];
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    const bf16x8 *src = lds + lane + (wave & 1) * 64;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            bf16x8 A[2][PLANES], B[2][PLANES];
#pragma unroll
            for (int p = 0; p < PLANES; p++) {
                A[0][p] = src[(p * 8 + ks * 2 + 0) * 64];
                A[1][p] = src[(p * 8 + ks * 2 + 1) * 64];
                B[0][p] = src[((p + 3) * 8 + ks * 2 + 0) * 64];
                B[1][p] = src[((p + 3) * 8 + ks * 2 + 1) * 64];
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (PLANES == 1) {
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][0], acc[i * 2 + j], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int pa = 0; pa < 3; pa++)
#pragma unroll
                            for (int pb = 0; pb + pa < 3; pb++)
                                acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][pa], B[j][pb], acc[i * 2 + j], 0, 0, 0);
                    }
                }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    float s = 0.f;
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) s += acc[q][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

static void report(const char *name, float ms, double flop, unsigned long long *clk_d) {
    unsigned long long c[2];
    hipMemcpy(c, clk_d, sizeof(c), hipMemcpyDeviceToHost);
    // s_memrealtime ticks at 100 MHz; s_memtime at the shader clock
    const double ghz = c[1] ? (double)c[0] / ((double)c[1] * 10.0) : 0.0;
    printf("%-64s %8.3f ms  %8.1f TFLOP/s  shader clock %.2f GHz\n", name, ms, flop / (ms * 1e-3) / 1e12, ghz);
}

template <int KIND, int NV, int WAVES>
void run(const char *name) {
    float *out;
    unsigned long long *clk;
    const int grid = 256 * 4, iters = 4000;
    hipMalloc(&out, (size_t)grid * WAVES * 64 * 4);
    hipMalloc(&clk, 16);
    k_rate<KIND, NV, WAVES><<<grid, WAVES * 64>>>(out, 10, 0.3f, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_rate<KIND, NV, WAVES><<<grid, WAVES * 64>>>(out, iters, 0.3f, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = KIND == 0 ? 2.0 * 32 * 32 * 2 : (KIND == 1 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
    report(name, ms, per * 16.0 * iters * WAVES * grid, clk);
    hipFree(out); hipFree(clk);
}

template <int PLANES>
void run_lds(const char *name) {
    float *out;
    unsigned long long *clk;
    const int grid = 256 * 4, iters = 2000;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipMalloc(&clk, 16);
    const int lds = 6 * 8 * 64 * 16;
    k_ldsfeed<PLANES><<<grid, 256, lds>>>(out, 10, 0.3f, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_ldsfeed<PLANES><<<grid, 256, lds>>>(out, iters, 0.3f, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 4.0 * 4 * (PLANES == 1 ? 1 : 6);
    report(name, ms, 2.0 * 32 * 32 * 16 * mfmas * iters * 4 * grid, clk);
    hipFree(out); hipFree(clk);
}


// The stage loop of conv_split.hip without its global loads: per stage a wave (64 x 32 output tile, six-product split) reads
// 18 fragments (2 k-steps x (2 x 3 A planes + 3 B planes)) and issues 24 MFMAs; NBAR barriers per stage (0, 1 or 2), WR LDS
// write instructions per stage (ds_write_b128 of a register), NW waves per workgroup.  What does the barrier-phased structure
// cost by itself?
template <int NW, int NBAR, int WR, int WPE = 1>
__global__ void __launch_bounds__(NW * 64, WPE) k_stage(float *out, int iters, float seed, unsigned long long *clk) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16x8 *lds = reinterpret_cast<bf16x8 *>(smem);
    for (int i = tid; i < 3072; i += NW * 64) {
        bf16x8 t;
        for (int e = 0; e < 8; e++) t[e] = (__bf16)(seed + (i & 7) + e);
        lds[i] = t;
    }
    __syncthreads();
    floatx16 acc[2];
    for (int q = 0; q < 2; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    const bf16x8 *a = lds + (wave & 1) * 64 + (lane & 31) + (lane >> 5) * 128;
    const bf16x8 *b = lds + 1536 + (wave >> 1) % 4 * 32 + (lane & 31) + (lane >> 5) * 128;
    bf16x8 wv;
    for (int e = 0; e < 8; e++) wv[e] = (__bf16)(seed + e);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 A[2][3], B[3];
#pragma unroll
            for (int p = 0; p < 3; p++) {
                A[0][p] = a[p * 512 + ks * 256];
                A[1][p] = a[p * 512 + ks * 256 + 32];
                B[p] = b[p * 512 + ks * 256];
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][2], B[0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[2], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[0], acc[i], 0, 0, 0);
            }
        }
        if (NBAR >= 1) __syncthreads();
#pragma unroll
        for (int w = 0; w < WR; w++) lds[3072 + (w * NW * 64 + tid) % 3072] = wv;   // a second 48-KB region: no hazard with the reads
        if (NBAR >= 2) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    float s = 0.f;
    for (int q = 0; q < 2; q++) for (int r = 0; r < 16; r++) s += acc[q][r];
    out[(size_t)blockIdx.x * NW * 64 + tid] = s;
}

template <int NW, int NBAR, int WR, int WPE = 1>
void run_stage(const char *name, int lds_bytes) {
    float *out;
    unsigned long long *clk;
    const int grid = 256 * 6, iters = 600;
    hipMalloc(&out, (size_t)grid * NW * 64 * 4);
    hipMalloc(&clk, 16);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stage<NW, NBAR, WR, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k_stage<NW, NBAR, WR, WPE><<<grid, NW * 64, lds_bytes>>>(out, 10, 0.3f, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_stage<NW, NBAR, WR, WPE><<<grid, NW * 64, lds_bytes>>>(out, iters, 0.3f, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    report(name, ms, 2.0 * 32 * 32 * 16 * 24.0 * iters * NW * grid, clk);
    hipFree(out); hipFree(clk);
}

int main() {
    run<0, 0, 4>("fp32 32x32x2, 1 wave/SIMD");
    run<0, 0, 8>("fp32 32x32x2, 2 waves/SIMD");
    run<1, 0, 4>("bf16 32x32x16, 1 wave/SIMD");
    run<1, 0, 8>("bf16 32x32x16, 2 waves/SIMD");
    run<2, 0, 4>("bf16 16x16x32, 1 wave/SIMD");
    run<2, 0, 8>("bf16 16x16x32, 2 waves/SIMD");
    run<1, 2, 4>("bf16 32x32x16 + 2 fma per MFMA, 1 wave/SIMD");
    run<1, 4, 4>("bf16 32x32x16 + 4 fma per MFMA, 1 wave/SIMD");
    run<1, 8, 4>("bf16 32x32x16 + 8 fma per MFMA, 1 wave/SIMD");
    run<1, 4, 8>("bf16 32x32x16 + 4 fma per MFMA, 2 waves/SIMD");
    run<1, 8, 8>("bf16 32x32x16 + 8 fma per MFMA, 2 waves/SIMD");
    run<2, 2, 8>("bf16 16x16x32 + 2 fma per MFMA, 2 waves/SIMD");
    run<0, 4, 8>("fp32 32x32x2 + 4 fma per MFMA, 2 waves/SIMD");
    run_lds<1>("bf16 32x32x16, 64x64 per wave, operands from LDS");
    run_lds<3>("six-product split, 64x64 per wave, 3+3 planes from LDS");
    // LDS per workgroup chosen so that 2 workgroups of 8 waves (or 1 of 16) share a CU, as in conv_split.hip
    run_stage<8, 0, 0>("stage loop, 8 waves x 2 WGs/CU, no barrier, no writes", 76 * 1024);
    run_stage<8, 1, 0>("stage loop, 8 waves x 2 WGs/CU, 1 barrier per stage", 76 * 1024);
    run_stage<8, 2, 0>("stage loop, 8 waves x 2 WGs/CU, 2 barriers per stage", 76 * 1024);
    run_stage<8, 2, 9>("stage loop, 8 waves x 2 WGs/CU, 2 barriers + 9 LDS writes", 76 * 1024);
    run_stage<8, 0, 9>("stage loop, 8 waves x 2 WGs/CU, no barrier, 9 LDS writes", 76 * 1024);
    run_stage<16, 1, 0>("stage loop, 16 waves x 1 WG/CU, 1 barrier per stage", 150 * 1024);
    run_stage<16, 1, 5>("stage loop, 16 waves x 1 WG/CU, 1 barrier + 5 LDS writes", 150 * 1024);
    run_stage<4, 2, 0>("stage loop, 4 waves x 3 WGs/CU, 2 barriers per stage", 50 * 1024);
    run_stage<8, 2, 9, 4>("stage loop, 8 waves x 2 WGs/CU, 2 barriers + 9 writes, <= 128 VGPRs", 76 * 1024);
    run_stage<8, 2, 0, 4>("stage loop, 8 waves x 2 WGs/CU, 2 barriers, <= 128 VGPRs", 76 * 1024);
    return 0;
}
