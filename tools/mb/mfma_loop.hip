// Micro-benchmark of the fp32-MFMA inner loop of k_conv3x3_halo: what limits it below 157 TFLOP/s?
//   V0 registers only            V1 + LDS fragment reads (16 ds_read_b128 / 64 MFMA)
//   V2 + one barrier per tap     V3 + weight-tile ds_write_b128 x4 per tap
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb/mfma_loop tools/mb/mfma_loop.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int LD = 36;

template <int V>
__global__ void __launch_bounds__(256) k(float *out, int ntaps, int lds_rows) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < lds_rows * LD; i += 256) sm[i] = (float)((i * 7 + 3) % 13) * 0.01f;
    __syncthreads();
    floatx16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const float *A = sm, *B = sm + 128 * LD;
    float4 wreg[4];
    for (int v = 0; v < 4; v++) wreg[v] = make_float4(0.1f * v, 0.2f, 0.3f, 0.4f);
    float4 af[2], bf[2];
    af[0] = af[1] = bf[0] = bf[1] = make_float4(0.5f, 0.25f, 0.125f, 1.f);
    for (int t = 0; t < ntaps; t++) {
        const int toff = (t % 9) * LD;  // shifted rows, as the taps do
        const int buf = t & 1;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            if (V >= 1) {
#pragma unroll
                for (int i = 0; i < 2; i++) af[i] = *reinterpret_cast<const float4 *>(A + ((wm * 64 + i * 32 + l31) * LD + toff) % (100 * LD) + kk * 8 + kh * 4);
#pragma unroll
                for (int j = 0; j < 2; j++) bf[j] = *reinterpret_cast<const float4 *>(B + buf * 128 * LD + (wn * 64 + j * 32 + l31) * LD + kk * 8 + kh * 4);
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (V >= 3) {
            float4 *db = reinterpret_cast<float4 *>(sm + 128 * LD + (buf ^ 1) * 128 * LD + (tid >> 1) * LD + (tid & 1) * 16);
#pragma unroll
            for (int v = 0; v < 4; v++) db[v] = wreg[v];
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(const char *name, int grid, int ntaps, size_t shmem) {
    float *out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    const int rows = (int)(shmem / 4 / LD);
    k<V><<<grid, 256, shmem>>>(out, ntaps, rows);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; r++) k<V><<<grid, 256, shmem>>>(out, ntaps, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    double flops = (double)grid * 4 * ntaps * 64 * 2.0 * 32 * 32 * 2;
    printf("%-44s grid %5d shmem %6zu: %8.3f ms  %7.1f TFLOP/s\n", name, grid, shmem, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    const int ntaps = 36 * 8;
    for (int wgs : {256, 512, 768}) {
        // shmem chosen so that 1 / 2 / 3 workgroups fit per CU
        size_t sh = wgs == 256 ? 100 * 1024 : (wgs == 512 ? 78 * 1024 : 52 * 1024);
        if (sh < (size_t)(128 + 256) * LD * 4) sh = (128 + 256) * LD * 4;
        printf("--- %d workgroups resident per CU (grid = %d x 8 rounds)\n", wgs / 256, wgs);
        run<0>("V0 MFMA only (registers)", wgs * 8, ntaps, sh);
        run<1>("V1 + LDS fragment reads", wgs * 8, ntaps, sh);
        run<2>("V2 + barrier per tap", wgs * 8, ntaps, sh);
        run<3>("V3 + W-tile ds_write per tap", wgs * 8, ntaps, sh);
    }
    return 0;
}
