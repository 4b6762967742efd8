// What HBM read rate do the head kernel's access patterns allow?  One workgroup per CU, every wave walks 16-KB tiles
// (32 pixels x 128 channels fp32, NHWC) one tile ahead in registers, as k_head_fused does, and only sums what it reads.
//   pattern 0: the transposing forms of the kernel -- a load instruction = 8 pixels x 128 B (32 channels), 512-B stride; the 4 channel
//              chunks of a tile are separate instructions
//   pattern 1: a load instruction = 1 KB contiguous (2 whole pixels)
//   pattern 3: the shipped kernel -- a load instruction = 32 pixels x 32 B (lane = pixel + 32 k-half, 16 B each): the MFMA's own operand
//              layout, no transposition
//   pattern 2: as 0, but the chunk loads of a tile are issued chunk-interleaved with ALU delay between them (as the pipelined kernel does)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb/stream_pattern tools/mb/stream_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PAT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1) k(const float *h, float *out, int tiles_per_wg, int nwg, int spin) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lp = lane >> 3, lc = lane & 7;
    float4 xb[16];
    float s = 0.f;
    auto src = [&](int wg, int t) { return h + ((int64_t)wg * tiles_per_wg + t) * 4096; };
    auto ld = [&](const float *p, int j) {
        if (PAT == 1) return *reinterpret_cast<const float4 *>(p + j * 256 + lane * 4);
        if (PAT == 3) return *reinterpret_cast<const float4 *>(p + (lane & 31) * 128 + j * 8 + (lane >> 5) * 4);
        const int c = j >> 2, i = j & 3;
        return *reinterpret_cast<const float4 *>(p + 32 * c + (8 * i + lp) * 128 + 4 * lc);
    };
    for (int wg = blockIdx.x; wg < nwg; wg += gridDim.x) {
        const float *p0 = src(wg, wave);
#pragma unroll
        for (int j = 0; j < 16; j++) xb[j] = ld(p0, j);
        for (int t = wave; t < tiles_per_wg; t += WAVES) {
            const float *pn = src(wg, t + WAVES < tiles_per_wg ? t + WAVES : t);
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int i = 0; i < 4; i++) { const float4 v = xb[4 * c + i]; s += v.x + v.y + v.z + v.w; }
#pragma unroll
                for (int i = 0; i < 4; i++) xb[4 * c + i] = ld(pn, 4 * c + i);
                if (PAT == 2) for (int q = 0; q < spin; q++) { s = __builtin_amdgcn_rcpf(s + 1.0f); asm volatile("" : "+v"(s)); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
}

template <int PAT, int WAVES>
void run(const char *name, const float *h, float *out, int nimg, int spin) {
    const int tiles = 32;   // a 32x32 image = 32 tiles
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<PAT, WAVES><<<256, WAVES * 64>>>(h, out, tiles, nimg, spin);
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) k<PAT, WAVES><<<256, WAVES * 64>>>(h, out, tiles, nimg, spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    const double bytes = (double)nimg * tiles * 16384;
    printf("%-72s %7.4f ms  %5.2f TB/s\n", name, ms, bytes / ms / 1e9);
}

int main() {
    const int nimg = 1024;
    float *h, *out;
    hipMalloc(&h, (size_t)nimg * 32 * 16384 + (1 << 20));
    hipMalloc(&out, 256 * 1024 * 4);
    hipMemset(h, 0, (size_t)nimg * 32 * 16384);
    run<1, 8>("1 KB contiguous per load instruction, 8 waves", h, out, nimg, 0);
    run<0, 8>("8 pixels x 128 B per load instruction (line-coalesced forms), 8 waves", h, out, nimg, 0);
    run<2, 8>("... + 16 dependent rcp between the chunks", h, out, nimg, 16);
    run<2, 8>("... + 64 dependent rcp between the chunks", h, out, nimg, 64);
    run<2, 8>("... + 128 dependent rcp between the chunks", h, out, nimg, 128);
    run<3, 8>("32 pixels x 32 B per load instruction (MFMA operand layout: the kernel's), 8 waves", h, out, nimg, 0);
    run<3, 4>("32 pixels x 32 B per load instruction (MFMA operand layout), 4 waves", h, out, nimg, 0);
    run<1, 16>("1 KB contiguous per load instruction, 16 waves", h, out, nimg, 0);
    run<0, 16>("8 pixels x 128 B per load instruction, 16 waves", h, out, nimg, 0);
    run<1, 4>("1 KB contiguous per load instruction, 4 waves", h, out, nimg, 0);
    run<0, 4>("8 pixels x 128 B per load instruction, 4 waves", h, out, nimg, 0);
    return 0;
}
