// Round 3 go / no-go for "Winograd F(2x2,3x3) with its 16 position GEMMs on the bf16 matrix pipe (fp32 operands split into
// three bf16 planes, six partial products)" -- the K loop of such a kernel WITHOUT everything a real kernel adds on top
// (raw staging with GroupNorm + SiLU, the input transform, prologue, output transform, epilogue), on random operands (the
// bf16 pipe is power-bound: all-zero operands run 27-31 % faster, profiles/r02/gemm_bf16x3_zero_operands_dvfs.txt).
//
//   part 1  does an fp32 MFMA (v_mfma_f32_16x16x4_f32: the VALU lanes, SQ_VALU_MFMA_COEXEC_CYCLES = 0) run beside a bf16 MFMA
//           (the matrix core proper) when the two come from the two waves of a SIMD, or interleaved inside one wave?
//   part 2  the F(2x2)-on-split K loop in the two block shapes the 256-KB accumulator block of a CU allows:
//             A  32 tiles x 128 channels x 16 positions, waves = 2 position halves x 4 channel quarters
//                (every weight fragment feeds ONE wave: 24 KB of U per wave and 16-channel step straight from L2);
//             B  64 tiles x 64 channels x 16 positions, waves = 2 position halves x 2 tile halves x 2 channel halves
//                (half the U stream per MFMA, twice the V to stage / transform / split per MFMA).
//           V fragments come from LDS (ds_read_b128, fragment order, conflict-free), U fragments through a register ring
//           from a real [n-tile][k-step][wave][position][plane][lane] stream in global memory (L2-resident, as in the
//           shipped Winograd kernels).  SIDE = independent VALU instructions per position (6 MFMAs) standing in for the
//           transform + split + staging arithmetic; WR = 1 adds the V writes of the next step (ds_write_b128) and the
//           step barrier.
// Output: executed bf16 TFLOP/s and the ALGORITHMIC (direct-convolution) TFLOP/s they amount to: 48 executed bf16 FLOP
// per 18 algorithmic ones.  The shipped F(4x4) fp32 kernel runs the same layers at 296-350 algorithmic TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb/wino2_split tools/mb/wino2_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two random bf16 in [2^-6, 1): random sign, 6 exponent values, random mantissa
__device__ __forceinline__ uint32_t rnd_bf16x2(uint32_t i) {
    const uint32_t h = hash32(i * 2654435761u + 12345u);
    auto one = [](uint32_t r) { return ((r & 1u) << 15) | ((121u + ((r >> 1) % 6u)) << 7) | ((r >> 4) & 0x7fu); };
    return one(h & 0xffffu) | (one(h >> 16) << 16);
}
__global__ void k_fill(uint32_t *p, size_t n, int zeros) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = zeros ? 0u : rnd_bf16x2((uint32_t)i);
}

// ---------------------------------------------------------------------------------------------------------- part 1
// MODE 0 all waves fp32 16x16x4 | 1 all waves bf16 32x32x16 | 2 waves 0-3 fp32, 4-7 bf16 (w and w+4 share a SIMD)
// 3 every wave alternates one fp32 and one bf16 MFMA | 4 waves 0-3 fp32, 4-7 idle | 5 waves 0-3 idle, 4-7 bf16
template <int MODE>
__global__ void __launch_bounds__(512, 1) k_coexec(float *out, int iters, float seed) {
    const int tid = threadIdx.x, wave = tid >> 6;
    floatx4 a4[8];
    floatx16 a16[4];
    for (int q = 0; q < 8; q++) for (int r = 0; r < 4; r++) a4[q][r] = 0.f;
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) a16[q][r] = 0.f;
    const float a = seed * 0.5f + tid * 1e-3f, b = seed * 0.25f - tid * 1e-3f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; i++) { ab[i] = (__bf16)(seed + i + tid * 0.01f); bb[i] = (__bf16)(seed - i - tid * 0.02f); }
    const bool lo = wave < 4;
    const bool f32 = MODE == 0 || ((MODE == 2 || MODE == 4) && lo);
    const bool b16 = MODE == 1 || ((MODE == 2 || MODE == 5) && !lo);
    if (MODE == 3) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                a4[j & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a4[j & 7], 0, 0, 0);
                a16[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a16[j & 3], 0, 0, 0);
            }
        }
    } else if (f32) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 32; j++) a4[j & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a4[j & 7], 0, 0, 0);
        }
    } else if (b16) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 32; j++) a16[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, a16[j & 3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 8; q++) for (int r = 0; r < 4; r++) s += a4[q][r];
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) s += a16[q][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

// ---------------------------------------------------------------------------------------------------------- part 2
// SHAPE 0 = A (32 tiles x 128 channels), 1 = B (64 tiles x 64 channels).  RING = positions of U fragments in flight.
template <int SHAPE, int SIDE, int WR, int RING, int NOU = 0>
__global__ void __launch_bounds__(512, 1) k_w2(const u32x4 *__restrict__ U, float *out, int nk16, int nmb, int flags) {
    constexpr int MT = SHAPE == 0 ? 32 : 64;
    constexpr int VPOS = 3 * 2 * MT;              // 16-byte chunks per position: [plane][k-half][tile]
    constexpr int VBUF = 16 * VPOS;               // chunks per V buffer (48 KB / 96 KB)
    constexpr int NBUF = SHAPE == 0 ? 3 : 1;      // distinct V images in LDS (A: a different one every step)
    extern __shared__ __align__(16) unsigned char smem[];
    u32x4 *V = reinterpret_cast<u32x4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int mb = blockIdx.x % nmb, nt = blockIdx.x / nmb;   // n-tile-major: workgroups in flight stream the same U slab
    for (int i = tid; i < NBUF * VBUF; i += 512) {
        u32x4 t;
        for (int e = 0; e < 4; e++) t[e] = (flags & 1) ? 0u : rnd_bf16x2((uint32_t)(i * 4 + e) + 977u * mb);
        V[i] = t;
    }
    __syncthreads();
    const int ph = wave >> 2;
    int th = 0, ustream;
    if (SHAPE == 0) ustream = wave;               // (ph, channel quarter): 8 distinct streams
    else { th = (wave >> 1) & 1; ustream = ph * 2 + (wave & 1); }   // (ph, channel half): 4 streams, each read by two waves
    constexpr int NSTREAM = SHAPE == 0 ? 8 : 4;
    // U[nt][k16][stream][pos 8][plane 3][lane 64] x 16 B
    const u32x4 *wp = U + ((size_t)nt * nk16 * NSTREAM + __builtin_amdgcn_readfirstlane(ustream)) * (8 * 3 * 64) + lane;
    const size_t wstep = (size_t)NSTREAM * 8 * 3 * 64;
    floatx16 acc[8];
    for (int q = 0; q < 8; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    float sv[8];
    for (int i = 0; i < 8; i++) sv[i] = 0.37f + i * 0.001f + tid * 1e-6f;

    u32x4 bq[RING][3];
#pragma unroll
    for (int a = 0; a < RING - 1; a++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) bq[a][pl] = wp[(a * 3 + pl) * 64];
    const u32x4 *wn = wp;   // fragment of position (step * 8 + pos) is wp_step[(pos * 3 + pl) * 64]
    for (int s = 0; s < nk16; s++) {
        const u32x4 *vb = V + (s % NBUF) * VBUF + (ph * 8) * VPOS + kh * MT + th * 32 + l31;
        u32x4 aq[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) aq[0][pl] = vb[pl * 2 * MT];
#pragma unroll
        for (int pos = 0; pos < 8; pos++) {
            // U fragments RING - 1 positions ahead (across the step boundary: the stream is linear per (wave, n-tile) only
            // within a step, so the look-ahead pointer wraps by hand)
            {
                const int ap = pos + RING - 1;
                const u32x4 *src = ap < 8 ? wn + (size_t)(ap * 3) * 64 : wn + wstep + (size_t)((ap - 8) * 3) * 64;
#pragma unroll
                for (int pl = 0; pl < 3; pl++) bq[(pos + RING - 1) % RING][pl] = NOU ? bq[pos % RING][pl] : src[pl * 64];
            }
            if (pos + 1 < 8) {
#pragma unroll
                for (int pl = 0; pl < 3; pl++) aq[(pos + 1) & 1][pl] = vb[(pos + 1) * VPOS + pl * 2 * MT];
            }
            bf16x8 A[3], B[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                A[pl] = __builtin_bit_cast(bf16x8, aq[pos & 1][pl]);
                B[pl] = __builtin_bit_cast(bf16x8, bq[pos % RING][pl]);
            }
            __builtin_amdgcn_s_setprio(1);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], acc[pos], 0, 0, 0);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], acc[pos], 0, 0, 0);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], acc[pos], 0, 0, 0);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], acc[pos], 0, 0, 0);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], acc[pos], 0, 0, 0);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], acc[pos], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int n = 0; n < SIDE; n++) sv[n % 8] = fmaf(sv[n % 8], 1.0001f, 0.5f);
            if (WR) {
                // the V of the next step: VBUF / 512 chunks per thread per step, spread over the positions
                constexpr int PER = (VBUF / 512 + 7) / 8;
#pragma unroll
                for (int w = 0; w < PER; w++) {
                    const int c = (pos * PER + w) * 512 + tid;
                    if (c < VBUF) V[((s + 1) % NBUF) * VBUF + c] = aq[pos & 1][w % 3];
                }
            }
        }
        wn += wstep;
        if (WR) __syncthreads();
    }
    float sum = 0.f;
    for (int q = 0; q < 8; q++) for (int r = 0; r < 16; r++) sum += acc[q][r];
    for (int i = 0; i < 8; i++) sum += sv[i];
    out[(size_t)blockIdx.x * 512 + tid] = sum;
}

template <int SHAPE, int SIDE, int WR, int RING, int NOU = 0>
static void run_w2(const char *name, const u32x4 *U, float *out, int B, int H, int Cin, int Cout, int flags) {
    constexpr int MT = SHAPE == 0 ? 32 : 64, NC = SHAPE == 0 ? 128 : 64;
    const int nk16 = Cin / 16, nmb = B * (H / 2) * (H / 2) / MT, ntn = Cout / NC;
    const size_t lds = (size_t)(SHAPE == 0 ? 3 * 48 : 96) * 1024;
    auto fn = &k_w2<SHAPE, SIDE, WR, RING, NOU>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    fn<<<nmb * ntn, 512, lds>>>(U, out, nk16, nmb, flags);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 5;
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) fn<<<nmb * ntn, 512, lds>>>(U, out, nk16, nmb, flags);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double alg = 2.0 * 9 * Cin * Cout * (double)B * H * H;
    const double exe = alg * 48.0 / 18.0;
    printf("%-58s H%-2d %3d->%-3d %8.3f ms  executed %7.1f bf16 TFLOP/s  = %6.1f algorithmic TFLOP/s%s\n", name, H, Cin, Cout, ms,
           exe / ms * 1e-9, alg / ms * 1e-9, hipGetLastError() == hipSuccess ? "" : "  [launch error]");
}

template <int MODE>
static float run_coexec(float *out, int iters) {
    k_coexec<MODE><<<256, 512>>>(out, 10, 0.3f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_coexec<MODE><<<256, 512>>>(out, iters, 0.3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char **argv) {
    const int zeros = argc > 1 && atoi(argv[1]) == 1;
    float *out;
    hipMalloc(&out, (size_t)16384 * 512 * 4);
    {
        const int iters = 4000;
        const char *nm[] = {"all 8 waves: 32 fp32 16x16x4 MFMAs / iteration", "all 8 waves: 32 bf16 32x32x16 MFMAs / iteration",
                            "waves 0-3 fp32, waves 4-7 bf16 (one of each per SIMD)", "every wave: 16 fp32 + 16 bf16 interleaved",
                            "waves 0-3 fp32, waves 4-7 idle", "waves 0-3 idle, waves 4-7 bf16"};
        float ms[6] = {run_coexec<0>(out, iters), run_coexec<1>(out, iters), run_coexec<2>(out, iters),
                       run_coexec<3>(out, iters), run_coexec<4>(out, iters), run_coexec<5>(out, iters)};
        for (int m = 0; m < 6; m++) printf("%-58s %8.3f ms  %7.1f cycles / iteration at 2.4 GHz\n", nm[m], ms[m], ms[m] * 1e-3 * 2.4e9 / iters);
    }
    // U: the largest stream used below: Cout 256 (2 or 4 n-tiles), Cin 512 -> 16 pos x 512 x 256 x 6 B = 12.6 MB
    const size_t ubytes = (size_t)16 * 512 * 256 * 6;
    uint32_t *U;
    hipMalloc(&U, ubytes + (1 << 20));
    k_fill<<<(unsigned)((ubytes / 4 + (1 << 18) + 255) / 256), 256>>>(U, ubytes / 4 + (1 << 18), zeros);
    hipDeviceSynchronize();
    const u32x4 *Uv = reinterpret_cast<const u32x4 *>(U);
    printf("-- operands: %s\n", zeros ? "all zero" : "random bf16 planes");
    const int B = 1024;
#define BOTH(SH, SIDE, WR, RING, nm)                                                   \
    run_w2<SH, SIDE, WR, RING>(nm, Uv, out, B, 16, 256, 256, zeros);                   \
    run_w2<SH, SIDE, WR, RING>(nm, Uv, out, B, 32, 128, 128, zeros);                   \
    run_w2<SH, SIDE, WR, RING>(nm, Uv, out, B, 16, 512, 256, zeros);
    BOTH(0, 0, 0, 3, "A 32t x 128c  bare loop, U ring 3");
    BOTH(0, 0, 0, 4, "A 32t x 128c  bare loop, U ring 4");
    BOTH(0, 0, 1, 3, "A 32t x 128c  + V writes + step barrier");
    BOTH(0, 24, 1, 3, "A 32t x 128c  + writes + barrier + 24 VALU / position");
    BOTH(0, 48, 1, 3, "A 32t x 128c  + writes + barrier + 48 VALU / position");
    BOTH(1, 0, 0, 3, "B 64t x 64c   bare loop, U ring 3");
    BOTH(1, 0, 1, 3, "B 64t x 64c   + V writes + step barrier");
    BOTH(1, 48, 1, 3, "B 64t x 64c   + writes + barrier + 48 VALU / position");
    BOTH(1, 96, 1, 3, "B 64t x 64c   + writes + barrier + 96 VALU / position");
    // no U loads at all (NOU): what the L2 -> CU weight stream costs
    run_w2<0, 0, 0, 3, 1>("A bare loop WITHOUT the U loads", Uv, out, B, 16, 256, 256, zeros);
    run_w2<1, 0, 0, 3, 1>("B bare loop WITHOUT the U loads", Uv, out, B, 16, 256, 256, zeros);
    return 0;
}
