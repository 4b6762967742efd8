// mlp.hip -- the toy-data score network (MLPModel of dlpm/configs/2d_data.yml) as ONE kernel.
//
// Replaces MLPModel.forward (dlpm/models/Model.py:148-211) and DiffusionBlockConditioned.forward
// (dlpm/models/DiffusionBlocks.py:125-136) for the shipped architecture: learnable time
// embedding Linear(1,TE)->SiLU->Linear(TE,TE)->SiLU, Linear(F,64)->LayerNorm->SiLU, nblocks+1
// time-conditioned residual blocks, Linear(64,F).  no_a=True, LayerNorm ("group_norm: true"),
// skip connections, dropout 0.
//
// The net is 55 k parameters and ~105 kFLOP per sample: launch- and latency-bound, not a GEMM
// problem.  Mapping: one wavefront owns 4 samples and the 64 lanes ARE the 64 hidden units, so a
// LayerNorm is a wave butterfly, a Linear is a k-loop of {coalesced weight row load, one 16-byte
// LDS broadcast of the 4 samples' activations, 4 FMAs}, and no workgroup barrier is ever needed.
// Weights (220 kB, transposed to [k][unit]) stay L2-resident across the whole sampling loop.
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "philox.h"

using namespace dlpm;

namespace {

constexpr int NU = 64;   // hidden units == wavefront width
constexpr int SPW = 4;   // samples per wave

struct MlpOffsets {      // offsets (floats) into the packed parameter blob
    int te_w, te_b, tm_w, tm_b, in_w, in_b, in_g, in_be, blk0, blk_stride, out_w, out_b;
    // per block: w1T[64*64] b1 g1 be1 twT[TE*64] tb w2T[64*64] b2 g2 be2
    int b_w1, b_b1, b_g1, b_be1, b_tw, b_tb, b_w2, b_b2, b_g2, b_be2;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }

// acc[s] += sum_k WT[k][lane] * act[k][s]
__device__ __forceinline__ void matvec(const float *__restrict__ WT, int K, const float *actT, int lane, float acc[SPW]) {
#pragma unroll 8
    for (int k = 0; k < K; k++) {
        const float w = WT[k * NU + lane];
        const float4 h = *reinterpret_cast<const float4 *>(actT + k * SPW);
        acc[0] = fmaf(w, h.x, acc[0]);
        acc[1] = fmaf(w, h.y, acc[1]);
        acc[2] = fmaf(w, h.z, acc[2]);
        acc[3] = fmaf(w, h.w, acc[3]);
    }
}

__device__ __forceinline__ void layer_norm(float v[SPW], float gam, float bet) {
#pragma unroll
    for (int s = 0; s < SPW; s++) {
        const float mean = wave_sum(v[s]) * (1.0f / NU);
        const float d = v[s] - mean;
        const float var = wave_sum(d * d) * (1.0f / NU);
        v[s] = d * (1.0f / sqrtf(var + 1e-5f)) * gam + bet;
    }
}

__global__ void __launch_bounds__(64) k_mlp_forward(const float *__restrict__ P, MlpOffsets o, const float *__restrict__ x,
                                                    const float *__restrict__ t, float *__restrict__ out, int64_t B, int F,
                                                    int TE, int nblk) {
    __shared__ __attribute__((aligned(16))) float actT[NU * SPW];   // [k][sample] activations
    __shared__ __attribute__((aligned(16))) float tembT[NU * SPW];  // [k][sample] time embedding
    const int lane = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * SPW;

    // ---- time embedding (Model.py:64,68-72,191): Linear(1,TE) -> SiLU -> Linear(TE,TE) -> SiLU
    float tv[SPW];
#pragma unroll
    for (int s = 0; s < SPW; s++) tv[s] = (s0 + s < B) ? t[s0 + s] : 0.f;
    if (lane < TE) {
        const float w = P[o.te_w + lane], b = P[o.te_b + lane];
#pragma unroll
        for (int s = 0; s < SPW; s++) actT[lane * SPW + s] = silu(fmaf(w, tv[s], b));
    }
    __syncthreads();
    if (lane < TE) {
        float acc[SPW];
        const float b = P[o.tm_b + lane];
#pragma unroll
        for (int s = 0; s < SPW; s++) acc[s] = b;
        for (int k = 0; k < TE; k++) {
            const float w = P[o.tm_w + k * TE + lane];
#pragma unroll
            for (int s = 0; s < SPW; s++) acc[s] = fmaf(w, actT[k * SPW + s], acc[s]);
        }
#pragma unroll
        for (int s = 0; s < SPW; s++) tembT[lane * SPW + s] = silu(acc[s]);
    }
    __syncthreads();

    // ---- input layer (Model.py:93-97): Linear(F,64) -> LayerNorm -> SiLU
    float h[SPW];
    {
        const float b = P[o.in_b + lane];
#pragma unroll
        for (int s = 0; s < SPW; s++) h[s] = b;
        for (int f = 0; f < F; f++) {
            const float w = P[o.in_w + f * NU + lane];
#pragma unroll
            for (int s = 0; s < SPW; s++) h[s] = fmaf(w, (s0 + s < B) ? x[(s0 + s) * F + f] : 0.f, h[s]);
        }
        layer_norm(h, P[o.in_g + lane], P[o.in_be + lane]);
#pragma unroll
        for (int s = 0; s < SPW; s++) h[s] = silu(h[s]);
    }

    // ---- conditioned residual blocks (DiffusionBlocks.py:125-136)
    for (int bi = 0; bi < nblk; bi++) {
        const float *Q = P + o.blk0 + (int64_t)bi * o.blk_stride;
#pragma unroll
        for (int s = 0; s < SPW; s++) actT[lane * SPW + s] = h[s];
        __syncthreads();
        float y[SPW], tp[SPW];
        const float b1 = Q[o.b_b1 + lane], tb = Q[o.b_tb + lane];
#pragma unroll
        for (int s = 0; s < SPW; s++) { y[s] = b1; tp[s] = tb; }
        matvec(Q + o.b_w1, NU, actT, lane, y);
        layer_norm(y, Q[o.b_g1 + lane], Q[o.b_be1 + lane]);
        matvec(Q + o.b_tw, TE, tembT, lane, tp);
#pragma unroll
        for (int s = 0; s < SPW; s++) y[s] = silu(y[s]) + silu(tp[s]);
        __syncthreads();  // everyone is done reading actT
#pragma unroll
        for (int s = 0; s < SPW; s++) actT[lane * SPW + s] = y[s];
        __syncthreads();
        float z[SPW];
        const float b2 = Q[o.b_b2 + lane];
#pragma unroll
        for (int s = 0; s < SPW; s++) z[s] = b2;
        matvec(Q + o.b_w2, NU, actT, lane, z);
        layer_norm(z, Q[o.b_g2 + lane], Q[o.b_be2 + lane]);
#pragma unroll
        for (int s = 0; s < SPW; s++) h[s] = silu(z[s] + h[s]);
        __syncthreads();
    }

    // ---- output layer Linear(64,F) (Model.py:129): a wave reduction per feature
    for (int f = 0; f < F; f++) {
        const float w = P[o.out_w + f * NU + lane];
        const float b = P[o.out_b + f];
#pragma unroll
        for (int s = 0; s < SPW; s++) {
            const float r = wave_sum(w * h[s]);
            if (lane == 0 && s0 + s < B) out[(s0 + s) * F + f] = r + b;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// The whole reverse loop for the toy net in ONE launch (BASELINE configs[0] is launch-bound: 4
// kernels of a few microseconds per step otherwise).  Samples are independent, so one wavefront
// owns one sample for all its steps: lanes = hidden units, state x (F <= 4 features) replicated in
// registers, Linear layers as 16 iterations of {one float4 weight load per lane, one 16-byte LDS
// broadcast of 4 activations, 4 FMAs}.  Everything that depends only on the step index -- the time
// embedding and every block's t_proj output -- comes from a [T][blocks][64] table built once.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_mlp_tp_table(const float *__restrict__ P, MlpOffsets o, float *__restrict__ tp, int T,
                                                     int TE, int nblk) {
    __shared__ float e0[NU], e1[NU];
    const int lane = threadIdx.x, ti = blockIdx.x;
    const float tv = __fmul_rn((float)ti, 1.0f / (float)T);  // t.float() * (1/T), GenerativeLevyProcess.py:92-96
    if (lane < TE) e0[lane] = silu(fmaf(P[o.te_w + lane], tv, P[o.te_b + lane]));
    __syncthreads();
    if (lane < TE) {
        float a = P[o.tm_b + lane];
        for (int k = 0; k < TE; k++) a = fmaf(P[o.tm_w + k * TE + lane], e0[k], a);
        e1[lane] = silu(a);
    }
    __syncthreads();
    for (int bi = 0; bi < nblk; bi++) {
        const float *Q = P + o.blk0 + (int64_t)bi * o.blk_stride;
        float a = Q[o.b_tb + lane];
        for (int k = 0; k < TE; k++) a = fmaf(Q[o.b_tw + k * NU + lane], e1[k], a);
        tp[((int64_t)ti * nblk + bi) * NU + lane] = silu(a);
    }
}

// acc += sum_k W[lane][k] act[k] with W stored as float4 per (k/4, lane)
__device__ __forceinline__ float matvec_q4(const float4 *__restrict__ Wq, const float *act, int lane, float acc) {
#pragma unroll 4
    for (int kq = 0; kq < NU / 4; kq++) {
        const float4 w = Wq[kq * NU + lane];
        const float4 a = *reinterpret_cast<const float4 *>(act + 4 * kq);
        acc = fmaf(w.x, a.x, acc);
        acc = fmaf(w.y, a.y, acc);
        acc = fmaf(w.z, a.z, acc);
        acc = fmaf(w.w, a.w, acc);
    }
    return acc;
}

__device__ __forceinline__ float layer_norm1(float v, float gam, float bet) {
    const float mean = wave_sum(v) * (1.0f / NU);
    const float d = v - mean;
    const float var = wave_sum(d * d) * (1.0f / NU);
    return d * (1.0f / sqrtf(var + 1e-5f)) * gam + bet;
}

struct MlpLoopArgs {
    const float *P;        // packed parameters ([k][unit] layout)
    const float4 *Q4;      // the 64x64 matrices of every block as float4 per (k/4, unit): [blk][2][16][64]
    const float *tp;       // [T][nblk][64]
    float *x;              // [B][F] state, updated in place
    const float *c_eps, *c_noise, *g;   // [T,B], [T,B], [T]
    int64_t B;
    int F, T, nblk, t_start, nsteps;
    const uint64_t *key_dev;
    uint64_t seed;
    int64_t sample_offset;
};

__global__ void __launch_bounds__(256) k_mlp_sample(MlpLoopArgs a, MlpOffsets o) {
    __shared__ __attribute__((aligned(16))) float acts[4][NU];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * 4 + wv;
    const bool live = b < a.B;
    const int64_t bb = live ? b : a.B - 1;            // dead waves shadow the last sample (barriers stay uniform)
    float *act = acts[wv];
    const float *P = a.P;
    const uint64_t seed = a.key_dev ? a.key_dev[0] : a.seed;
    const uint64_t gidx = (uint64_t)((a.key_dev ? (int64_t)a.key_dev[1] : a.sample_offset) + bb);
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    for (int f = 0; f < a.F; f++) x[f] = a.x[bb * a.F + f];
    const float in_b = P[o.in_b + lane], in_g = P[o.in_g + lane], in_be = P[o.in_be + lane];
    float in_w[4] = {0.f, 0.f, 0.f, 0.f}, out_w[4] = {0.f, 0.f, 0.f, 0.f};
    for (int f = 0; f < a.F; f++) {
        in_w[f] = P[o.in_w + f * NU + lane];
        out_w[f] = P[o.out_w + f * NU + lane];
    }
    for (int step = 0; step < a.nsteps; step++) {
        const int t = a.t_start - step;
        // ---- model forward at step t
        float h = in_b;
        for (int f = 0; f < a.F; f++) h = fmaf(in_w[f], x[f], h);
        h = silu(layer_norm1(h, in_g, in_be));
        for (int bi = 0; bi < a.nblk; bi++) {
            const float *Q = P + o.blk0 + (int64_t)bi * o.blk_stride;
            const float4 *W1 = a.Q4 + ((int64_t)bi * 2 + 0) * (NU / 4) * NU, *W2 = W1 + (NU / 4) * NU;
            act[lane] = h;
            __syncthreads();
            float y = matvec_q4(W1, act, lane, Q[o.b_b1 + lane]);
            y = silu(layer_norm1(y, Q[o.b_g1 + lane], Q[o.b_be1 + lane])) + a.tp[((int64_t)t * a.nblk + bi) * NU + lane];
            __syncthreads();
            act[lane] = y;
            __syncthreads();
            float z = matvec_q4(W2, act, lane, Q[o.b_b2 + lane]);
            z = layer_norm1(z, Q[o.b_g2 + lane], Q[o.b_be2 + lane]);
            h = silu(z + h);
            __syncthreads();
        }
        // ---- eps, then the DLPM update (dlpm.py:272-278, GenerativeLevyProcess.py:236-238)
        const float g = a.g[t];
        const float ce = a.c_eps[(int64_t)t * a.B + bb], cn = a.c_noise[(int64_t)t * a.B + bb];
        float zz[4] = {0.f, 0.f, 0.f, 0.f};
        if (cn != 0.0f) {
            const float4 z4 = philox_normal4(seed, gidx, 0u, 4u /* kPurposeStepZ */, (uint32_t)t);
            zz[0] = z4.x; zz[1] = z4.y; zz[2] = z4.z; zz[3] = z4.w;
        }
        for (int f = 0; f < a.F; f++) {
            const float eps = wave_sum(out_w[f] * h) + P[o.out_b + f];
            const float m = __fdiv_rn(x[f] - __fmul_rn(ce, eps), g);
            x[f] = __fadd_rn(m, __fmul_rn(cn, zz[f]));
        }
    }
    if (live && lane == 0)
        for (int f = 0; f < a.F; f++) a.x[b * a.F + f] = x[f];
}

}  // namespace

struct dlpm_mlp {
    int F, nunits, nblocks, TE;
    std::vector<float> host;       // packed blob, filled by set_param
    std::map<std::string, bool> seen;
    MlpOffsets off;
    float *dev = nullptr;
    float4 *q4 = nullptr;      // float4-interleaved copies of the 64x64 matrices (k_mlp_sample)
    float *tp = nullptr;       // [T][nblocks+1][64] step-only terms
    int tp_T = 0;
    bool finalized = false;
};

extern "C" int dlpm_mlp_create(int32_t nfeatures, int32_t nunits, int32_t nblocks, int32_t time_emb_size, dlpm_mlp **out) {
    DLPM_CHECK_ARG(out, "dlpm_mlp_create: null out");
    DLPM_CHECK_ARG(nfeatures > 0 && nblocks >= 0, "dlpm_mlp_create: bad nfeatures / nblocks");
    if (nunits != NU || time_emb_size <= 0 || time_emb_size > NU) {
        set_error("dlpm_mlp_create: this build maps hidden units onto the 64-lane wavefront: nunits must be 64 "
                  "(got %d) and time_emb_size <= 64 (got %d)", nunits, time_emb_size);
        return DLPM_ERR_UNSUPPORTED;
    }
    dlpm_mlp *m = new dlpm_mlp();
    m->F = nfeatures; m->nunits = nunits; m->nblocks = nblocks; m->TE = time_emb_size;
    MlpOffsets &o = m->off;
    int p = 0;
    auto take = [&](int n) { int r = p; p += (n + 3) / 4 * 4; return r; };
    const int TE = m->TE, F = m->F;
    o.te_w = take(TE); o.te_b = take(TE); o.tm_w = take(TE * TE); o.tm_b = take(TE);
    o.in_w = take(F * NU); o.in_b = take(NU); o.in_g = take(NU); o.in_be = take(NU);
    int q = 0;
    auto takeb = [&](int n) { int r = q; q += (n + 3) / 4 * 4; return r; };
    o.b_w1 = takeb(NU * NU); o.b_b1 = takeb(NU); o.b_g1 = takeb(NU); o.b_be1 = takeb(NU);
    o.b_tw = takeb(TE * NU); o.b_tb = takeb(NU);
    o.b_w2 = takeb(NU * NU); o.b_b2 = takeb(NU); o.b_g2 = takeb(NU); o.b_be2 = takeb(NU);
    o.blk_stride = q;
    o.blk0 = take(q * (nblocks + 1));
    o.out_w = take(F * NU); o.out_b = take(F);
    m->host.assign(p, 0.f);
    *out = m;
    return DLPM_OK;
}

namespace {
// copy a Linear weight [N][K] (row-major, as in the state_dict) transposed to [K][ld]
void put_T(std::vector<float> &h, int off, const float *w, int N, int K, int ld) {
    for (int n = 0; n < N; n++)
        for (int k = 0; k < K; k++) h[off + k * ld + n] = w[n * K + k];
}
}  // namespace

extern "C" int dlpm_mlp_set_param(dlpm_mlp *m, const char *key_c, const float *w, int64_t numel) {
    DLPM_CHECK_ARG(m && key_c && w, "dlpm_mlp_set_param: null argument");
    const std::string key = key_c;
    const MlpOffsets &o = m->off;
    const int TE = m->TE, F = m->F;
    auto need = [&](int64_t n) -> bool {
        if (n == numel) return true;
        set_error("dlpm_mlp_set_param: '%s' has %lld elements, expected %lld", key_c, (long long)numel, (long long)n);
        return false;
    };
    auto vec = [&](int off, int n) { for (int i = 0; i < n; i++) m->host[off + i] = w[i]; };
    // aliases the reference's state_dict repeats (same tensors registered twice: Model.py:68-72,93-97,
    // DiffusionBlocks.py:108-123): accepted and ignored
    static const char *alias_suffix[] = {"time_mlp.0.weight", "time_mlp.0.bias", "inblock.0.weight", "inblock.0.bias",
                                         "inblock.1.weight", "inblock.1.bias", "mlp_1.2.weight", "mlp_1.2.bias",
                                         "mlp_2.2.weight", "mlp_2.2.bias"};
    for (const char *a : alias_suffix) {
        const std::string s = a;
        if (key.size() >= s.size() && key.compare(key.size() - s.size(), s.size(), s) == 0) return DLPM_OK;
    }
    m->finalized = false;
    if (key == "time_emb.weight") { if (!need(TE)) return DLPM_ERR_ARG; vec(o.te_w, TE); }
    else if (key == "time_emb.bias") { if (!need(TE)) return DLPM_ERR_ARG; vec(o.te_b, TE); }
    else if (key == "time_mlp.2.weight") { if (!need(TE * TE)) return DLPM_ERR_ARG; put_T(m->host, o.tm_w, w, TE, TE, TE); }
    else if (key == "time_mlp.2.bias") { if (!need(TE)) return DLPM_ERR_ARG; vec(o.tm_b, TE); }
    else if (key == "linear_in.weight") { if (!need(NU * F)) return DLPM_ERR_ARG; put_T(m->host, o.in_w, w, NU, F, NU); }
    else if (key == "linear_in.bias") { if (!need(NU)) return DLPM_ERR_ARG; vec(o.in_b, NU); }
    else if (key == "group_norm_in.weight") { if (!need(NU)) return DLPM_ERR_ARG; vec(o.in_g, NU); }
    else if (key == "group_norm_in.bias") { if (!need(NU)) return DLPM_ERR_ARG; vec(o.in_be, NU); }
    else if (key == "outblocks_mean.1.weight") {
        if (!need(F * NU)) return DLPM_ERR_ARG;
        for (int f = 0; f < F; f++) for (int k = 0; k < NU; k++) m->host[o.out_w + f * NU + k] = w[f * NU + k];
    }
    else if (key == "outblocks_mean.1.bias") { if (!need(F)) return DLPM_ERR_ARG; vec(o.out_b, F); }
    else {
        int bi = -1;
        std::string rest;
        if (key.rfind("midblocks.", 0) == 0) {
            size_t dot = key.find('.', 10);
            DLPM_CHECK_ARG(dot != std::string::npos, "dlpm_mlp_set_param: malformed key '%s'", key_c);
            bi = std::stoi(key.substr(10, dot - 10));
            rest = key.substr(dot + 1);
            DLPM_CHECK_ARG(bi >= 0 && bi < m->nblocks, "dlpm_mlp_set_param: block index out of range in '%s'", key_c);
        } else if (key.rfind("outblocks_mean.0.", 0) == 0) {
            bi = m->nblocks;
            rest = key.substr(17);
        } else {
            set_error("dlpm_mlp_set_param: unexpected key '%s' for this architecture", key_c);
            return DLPM_ERR_ARG;
        }
        const int base = o.blk0 + bi * o.blk_stride;
        if (rest == "mlp_1.1.weight") { if (!need(NU * NU)) return DLPM_ERR_ARG; put_T(m->host, base + o.b_w1, w, NU, NU, NU); }
        else if (rest == "mlp_1.1.bias") { if (!need(NU)) return DLPM_ERR_ARG; vec(base + o.b_b1, NU); }
        else if (rest == "group_norm1.weight") { if (!need(NU)) return DLPM_ERR_ARG; vec(base + o.b_g1, NU); }
        else if (rest == "group_norm1.bias") { if (!need(NU)) return DLPM_ERR_ARG; vec(base + o.b_be1, NU); }
        else if (rest == "t_proj.1.weight") { if (!need(NU * TE)) return DLPM_ERR_ARG; put_T(m->host, base + o.b_tw, w, NU, TE, NU); }
        else if (rest == "t_proj.1.bias") { if (!need(NU)) return DLPM_ERR_ARG; vec(base + o.b_tb, NU); }
        else if (rest == "mlp_2.1.weight") { if (!need(NU * NU)) return DLPM_ERR_ARG; put_T(m->host, base + o.b_w2, w, NU, NU, NU); }
        else if (rest == "mlp_2.1.bias") { if (!need(NU)) return DLPM_ERR_ARG; vec(base + o.b_b2, NU); }
        else if (rest == "group_norm2.weight") { if (!need(NU)) return DLPM_ERR_ARG; vec(base + o.b_g2, NU); }
        else if (rest == "group_norm2.bias") { if (!need(NU)) return DLPM_ERR_ARG; vec(base + o.b_be2, NU); }
        else {
            set_error("dlpm_mlp_set_param: unexpected key '%s' for this architecture", key_c);
            return DLPM_ERR_ARG;
        }
    }
    m->seen[key] = true;
    return DLPM_OK;
}

extern "C" int dlpm_mlp_finalize(dlpm_mlp *m) {
    DLPM_CHECK_ARG(m, "dlpm_mlp_finalize: null handle");
    const size_t expected = 8 + 10 * (size_t)(m->nblocks + 1) + 2;
    if (m->seen.size() != expected) {
        set_error("dlpm_mlp_finalize: %zu of %zu parameter tensors were set", m->seen.size(), expected);
        return DLPM_ERR_STATE;
    }
    if (!m->dev) DLPM_HIP(hipMalloc(&m->dev, m->host.size() * sizeof(float)));
    DLPM_HIP(hipMemcpy(m->dev, m->host.data(), m->host.size() * sizeof(float), hipMemcpyHostToDevice));
    {   // [blk][2][k/4][unit] float4 = (W[unit][4kq .. 4kq+3]); host holds W transposed: host[off + k*64 + unit]
        const int nb = m->nblocks + 1;
        std::vector<float> q((size_t)nb * 2 * NU * NU);
        for (int bi = 0; bi < nb; bi++)
            for (int which = 0; which < 2; which++) {
                const int src = m->off.blk0 + bi * m->off.blk_stride + (which ? m->off.b_w2 : m->off.b_w1);
                float *dst = q.data() + ((size_t)bi * 2 + which) * NU * NU;
                for (int kq = 0; kq < NU / 4; kq++)
                    for (int u = 0; u < NU; u++)
                        for (int j = 0; j < 4; j++) dst[((size_t)kq * NU + u) * 4 + j] = m->host[src + (4 * kq + j) * NU + u];
            }
        if (!m->q4) DLPM_HIP(hipMalloc(&m->q4, q.size() * sizeof(float)));
        DLPM_HIP(hipMemcpy(m->q4, q.data(), q.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    m->tp_T = 0;
    m->finalized = true;
    return DLPM_OK;
}

extern "C" int dlpm_mlp_sample_steps_f32(dlpm_mlp *m, float *x_dev, const float *c_eps_dev, const float *c_noise_dev,
                                         const float *g_dev, int32_t T, int64_t B, int32_t t_start, int32_t nsteps,
                                         uint64_t seed, int64_t sample_offset, const uint64_t *key_dev, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(m && x_dev && c_eps_dev && c_noise_dev && g_dev, "dlpm_mlp_sample_steps_f32: null argument");
    DLPM_CHECK_ARG(T >= 2 && B > 0 && nsteps >= 0 && t_start < T && t_start - nsteps >= 0, "dlpm_mlp_sample_steps_f32: bad step range");
    if (!m->finalized) {
        set_error("dlpm_mlp_sample_steps_f32: call dlpm_mlp_finalize first");
        return DLPM_ERR_STATE;
    }
    if (m->F > 4) {
        set_error("dlpm_mlp_sample_steps_f32: the fused loop keeps the state in registers: nfeatures <= 4 (got %d)", m->F);
        return DLPM_ERR_UNSUPPORTED;
    }
    if (nsteps == 0) return DLPM_OK;
    hipStream_t st = as_stream(stream);
    const int nb = m->nblocks + 1;
    if (m->tp_T != T) {
        if (m->tp) DLPM_HIP(hipFree(m->tp));
        m->tp = nullptr;
        DLPM_HIP(hipMalloc(&m->tp, (size_t)T * nb * NU * sizeof(float)));
        k_mlp_tp_table<<<(unsigned)T, 64, 0, st>>>(m->dev, m->off, m->tp, T, m->TE, nb);
        DLPM_LAUNCH_CHECK();
        m->tp_T = T;
    }
    MlpLoopArgs a;
    a.P = m->dev; a.Q4 = m->q4; a.tp = m->tp; a.x = x_dev; a.c_eps = c_eps_dev; a.c_noise = c_noise_dev; a.g = g_dev;
    a.B = B; a.F = m->F; a.T = T; a.nblk = nb; a.t_start = t_start; a.nsteps = nsteps;
    a.key_dev = key_dev; a.seed = seed; a.sample_offset = sample_offset;
    k_mlp_sample<<<(unsigned)ceil_div(B, 4), 256, 0, st>>>(a, m->off);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" int dlpm_mlp_forward(dlpm_mlp *m, const float *x_dev, const float *t_dev, float *eps_dev, int64_t B,
                                dlpm_stream_t stream) {
    DLPM_CHECK_ARG(m && x_dev && t_dev && eps_dev && B > 0, "dlpm_mlp_forward: bad argument");
    if (!m->finalized) {
        set_error("dlpm_mlp_forward: call dlpm_mlp_finalize first");
        return DLPM_ERR_STATE;
    }
    k_mlp_forward<<<(unsigned)ceil_div(B, SPW), 64, 0, as_stream(stream)>>>(m->dev, m->off, x_dev, t_dev, eps_dev, B, m->F,
                                                                          m->TE, m->nblocks + 1);
    DLPM_LAUNCH_CHECK();
    return DLPM_OK;
}

extern "C" void dlpm_mlp_destroy(dlpm_mlp *m) {
    if (!m) return;
    if (m->dev) (void)hipFree(m->dev);
    if (m->q4) (void)hipFree(m->q4);
    if (m->tp) (void)hipFree(m->tp);
    delete m;
}
