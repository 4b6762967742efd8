// conv.h -- launch descriptors shared by the convolution kernels and the UNet plan.
#pragma once
#include "common.h"

namespace dlpm {

// One conv2d / conv1d(k=1) / Linear launch over NHWC fp32 activations.
struct ConvLaunch {
    const float *src0 = nullptr, *src1 = nullptr;  // virtual channel concat [src0 | src1]
    int C0 = 0, C1 = 0;
    int B = 0, Hin = 0, Win = 0, Hout = 0, Wout = 0;
    int ks = 1, stride = 1, ups = 0;
    const float *w = nullptr;      // igemm: [ks*ks][Cout][Cin]   direct: [ks*ks][Cin][Cout]
    const float *w_frag = nullptr; // optional, 3x3 only: MFMA B-fragment order [Cout/32][Cin/32][9][4][64 lanes][4]
                                   // (see k_conv3x3_halo_ws): each wave streams it straight into registers
    const float *w_wino = nullptr; // optional, 3x3 s1 only: Winograd-domain weights U = G g G^T in fragment order
                                   // (see conv_wino.hip); when the shape qualifies the F(2x2,3x3) kernel runs
    const float *w_wino4 = nullptr;// optional, 3x3 s1 only: F(4x4,3x3) Winograd-domain weights (conv_wino4.hip); preferred over w_wino
    const float *w_wino4_n64 = nullptr, *w_wino4_n32 = nullptr;   // optional, Cout % 128 == 0 only: the same weights in the fragment order of 64- / 32-channel
                                   // n-tiles (round 6: at a small DECLARED batch the narrow shapes give 2-4x as many workgroups; same bits)
    const void *w_split = nullptr; // optional: the weights as three bf16 planes in stage-tile order (conv_split.hip); a 3x3 launch
                                   // that carries it is a stride-2 downsampling convolution, or a test forcing the path
    const float *w_small = nullptr;// optional, 3x3 with Cout <= 4 (the head): [tap][Cin][4] for k_conv3x3_head
    const float *w_taps = nullptr; // optional, 3x3 with Cout <= 3 (the head): [9 Cout -> 32][Cin], the head as a 1x1 GEMM + gather (launch_conv_head_gemm)
    const float *w_hfused = nullptr; // optional, the head: W' in MFMA fragment order for the one-pass head + update kernel (head_fused.hip)
    const float *bias = nullptr;   // [Cout] or null
    const float *coefA = nullptr, *coefB = nullptr;  // [B, Cin] fused GroupNorm affine, or null
    int act_silu = 0;
    const float *res0 = nullptr, *res1 = nullptr;    // residual (virtual concat), NHWC at output size
    int R0 = 0;
    float *out = nullptr;
    int Cout = 0;
    int in_nchw = 0, out_nchw = 0;  // boundary layouts
    int ws_gemm = 0;                // 1x1 only: use the weight-streaming kernel (TAPS = 1) instead of k_conv_igemm
    int abl = 0;                    // timing-only ablation bits (DLPM_ABL env, only in builds with -DDLPM_IGEMM_ABLATIONS; results are wrong when set)
    // Which kernel generation a 3x3 stride-1 launch takes is a function of the LAYER (geometry + this policy), never of the
    // batch the launch happens to carry: the generations round differently, and a sample must not depend on how its batch
    // was sharded or chunked.  gen = DLPM_CONV_AUTO / _F4 / _F2 / _IGEMM (include/dlpm_amd.h); dispatch_B > 0 lets AUTO
    // weigh grid occupancy for a caller-declared batch (a property of the configuration, the same on every rank / chunk).
    int gen = 0;
    int64_t dispatch_B = 0;
    int gemm = 0;                   // DLPM_GEMM_AUTO / _F32 / _BF16X3: which matrix pipe the 1x1 and downsampling convolutions take (conv_split.hip)
    // Round 6, split-K (conv_ksplit_for): > 1 = the launch runs `ksplit` copies of its grid, copy s walking the input channels
    // [s Cin / ksplit, (s + 1) Cin / ksplit) and writing its partial outputs to out + s B Hout Wout Cout (bias / residual / statistics
    // are null in such a launch: launch_splitk_reduce adds them).  Taken by the narrow F(4x4) shapes and the F(2x2) kernel only.
    int ksplit = 0;
    int pers_total = 0;             // persistent form of the 8-wave F(4x4) shape (DLPM_WINO4_PERSIST): tiles of the launch (the grid is one workgroup per CU)
    // Optional fused GroupNorm statistics of the OUTPUT: per (image, pixel tile, channel) the
    // pair (mean, centred sum of squares) over the tile's pixels, written by the MFMA kernels'
    // epilogue when the tile lies inside one image.  [B][HW/tile][Cout] float2 with tile =
    // conv_stats_pixels(launch): 128 for the implicit-GEMM kernels, 256 for the Winograd kernel.
    float2 *stats_out = nullptr;
#ifdef DLPM_PHASE_TIMING
    unsigned long long *phase = nullptr;
#endif
};

// true when the MFMA implicit-GEMM kernel covers this shape
bool igemm_supported(const ConvLaunch &c);
int launch_conv_igemm(const ConvLaunch &c, hipStream_t st);
int launch_conv_direct(const ConvLaunch &c, hipStream_t st);
int launch_conv_stem(const ConvLaunch &c, hipStream_t st);
// head convolution (Cout <= 4) as a VALU kernel (conv_direct.hip), optionally with the sampler's reverse update fused into
// its epilogue: x <- (x - c_eps[t,b] eps) / g[t] + c_noise[t,b] z on the NCHW state the eps would have been subtracted from
struct HeadUpdate {
    float *x = nullptr;                 // [B, Cout*H*W] state, updated in place (null: plain convolution, eps -> ConvLaunch::out)
    const float *z = nullptr;           // injected normals, or null = in-kernel Philox (same counters as k_update_rows)
    const int32_t *t = nullptr;         // device step counter
    const float *g = nullptr, *c_eps = nullptr, *c_noise = nullptr;   // [T], [T,B], [T,B]
    const uint64_t *key = nullptr;      // optional device {seed, sample_offset}
    uint64_t seed = 0;
    int64_t sample_offset = 0;
    float *const *hist_pp = nullptr;    // optional history cell (dlpm_update_args::hist_pp)
    float *eps_out = nullptr;           // optional: also keep eps
    int32_t T = 0;
    int64_t B = 0;
};
bool head_conv_ok(const ConvLaunch &c);
int launch_conv_head(const ConvLaunch &c, const HeadUpdate *hu, hipStream_t st);
int relayout_weight_head(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st);
// the head as a 1x1 GEMM onto 9 Cout (padded to 32) tap channels + a 9-point gather that also carries the update (conv_direct.hip);
// P = scratch of head_gemm_scratch_floats(c) floats
bool head_gemm_ok(const ConvLaunch &c);
int head_taps_rows(int Cout);
int64_t head_gemm_scratch_floats(const ConvLaunch &c);
int launch_conv_head_gemm(const ConvLaunch &c, const HeadUpdate *hu, float *P, hipStream_t st);
int relayout_weight_head_taps(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st);
// the head and the reverse update in ONE pass over HBM (head_fused.hip): the tap-channel tensor stays in LDS
bool head_fused_ok(const ConvLaunch &c);
int64_t head_fused_weight_floats(int Cin);
int relayout_weight_head_fused(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st);
int launch_conv_head_fused(const ConvLaunch &c, const HeadUpdate *hu, hipStream_t st);
// Winograd F(2x2,3x3) path (conv_wino.hip)
bool wino_geometry(const ConvLaunch &c, int *bh, int *bw, int *nimg);
int wino_tiles(const ConvLaunch &c);   // 2x2 output tiles per workgroup (64 or 32)
int launch_conv_wino(const ConvLaunch &c, hipStream_t st);
int64_t wino_weight_floats(int Cout, int Cin);
int relayout_weight_wino(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st);
// Winograd F(4x4,3x3) path (conv_wino4.hip): 16 tiles of 4x4 outputs x 128 channels per workgroup
bool wino4_enabled();                  // DLPM_WINO_F4
bool wino4_vsplit();                   // DLPM_WINO_VS: waves = position halves x channel quarters (weight fragment order follows)
bool wino4_image_stats(int cout);      // this layer's epilogue emits per-image statistics for blocks of four 8x8 images
bool wino4_geometry(const ConvLaunch &c, int *bh, int *bw, int *nimg);
bool wino4_preferred(const ConvLaunch &c, int *bh, int *bw, int *nimg);   // geometry + dispatch policy
int launch_conv_wino4(const ConvLaunch &c, hipStream_t st);
int64_t wino4_weight_floats(int Cout, int Cin);
int relayout_weight_wino4(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st, int nq = 0);   // nq: n-tile width (0: the layer's own)
// A WHOLE ResBlock with 32 output channels on 32x32 images in one launch (round 6, conv_wino4.hip: k_resblock_wino4_img): GroupNorm-1
// coefficients from the producers' statistics (or given), conv1 as F(4x4,3x3), GroupNorm-2 with scale-shift inside the workgroup (an image
// is one workgroup), conv2 over the activated intermediate (which makes one trip through L2, never a second launch), skip, output
// statistics.  The 1x1 skip convolution of the concat blocks stays a launch of its own (its result arrives as `res`).
struct ResImgLaunch {
    const float *x0 = nullptr, *x1 = nullptr;        // NHWC input, virtual concat [x0 | x1]
    int C0 = 0, C1 = 0, B = 0;
    int H = 32;                                      // 32: 32 output channels on 32x32 images (k_resblock_wino4_img); 16: 64 output channels on 16x16
                                                     // images (k_resblock_wino4_img16: the intermediate stays in LDS, hbuf unused, stats_out = [B][64])
    const float2 *st0 = nullptr, *st1 = nullptr;     // producers' statistics ([B][nt][C] float2) -> GroupNorm-1 inside the kernel, or
    int nt0 = 1, nt1 = 1;
    const float *coefA1 = nullptr, *coefB1 = nullptr;   // ... its coefficients [B][Cin] computed by a launch in front (sources without statistics)
    const float *gn1_w = nullptr, *gn1_b = nullptr, *gn2_w = nullptr, *gn2_b = nullptr;
    const float *w1 = nullptr, *b1 = nullptr;        // in_layers.2 (3x3, Cin -> 32): F(4x4) fragment stream (relayout_weight_wino4)
    const float *w2 = nullptr, *b2 = nullptr;        // out_layers.3 (3x3, 32 -> 32)
    const float *emb = nullptr;                      // emb_layers output rows: scale at [emb_off + c], shift at [emb_off + 32 + c]
    int64_t emb_stride = 0;
    int emb_off = 0;
    const float *res = nullptr;                      // [B][1024][32]: x itself (identity skip) or the 1x1 skip convolution's output
    float *hbuf = nullptr;                           // [B][1024][32] scratch: silu(GN2(conv1(..)) (1 + scale) + shift)
    float *out = nullptr;                            // [B][1024][32]
    float2 *stats_out = nullptr;                     // optional [B][4][32]: (mean, M2) per 256-pixel quadrant, as the convolution kernels emit
#ifdef DLPM_PHASE_TIMING
    unsigned long long *phase = nullptr;
#endif
};
bool res_img_ok(const ResImgLaunch &r);
int launch_resblock_img(const ResImgLaunch &r, hipStream_t st);

// 1x1 convolutions, and 3x3 ones as an implicit GEMM, on the bf16 matrix pipe with fp32 operands split exactly into three
// bf16 planes (conv_split.hip); taps = 1 or 9
bool conv_split_ok(const ConvLaunch &c);
int launch_conv_split(const ConvLaunch &c, hipStream_t st);
int64_t split_weight_floats(int Cout, int Cin, int taps);
int relayout_weight_split(const float *oihw_dev, void *dst_dev, int Cout, int Cin, int taps, hipStream_t st);
// pixels behind one stats_out partial for this launch (0: the launch cannot emit statistics)
int conv_stats_pixels(const ConvLaunch &c);
// Split-K factor of a 3x3 stride-1 launch under the AUTO policy with a DECLARED batch (1: none): when the kernel this layer takes would
// put workgroups on at most half of the 256 CUs at that batch (8x8 / 4x4 levels at batch <= 128), its K loop -- 32-64 serial phases --
// is cut over 2 / 4 / 8 grid copies.  A function of the layer and the declaration only.  DLPM_KSPLIT=0 switches it off.
int conv_ksplit_for(const ConvLaunch &c);
int wino4_launch_nq(const ConvLaunch &c);      // the n-tile width launch_conv_wino4 would use
int64_t wino_grid_at(const ConvLaunch &c, int64_t B);   // workgroups of the F(2x2) kernel at batch B
int wino_chunk_channels(const ConvLaunch &c);            // input channels per chunk of its K loop (8; 16 under DLPM_WINO_KC=16)
// out = bias + sum_s part[s] (+ residual [res0 | res1]) over n = B Hout Wout pixels x Cout channels, partials summed in ascending s
int launch_splitk_reduce(const float *part, int S, int64_t npix, int Cout, const float *bias, const float *res0, const float *res1, int R0,
                         float *out, hipStream_t st);
// non-MFMA shapes: the stem kernel when it applies, the generic direct kernel otherwise
inline int launch_conv_fallback(const ConvLaunch &L, hipStream_t st) {
    if (L.in_nchw && !L.out_nchw && L.ks == 3 && L.stride == 1 && !L.ups && L.C1 == 0 && L.Cout % 4 == 0 && !L.coefA &&
        !L.act_silu && !L.res0 && L.bias)
        return launch_conv_stem(L, st);
    return launch_conv_direct(L, st);
}  // NCHW in, 3x3 s1, Cout % 4 == 0, direct weight layout

// weight re-layout kernels: OIHW -> [tap][Cout][Cin] (igemm) or [tap][Cin][Cout] (direct)
int relayout_weight(const float *oihw_dev, float *dst_dev, int Cout, int Cin, int ks, bool for_igemm, hipStream_t st);
// OIHW 3x3 -> fragment order for k_conv3x3_halo_ws; dst holds frag_weight_floats(Cout, Cin) floats
int64_t frag_weight_floats(int Cout, int Cin, int taps = 9);
int relayout_weight_frag(const float *oihw_dev, float *dst_dev, int Cout, int Cin, hipStream_t st, int taps = 9);

int launch_gn_coeffs(const float *src0, const float *src1, int C0, int C1, int B, int HW, int groups,
                     const float *gamma, const float *beta, const float *ss, int64_t ss_stride, int64_t ss_offset,
                     float *coefA, float *coefB, hipStream_t st);
// GroupNorm coefficients from per-tile channel statistics (see ConvLaunch::stats_out) instead of the
// activations: st0/st1 = statistics of the two concat sources, nt0/nt1 = their tiles per image.
int launch_gn_coeffs_from_stats(const float2 *st0, const float2 *st1, int C0, int C1, int B, int nt0, int nt1, int HW, int groups,
                                const float *gamma, const float *beta, const float *ss, int64_t ss_stride,
                                int64_t ss_offset, float *coefA, float *coefB, hipStream_t st);
int launch_attention(const float *qkv, float *out, int B, int T, int C, int heads, hipStream_t st);

// Whole blocks of small images in one launch, activations resident in LDS (block_small.hip): 64-output-channel ResBlocks on
// 8x8 / 4x4 images, one workgroup per image.
struct ResSmallLaunch {
    const float *x0 = nullptr, *x1 = nullptr;   // NHWC input, virtual concat [x0 | x1]
    int C0 = 0, C1 = 0, B = 0, H = 0, W = 0;
    const float *gn1_w = nullptr, *gn1_b = nullptr, *gn2_w = nullptr, *gn2_b = nullptr;
    const float *w1f = nullptr, *b1 = nullptr;   // in_layers.2 (3x3, Cin -> 64), fragment order (relayout_weight_small)
    const float *w2f = nullptr, *b2 = nullptr;   // out_layers.3 (3x3, 64 -> 64)
    const float *wsf = nullptr, *bs = nullptr;   // skip_connection (1x1, Cin -> 64) or null = identity
    const float *emb = nullptr;                  // emb_layers output rows: scale at [emb_off + c], shift at [emb_off + 64 + c]
    int64_t emb_stride = 0;
    int emb_off = 0;
    float *out = nullptr;                        // [B][H W][64]
    float2 *stats_out = nullptr;                 // optional [B][64] (mean, M2) of the output per image and channel
};
struct AttnSmallLaunch {                         // AttentionBlock, 64 channels, 4 heads, 8x8 / 4x4 images
    const float *x = nullptr;                    // [B][H W][64]
    int C = 0, heads = 0, B = 0, H = 0, W = 0;
    const float *gn_w = nullptr, *gn_b = nullptr;
    const float *wqkv = nullptr, *bqkv = nullptr;    // qkv (1x1, 64 -> 192), fragment order
    const float *wproj = nullptr, *bproj = nullptr;  // proj_out (1x1, 64 -> 64), fragment order
    float *out = nullptr;
    float2 *stats_out = nullptr;
#ifdef DLPM_PHASE_TIMING
    unsigned long long *phase = nullptr;
#endif
};
bool attn_small_ok(const AttnSmallLaunch &a);
int launch_attnblock_small(const AttnSmallLaunch &a, hipStream_t st);
bool attn16_ok(const AttnSmallLaunch &a);        // 16x16 images (round 6): the whole block, one head at a time
int launch_attnblock16(const AttnSmallLaunch &a, hipStream_t st);
bool gnqkv_small_ok(const AttnSmallLaunch &a);   // 16x16 images: GroupNorm + qkv only, out = qkv [B][256][192]
int launch_gnqkv_small(const AttnSmallLaunch &a, hipStream_t st);
bool small_blocks_enabled();                     // DLPM_NO_FUSED_BLOCKS
bool small_weight_ok(int Cout, int Cin, int ks);
int64_t small_weight_floats(int Cout, int Cin, int ks);
int relayout_weight_small(const float *oihw_dev, float *dst_dev, int Cout, int Cin, int ks, hipStream_t st);
bool res_small_ok(const ResSmallLaunch &r);
int launch_resblock_small(const ResSmallLaunch &r, hipStream_t st);
int launch_timestep_embedding(const float *t, float *emb, int64_t B, int dim, hipStream_t st);

__device__ __forceinline__ float silu_f(float v) {
    // x * sigmoid(x), sigmoid = 1/(1+exp(-x))  (nn.py:12-14).  v_exp_f32 + v_rcp_f32 (1 ulp each): the IEEE
    // division sequence costs ~10 instructions per element and this runs once per staged activation.
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}

}  // namespace dlpm
