// unet.hip -- the improved-DDPM UNet score network as a launch plan over the HIP kernels.
//
// Replaces UNetModel.forward and its blocks (dlpm/models/unet.py:33-250,276-492) for the
// constructor arguments of _unet_model (dlpm/dlpm_experiment.py:38-56: use_scale_shift_norm=True,
// dims=2, no class conditioning, conv_resample=True).  Weights arrive under the reference's
// state_dict keys, so reference checkpoints load unchanged.
//
// Data layout in HBM: activations NHWC fp32 in a caller-provided workspace, handed out by an arena
// that recycles every buffer after its last consumer has been enqueued (skip-connection tensors live
// until their output block); the (B,C,H,W) boundary tensors are read/written in place by the
// stem/head kernels.  Fusions:
//   GroupNorm(+scale-shift)+SiLU  -> coefficient kernel + the consumer conv's tile staging
//   cat([h, skip])                -> two-pointer reads in the conv / GroupNorm kernels
//   nearest upsample              -> index arithmetic in the conv
//   bias, residual add            -> conv epilogue
//   22 per-ResBlock Linear(4mc -> 2C) -> one GEMM against the row-concatenated weights
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "conv.h"

using namespace dlpm;

namespace {

struct Param {
    std::string key;
    int64_t numel = 0;
    float *dev = nullptr;   // reference layout
    bool set = false;
};

enum LayerKind { L_STEM, L_RES, L_ATTN, L_DOWN, L_UP };

struct ConvW {            // one convolution's weights
    int p_w = -1, p_b = -1;  // param indices
    int cout = 0, cin = 0, ks = 1;
    bool use_igemm = false;
    float *w_dev = nullptr;  // re-laid-out copy (or the original for 1x1 igemm)
    float *w_frag = nullptr; // 3x3 only: MFMA fragment order for the weight-streaming halo kernel
    float *w_wino = nullptr; // 3x3 only: Winograd-domain weights in fragment order (conv_wino.hip)
    float *w_wino4 = nullptr;// 3x3 only: F(4x4,3x3) Winograd-domain weights (conv_wino4.hip), when DLPM_WINO_F4 is on
    float *w_wino4_n64 = nullptr, *w_wino4_n32 = nullptr;   // cout % 128 == 0, built while a SMALL dispatch batch is declared: the same weights in the
                             // fragment order of 64- / 32-channel n-tiles (narrow_wino4_copies)
    float *w_small = nullptr;// 3x3 with cout <= 4 (head): [tap][cin][4]
    float *w_taps = nullptr; // 3x3 with cout <= 3 (head): [9 cout -> 32][cin], the head as a 1x1 GEMM + gather (conv_direct.hip)
    void *w_split = nullptr; // 1x1 with cout % 128 == 0, cin % 32 == 0: three bf16 planes in stage-tile order (conv_split.hip)
    float *w_rs = nullptr;   // 64-channel layers: fragment order of the fused small-image blocks (block_small.hip)
    float *w_hfused = nullptr;  // head: fragment order of the one-pass head + update kernel (head_fused.hip)
    bool owns = false;
};

struct Layer {
    LayerKind kind;
    std::string pre;
    int cin = 0, cout = 0;       // RES: cin = total input channels
    int emb_off = 0;             // RES: column offset into the fused emb GEMM output
    ConvW c1, c2, skip;          // RES: in conv, out conv, 1x1 skip | ATTN: c1 = qkv, c2 = proj | STEM/DOWN/UP: c1
    bool has_skip = false;
    int p_gn1_w = -1, p_gn1_b = -1, p_gn2_w = -1, p_gn2_b = -1;
    int p_emb_w = -1, p_emb_b = -1;
};

struct Tensor4 {
    float *p = nullptr;
    int C = 0, H = 0, W = 0;
    float2 *stats = nullptr;  // per (image, pixel tile, channel) (mean, M2) emitted by the producing conv, or null
    int stats_px = 0;         // pixels per tile of `stats` (128: implicit-GEMM kernels, 256: Winograd kernel)
};

// Activation arena: a first-fit free list over the caller's workspace.  Launches are stream-ordered, so a buffer can be
// handed out again as soon as every launch that touches it has been ENQUEUED; the plan releases each tensor right after
// its last consumer (skip-connection tensors when their output block pops them).  A dry run (no base pointer: offsets
// from a fake base, never dereferenced) replays the same alloc / release sequence to size the workspace = the peak.
struct Bump {
    char *base = nullptr;
    int64_t cap = 0, off = 0, peak = 0;
    bool dry = false, overflow = false, reuse = true;
    std::vector<std::pair<int64_t, int64_t>> holes;     // (offset, bytes), sorted by offset, coalesced
    std::map<int64_t, int64_t> live;                    // offset -> bytes
    char *origin() const { return dry ? reinterpret_cast<char *>(uintptr_t(1) << 20) : base; }
    float *alloc(int64_t nfloats) {
        const int64_t bytes = (nfloats * 4 + 255) / 256 * 256;
        int64_t at = -1;
        if (reuse) {
            size_t best = holes.size();                  // best fit: the smallest hole that holds the request
            for (size_t i = 0; i < holes.size(); i++)
                if (holes[i].second >= bytes && (best == holes.size() || holes[i].second < holes[best].second)) best = i;
            if (best != holes.size()) {
                at = holes[best].first;
                if (holes[best].second == bytes) holes.erase(holes.begin() + best);
                else { holes[best].first += bytes; holes[best].second -= bytes; }
            }
        }
        if (at < 0) {
            at = off;
            off += bytes;
            if (off > peak) peak = off;
            if (!dry && off > cap) overflow = true;
        }
        live[at] = bytes;
        return reinterpret_cast<float *>(origin() + at);
    }
    void release(const void *p) {
        if (!p || !reuse) return;
        const int64_t at = reinterpret_cast<const char *>(p) - origin();
        auto it = live.find(at);
        if (it == live.end()) return;
        int64_t lo = at, n = it->second;
        live.erase(it);
        size_t i = 0;
        while (i < holes.size() && holes[i].first < lo) i++;
        if (i > 0 && holes[i - 1].first + holes[i - 1].second == lo) { lo = holes[i - 1].first; n += holes[i - 1].second; holes.erase(holes.begin() + --i); }
        if (i < holes.size() && lo + n == holes[i].first) { n += holes[i].second; holes.erase(holes.begin() + i); }
        if (lo + n == off) { off = lo; return; }         // the top of the arena shrinks back
        holes.insert(holes.begin() + i, std::make_pair(lo, n));
    }
};

void policy_of(const dlpm_unet *u, ConvLaunch &L);   // copies the net's conv policy into a launch (defined below the struct)

// GroupNorm statistics can ride on the producing conv's epilogue when its pixel tiles stay inside one image
// and the conv runs on the MFMA kernels with the row epilogue.  Decides (at plan time) whether `t`, produced by
// conv `c` with the given geometry, carries statistics, and allocates them: tile size per conv_stats_pixels.
void plan_stats(const dlpm_unet *u, Bump &ws, Tensor4 &t, const ConvW &c, int B, int C0, int stride, int ups, bool stem = false) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("DLPM_NO_GN_FUSION"); off = (e && e[0] == '1') ? 1 : 0; }
    t.stats = nullptr;
    t.stats_px = 0;
    if (off || (!c.use_igemm && !stem)) return;
    ConvLaunch L;
    L.w_wino = c.w_wino; L.w_wino4 = c.w_wino4; L.w_wino4_n64 = c.w_wino4_n64; L.w_wino4_n32 = c.w_wino4_n32; L.w_split = c.w_split; L.ks = c.ks; L.stride = stride; L.ups = ups; L.Hout = t.H; L.Wout = t.W; L.Cout = c.cout;
    L.Hin = ups ? t.H / 2 : t.H * stride; L.Win = ups ? t.W / 2 : t.W * stride;
    L.C0 = C0; L.C1 = c.cin - C0; L.B = B;
    L.in_nchw = stem ? 1 : 0;
    policy_of(u, L);
    const int px = conv_stats_pixels(L);
    if (px <= 0) return;
    t.stats_px = px;
    t.stats = reinterpret_cast<float2 *>(ws.alloc((int64_t)2 * B * (t.H * t.W / px) * t.C));
}

// GroupNorm(+scale/shift) coefficients of the virtual concat [x0 | x1]
int gn_any(Tensor4 x0, Tensor4 x1, int B, int groups, const float *gamma, const float *beta, const float *ss,
           int64_t ss_stride, int64_t ss_offset, float *cA, float *cB, hipStream_t st) {
    const int HW = x0.H * x0.W;
    if (x0.stats && (x1.C == 0 || x1.stats))
        return launch_gn_coeffs_from_stats(x0.stats, x1.stats, x0.C, x1.C, B, HW / x0.stats_px, x1.C ? HW / x1.stats_px : 1, HW,
                                           groups, gamma, beta, ss, ss_stride, ss_offset, cA, cB, st);
    return launch_gn_coeffs(x0.p, x1.p, x0.C, x1.C, B, HW, groups, gamma, beta, ss, ss_stride, ss_offset, cA, cB, st);
}


void drop(Bump &ws, Tensor4 &t) {   // release a tensor and the statistics that ride with it
    ws.release(t.p);
    ws.release(t.stats);
    t.p = nullptr;
    t.stats = nullptr;
}

}  // namespace

struct dlpm_unet {
    dlpm_unet_config cfg;
    std::vector<Param> params;
    std::map<std::string, int> index;
    std::vector<std::vector<Layer>> in_blocks, out_blocks;
    std::vector<Layer> mid;
    std::vector<int> skip_ch;        // channels pushed by each input block
    ConvW te0, te2, embcat, head;
    int p_head_gn_w = -1, p_head_gn_b = -1;
    float *embcat_w = nullptr, *embcat_b = nullptr;
    int emb_total = 0, ted = 0, final_ch = 0;
    bool finalized = false;
    std::vector<Tensor4> feats;      // block outputs of the last forward
    int64_t flops = 0;
    bool keep_feats = false;         // dlpm_unet_keep_features: block outputs stay valid after the forward (no arena reuse)
    int gen = DLPM_CONV_AUTO;        // dlpm_unet_set_conv_policy
    int64_t dispatch_B = 0;
    int gemm = DLPM_GEMM_AUTO;       // dlpm_unet_set_gemm_policy
    int64_t plan_version = 0;

    int add(const std::string &key, int64_t numel) {
        Param p;
        p.key = key;
        p.numel = numel;
        params.push_back(p);
        index[key] = (int)params.size() - 1;
        return (int)params.size() - 1;
    }
    ConvW conv(const std::string &pre, int cout, int cin, int ks) {
        ConvW c;
        c.cout = cout; c.cin = cin; c.ks = ks;
        c.p_w = add(pre + "weight", (int64_t)cout * cin * ks * ks);
        c.p_b = add(pre + "bias", cout);
        return c;
    }
    Layer res(const std::string &pre, int cin, int cout) {
        Layer L;
        L.kind = L_RES; L.pre = pre; L.cin = cin; L.cout = cout;
        L.p_gn1_w = add(pre + "in_layers.0.weight", cin);
        L.p_gn1_b = add(pre + "in_layers.0.bias", cin);
        L.c1 = conv(pre + "in_layers.2.", cout, cin, 3);
        L.p_emb_w = add(pre + "emb_layers.1.weight", (int64_t)2 * cout * ted);
        L.p_emb_b = add(pre + "emb_layers.1.bias", 2 * cout);
        L.p_gn2_w = add(pre + "out_layers.0.weight", cout);
        L.p_gn2_b = add(pre + "out_layers.0.bias", cout);
        L.c2 = conv(pre + "out_layers.3.", cout, cout, 3);
        L.has_skip = cin != cout;
        if (L.has_skip) L.skip = conv(pre + "skip_connection.", cout, cin, 1);
        L.emb_off = emb_total;
        emb_total += 2 * cout;
        return L;
    }
    Layer attn(const std::string &pre, int ch) {
        Layer L;
        L.kind = L_ATTN; L.pre = pre; L.cin = L.cout = ch;
        L.p_gn1_w = add(pre + "norm.weight", ch);
        L.p_gn1_b = add(pre + "norm.bias", ch);
        L.c1 = conv(pre + "qkv.", 3 * ch, ch, 1);
        L.c2 = conv(pre + "proj_out.", ch, ch, 1);
        return L;
    }
    bool has_attn(int ds) const {
        for (int i = 0; i < cfg.n_attn; i++)
            if (cfg.attention_resolutions[i] == ds) return true;
        return false;
    }
};

namespace {

void policy_of(const dlpm_unet *u, ConvLaunch &L) {
    L.gen = u->gen;
    L.dispatch_B = u->dispatch_B;
    L.gemm = u->gemm;
}

// The block structure the reference constructor produces (unet.py:334-436).
void build_arch(dlpm_unet *u) {
    const dlpm_unet_config &c = u->cfg;
    const int mc = c.model_channels;
    u->ted = 4 * mc;
    u->te0 = u->conv("time_embed.0.", u->ted, mc, 1);
    u->te2 = u->conv("time_embed.2.", u->ted, u->ted, 1);
    {
        Layer L;
        L.kind = L_STEM; L.pre = "input_blocks.0.0."; L.cin = c.in_channels; L.cout = mc;
        L.c1 = u->conv(L.pre, mc, c.in_channels, 3);
        u->in_blocks.push_back({L});
        u->skip_ch.push_back(mc);
    }
    int ch = mc, ds = 1;
    for (int level = 0; level < c.n_mult; level++) {
        const int mult = c.channel_mult[level];
        for (int r = 0; r < c.num_res_blocks; r++) {
            const std::string pre = "input_blocks." + std::to_string(u->in_blocks.size()) + ".";
            std::vector<Layer> seq;
            seq.push_back(u->res(pre + "0.", ch, mult * mc));
            ch = mult * mc;
            if (u->has_attn(ds)) seq.push_back(u->attn(pre + "1.", ch));
            u->in_blocks.push_back(seq);
            u->skip_ch.push_back(ch);
        }
        if (level != c.n_mult - 1) {
            Layer L;
            L.kind = L_DOWN; L.pre = "input_blocks." + std::to_string(u->in_blocks.size()) + ".0.op.";
            L.cin = L.cout = ch;
            L.c1 = u->conv(L.pre, ch, ch, 3);
            u->in_blocks.push_back({L});
            u->skip_ch.push_back(ch);
            ds *= 2;
        }
    }
    u->mid.push_back(u->res("middle_block.0.", ch, ch));
    u->mid.push_back(u->attn("middle_block.1.", ch));
    u->mid.push_back(u->res("middle_block.2.", ch, ch));
    std::vector<int> stack = u->skip_ch;
    for (int level = c.n_mult - 1; level >= 0; level--) {
        const int mult = c.channel_mult[level];
        for (int i = 0; i <= c.num_res_blocks; i++) {
            const std::string pre = "output_blocks." + std::to_string(u->out_blocks.size()) + ".";
            const int ich = stack.back();
            stack.pop_back();
            std::vector<Layer> seq;
            seq.push_back(u->res(pre + "0.", ch + ich, mc * mult));
            ch = mc * mult;
            int k = 1;
            if (u->has_attn(ds)) seq.push_back(u->attn(pre + std::to_string(k++) + ".", ch));
            if (level && i == c.num_res_blocks) {
                Layer L;
                L.kind = L_UP; L.pre = pre + std::to_string(k++) + ".conv.";
                L.cin = L.cout = ch;
                L.c1 = u->conv(L.pre, ch, ch, 3);
                seq.push_back(L);
                ds /= 2;
            }
            u->out_blocks.push_back(seq);
        }
    }
    u->final_ch = ch;
    u->p_head_gn_w = u->add("out.0.weight", ch);
    u->p_head_gn_b = u->add("out.0.bias", ch);
    u->head = u->conv("out.2.", c.out_channels, mc, 3);
}

bool ws_gemm_enabled() {   // DLPM_WS1X1=1: route the UNet's 1x1 convolutions through the weight-streaming kernel (experiment)
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLPM_WS1X1"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

int prep_conv(dlpm_unet *u, ConvW &c, int C0, int boundary) {  // boundary: 0 none, 1 NCHW input (stem), 2 NCHW output (head), 3 time MLP, 4 stride-2 downsampling
    ConvLaunch probe;
    probe.C0 = C0; probe.C1 = c.cin - C0; probe.Cout = c.cout; probe.ks = c.ks;
    probe.in_nchw = boundary == 1; probe.out_nchw = boundary == 2;
    c.use_igemm = igemm_supported(probe);
    const float *src = u->params[c.p_w].dev;
    if (c.use_igemm && c.ks == 1) {  // [O][I] row-major is already the igemm layout
        c.w_dev = const_cast<float *>(src);
        if (small_weight_ok(c.cout, c.cin, 1) && boundary == 0) {
            DLPM_HIP(hipMalloc(&c.w_rs, (size_t)small_weight_floats(c.cout, c.cin, 1) * sizeof(float)));
            int r = relayout_weight_small(src, c.w_rs, c.cout, c.cin, 1, nullptr);
            if (r != DLPM_OK) return r;
        }
        if (c.cout % 128 == 0 && c.cin % 32 == 0 && boundary == 0) {
            DLPM_HIP(hipMalloc(&c.w_split, (size_t)split_weight_floats(c.cout, c.cin, 1) * sizeof(float)));
            int r = relayout_weight_split(src, c.w_split, c.cout, c.cin, 1, nullptr);
            if (r != DLPM_OK) return r;
        }
        if (ws_gemm_enabled() && c.cin % 32 == 0 && boundary == 0) {   // fragment order for the weight-streaming GEMM (TAPS = 1)
            DLPM_HIP(hipMalloc(&c.w_frag, (size_t)frag_weight_floats(c.cout, c.cin, 1) * sizeof(float)));
            return relayout_weight_frag(src, c.w_frag, c.cout, c.cin, nullptr, 1);
        }
        return DLPM_OK;
    }
    DLPM_HIP(hipMalloc(&c.w_dev, (size_t)c.cout * c.cin * c.ks * c.ks * sizeof(float)));
    c.owns = true;
    if (c.ks == 3 && c.cout <= 4 && c.cin % 32 == 0) {
        DLPM_HIP(hipMalloc(&c.w_small, (size_t)9 * c.cin * 4 * sizeof(float)));
        int r = relayout_weight_head(src, c.w_small, c.cout, c.cin, nullptr);
        if (r != DLPM_OK) return r;
        if (c.cout <= 3) {
            DLPM_HIP(hipMalloc(&c.w_taps, (size_t)head_taps_rows(c.cout) * c.cin * sizeof(float)));
            r = relayout_weight_head_taps(src, c.w_taps, c.cout, c.cin, nullptr);
            if (r != DLPM_OK) return r;
            if (c.cin <= 128) {
                DLPM_HIP(hipMalloc(&c.w_hfused, (size_t)head_fused_weight_floats(c.cin) * sizeof(float)));
                r = relayout_weight_head_fused(src, c.w_hfused, c.cout, c.cin, nullptr);
                if (r != DLPM_OK) return r;
            }
        }
    }
    if (c.use_igemm && c.ks == 3 && boundary == 4 && c.cout % 128 == 0 && c.cin % 32 == 0) {   // stride-2 downsampling convolution
        DLPM_HIP(hipMalloc(&c.w_split, (size_t)split_weight_floats(c.cout, c.cin, 9) * sizeof(float)));
        int r = relayout_weight_split(src, c.w_split, c.cout, c.cin, 9, nullptr);
        if (r != DLPM_OK) return r;
    }
    if (c.use_igemm && c.ks == 3 && c.cin % 32 == 0) {
        DLPM_HIP(hipMalloc(&c.w_frag, (size_t)frag_weight_floats(c.cout, c.cin) * sizeof(float)));
        int r = relayout_weight_frag(src, c.w_frag, c.cout, c.cin, nullptr);
        if (r != DLPM_OK) return r;
        if (c.cout % 64 == 0 && c.cin % 16 == 0 && boundary == 0) {   // Winograd-domain copy (stride-1 launches pick it up)
            DLPM_HIP(hipMalloc(&c.w_wino, (size_t)wino_weight_floats(c.cout, c.cin) * sizeof(float)));
            r = relayout_weight_wino(src, c.w_wino, c.cout, c.cin, nullptr);
            if (r != DLPM_OK) return r;
        }
        if (wino4_enabled() && c.cout % 32 == 0 && c.cin % 8 == 0 && boundary == 0) {   // F(4x4,3x3) copy (128-, 64- or 32-channel n-tiles)
            DLPM_HIP(hipMalloc(&c.w_wino4, (size_t)wino4_weight_floats(c.cout, c.cin) * sizeof(float)));
            r = relayout_weight_wino4(src, c.w_wino4, c.cout, c.cin, nullptr);
            if (r != DLPM_OK) return r;
        }
    }
    if (small_weight_ok(c.cout, c.cin, c.ks) && boundary == 0) {
        DLPM_HIP(hipMalloc(&c.w_rs, (size_t)small_weight_floats(c.cout, c.cin, c.ks) * sizeof(float)));
        int r = relayout_weight_small(src, c.w_rs, c.cout, c.cin, c.ks, nullptr);
        if (r != DLPM_OK) return r;
    }
    return relayout_weight(src, c.w_dev, c.cout, c.cin, c.ks, c.use_igemm, nullptr);
}

// the launch descriptor with the net's policy and this convolution's weight layouts attached
void attach_conv(const dlpm_unet *u, const ConvW &c, ConvLaunch &L) {
    policy_of(u, L);
    L.w = c.w_dev;
    L.w_frag = c.w_frag;
    L.w_wino = c.w_wino;
    L.w_wino4 = c.w_wino4;
    L.w_wino4_n64 = c.w_wino4_n64;
    L.w_wino4_n32 = c.w_wino4_n32;
    L.w_small = c.w_small;
    L.w_taps = c.w_taps;
    L.w_hfused = c.w_hfused;
    L.w_split = c.w_split;
    L.ws_gemm = (c.ks == 1 && c.w_frag) ? 1 : 0;
    L.ks = c.ks;
    L.Cout = c.cout;
}

int run_conv(const dlpm_unet *u, const ConvW &c, ConvLaunch L, hipStream_t st, const HeadUpdate *hu = nullptr, float *head_scratch = nullptr) {
    policy_of(u, L);
    L.w = c.w_dev;
    L.w_frag = c.w_frag;
    L.w_wino = c.w_wino;
    L.w_wino4 = c.w_wino4;
    L.w_wino4_n64 = c.w_wino4_n64;
    L.w_wino4_n32 = c.w_wino4_n32;
    L.w_small = c.w_small;
    L.w_taps = c.w_taps;
    L.w_hfused = c.w_hfused;
    L.w_split = c.w_split;
    L.ws_gemm = (c.ks == 1 && c.w_frag) ? 1 : 0;
    L.ks = c.ks;
    L.Cout = c.cout;
    if (head_fused_ok(L)) return launch_conv_head_fused(L, hu, st);
    if (head_scratch && head_gemm_ok(L)) return launch_conv_head_gemm(L, hu, head_scratch, st);
    if (head_conv_ok(L)) return launch_conv_head(L, hu, st);
    if (hu) { set_error("unet: the head convolution of this net cannot carry the fused update"); return DLPM_ERR_UNSUPPORTED; }
    return c.use_igemm ? launch_conv_igemm(L, st) : launch_conv_fallback(L, st);
}

#define TRY(expr)                  \
    do {                           \
        int _r = (expr);           \
        if (_r != DLPM_OK) return _r; \
    } while (0)

struct Ctx {
    dlpm_unet *u;
    int B;
    Bump ws;
    hipStream_t st;
    float *embout = nullptr;
    bool uniform_t = false;      // every sample has the same timestep: the time MLP and the emb linears run on ONE row
    const HeadUpdate *hu = nullptr;   // the sampler's reverse update, fused into the head convolution (dlpm_unet_forward_update)
    bool dry() const { return ws.dry; }
};

// Split-K factor of a 3x3 stride-1 convolution of the plan (conv_splitk.hip: conv_ksplit_for): a function of the layer and the declared batch
int ksplit_of(const Ctx &cx, const ConvW &c, ConvLaunch L) {
    attach_conv(cx.u, c, L);
    L.bias = nullptr;
    return conv_ksplit_for(L);
}
// ... and the launch: S grid copies write their partial outputs into a workspace buffer, launch_splitk_reduce adds bias / residual.  (The
// caller's dry run reserves the buffer with reserve_ksplit at the point where this runs.)
int run_conv3(Ctx &cx, const ConvW &c, const ConvLaunch &L) {
    const int S = ksplit_of(cx, c, L);
    if (S <= 1) return run_conv(cx.u, c, L, cx.st);
    const int64_t npix = (int64_t)L.B * L.Hout * L.Wout;
    float *part = cx.ws.alloc((int64_t)S * npix * c.cout);
    ConvLaunch P = L;
    P.bias = nullptr; P.res0 = nullptr; P.res1 = nullptr; P.R0 = 0; P.stats_out = nullptr; P.out = part; P.ksplit = S;
    int rc = run_conv(cx.u, c, P, cx.st);
    if (rc == DLPM_OK) rc = launch_splitk_reduce(part, S, npix, c.cout, L.bias, L.res0, L.res1, L.R0, L.out, cx.st);
    cx.ws.release(part);
    return rc;
}
void reserve_ksplit(Ctx &cx, const ConvW &c, const ConvLaunch &L) {     // dry run: the partial buffer's share of the workspace peak
    const int S = ksplit_of(cx, c, L);
    if (S > 1) cx.ws.release(cx.ws.alloc((int64_t)S * L.B * L.Hout * L.Wout * c.cout));
}

void release_res_temps(Ctx &cx, float *cA1, float *cB1, Tensor4 &h1, float *cA2, float *cB2, float *sk) {
    cx.ws.release(cA1);
    cx.ws.release(cB1);
    drop(cx.ws, h1);
    cx.ws.release(cA2);
    cx.ws.release(cB2);
    cx.ws.release(sk);
}

int run_res(Ctx &cx, const Layer &L, Tensor4 x0, Tensor4 x1, Tensor4 *out) {
    dlpm_unet *u = cx.u;
    const int B = cx.B, H = x0.H, W = x0.W, HW = H * W;
    const int C0 = x0.C, C1 = x1.C, Cin = C0 + C1, Co = L.cout;
    {
        // small images, 64 channels: the whole block in one launch (block_small.hip).  A function of the layer and of the net's
        // policy only -- never of the batch -- like every other kernel choice (the fused block rounds differently).
        ResSmallLaunch r;
        r.x0 = x0.p; r.x1 = x1.p; r.C0 = C0; r.C1 = C1; r.B = B; r.H = H; r.W = W;
        r.w1f = L.c1.w_rs; r.w2f = L.c2.w_rs; r.wsf = L.has_skip ? L.skip.w_rs : nullptr;
        if (u->gen == DLPM_CONV_AUTO && Co == 64 && (!L.has_skip || r.wsf) && res_small_ok(r)) {
            float *o = cx.ws.alloc((int64_t)B * HW * Co);
            out->p = o; out->C = Co; out->H = H; out->W = W;
            out->stats = reinterpret_cast<float2 *>(cx.ws.alloc((int64_t)2 * B * Co));
            out->stats_px = HW;
            if (cx.dry()) return DLPM_OK;
            r.gn1_w = u->params[L.p_gn1_w].dev; r.gn1_b = u->params[L.p_gn1_b].dev;
            r.gn2_w = u->params[L.p_gn2_w].dev; r.gn2_b = u->params[L.p_gn2_b].dev;
            r.b1 = u->params[L.c1.p_b].dev; r.b2 = u->params[L.c2.p_b].dev;
            r.bs = L.has_skip ? u->params[L.skip.p_b].dev : nullptr;
            r.emb = cx.embout; r.emb_stride = cx.uniform_t ? 0 : u->emb_total; r.emb_off = L.emb_off;
            r.out = o; r.stats_out = out->stats;
            return launch_resblock_small(r, cx.st);
        }
    }
    if (u->gen == DLPM_CONV_AUTO && ((Co == 32 && H == 32 && W == 32) || (Co == 64 && H == 16 && W == 16)) && L.c1.w_wino4 && L.c2.w_wino4) {
        // 32 channels on 32x32 images: the whole block in ONE launch, one workgroup per image (conv_wino4.hip: k_resblock_wino4_img) -- the
        // bits of the separate launches below (a function of the layer and the policy only).  GroupNorm-1 runs inside the kernel when both
        // sources carry statistics; the 1x1 skip convolution of the concat blocks stays a launch of its own.
        // 64 channels on 16x16 images likewise (k_resblock_wino4_img16: the intermediate stays in LDS; it rounds differently from the launches below).
        ResImgLaunch r;
        r.x0 = x0.p; r.x1 = x1.p; r.C0 = C0; r.C1 = C1; r.B = B; r.H = H;
        const bool from_stats = x0.stats && (C1 == 0 || x1.stats);
        r.w1 = L.c1.w_wino4; r.w2 = L.c2.w_wino4;
        float *cA1 = from_stats ? nullptr : cx.ws.alloc((int64_t)B * Cin), *cB1 = from_stats ? nullptr : cx.ws.alloc((int64_t)B * Cin);
        float *sk = L.has_skip ? cx.ws.alloc((int64_t)B * HW * Co) : nullptr;
        float *hb = H == 32 ? cx.ws.alloc((int64_t)B * HW * Co) : nullptr;
        float *o = cx.ws.alloc((int64_t)B * HW * Co);
        r.hbuf = hb; r.res = L.has_skip ? sk : x0.p;
        if (from_stats) {
            r.st0 = x0.stats; r.nt0 = HW / x0.stats_px;
            if (C1) { r.st1 = x1.stats; r.nt1 = HW / x1.stats_px; }
        } else {
            r.coefA1 = cA1; r.coefB1 = cB1;
        }
        if ((L.has_skip || C1 == 0) && res_img_ok(r)) {
            out->p = o; out->C = Co; out->H = H; out->W = W;
            plan_stats(cx.u, cx.ws, *out, L.c2, B, Co, 1, 0);
            auto done = [&]() { cx.ws.release(cA1); cx.ws.release(cB1); cx.ws.release(sk); cx.ws.release(hb); };
            if (cx.dry()) { done(); return DLPM_OK; }
            if (!from_stats)
                TRY(gn_any(x0, x1, B, Cin < 32 ? Cin : 32, u->params[L.p_gn1_w].dev, u->params[L.p_gn1_b].dev, nullptr, 0, 0, cA1, cB1, cx.st));
            if (L.has_skip) {
                ConvLaunch s;
                s.src0 = x0.p; s.src1 = x1.p; s.C0 = C0; s.C1 = C1; s.B = B; s.Hin = s.Hout = H; s.Win = s.Wout = W;
                s.bias = u->params[L.skip.p_b].dev; s.out = sk;
                TRY(run_conv(u, L.skip, s, cx.st));
            }
            r.gn1_w = u->params[L.p_gn1_w].dev; r.gn1_b = u->params[L.p_gn1_b].dev;
            r.gn2_w = u->params[L.p_gn2_w].dev; r.gn2_b = u->params[L.p_gn2_b].dev;
            r.b1 = u->params[L.c1.p_b].dev; r.b2 = u->params[L.c2.p_b].dev;
            r.emb = cx.embout; r.emb_stride = cx.uniform_t ? 0 : u->emb_total; r.emb_off = L.emb_off;
            r.out = o; r.stats_out = out->stats;
            TRY(launch_resblock_img(r, cx.st));
            done();
            return DLPM_OK;
        }
        cx.ws.release(cA1); cx.ws.release(cB1); cx.ws.release(sk); cx.ws.release(hb); cx.ws.release(o);
    }
    float *cA1 = cx.ws.alloc((int64_t)B * Cin), *cB1 = cx.ws.alloc((int64_t)B * Cin);
    Tensor4 h1;
    h1.C = Co; h1.H = H; h1.W = W;
    h1.p = cx.ws.alloc((int64_t)B * HW * Co);
    plan_stats(cx.u, cx.ws, h1, L.c1, B, C0, 1, 0);
    float *cA2 = cx.ws.alloc((int64_t)B * Co), *cB2 = cx.ws.alloc((int64_t)B * Co);
    float *sk = L.has_skip ? cx.ws.alloc((int64_t)B * HW * Co) : nullptr;
    float *o = cx.ws.alloc((int64_t)B * HW * Co);
    out->p = o; out->C = Co; out->H = H; out->W = W;
    plan_stats(cx.u, cx.ws, *out, L.c2, B, Co, 1, 0);
    if (cx.dry()) {
        ConvLaunch g;      // (geometry of the block's two 3x3 convolutions: what ksplit_of looks at)
        g.C0 = C0; g.C1 = C1; g.B = B; g.Hin = g.Hout = H; g.Win = g.Wout = W;
        reserve_ksplit(cx, L.c1, g);
        g.C0 = Co; g.C1 = 0;
        reserve_ksplit(cx, L.c2, g);
        release_res_temps(cx, cA1, cB1, h1, cA2, cB2, sk);
        return DLPM_OK;
    }
    const int G1 = Cin < 32 ? Cin : 32, G2 = Co < 32 ? Co : 32;
    TRY(gn_any(x0, x1, B, G1, u->params[L.p_gn1_w].dev, u->params[L.p_gn1_b].dev, nullptr, 0, 0, cA1, cB1, cx.st));
    ConvLaunch a;
    a.src0 = x0.p; a.src1 = x1.p; a.C0 = C0; a.C1 = C1; a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W;
    a.bias = u->params[L.c1.p_b].dev; a.coefA = cA1; a.coefB = cB1; a.act_silu = 1; a.out = h1.p; a.stats_out = h1.stats;
    TRY(run_conv3(cx, L.c1, a));
    TRY(gn_any(h1, Tensor4(), B, G2, u->params[L.p_gn2_w].dev, u->params[L.p_gn2_b].dev, cx.embout,
               cx.uniform_t ? 0 : u->emb_total, L.emb_off, cA2, cB2, cx.st));   // row pitch 0: all samples read the one emb row
    ConvLaunch b;
    b.src0 = h1.p; b.C0 = Co; b.B = B; b.Hin = b.Hout = H; b.Win = b.Wout = W;
    b.bias = u->params[L.c2.p_b].dev; b.coefA = cA2; b.coefB = cB2; b.act_silu = 1; b.out = o; b.stats_out = out->stats;
    if (L.has_skip) {
        ConvLaunch s;
        s.src0 = x0.p; s.src1 = x1.p; s.C0 = C0; s.C1 = C1; s.B = B; s.Hin = s.Hout = H; s.Win = s.Wout = W;
        s.bias = u->params[L.skip.p_b].dev; s.out = sk;
        TRY(run_conv(u, L.skip, s, cx.st));
        b.res0 = sk; b.R0 = Co;
    } else {
        b.res0 = x0.p; b.res1 = x1.p; b.R0 = C0;
    }
    TRY(run_conv3(cx, L.c2, b));
    release_res_temps(cx, cA1, cB1, h1, cA2, cB2, sk);
    return DLPM_OK;
}

int run_attn(Ctx &cx, const Layer &L, Tensor4 x, Tensor4 *out) {
    dlpm_unet *u = cx.u;
    const int B = cx.B, C = x.C, T = x.H * x.W;
    {
        // small images: GroupNorm -> qkv -> attention -> proj -> + x in one launch, one workgroup per image (block_small.hip)
        AttnSmallLaunch a;
        a.x = x.p; a.C = C; a.heads = u->cfg.num_heads; a.B = B; a.H = x.H; a.W = x.W;
        a.wqkv = L.c1.w_rs; a.wproj = L.c2.w_rs;
        const bool whole16 = attn16_ok(a);
        if (u->gen == DLPM_CONV_AUTO && (attn_small_ok(a) || whole16)) {
            *out = x;
            out->p = cx.ws.alloc((int64_t)B * T * C);
            out->stats = reinterpret_cast<float2 *>(cx.ws.alloc((int64_t)2 * B * C));
            out->stats_px = T;
            if (cx.dry()) return DLPM_OK;
            a.gn_w = u->params[L.p_gn1_w].dev; a.gn_b = u->params[L.p_gn1_b].dev;
            a.bqkv = u->params[L.c1.p_b].dev; a.bproj = u->params[L.c2.p_b].dev;
            a.out = out->p; a.stats_out = out->stats;
            return whole16 ? launch_attnblock16(a, cx.st) : launch_attnblock_small(a, cx.st);
        }
    }
    float *cA = cx.ws.alloc((int64_t)B * C), *cB = cx.ws.alloc((int64_t)B * C);
    float *qkv = cx.ws.alloc((int64_t)B * T * 3 * C);
    float *av = cx.ws.alloc((int64_t)B * T * C);
    float *o = cx.ws.alloc((int64_t)B * T * C);
    const Tensor4 xin = x;
    *out = x;
    out->p = o;
    plan_stats(cx.u, cx.ws, *out, L.c2, B, C, 1, 0);
    auto done = [&]() { cx.ws.release(cA); cx.ws.release(cB); cx.ws.release(qkv); cx.ws.release(av); };
    if (cx.dry()) {
        done();
        return DLPM_OK;
    }
    AttnSmallLaunch gq;
    gq.x = x.p; gq.C = C; gq.heads = u->cfg.num_heads; gq.B = B; gq.H = x.H; gq.W = x.W; gq.wqkv = L.c1.w_rs;
    if (u->gen == DLPM_CONV_AUTO && gnqkv_small_ok(gq)) {
        // 16x16 images: GroupNorm + qkv of one image in one launch (block_small.hip)
        gq.gn_w = u->params[L.p_gn1_w].dev; gq.gn_b = u->params[L.p_gn1_b].dev; gq.bqkv = u->params[L.c1.p_b].dev; gq.out = qkv;
        TRY(launch_gnqkv_small(gq, cx.st));
    } else {
        TRY(gn_any(xin, Tensor4(), B, C < 32 ? C : 32, u->params[L.p_gn1_w].dev, u->params[L.p_gn1_b].dev, nullptr, 0, 0, cA, cB,
                   cx.st));
        ConvLaunch q;
        q.src0 = x.p; q.C0 = C; q.B = B; q.Hin = q.Hout = x.H; q.Win = q.Wout = x.W;
        q.bias = u->params[L.c1.p_b].dev; q.coefA = cA; q.coefB = cB; q.out = qkv;
        TRY(run_conv(u, L.c1, q, cx.st));
    }
    TRY(launch_attention(qkv, av, B, T, C, u->cfg.num_heads, cx.st));
    ConvLaunch p;
    p.src0 = av; p.C0 = C; p.B = B; p.Hin = p.Hout = x.H; p.Win = p.Wout = x.W;
    p.bias = u->params[L.c2.p_b].dev; p.res0 = x.p; p.R0 = C; p.out = o; p.stats_out = out->stats;
    TRY(run_conv(u, L.c2, p, cx.st));
    done();
    return DLPM_OK;
}

// free0 / free1: the sequence is the last consumer of x0 / x1 (released once its first layer has been enqueued)
int run_seq(Ctx &cx, const std::vector<Layer> &seq, Tensor4 x0, Tensor4 x1, const float *x_nchw, Tensor4 *out, bool free0 = false,
            bool free1 = false) {
    dlpm_unet *u = cx.u;
    const int B = cx.B;
    Tensor4 h = x0;
    for (size_t i = 0; i < seq.size(); i++) {
        const Layer &L = seq[i];
        Tensor4 o;
        switch (L.kind) {
            case L_STEM: {
                const int S = u->cfg.image_size;
                o.C = L.cout; o.H = S; o.W = S;
                o.p = cx.ws.alloc((int64_t)B * S * S * L.cout);
                plan_stats(cx.u, cx.ws, o, L.c1, B, L.cin, 1, 0, true);
                if (!cx.dry()) {
                    ConvLaunch a;
                    a.src0 = x_nchw; a.C0 = L.cin; a.B = B; a.Hin = a.Hout = S; a.Win = a.Wout = S;
                    a.bias = u->params[L.c1.p_b].dev; a.out = o.p; a.in_nchw = 1; a.stats_out = o.stats;
                    TRY(run_conv(u, L.c1, a, cx.st));
                }
                break;
            }
            case L_RES:
                TRY(run_res(cx, L, h, (i == 0) ? x1 : Tensor4(), &o));
                break;
            case L_ATTN:
                TRY(run_attn(cx, L, h, &o));
                break;
            case L_DOWN:
            case L_UP: {
                const bool up = L.kind == L_UP;
                o.C = h.C;
                o.H = up ? h.H * 2 : (h.H - 1) / 2 + 1;
                o.W = up ? h.W * 2 : (h.W - 1) / 2 + 1;
                o.p = cx.ws.alloc((int64_t)B * o.H * o.W * o.C);
                plan_stats(cx.u, cx.ws, o, L.c1, B, h.C, up ? 1 : 2, up ? 1 : 0);
                {
                    ConvLaunch a;
                    a.src0 = h.p; a.C0 = h.C; a.B = B; a.Hin = h.H; a.Win = h.W; a.Hout = o.H; a.Wout = o.W;
                    a.stride = up ? 1 : 2; a.ups = up ? 1 : 0;
                    a.bias = u->params[L.c1.p_b].dev; a.out = o.p; a.stats_out = o.stats;
                    if (cx.dry()) { if (up) reserve_ksplit(cx, L.c1, a); }
                    else TRY(up ? run_conv3(cx, L.c1, a) : run_conv(u, L.c1, a, cx.st));
                }
                break;
            }
        }
        if (i == 0) {
            if (free0) drop(cx.ws, x0);
            if (free1) drop(cx.ws, x1);
        } else {
            drop(cx.ws, h);      // an intermediate of this sequence: the layer just enqueued was its only reader
        }
        h = o;
    }
    *out = h;
    return DLPM_OK;
}

__global__ void k_table_row(const float *__restrict__ table, const int32_t *__restrict__ row, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table[(int64_t)(*row) * n + i];
}

// time embedding -> time MLP -> the row-concatenated per-ResBlock emb linears, for M rows of t (unet.py:147-150, 336-338, 470)
int time_path(dlpm_unet *u, const float *t, int M, float *e0, float *e1, float *e2, float *out, hipStream_t st) {
    const int mc = u->cfg.model_channels, ted = u->ted;
    TRY(launch_timestep_embedding(t, e0, M, mc, st));
    ConvLaunch g;
    g.B = M; g.Hin = g.Win = g.Hout = g.Wout = 1;
    g.src0 = e0; g.C0 = mc; g.bias = u->params[u->te0.p_b].dev; g.out = e1;
    TRY(run_conv(u, u->te0, g, st));
    g.src0 = e1; g.C0 = ted; g.bias = u->params[u->te2.p_b].dev; g.out = e2; g.act_silu = 1;
    TRY(run_conv(u, u->te2, g, st));
    g.src0 = e2; g.bias = u->embcat_b; g.out = out;
    return run_conv(u, u->embcat, g, st);
}

// optional [T][emb_total] table of the time path's output (time embedding -> time MLP -> the per-ResBlock emb linears) and the device
// step counter that picks its row: bound by the sampler around its forward calls (dlpm_unet_bind_time_table), per host thread
thread_local const dlpm_unet *tl_time_net = nullptr;
thread_local const float *tl_time_table = nullptr;
thread_local const int32_t *tl_time_index = nullptr;

int walk(dlpm_unet *u, Ctx &cx, const float *x, const float *t, float *eps) {
    const int B = cx.B, mc = u->cfg.model_channels, ted = u->ted;
    // time embedding -> time MLP -> the per-ResBlock emb linears (unet.py:147-150, 336-338, 470).  In the sampling loop t is
    // the same for the whole batch (GenerativeLevyProcess.py:319), so these run for one row instead of B identical ones
    // (a row's dot products do not depend on how many rows the GEMM has: bit-identical to the batched form)
    const int Bt = cx.uniform_t ? 1 : B;
    float *e0 = cx.ws.alloc((int64_t)B * mc), *e1 = cx.ws.alloc((int64_t)B * ted), *e2 = cx.ws.alloc((int64_t)B * ted);
    cx.embout = cx.ws.alloc((int64_t)B * u->emb_total);
    if (!cx.dry()) {
        if (cx.uniform_t && tl_time_net == u && tl_time_table) {
            // the whole time path is a function of the step index alone: its output for every step was computed once
            // (dlpm_unet_time_embeddings, same kernels, one row per step -- same bits) and the step reads its row
            k_table_row<<<(unsigned)ceil_div(u->emb_total, 256), 256, 0, cx.st>>>(tl_time_table, tl_time_index, u->emb_total, cx.embout);
            DLPM_LAUNCH_CHECK();
        } else {
            TRY(time_path(u, t, Bt, e0, e1, e2, cx.embout, cx.st));
        }
    }
    cx.ws.release(e0);
    cx.ws.release(e1);
    cx.ws.release(e2);
    u->feats.clear();
    std::vector<Tensor4> hs;
    Tensor4 h;
    // every input-block output waits on the skip stack for its output block; the middle block's input is the last of them
    for (auto &seq : u->in_blocks) {
        TRY(run_seq(cx, seq, h, Tensor4(), x, &h));
        hs.push_back(h);
        u->feats.push_back(h);
    }
    TRY(run_seq(cx, u->mid, h, Tensor4(), nullptr, &h));
    u->feats.push_back(h);
    for (auto &seq : u->out_blocks) {
        Tensor4 skip = hs.back();
        hs.pop_back();
        if (skip.H != h.H || skip.W != h.W) {
            set_error("unet: skip connection %dx%d does not match %dx%d (image_size must be divisible by 2^(levels-1))",
                      skip.H, skip.W, h.H, h.W);
            return DLPM_ERR_ARG;
        }
        TRY(run_seq(cx, seq, h, skip, nullptr, &h, true, true));   // last readers of the previous output and of the skip tensor
        u->feats.push_back(h);
    }
    float *cA = cx.ws.alloc((int64_t)B * h.C), *cB = cx.ws.alloc((int64_t)B * h.C);
    // the head as a GEMM onto its 9 Cout tap channels + gather: the tap-channel tensor P lives in the arena
    float *P = nullptr;
    {
        ConvLaunch a;
        a.C0 = h.C; a.B = B; a.Hin = a.Hout = h.H; a.Win = a.Wout = h.W; a.ks = 3; a.Cout = u->head.cout; a.w_taps = u->head.w_taps; a.out_nchw = 1;
        a.w_hfused = u->head.w_hfused; a.coefA = cA; a.coefB = cB; a.act_silu = 1;
        if (!head_fused_ok(a) && head_gemm_ok(a)) P = cx.ws.alloc(head_gemm_scratch_floats(a));
    }
    if (!cx.dry()) {
        TRY(gn_any(h, Tensor4(), B, h.C < 32 ? h.C : 32, u->params[u->p_head_gn_w].dev, u->params[u->p_head_gn_b].dev, nullptr,
                   0, 0, cA, cB, cx.st));
        ConvLaunch a;
        a.src0 = h.p; a.C0 = h.C; a.B = B; a.Hin = a.Hout = h.H; a.Win = a.Wout = h.W;
        a.bias = u->params[u->head.p_b].dev; a.coefA = cA; a.coefB = cB; a.act_silu = 1; a.out = eps; a.out_nchw = 1;
        TRY(run_conv(u, u->head, a, cx.st, cx.hu, P));
    }
    if (!cx.ws.reuse) return DLPM_OK;
    for (auto &f : u->feats) f.p = nullptr;     // the arena has recycled them: dlpm_unet_get_feature needs dlpm_unet_keep_features
    return DLPM_OK;
}

int64_t count_flops(dlpm_unet *u) {
    // 2*MAC per sample: conv / linear / attention matmuls
    int64_t f = 0;
    const int S = u->cfg.image_size;
    auto convf = [&](const ConvW &c, int H, int W) { f += 2LL * c.cout * c.cin * c.ks * c.ks * H * W; };
    convf(u->te0, 1, 1);
    convf(u->te2, 1, 1);
    f += 2LL * u->emb_total * u->ted;
    int H = S;
    auto seqf = [&](const std::vector<Layer> &seq) {
        for (auto &L : seq) {
            switch (L.kind) {
                case L_STEM: convf(L.c1, H, H); break;
                case L_RES:
                    convf(L.c1, H, H); convf(L.c2, H, H);
                    if (L.has_skip) convf(L.skip, H, H);
                    break;
                case L_ATTN:
                    convf(L.c1, H, H); convf(L.c2, H, H);
                    f += 4LL * (H * H) * (H * H) * L.cin;
                    break;
                case L_DOWN: H = (H - 1) / 2 + 1; convf(L.c1, H, H); break;
                case L_UP: H *= 2; convf(L.c1, H, H); break;
            }
        }
    };
    for (auto &s : u->in_blocks) seqf(s);
    seqf(u->mid);
    for (auto &s : u->out_blocks) seqf(s);
    convf(u->head, H, H);
    return f;
}

}  // namespace

extern "C" int dlpm_unet_create(const dlpm_unet_config *cfg, dlpm_unet **out) {
    DLPM_CHECK_ARG(cfg && out, "dlpm_unet_create: null argument");
    DLPM_CHECK_ARG(cfg->in_channels > 0 && cfg->model_channels > 0 && cfg->out_channels > 0 && cfg->num_res_blocks > 0,
                   "dlpm_unet_create: channels / res blocks must be positive");
    DLPM_CHECK_ARG(cfg->n_mult > 0 && cfg->n_mult <= 8 && cfg->n_attn >= 0 && cfg->n_attn <= 8,
                   "dlpm_unet_create: bad channel_mult / attention_resolutions length");
    DLPM_CHECK_ARG(cfg->num_heads > 0 && cfg->image_size > 0, "dlpm_unet_create: bad num_heads / image_size");
    DLPM_CHECK_ARG(cfg->image_size % (1 << (cfg->n_mult - 1)) == 0,
                   "dlpm_unet_create: image_size %d not divisible by 2^%d (the reference UNet fails on such sizes too)",
                   cfg->image_size, cfg->n_mult - 1);
    dlpm_unet *u = new dlpm_unet();
    u->cfg = *cfg;
    build_arch(u);
    *out = u;
    return DLPM_OK;
}

extern "C" int dlpm_unet_num_params(const dlpm_unet *net) { return net ? (int)net->params.size() : 0; }

extern "C" const char *dlpm_unet_param_key(const dlpm_unet *net, int i, int64_t *numel_out) {
    if (!net || i < 0 || i >= (int)net->params.size()) return nullptr;
    if (numel_out) *numel_out = net->params[i].numel;
    return net->params[i].key.c_str();
}

extern "C" int dlpm_unet_set_param(dlpm_unet *net, const char *key, const float *host, int64_t numel) {
    DLPM_CHECK_ARG(net && key && host, "dlpm_unet_set_param: null argument");
    auto it = net->index.find(key);
    DLPM_CHECK_ARG(it != net->index.end(), "dlpm_unet_set_param: unexpected key '%s' for this architecture", key);
    Param &p = net->params[it->second];
    DLPM_CHECK_ARG(p.numel == numel, "dlpm_unet_set_param: '%s' has %lld elements, expected %lld", key, (long long)numel,
                   (long long)p.numel);
    if (!p.dev) DLPM_HIP(hipMalloc(&p.dev, (size_t)numel * sizeof(float)));
    DLPM_HIP(hipMemcpy(p.dev, host, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
    p.set = true;
    net->finalized = false;
    return DLPM_OK;
}

static void free_conv(ConvW &c) {
    if (c.owns && c.w_dev) (void)hipFree(c.w_dev);
    if (c.w_frag) (void)hipFree(c.w_frag);
    c.w_frag = nullptr;
    if (c.w_wino) (void)hipFree(c.w_wino);
    c.w_wino = nullptr;
    if (c.w_wino4) (void)hipFree(c.w_wino4);
    c.w_wino4 = nullptr;
    if (c.w_wino4_n64) (void)hipFree(c.w_wino4_n64);
    if (c.w_wino4_n32) (void)hipFree(c.w_wino4_n32);
    c.w_wino4_n64 = c.w_wino4_n32 = nullptr;
    if (c.w_small) (void)hipFree(c.w_small);
    c.w_small = nullptr;
    if (c.w_taps) (void)hipFree(c.w_taps);
    c.w_taps = nullptr;
    if (c.w_rs) (void)hipFree(c.w_rs);
    c.w_rs = nullptr;
    if (c.w_hfused) (void)hipFree(c.w_hfused);
    c.w_hfused = nullptr;
    if (c.w_split) (void)hipFree(c.w_split);
    c.w_split = nullptr;
    c.w_dev = nullptr;
    c.owns = false;
}

static void for_each_conv(dlpm_unet *u, void (*fn)(ConvW &)) {
    auto seq = [&](std::vector<Layer> &s) {
        for (auto &L : s) {
            fn(L.c1);
            if (L.kind == L_RES || L.kind == L_ATTN) fn(L.c2);
            if (L.kind == L_RES && L.has_skip) fn(L.skip);
        }
    };
    for (auto &s : u->in_blocks) seq(s);
    seq(u->mid);
    for (auto &s : u->out_blocks) seq(s);
    fn(u->te0);
    fn(u->te2);
    fn(u->head);
}

// Round 6: under a small DECLARED batch (dlpm_unet_set_conv_policy's dispatch_batch) the 128-channel-multiple F(4x4) layers may run on 64- / 32-
// channel n-tiles (conv_wino4.hip: wino4_nq_for), which read the same Winograd-domain weights in another fragment order.  The copies
// exist only while such a batch is declared (3x the F(4x4) weight bytes of those layers); built from the parameters, on the null stream.
constexpr int64_t NARROW_COPIES_MAX_BATCH = 512;
static int narrow_wino4_copies(dlpm_unet *u) {
    const bool want = u->dispatch_B > 0 && u->dispatch_B <= NARROW_COPIES_MAX_BATCH;
    int rc = DLPM_OK;
    auto one = [&](ConvW &c) {
        if (rc != DLPM_OK || !c.w_wino4 || c.cout % 128 != 0) return;
        if (!want) {
            if (c.w_wino4_n64) (void)hipFree(c.w_wino4_n64);
            if (c.w_wino4_n32) (void)hipFree(c.w_wino4_n32);
            c.w_wino4_n64 = c.w_wino4_n32 = nullptr;
            return;
        }
        for (int nq : {64, 32}) {
            float *&dst = nq == 64 ? c.w_wino4_n64 : c.w_wino4_n32;
            if (dst) continue;
            if (hipMalloc(&dst, (size_t)wino4_weight_floats(c.cout, c.cin) * sizeof(float)) != hipSuccess) { dst = nullptr; rc = DLPM_ERR_HIP; set_error("narrow_wino4_copies: hipMalloc failed"); return; }
            rc = relayout_weight_wino4(u->params[c.p_w].dev, dst, c.cout, c.cin, nullptr, nq);
            if (rc != DLPM_OK) return;
        }
    };
    auto seq = [&](std::vector<Layer> &s) {
        for (auto &L : s) {
            one(L.c1);
            if (L.kind == L_RES) one(L.c2);
        }
    };
    for (auto &s : u->in_blocks) seq(s);
    seq(u->mid);
    for (auto &s : u->out_blocks) seq(s);
    if (rc == DLPM_OK && hipDeviceSynchronize() != hipSuccess) { set_error("narrow_wino4_copies: synchronize failed"); rc = DLPM_ERR_HIP; }
    return rc;
}

extern "C" int dlpm_unet_finalize(dlpm_unet *u) {
    DLPM_CHECK_ARG(u, "dlpm_unet_finalize: null handle");
    for (auto &p : u->params)
        if (!p.set) {
            set_error("dlpm_unet_finalize: parameter '%s' was never set", p.key.c_str());
            return DLPM_ERR_STATE;
        }
    for_each_conv(u, free_conv);
    // per-conv weight layouts; C0 matters only for the concat inputs of the output blocks
    auto prep_seq = [&](std::vector<Layer> &s, int C0_first) -> int {
        for (size_t i = 0; i < s.size(); i++) {
            Layer &L = s[i];
            const bool cat = (i == 0 && L.kind == L_RES && C0_first > 0);
            const int C0 = cat ? C0_first : L.c1.cin;
            TRY(prep_conv(u, L.c1, L.kind == L_STEM ? L.c1.cin : C0, L.kind == L_STEM ? 1 : L.kind == L_DOWN ? 4 : 0));
            if (L.kind == L_RES || L.kind == L_ATTN) TRY(prep_conv(u, L.c2, L.c2.cin, 0));
            if (L.kind == L_RES && L.has_skip) TRY(prep_conv(u, L.skip, cat ? C0_first : L.skip.cin, 0));
        }
        return DLPM_OK;
    };
    for (auto &s : u->in_blocks) TRY(prep_seq(s, 0));
    TRY(prep_seq(u->mid, 0));
    {
        // channels of h entering each output block = previous block's cout (or the middle's)
        int ch = u->mid.back().cout;
        for (auto &s : u->out_blocks) {
            TRY(prep_seq(s, ch));
            ch = s[0].cout;
        }
    }
    TRY(prep_conv(u, u->te0, u->te0.cin, 3));   // 3: the time path stays on the fp32 pipe whatever its row count
    TRY(prep_conv(u, u->te2, u->te2.cin, 3));
    TRY(prep_conv(u, u->head, u->head.cin, 2));
    // fused emb GEMM: rows of every emb_layers.1.weight stacked in ResBlock order
    if (u->embcat_w) (void)hipFree(u->embcat_w);
    if (u->embcat_b) (void)hipFree(u->embcat_b);
    DLPM_HIP(hipMalloc(&u->embcat_w, (size_t)u->emb_total * u->ted * sizeof(float)));
    DLPM_HIP(hipMalloc(&u->embcat_b, (size_t)u->emb_total * sizeof(float)));
    auto cat_seq = [&](std::vector<Layer> &s) -> int {
        for (auto &L : s)
            if (L.kind == L_RES) {
                DLPM_HIP(hipMemcpy(u->embcat_w + (size_t)L.emb_off * u->ted, u->params[L.p_emb_w].dev,
                                   (size_t)2 * L.cout * u->ted * sizeof(float), hipMemcpyDeviceToDevice));
                DLPM_HIP(hipMemcpy(u->embcat_b + L.emb_off, u->params[L.p_emb_b].dev, (size_t)2 * L.cout * sizeof(float),
                                   hipMemcpyDeviceToDevice));
            }
        return DLPM_OK;
    };
    for (auto &s : u->in_blocks) TRY(cat_seq(s));
    TRY(cat_seq(u->mid));
    for (auto &s : u->out_blocks) TRY(cat_seq(s));
    u->embcat.cout = u->emb_total; u->embcat.cin = u->ted; u->embcat.ks = 1;
    {
        ConvLaunch probe;
        probe.C0 = u->ted; probe.Cout = u->emb_total; probe.ks = 1;
        u->embcat.use_igemm = igemm_supported(probe);
        if (u->embcat.use_igemm) {
            u->embcat.w_dev = u->embcat_w;
        } else {
            DLPM_HIP(hipMalloc(&u->embcat.w_dev, (size_t)u->emb_total * u->ted * sizeof(float)));
            u->embcat.owns = true;
            TRY(relayout_weight(u->embcat_w, u->embcat.w_dev, u->emb_total, u->ted, 1, false, nullptr));
        }
    }
    TRY(narrow_wino4_copies(u));     // (a dispatch batch declared before finalize, or weights re-uploaded under one)
    DLPM_HIP(hipDeviceSynchronize());
    u->flops = count_flops(u);
    u->finalized = true;
    u->plan_version++;
    return DLPM_OK;
}

extern "C" int dlpm_unet_set_conv_policy(dlpm_unet *u, int32_t generation, int64_t dispatch_batch) {
    DLPM_CHECK_ARG(u, "dlpm_unet_set_conv_policy: null handle");
    DLPM_CHECK_ARG(generation >= DLPM_CONV_AUTO && generation <= DLPM_CONV_IGEMM, "dlpm_unet_set_conv_policy: unknown generation %d",
                   generation);
    DLPM_CHECK_ARG(dispatch_batch >= 0, "dlpm_unet_set_conv_policy: negative dispatch batch");
    if (u->gen == generation && u->dispatch_B == dispatch_batch) return DLPM_OK;
    u->gen = generation;
    u->dispatch_B = dispatch_batch;
    u->plan_version++;
    if (u->finalized) TRY(narrow_wino4_copies(u));
    return DLPM_OK;
}

extern "C" int dlpm_unet_set_gemm_policy(dlpm_unet *u, int32_t mode) {
    DLPM_CHECK_ARG(u, "dlpm_unet_set_gemm_policy: null handle");
    DLPM_CHECK_ARG(mode >= DLPM_GEMM_AUTO && mode <= DLPM_GEMM_BF16X3, "dlpm_unet_set_gemm_policy: unknown mode %d", mode);
    if (u->gemm == mode) return DLPM_OK;
    u->gemm = mode;
    u->plan_version++;
    return DLPM_OK;
}

extern "C" int64_t dlpm_unet_time_embedding_width(const dlpm_unet *u) { return u ? u->emb_total : -1; }
extern "C" int64_t dlpm_unet_time_embeddings_scratch_bytes(const dlpm_unet *u, int64_t M) {
    return (u && M > 0) ? M * (u->cfg.model_channels + 2 * (int64_t)u->ted) * (int64_t)sizeof(float) : -1;
}

extern "C" int dlpm_unet_time_embeddings(dlpm_unet *net, const float *t_dev, int64_t M, float *out_dev, void *scratch_dev, int64_t scratch_bytes,
                                         dlpm_stream_t stream) {
    DLPM_CHECK_ARG(net && t_dev && out_dev && scratch_dev && M > 0 && M < (1 << 24), "dlpm_unet_time_embeddings: bad argument");
    if (!net->finalized) {
        set_error("dlpm_unet_time_embeddings: call dlpm_unet_finalize first");
        return DLPM_ERR_STATE;
    }
    const int64_t need = dlpm_unet_time_embeddings_scratch_bytes(net, M);
    if (scratch_bytes < need) {
        set_error("dlpm_unet_time_embeddings: scratch of %lld bytes, need %lld", (long long)scratch_bytes, (long long)need);
        return DLPM_ERR_NOMEM;
    }
    float *e0 = static_cast<float *>(scratch_dev), *e1 = e0 + M * net->cfg.model_channels, *e2 = e1 + M * net->ted;
    return time_path(net, t_dev, (int)M, e0, e1, e2, out_dev, as_stream(stream));
}

extern "C" int dlpm_unet_bind_time_table(dlpm_unet *net, const float *table_dev, const int32_t *row_index_dev) {
    DLPM_CHECK_ARG(net && ((table_dev == nullptr) == (row_index_dev == nullptr)), "dlpm_unet_bind_time_table: give both pointers or neither");
    // per HOST THREAD, not per net: two samplers that share one net from different threads each see their own binding while
    // they enqueue (round 3 kept it in the net, where one thread's table could be baked into the other's captured graph)
    // ONE binding per host thread: binding a second net replaces the first; un-binding (null pointers) clears the slot only if
    // `net` is the net it holds -- un-binding net A must not drop net B's table
    if (!table_dev && tl_time_net != net) return DLPM_OK;
    tl_time_net = table_dev ? net : nullptr;
    tl_time_table = table_dev;
    tl_time_index = row_index_dev;
    return DLPM_OK;
}

extern "C" int64_t dlpm_unet_plan_version(const dlpm_unet *u) { return u ? u->plan_version : -1; }

extern "C" int64_t dlpm_unet_workspace_bytes(const dlpm_unet *net, int64_t B) {
    if (!net || B <= 0) return -1;
    Ctx cx;
    cx.u = const_cast<dlpm_unet *>(net);
    cx.B = (int)B;
    cx.ws.dry = true;
    cx.ws.reuse = !net->keep_feats;
    cx.st = nullptr;
    std::vector<Tensor4> keep = net->feats;
    int r = walk(cx.u, cx, nullptr, nullptr, nullptr);
    cx.u->feats = keep;
    return r == DLPM_OK ? cx.ws.peak : -1;
}

static int unet_forward(dlpm_unet *net, const float *x_dev, const float *t_dev, float *eps_dev, int64_t B, void *workspace_dev,
                        int64_t workspace_bytes, dlpm_stream_t stream, bool uniform_t, const HeadUpdate *hu = nullptr);

extern "C" int dlpm_unet_forward(dlpm_unet *net, const float *x_dev, const float *t_dev, float *eps_dev, int64_t B,
                                 void *workspace_dev, int64_t workspace_bytes, dlpm_stream_t stream) {
    return unet_forward(net, x_dev, t_dev, eps_dev, B, workspace_dev, workspace_bytes, stream, false);
}

extern "C" int dlpm_unet_forward_uniform_t(dlpm_unet *net, const float *x_dev, const float *t_dev, float *eps_dev, int64_t B,
                                           void *workspace_dev, int64_t workspace_bytes, dlpm_stream_t stream) {
    return unet_forward(net, x_dev, t_dev, eps_dev, B, workspace_dev, workspace_bytes, stream, true);
}

// Whether this net's head convolution runs on the kernel that can carry the update (a property of the architecture).
static bool head_fusable(const dlpm_unet *u) {
    ConvLaunch L;
    const int H = u->cfg.image_size;
    L.C0 = u->head.cin; L.Hin = L.Hout = H; L.Win = L.Wout = H; L.ks = 3; L.Cout = u->head.cout; L.out_nchw = 1;
    L.w_small = u->head.w_small;
    L.w_taps = u->head.w_taps;
    L.w_hfused = u->head.w_hfused;
    L.coefA = L.coefB = reinterpret_cast<const float *>(u);   // (the head always carries its GroupNorm: only non-null matters here)
    L.act_silu = 1;
    return head_fused_ok(L) || head_gemm_ok(L) || head_conv_ok(L);
}

extern "C" int dlpm_unet_forward_update(dlpm_unet *net, const float *x_in_dev, const float *t_dev, const dlpm_update_args *upd,
                                        float *eps_scratch_dev, int64_t B, void *workspace_dev, int64_t workspace_bytes,
                                        dlpm_stream_t stream) {
    DLPM_CHECK_ARG(net && x_in_dev && t_dev && upd && upd->x_dev && upd->t_dev && upd->g_dev && upd->c_eps_dev && upd->c_noise_dev,
                   "dlpm_unet_forward_update: null argument");
    DLPM_CHECK_ARG(upd->B == B, "dlpm_unet_forward_update: update batch %lld != %lld", (long long)upd->B, (long long)B);
    const int64_t D = (int64_t)net->cfg.out_channels * net->cfg.image_size * net->cfg.image_size;
    DLPM_CHECK_ARG(upd->D == D, "dlpm_unet_forward_update: state of %lld elements per sample, the net emits %lld", (long long)upd->D,
                   (long long)D);
    const bool aligned = ((reinterpret_cast<uintptr_t>(upd->x_dev) | reinterpret_cast<uintptr_t>(upd->z_dev)) % 16) == 0;
    const bool fuse = net->finalized && head_fusable(net) && aligned &&
                      !(upd->flags & (DLPM_UPD_DLIM | DLPM_UPD_CLIP | DLPM_UPD_ELEMENTWISE));
    if (!fuse) {   // the variants the head's epilogue does not carry: eps through HBM, then the update kernels
        DLPM_CHECK_ARG(eps_scratch_dev, "dlpm_unet_forward_update: this update variant needs the eps scratch buffer");
        int r = unet_forward(net, x_in_dev, t_dev, eps_scratch_dev, B, workspace_dev, workspace_bytes, stream, true);
        if (r != DLPM_OK) return r;
        dlpm_update_args a = *upd;
        a.eps_dev = eps_scratch_dev;
        return dlpm_update_f32(&a, stream);
    }
    HeadUpdate hu;
    hu.x = upd->x_dev; hu.z = upd->z_dev; hu.t = upd->t_dev; hu.g = upd->g_dev; hu.c_eps = upd->c_eps_dev; hu.c_noise = upd->c_noise_dev;
    hu.key = upd->key_dev; hu.seed = upd->seed; hu.sample_offset = upd->sample_offset; hu.hist_pp = upd->hist_pp;
    hu.T = upd->T; hu.B = B;
    float dummy;   // walk() wants a non-null eps pointer; the fused head never writes it
    int r = unet_forward(net, x_in_dev, t_dev, eps_scratch_dev ? eps_scratch_dev : &dummy, B, workspace_dev, workspace_bytes, stream, true, &hu);
    if (r != DLPM_OK) return r;
    if (upd->flags & DLPM_UPD_ADVANCE) return launch_step_advance(const_cast<int32_t *>(upd->t_dev), as_stream(stream));
    return DLPM_OK;
}

static int unet_forward(dlpm_unet *net, const float *x_dev, const float *t_dev, float *eps_dev, int64_t B, void *workspace_dev,
                        int64_t workspace_bytes, dlpm_stream_t stream, bool uniform_t, const HeadUpdate *hu) {
    DLPM_CHECK_ARG(net && x_dev && t_dev && eps_dev && workspace_dev, "dlpm_unet_forward: null argument");
    DLPM_CHECK_ARG(B > 0 && B < (1 << 24), "dlpm_unet_forward: bad batch %lld", (long long)B);
    if (!net->finalized) {
        set_error("dlpm_unet_forward: call dlpm_unet_finalize first");
        return DLPM_ERR_STATE;
    }
    const int64_t need = dlpm_unet_workspace_bytes(net, B);
    if (need < 0) return DLPM_ERR_ARG;
    if (workspace_bytes < need) {
        set_error("dlpm_unet_forward: workspace of %lld bytes, need %lld", (long long)workspace_bytes, (long long)need);
        return DLPM_ERR_NOMEM;
    }
    Ctx cx;
    cx.u = net;
    cx.B = (int)B;
    cx.ws.base = static_cast<char *>(workspace_dev);
    cx.ws.cap = workspace_bytes;
    cx.ws.reuse = !net->keep_feats;
    cx.st = as_stream(stream);
    cx.uniform_t = uniform_t;
    cx.hu = hu;
    return walk(net, cx, x_dev, t_dev, eps_dev);
}

extern "C" int dlpm_unet_keep_features(dlpm_unet *net, int on) {
    DLPM_CHECK_ARG(net, "dlpm_unet_keep_features: null handle");
    net->keep_feats = on != 0;
    net->plan_version++;       // the workspace size changes with it
    return DLPM_OK;
}

extern "C" int dlpm_unet_num_features(const dlpm_unet *net) {
    return net ? (int)(net->in_blocks.size() + 1 + net->out_blocks.size()) : 0;
}

extern "C" int dlpm_unet_feature_shape(const dlpm_unet *net, int i, int32_t *C, int32_t *H, int32_t *W) {
    DLPM_CHECK_ARG(net && i >= 0 && i < (int)net->feats.size(), "dlpm_unet_feature_shape: no such feature (run a forward first)");
    if (C) *C = net->feats[i].C;
    if (H) *H = net->feats[i].H;
    if (W) *W = net->feats[i].W;
    return DLPM_OK;
}

extern "C" int dlpm_unet_get_feature(dlpm_unet *net, int i, float *out_nchw_dev, int64_t B, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(net && out_nchw_dev && i >= 0 && i < (int)net->feats.size() && net->feats[i].p,
                   "dlpm_unet_get_feature: no such feature (call dlpm_unet_keep_features(net, 1), then run a forward)");
    const Tensor4 &f = net->feats[i];
    return dlpm_nhwc_to_nchw_f32(f.p, out_nchw_dev, (int32_t)B, f.C, f.H, f.W, stream);
}

extern "C" int64_t dlpm_unet_flops_per_sample(const dlpm_unet *net) { return net ? net->flops : 0; }

extern "C" void dlpm_unet_destroy(dlpm_unet *u) {
    if (!u) return;
    for_each_conv(u, free_conv);
    if (u->embcat.owns && u->embcat.w_dev) (void)hipFree(u->embcat.w_dev);
    if (u->embcat_w) (void)hipFree(u->embcat_w);
    if (u->embcat_b) (void)hipFree(u->embcat_b);
    for (auto &p : u->params)
        if (p.dev) (void)hipFree(p.dev);
    delete u;
}

// ---------------------------------------------------------------------------------------------
// building-block entry points (parity tests call each kernel through the C ABI)
// ---------------------------------------------------------------------------------------------
extern "C" int dlpm_conv2d_f32(const dlpm_conv_args *a, float *scratch_dev, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(a && a->src0 && a->weight && a->out && scratch_dev, "dlpm_conv2d_f32: null argument");
    DLPM_CHECK_ARG(a->ksize == 1 || a->ksize == 3, "dlpm_conv2d_f32: ksize must be 1 or 3");
    DLPM_CHECK_ARG(a->stride == 1 || a->stride == 2, "dlpm_conv2d_f32: stride must be 1 or 2");
    DLPM_CHECK_ARG((a->C1 == 0) == (a->src1 == nullptr), "dlpm_conv2d_f32: src1/C1 mismatch");
    ConvLaunch L;
    L.src0 = a->src0; L.src1 = a->src1; L.C0 = a->C0; L.C1 = a->C1; L.B = a->B;
    L.Hin = a->Hin; L.Win = a->Win; L.Hout = a->Hout; L.Wout = a->Wout;
    L.ks = a->ksize; L.stride = a->stride; L.ups = a->upsample;
    L.bias = a->bias; L.coefA = a->coefA; L.coefB = a->coefB; L.act_silu = a->act_silu;
    L.res0 = a->res0; L.res1 = a->res1; L.R0 = a->R0; L.out = a->out; L.Cout = a->Cout;
    L.in_nchw = a->in_nchw; L.out_nchw = a->out_nchw;
    const bool ig = !(a->force_direct & 1) && igemm_supported(L);
    if ((a->in_nchw || a->out_nchw) && a->C1 != 0) {
        set_error("dlpm_conv2d_f32: NCHW boundary layouts do not combine with a concat input");
        return DLPM_ERR_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    TRY(relayout_weight(a->weight, scratch_dev, a->Cout, a->C0 + a->C1, a->ksize, ig, st));
    L.w = scratch_dev;
    if (ig && a->ksize == 1 && (a->C0 + a->C1) % 32 == 0 && (a->force_direct & 4) &&
        a->scratch_floats >= (int64_t)a->Cout * (a->C0 + a->C1) + frag_weight_floats(a->Cout, a->C0 + a->C1, 1)) {
        float *wf = scratch_dev + (int64_t)a->Cout * (a->C0 + a->C1);
        TRY(relayout_weight_frag(a->weight, wf, a->Cout, a->C0 + a->C1, st, 1));
        L.w_frag = wf;
        L.ws_gemm = 1;
    }
    L.gemm = DLPM_GEMM_F32;
    const int taps = a->ksize * a->ksize;
    if (ig && (a->force_direct & 16) && a->Cout % 128 == 0 && (a->C0 + a->C1) % 32 == 0 &&
        a->scratch_floats >= (int64_t)a->Cout * (a->C0 + a->C1) * taps + split_weight_floats(a->Cout, a->C0 + a->C1, taps)) {
        float *wsp = scratch_dev + (int64_t)a->Cout * (a->C0 + a->C1) * taps;
        TRY(relayout_weight_split(a->weight, wsp, a->Cout, a->C0 + a->C1, taps, st));
        L.w_split = wsp;
        L.gemm = DLPM_GEMM_BF16X3;
    }
    if (ig && a->ksize == 3 && !L.w_split && (a->C0 + a->C1) % 32 == 0 &&   // (the split copy sits where these would go)
        a->scratch_floats >= (int64_t)a->Cout * (a->C0 + a->C1) * 9 + frag_weight_floats(a->Cout, a->C0 + a->C1)) {
        float *wf = scratch_dev + (int64_t)a->Cout * (a->C0 + a->C1) * 9;
        TRY(relayout_weight_frag(a->weight, wf, a->Cout, a->C0 + a->C1, st));
        L.w_frag = wf;
        int64_t used = (int64_t)a->Cout * (a->C0 + a->C1) * 9 + frag_weight_floats(a->Cout, a->C0 + a->C1);
        const bool wino_shape = !(a->force_direct & 2) && a->stride == 1 && !a->in_nchw && !a->out_nchw;
        if (wino_shape && a->Cout % 64 == 0 && a->scratch_floats >= used + wino_weight_floats(a->Cout, a->C0 + a->C1)) {
            float *ww = scratch_dev + used;
            TRY(relayout_weight_wino(a->weight, ww, a->Cout, a->C0 + a->C1, st));
            L.w_wino = ww;
            used += wino_weight_floats(a->Cout, a->C0 + a->C1);
        }
        // bit 8: the F(4x4,3x3) kernel where the shape qualifies (128-, 64- or 32-channel n-tiles; its weights go behind the F(2x2) copy)
        if (wino_shape && (a->force_direct & 8) && a->Cout % 32 == 0 && (a->C0 + a->C1) % 8 == 0 &&
            a->scratch_floats >= used + wino4_weight_floats(a->Cout, a->C0 + a->C1)) {
            float *w4 = scratch_dev + used;
            TRY(relayout_weight_wino4(a->weight, w4, a->Cout, a->C0 + a->C1, st));
            L.w_wino4 = w4;
            // bit 8 ASKS for the F(4x4) kernel: a geometry it rejects (e.g. 16 one-tile 4x4 images on a 64- / 32-channel n-tile: 576 halo
            // pixels > F4_RAWPIX_N) is an error, not a silent drop to the implicit GEMM -- tests that compare "the F(4x4) kernel" with the
            // default path must not pass for the wrong reason
            int bh_, bw_, ni_;
            if (!wino4_geometry(L, &bh_, &bw_, &ni_)) {
                set_error("dlpm_conv2d_f32: force_direct bit 8 asks for the F(4x4,3x3) kernel, which does not take this geometry");
                return DLPM_ERR_UNSUPPORTED;
            }
            L.w_wino = nullptr;   // no F(2x2) alternative: the launch takes the F(4x4) kernel
        }
    }
    if ((a->force_direct & 64) && a->ksize == 3 && a->Cout <= 3 && a->C0 % 32 == 0 && a->C1 == 0) {
        // bit 64: the one-pass head kernel (head_fused.hip; in the UNet it also carries the sampler's update): W' fragments in scratch
        L.w_hfused = scratch_dev;
        L.gemm = (a->force_direct & 128) ? DLPM_GEMM_F32 : DLPM_GEMM_AUTO;   // bit 128: its fp32-MFMA form instead of the bf16 x 3 one
        if (a->scratch_floats >= head_fused_weight_floats(a->C0) && head_fused_ok(L)) {
            TRY(relayout_weight_head_fused(a->weight, scratch_dev, a->Cout, a->C0, st));
            return launch_conv_head_fused(L, nullptr, st);
        }
        set_error("dlpm_conv2d_f32: the one-pass head kernel does not take this shape / scratch size");
        return DLPM_ERR_UNSUPPORTED;
    }
    if ((a->force_direct & 32) && a->ksize == 3 && a->Cout <= 3 && a->C0 % 32 == 0 && a->C1 == 0) {
        // bit 32: the head as a 1x1 GEMM onto 9 Cout tap channels + gather: W' at the front of the scratch buffer, P behind it
        const int64_t wsz = (int64_t)head_taps_rows(a->Cout) * a->C0;
        L.w_taps = scratch_dev;   // (head_gemm_ok only looks at the pointer)
        if (a->scratch_floats >= wsz + head_gemm_scratch_floats(L) && head_gemm_ok(L)) {
            TRY(relayout_weight_head_taps(a->weight, scratch_dev, a->Cout, a->C0, st));
            return launch_conv_head_gemm(L, nullptr, scratch_dev + wsz, st);
        }
        set_error("dlpm_conv2d_f32: the GEMM + gather head does not take this shape / scratch size");
        return DLPM_ERR_UNSUPPORTED;
    }
    if (!(a->force_direct & 7) && a->ksize == 3 && a->Cout <= 4 && a->C0 % 32 == 0 && a->C1 == 0) {
        // [tap][cin][4] copy at the END of the scratch buffer (the front holds the layouts built above)
        const int64_t front = (int64_t)a->Cout * a->C0 * 9 + frag_weight_floats(a->Cout, a->C0);
        const int64_t wsz = (int64_t)9 * a->C0 * 4;
        if (a->scratch_floats >= front + wsz) {
            float *ws = scratch_dev + (a->scratch_floats - wsz);
            TRY(relayout_weight_head(a->weight, ws, a->Cout, a->C0, st));
            L.w_small = ws;
            if (head_conv_ok(L)) return launch_conv_head(L, nullptr, st);
        }
    }
    if (ig) return launch_conv_igemm(L, st);
    return (a->force_direct & 1) ? launch_conv_direct(L, st) : launch_conv_fallback(L, st);
}

extern "C" int dlpm_groupnorm_coeffs_f32(const float *src0, const float *src1, int32_t C0, int32_t C1, int32_t B, int32_t HW,
                                         int32_t groups, const float *gamma, const float *beta, const float *ss,
                                         int64_t ss_stride, int64_t ss_offset, float *coefA, float *coefB,
                                         dlpm_stream_t stream) {
    DLPM_CHECK_ARG(src0 && gamma && beta && coefA && coefB, "dlpm_groupnorm_coeffs_f32: null argument");
    DLPM_CHECK_ARG(groups > 0 && (C0 + C1) % groups == 0, "dlpm_groupnorm_coeffs_f32: %d channels not divisible by %d groups",
                   C0 + C1, groups);
    return launch_gn_coeffs(src0, src1, C0, C1, B, HW, groups, gamma, beta, ss, ss_stride, ss_offset, coefA, coefB,
                            as_stream(stream));
}

extern "C" int dlpm_attention_f32(const float *qkv, float *out, int32_t B, int32_t T, int32_t C, int32_t heads,
                                  dlpm_stream_t stream) {
    DLPM_CHECK_ARG(qkv && out && B > 0, "dlpm_attention_f32: null argument");
    return launch_attention(qkv, out, B, T, C, heads, as_stream(stream));
}

extern "C" int dlpm_resblock_small_f32(const dlpm_resblock_args *a, float *scratch_dev, int64_t scratch_floats, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(a && a->x0 && a->conv1_w && a->conv2_w && a->ss && a->out && scratch_dev, "dlpm_resblock_small_f32: null argument");
    // every parameter the kernel reads at entry (it has no optional ones but the skip convolution and the statistics)
    DLPM_CHECK_ARG(a->gn1_w && a->gn1_b && a->gn2_w && a->gn2_b && a->conv1_b && a->conv2_b, "dlpm_resblock_small_f32: null GroupNorm / bias parameter");
    DLPM_CHECK_ARG(!a->skip_w || a->skip_b, "dlpm_resblock_small_f32: skip_w without skip_b");
    DLPM_CHECK_ARG(a->B > 0 && a->H == a->W && (a->H == 8 || a->H == 4), "dlpm_resblock_small_f32: B %d, %d x %d images (8x8 or 4x4)", a->B, a->H, a->W);
    DLPM_CHECK_ARG((a->C1 == 0) == (a->x1 == nullptr), "dlpm_resblock_small_f32: x1/C1 mismatch");
    const int Cin = a->C0 + a->C1;
    const int64_t n1 = (int64_t)64 * Cin * 9, n2 = (int64_t)64 * 64 * 9, ns = a->skip_w ? (int64_t)64 * Cin : 0;
    DLPM_CHECK_ARG(scratch_floats >= n1 + n2 + ns, "dlpm_resblock_small_f32: scratch of %lld floats, need %lld", (long long)scratch_floats,
                   (long long)(n1 + n2 + ns));
    DLPM_CHECK_ARG(small_weight_ok(64, Cin, 3), "dlpm_resblock_small_f32: %d input channels (64 or 128)", Cin);
    hipStream_t st = as_stream(stream);
    TRY(relayout_weight_small(a->conv1_w, scratch_dev, 64, Cin, 3, st));
    TRY(relayout_weight_small(a->conv2_w, scratch_dev + n1, 64, 64, 3, st));
    if (a->skip_w) TRY(relayout_weight_small(a->skip_w, scratch_dev + n1 + n2, 64, Cin, 1, st));
    ResSmallLaunch r;
    r.x0 = a->x0; r.x1 = a->x1; r.C0 = a->C0; r.C1 = a->C1; r.B = a->B; r.H = a->H; r.W = a->W;
    r.gn1_w = a->gn1_w; r.gn1_b = a->gn1_b; r.gn2_w = a->gn2_w; r.gn2_b = a->gn2_b;
    r.w1f = scratch_dev; r.b1 = a->conv1_b; r.w2f = scratch_dev + n1; r.b2 = a->conv2_b;
    r.wsf = a->skip_w ? scratch_dev + n1 + n2 : nullptr; r.bs = a->skip_b;
    r.emb = a->ss; r.emb_stride = a->ss_stride; r.emb_off = 0;
    r.out = a->out; r.stats_out = reinterpret_cast<float2 *>(a->stats_out);
    return launch_resblock_small(r, st);
}

extern "C" int64_t dlpm_resblock_img_scratch_floats(int64_t B, int32_t Cin) {
    if (B <= 0 || Cin <= 0) return -1;
    // (sized for either shape: 32 channels on 32x32 images, 64 channels on 16x16 ones)
    return wino4_weight_floats(64, Cin) + wino4_weight_floats(64, 64) + 2 * B * Cin + 2 * B * 1024 * 32 + 64;
}

extern "C" int dlpm_resblock_img_f32(const dlpm_resblock_args *a, float *scratch_dev, int64_t scratch_floats, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(a && a->x0 && a->conv1_w && a->conv2_w && a->ss && a->out && scratch_dev, "dlpm_resblock_img_f32: null argument");
    DLPM_CHECK_ARG(a->gn1_w && a->gn1_b && a->gn2_w && a->gn2_b && a->conv1_b && a->conv2_b, "dlpm_resblock_img_f32: null GroupNorm / bias parameter");
    DLPM_CHECK_ARG(a->B > 0 && a->H == a->W && (a->H == 32 || a->H == 16), "dlpm_resblock_img_f32: B %d, %d x %d images (32x32 or 16x16)", a->B, a->H, a->W);
    const int CO = a->H == 32 ? 32 : 64, HW = a->H * a->W;
    DLPM_CHECK_ARG((a->C1 == 0) == (a->x1 == nullptr), "dlpm_resblock_img_f32: x1/C1 mismatch");
    const int Cin = a->C0 + a->C1;
    DLPM_CHECK_ARG(Cin >= 32 && Cin <= 128 && Cin % 8 == 0 && a->C0 % 8 == 0, "dlpm_resblock_img_f32: %d + %d input channels", a->C0, a->C1);
    DLPM_CHECK_ARG((Cin == CO) == (a->skip_w == nullptr) && (!a->skip_w || a->skip_b), "dlpm_resblock_img_f32: skip_w / skip_b (required unless C0 + C1 = the output channels)");
    DLPM_CHECK_ARG(scratch_floats >= dlpm_resblock_img_scratch_floats(a->B, Cin), "dlpm_resblock_img_f32: scratch of %lld floats, need %lld",
                   (long long)scratch_floats, (long long)dlpm_resblock_img_scratch_floats(a->B, Cin));
    hipStream_t st = as_stream(stream);
    float *w1 = scratch_dev, *w2 = w1 + wino4_weight_floats(CO, Cin), *cA = w2 + wino4_weight_floats(CO, CO), *cB = cA + (int64_t)a->B * Cin;
    float *hb = cB + (int64_t)a->B * Cin, *sk = hb + (int64_t)a->B * HW * CO;
    hb += (64 - ((hb - scratch_dev) & 63)) & 63;          // (16-byte rows: float4 staging loads)
    sk = hb + (int64_t)a->B * HW * CO;
    TRY(relayout_weight_wino4(a->conv1_w, w1, CO, Cin, st));
    TRY(relayout_weight_wino4(a->conv2_w, w2, CO, CO, st));
    TRY(launch_gn_coeffs(a->x0, a->x1, a->C0, a->C1, a->B, HW, 32, a->gn1_w, a->gn1_b, nullptr, 0, 0, cA, cB, st));
    ResImgLaunch r;
    r.x0 = a->x0; r.x1 = a->x1; r.C0 = a->C0; r.C1 = a->C1; r.B = a->B; r.H = a->H;
    r.coefA1 = cA; r.coefB1 = cB;
    r.gn1_w = a->gn1_w; r.gn1_b = a->gn1_b; r.gn2_w = a->gn2_w; r.gn2_b = a->gn2_b;
    r.w1 = w1; r.b1 = a->conv1_b; r.w2 = w2; r.b2 = a->conv2_b;
    r.emb = a->ss; r.emb_stride = a->ss_stride; r.emb_off = 0;
    r.hbuf = hb; r.out = a->out; r.stats_out = reinterpret_cast<float2 *>(a->stats_out);
    if (a->skip_w) {
        ConvLaunch s;
        s.src0 = a->x0; s.src1 = a->x1; s.C0 = a->C0; s.C1 = a->C1; s.B = a->B; s.Hin = s.Hout = a->H; s.Win = s.Wout = a->W;
        s.ks = 1; s.Cout = CO; s.w = a->skip_w; s.bias = a->skip_b; s.out = sk; s.gemm = DLPM_GEMM_F32;
        TRY(launch_conv_igemm(s, st));
        r.res = sk;
    } else {
        r.res = a->x0;
    }
    if (!res_img_ok(r)) {
        set_error("dlpm_resblock_img_f32: the whole-image ResBlock kernel is switched off (DLPM_RES_IMG=0 / DLPM_WINO_F4=0)");
        return DLPM_ERR_UNSUPPORTED;
    }
    return launch_resblock_img(r, st);
}

extern "C" int dlpm_attnblock_small_f32(const dlpm_attnblock_args *a, float *scratch_dev, int64_t scratch_floats, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(a && a->x && a->qkv_w && a->proj_w && a->out && scratch_dev, "dlpm_attnblock_small_f32: null argument");
    DLPM_CHECK_ARG(a->gn_w && a->gn_b && a->qkv_b && a->proj_b, "dlpm_attnblock_small_f32: null GroupNorm / bias parameter");
    DLPM_CHECK_ARG(a->B > 0 && a->H == a->W && (a->H == 16 || a->H == 8 || a->H == 4), "dlpm_attnblock_small_f32: B %d, %d x %d images (16x16, 8x8 or 4x4)", a->B, a->H, a->W);
    DLPM_CHECK_ARG(a->C == 64 && a->heads == 4, "dlpm_attnblock_small_f32: 64 channels, 4 heads (got %d, %d)", a->C, a->heads);
    const int64_t nq = (int64_t)192 * 64, np = (int64_t)64 * 64;
    DLPM_CHECK_ARG(scratch_floats >= nq + np, "dlpm_attnblock_small_f32: scratch of %lld floats, need %lld", (long long)scratch_floats,
                   (long long)(nq + np));
    hipStream_t st = as_stream(stream);
    TRY(relayout_weight_small(a->qkv_w, scratch_dev, 192, 64, 1, st));
    TRY(relayout_weight_small(a->proj_w, scratch_dev + nq, 64, 64, 1, st));
    AttnSmallLaunch l;
    l.x = a->x; l.C = a->C; l.heads = a->heads; l.B = a->B; l.H = a->H; l.W = a->W;
    l.gn_w = a->gn_w; l.gn_b = a->gn_b; l.wqkv = scratch_dev; l.bqkv = a->qkv_b; l.wproj = scratch_dev + nq; l.bproj = a->proj_b;
    l.out = a->out; l.stats_out = reinterpret_cast<float2 *>(a->stats_out);
    return a->H == 16 ? launch_attnblock16(l, st) : launch_attnblock_small(l, st);
}

extern "C" int dlpm_timestep_embedding_f32(const float *t, float *emb, int64_t B, int32_t dim, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(t && emb && B > 0 && dim > 0, "dlpm_timestep_embedding_f32: bad argument");
    return launch_timestep_embedding(t, emb, B, dim, as_stream(stream));
}
