// sampler.hip -- the reverse-time loop x_T -> x_0 as a replayed hipGraph.
//
// Replaces p_sample_loop_progressive / ddim_sample_loop_progressive
// (dlpm/methods/GenerativeLevyProcess.py:291-330, 418-452).  One reverse step is
//     tvec <- t/T ; eps <- model(x, tvec) ; x <- update(x, eps, tables[t]) ; t <- t - 1
// Every per-step scalar is read from device tables indexed by a device-resident step counter, so
// the captured graph of ONE step is replayed T-1 times with no host patching in between.
#include <vector>

#include "common.h"

using namespace dlpm;

extern "C" int dlpm_mlp_forward(dlpm_mlp *, const float *, const float *, float *, int64_t, dlpm_stream_t);

struct dlpm_sampler {
    dlpm_sampler_config cfg;
    int64_t D = 0;
    float *g = nullptr, *bg = nullptr, *s = nullptr, *bs = nullptr;  // device schedule [T]
    float bs_last = 0.f;
    float *A = nullptr, *c_eps = nullptr, *c_noise = nullptr;         // [T,B]; [T,B,D] when non-isotropic
    bool elem = false;             // DLPM_UPD_ELEMENTWISE: per-element tables
    int64_t cols = 0;              // table columns: B, or B*D when non-isotropic
    bool lim = false;              // DLPM_SMP_LIM: continuous-time LIM updates (T = steps + 1)
    float *lim_ts = nullptr, *lim_tmp = nullptr, *lim_cx = nullptr, *lim_cs = nullptr, *lim_cn = nullptr;
    float *in_scale = nullptr, *xin = nullptr;   // input_scaling table [T] and the scaled copy the net reads
    float **hist_cell = nullptr;   // device cell with the history base (see dlpm_update_args::hist_pp)
    float *hist = nullptr;         // its current value (caller-owned [T,B,D] buffer or null)
    float *x = nullptr, *eps = nullptr, *tvec = nullptr;
    float *emb_tab = nullptr;      // UNet, DLPM loop: the time path's output for every step, [T][emb width] (dlpm_unet_time_embeddings)
    int32_t *t_dev = nullptr;
    uint64_t *key_dev = nullptr;   // {seed, sample_offset}: read by the update kernel, so reseeding keeps the graph
    int graph_steps = 0;           // steps inside the captured graph
    void *ws = nullptr;
    int64_t ws_bytes = 0;
    int32_t t_host = 0;
    int64_t plan_version = 0;      // dlpm_unet_plan_version at capture / workspace sizing time
    bool key_dirty = false;        // dlpm_sampler_reseed happened: key_dev is rewritten by the next begin*()
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // Graph capture is illegal on the legacy default stream (which is what torch hands out by
    // default), so graph work runs on a private non-blocking stream fenced to the caller's stream
    // with events at both ends of every dlpm_sampler_steps call.
    hipStream_t own = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
};

namespace {

__global__ void k_set_t(int32_t *t, int32_t v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *t = v;
}
__global__ void k_set_key(uint64_t *key, uint64_t seed, uint64_t offset) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { key[0] = seed; key[1] = offset; }
}
__global__ void k_set_ptr(float **cell, float *v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *cell = v;
}

#define TRY(expr)                     \
    do {                              \
        int _r = (expr);              \
        if (_r != DLPM_OK) return _r; \
    } while (0)

// The time path (time embedding -> MLP -> per-ResBlock emb linears) is a function of the step index alone: one row per step,
// computed with the kernels the forward would run (t = float(i) * (1 / T), k_fill_t's arithmetic), replaces four launches of
// every step.  DLPM loop only (the LIM loop feeds its own time grid).  DLPM_NO_TIME_TABLE=1: off.  The rows depend on the net's
// WEIGHTS and on its GEMM policy, so sync_plan() rebuilds them whenever the net's plan version moved under a live sampler
// (dlpm_unet_set_param + dlpm_unet_finalize, dlpm_unet_set_gemm_policy): round 3 built them once at create, and a sampler that
// outlived a weight upload or a pipe switch kept conditioning on the old rows.
int build_time_table(dlpm_sampler *s) {
    if (!s->cfg.unet || s->lim) return DLPM_OK;
    const char *nt = getenv("DLPM_NO_TIME_TABLE");
    const int64_t ew = dlpm_unet_time_embedding_width(s->cfg.unet);
    if ((nt && nt[0] == '1') || ew <= 0) return DLPM_OK;
    const int T = s->cfg.T;
    std::vector<float> tv(T);
    for (int i = 0; i < T; i++) tv[i] = (float)i * (1.0f / (float)T);
    float *tv_dev = nullptr;
    void *scr = nullptr;
    const int64_t scr_bytes = dlpm_unet_time_embeddings_scratch_bytes(s->cfg.unet, T);
    int r = DLPM_OK;
    hipError_t e = hipMalloc(&tv_dev, T * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(tv_dev, tv.data(), T * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&scr, (size_t)scr_bytes);
    if (e == hipSuccess && !s->emb_tab) e = hipMalloc(&s->emb_tab, (size_t)T * ew * sizeof(float));
    if (e == hipSuccess) {
        r = dlpm_unet_time_embeddings(s->cfg.unet, tv_dev, T, s->emb_tab, scr, scr_bytes, nullptr);
        e = hipDeviceSynchronize();
    }
    if (tv_dev) (void)hipFree(tv_dev);
    if (scr) (void)hipFree(scr);
    if (e != hipSuccess) {
        set_error("dlpm_sampler: time table: %s", hipGetErrorString(e));
        return DLPM_ERR_HIP;
    }
    return r;
}

// The net's launch plan may change under a live sampler (dlpm_unet_set_conv_policy / set_gemm_policy, re-uploaded weights):
// drop the captured graph, re-size the activation workspace and rebuild the time table before the next step.
int sync_plan(dlpm_sampler *s) {
    if (!s->cfg.unet) return DLPM_OK;
    const int64_t v = dlpm_unet_plan_version(s->cfg.unet);
    if (v == s->plan_version) return DLPM_OK;
    DLPM_HIP(hipDeviceSynchronize());
    if (s->exec) (void)hipGraphExecDestroy(s->exec);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    s->exec = nullptr;
    s->graph = nullptr;
    const int64_t need = dlpm_unet_workspace_bytes(s->cfg.unet, s->cfg.B);
    if (need < 0) return DLPM_ERR_STATE;
    if (need > s->ws_bytes) {
        if (s->ws) DLPM_HIP(hipFree(s->ws));
        s->ws = nullptr;
        DLPM_HIP(hipMalloc(&s->ws, (size_t)need));
        s->ws_bytes = need;
    }
    if (s->emb_tab) TRY(build_time_table(s));
    s->plan_version = v;
    return DLPM_OK;
}

// RAII: the net reads its time path from the sampler's table while one of the sampler's forward calls is being enqueued
struct TimeTableBinding {
    dlpm_unet *u = nullptr;
    explicit TimeTableBinding(const dlpm_sampler *s) {
        if (s->cfg.unet && s->emb_tab) {
            u = s->cfg.unet;
            (void)dlpm_unet_bind_time_table(u, s->emb_tab, s->t_dev);
        }
    }
    ~TimeTableBinding() {
        if (u) (void)dlpm_unet_bind_time_table(u, nullptr, nullptr);
    }
};

int model_forward(dlpm_sampler *s, hipStream_t st) {
    TimeTableBinding bound(s);
    const float *xin = s->x;
    if (s->in_scale) {
        TRY(dlpm_scale_by_table_f32(s->x, s->xin, s->cfg.B * s->D, s->t_dev, s->in_scale, st));
        xin = s->xin;
    }
    if (s->cfg.unet)
        return dlpm_unet_forward_uniform_t(s->cfg.unet, xin, s->tvec, s->eps, s->cfg.B, s->ws, s->ws_bytes, st);   // t = [i] * B
    return dlpm_mlp_forward(s->cfg.mlp, xin, s->tvec, s->eps, s->cfg.B, st);
}

int one_step(dlpm_sampler *s, const float *z, bool advance, hipStream_t st) {
    if (s->lim) {
        TRY(dlpm_fill_table_t_f32(s->tvec, s->t_dev, s->lim_ts, s->cfg.T, s->cfg.B, st));
        TRY(model_forward(s, st));
        dlpm_lim_update_args a{};
        a.x_dev = s->x; a.eps_dev = s->eps; a.z_dev = z; a.t_dev = s->t_dev;
        a.tmp_dev = s->lim_tmp; a.cx_dev = s->lim_cx; a.cs_dev = s->lim_cs; a.cn_dev = s->lim_cn;
        a.A_dev = s->cfg.alpha == 2.0 ? nullptr : s->A;      // alpha = 2: e_B = randn_like(x) (sampler.py:135)
        a.B = s->cfg.B; a.D = s->D; a.T = s->cfg.T;
        a.flags = (s->cfg.flags & DLPM_UPD_DLIM) | (advance ? DLPM_UPD_ADVANCE : 0) | (s->hist ? DLPM_UPD_HIST_ON : 0);
        a.clamp_eps = (float)s->cfg.clamp_eps;
        a.seed = s->cfg.seed; a.sample_offset = s->cfg.sample_offset; a.key_dev = s->key_dev;
        a.hist_pp = s->hist_cell;
        return dlpm_lim_update_f32(&a, st);
    }
    // (with the time table bound the UNet reads its row by the step counter and never looks at tvec: one launch less per step)
    if (!(s->cfg.unet && s->emb_tab)) TRY(dlpm_fill_scaled_t_f32(s->tvec, s->t_dev, s->cfg.T, s->cfg.B, st));
    dlpm_update_args a{};
    a.x_dev = s->x; a.eps_dev = s->eps; a.z_dev = z; a.t_dev = s->t_dev;
    a.g_dev = s->g; a.bg_dev = s->bg; a.bs_dev = s->bs;
    a.c_eps_dev = s->c_eps; a.c_noise_dev = s->c_noise; a.A_dev = s->A;
    a.B = s->cfg.B; a.D = s->D; a.T = s->cfg.T;
    a.flags = (s->cfg.flags & (DLPM_UPD_DLIM | DLPM_UPD_CLIP | DLPM_UPD_ELEMENTWISE)) | (advance ? DLPM_UPD_ADVANCE : 0) |
              (s->hist ? DLPM_UPD_HIST_ON : 0);
    a.hist_pp = s->hist_cell;
    a.dlim_eta = s->cfg.dlim_eta; a.alpha = (float)s->cfg.alpha;
    a.seed = s->cfg.seed; a.sample_offset = s->cfg.sample_offset; a.key_dev = s->key_dev;
    if (s->cfg.mean_type != DLPM_MEAN_EPSILON) {
        // the net predicts x_0 / z_t / the anterior mean: model output -> x_0 -> (clamp) -> eps in one pass over the eps buffer
        // (p_mean_variance's else-branch, GenerativeLevyProcess.py:185-207), then the plain step
        TRY(model_forward(s, st));
        dlpm_predict_args q{};
        q.x_dev = s->x; q.in_dev = s->eps; q.out_dev = s->eps; q.t_dev = s->t_dev;
        q.g_dev = s->g; q.bg_dev = s->bg; q.bs_dev = s->bs; q.c_eps_dev = s->c_eps; q.A_dev = s->A;
        q.B = s->cfg.B; q.D = s->D; q.T = s->cfg.T; q.mean_type = s->cfg.mean_type;
        q.flags = DLPM_PRED_TO_XSTART | DLPM_PRED_TO_EPS | ((s->cfg.flags & DLPM_UPD_CLIP) ? DLPM_PRED_CLIP : 0) |
                  (s->elem ? DLPM_PRED_ELEMENTWISE : 0);
        TRY(dlpm_predict_f32(&q, st));
        a.flags &= ~DLPM_UPD_CLIP;
        return dlpm_update_f32(&a, st);
    }
    if (s->cfg.unet) {   // the UNet's head convolution applies the update itself where the variant allows (eps stays on chip)
        const float *xin = s->x;
        if (s->in_scale) {
            TRY(dlpm_scale_by_table_f32(s->x, s->xin, s->cfg.B * s->D, s->t_dev, s->in_scale, st));
            xin = s->xin;
        }
        TimeTableBinding bound(s);
        return dlpm_unet_forward_update(s->cfg.unet, xin, s->tvec, &a, s->eps, s->cfg.B, s->ws, s->ws_bytes, st);
    }
    TRY(model_forward(s, st));
    return dlpm_update_f32(&a, st);
}

int flush_key(dlpm_sampler *s, hipStream_t st) {
    if (!s->key_dirty) return DLPM_OK;
    k_set_key<<<1, 64, 0, st>>>(s->key_dev, s->cfg.seed, (uint64_t)s->cfg.sample_offset);
    DLPM_LAUNCH_CHECK();
    s->key_dirty = false;
    return DLPM_OK;
}

int build_tables(dlpm_sampler *s, hipStream_t st) {
    TRY(flush_key(s, st));
    if (!s->lim)
        TRY(dlpm_coeff_tables_f32(s->A, s->g, s->s, s->bs, s->cfg.T, s->cols, s->c_eps, s->c_noise, nullptr, st));
    if (s->hist)   // row 0 of the history is x_T (GenerativeLevyProcess.py:314)
        DLPM_HIP(hipMemcpyAsync(s->hist, s->x, (size_t)s->cfg.B * s->D * sizeof(float), hipMemcpyDeviceToDevice, st));
    k_set_t<<<1, 64, 0, st>>>(s->t_dev, s->cfg.T - 1);
    DLPM_LAUNCH_CHECK();
    s->t_host = s->cfg.T - 1;
    return DLPM_OK;
}

}  // namespace

extern "C" int dlpm_sampler_create(const dlpm_sampler_config *cfg, dlpm_sampler **out) {
    DLPM_CHECK_ARG(cfg && out, "dlpm_sampler_create: null argument");
    DLPM_CHECK_ARG((cfg->unet != nullptr) != (cfg->mlp != nullptr), "dlpm_sampler_create: exactly one of unet / mlp");
    DLPM_CHECK_ARG(cfg->B > 0 && cfg->C > 0 && cfg->H > 0 && cfg->W > 0, "dlpm_sampler_create: bad shape");
    DLPM_CHECK_ARG(cfg->T >= 2, "dlpm_sampler_create: reverse_steps must be >= 2, got %d", cfg->T);
    DLPM_CHECK_ARG(cfg->alpha > 0.0 && cfg->alpha <= 2.0, "Wrong value of alpha (%g) for skewed levy r.v generation", cfg->alpha);
    const bool have = cfg->g && cfg->bg && cfg->s && cfg->bs;
    DLPM_CHECK_ARG(have || (!cfg->g && !cfg->bg && !cfg->s && !cfg->bs), "dlpm_sampler_create: give all four schedule arrays or none");
    DLPM_CHECK_ARG(!(cfg->in_scale && (cfg->flags & DLPM_SMP_LIM)), "dlpm_sampler_create: input scaling belongs to the DLPM loop");
    const bool is_lim = (cfg->flags & DLPM_SMP_LIM) != 0;
    const bool have_lim = cfg->lim_ts && cfg->lim_tmp && cfg->lim_cx && cfg->lim_cs && cfg->lim_cn;
    DLPM_CHECK_ARG(!is_lim || have_lim || (!cfg->lim_ts && !cfg->lim_tmp && !cfg->lim_cx && !cfg->lim_cs && !cfg->lim_cn),
                   "dlpm_sampler_create: give all five LIM tables or none");
    DLPM_CHECK_ARG(cfg->mean_type >= DLPM_MEAN_EPSILON && cfg->mean_type <= DLPM_MEAN_PREVIOUS_X, "dlpm_sampler_create: unknown mean type %d", cfg->mean_type);
    DLPM_CHECK_ARG(!is_lim || cfg->mean_type == DLPM_MEAN_EPSILON, "LIM only supports epsilon prediction, fixed variance and rescaled timesteps");
    DLPM_CHECK_ARG(!is_lim || !(cfg->flags & (DLPM_UPD_CLIP | DLPM_UPD_ELEMENTWISE)),
                   "dlpm_sampler_create: the LIM sampler has no clip_denoised / non-isotropic variant (the reference's are commented out)");
    dlpm_sampler *s = new dlpm_sampler();
    s->cfg = *cfg;
    s->D = (int64_t)cfg->C * cfg->H * cfg->W;
    s->lim = is_lim;
    s->elem = (cfg->flags & DLPM_UPD_ELEMENTWISE) != 0;
    s->cols = s->elem ? cfg->B * s->D : cfg->B;
    if (s->elem && cfg->mlp) {
        set_error("dlpm_sampler_create: the toy MLP takes isotropic noise only (the reference asserts the same, Model.py:44)");
        delete s;
        return DLPM_ERR_UNSUPPORTED;
    }
    const int T = cfg->T;
    const int64_t B = cfg->B;
    std::vector<float> hg(T), hbg(T), hs(T), hbs(T);
    std::vector<float> lts(T), ltmp(T), lcx(T), lcs(T), lcn(T);
    if (is_lim) {
        if (have_lim) {
            for (int i = 0; i < T; i++) lts[i] = cfg->lim_ts[i];
            for (int i = 0; i < T - 1; i++) { ltmp[i] = cfg->lim_tmp[i]; lcx[i] = cfg->lim_cx[i]; lcs[i] = cfg->lim_cs[i]; lcn[i] = cfg->lim_cn[i]; }
        } else {
            int r = dlpm_lim_tables_f32(cfg->alpha, T - 1, (cfg->flags & DLPM_UPD_DLIM) != 0, lts.data(), ltmp.data(), lcx.data(),
                                        lcs.data(), lcn.data());
            if (r != DLPM_OK) { delete s; return r; }
        }
        hbs[T - 1] = 1.0f;   // x_0 = gen_eps.generate(shape), unscaled (GenerativeLevyProcess.py:464)
    } else if (have) {
        for (int i = 0; i < T; i++) { hg[i] = cfg->g[i]; hbg[i] = cfg->bg[i]; hs[i] = cfg->s[i]; hbs[i] = cfg->bs[i]; }
    } else {
        int r = dlpm_schedule_f32(T, cfg->alpha, hg.data(), hbg.data(), hs.data(), hbs.data());
        if (r != DLPM_OK) { delete s; return r; }
    }
    s->cfg.g = s->cfg.bg = s->cfg.s = s->cfg.bs = nullptr;  // host pointers are not retained
    s->cfg.lim_ts = s->cfg.lim_tmp = s->cfg.lim_cx = s->cfg.lim_cs = s->cfg.lim_cn = nullptr;
    const float *h_in_scale = cfg->in_scale;
    s->cfg.in_scale = nullptr;
    s->bs_last = hbs[T - 1];
    auto fail = [&](hipError_t e) {
        set_error("dlpm_sampler_create: %s", hipGetErrorString(e));
        dlpm_sampler_destroy(s);
        return DLPM_ERR_HIP;
    };
    hipError_t e;
    float **sched[4] = {&s->g, &s->bg, &s->s, &s->bs};
    std::vector<float> *hsrc[4] = {&hg, &hbg, &hs, &hbs};
    for (int i = 0; i < 4; i++) {
        if ((e = hipMalloc(sched[i], T * sizeof(float))) != hipSuccess) return fail(e);
        if ((e = hipMemcpy(*sched[i], hsrc[i]->data(), T * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    }
    if (is_lim) {
        float **ld[5] = {&s->lim_ts, &s->lim_tmp, &s->lim_cx, &s->lim_cs, &s->lim_cn};
        std::vector<float> *lsrc[5] = {&lts, &ltmp, &lcx, &lcs, &lcn};
        for (int i = 0; i < 5; i++) {
            if ((e = hipMalloc(ld[i], T * sizeof(float))) != hipSuccess) return fail(e);
            if ((e = hipMemcpy(*ld[i], lsrc[i]->data(), T * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
        }
    }
    if (h_in_scale) {
        if ((e = hipMalloc(&s->in_scale, T * sizeof(float))) != hipSuccess) return fail(e);
        if ((e = hipMemcpy(s->in_scale, h_in_scale, T * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
        if ((e = hipMalloc(&s->xin, (size_t)B * s->D * sizeof(float))) != hipSuccess) return fail(e);
    }
    const size_t tb = (size_t)T * s->cols * sizeof(float);
    if ((e = hipMalloc(&s->A, tb)) != hipSuccess) return fail(e);
    // Non-isotropic tables are T*B*D floats each (12.6 GB for [1024,3,32,32], T = 1000): A is only read again by
    // DLIM with eta > 0, otherwise c_eps is computed in place over it.
    const bool keepA = !s->elem || ((cfg->flags & DLPM_UPD_DLIM) && cfg->dlim_eta != 0.0f) || cfg->mean_type == DLPM_MEAN_Z;
    if (is_lim) {
        s->c_eps = s->A;     // LIM has no coefficient tables: A[i,b] is read directly by the update
        s->c_noise = nullptr;
    } else if (keepA) {
        if ((e = hipMalloc(&s->c_eps, tb)) != hipSuccess) return fail(e);
    } else {
        s->c_eps = s->A;
    }
    if (!is_lim && (e = hipMalloc(&s->c_noise, tb)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->hist_cell, sizeof(float *))) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->hist_cell, 0, sizeof(float *))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->x, (size_t)B * s->D * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->eps, (size_t)B * s->D * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->tvec, (size_t)B * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->t_dev, sizeof(int32_t))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&s->key_dev, 2 * sizeof(uint64_t))) != hipSuccess) return fail(e);
    {
        uint64_t key[2] = {cfg->seed, (uint64_t)cfg->sample_offset};
        if ((e = hipMemcpy(s->key_dev, key, sizeof(key), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    }
    if (cfg->unet) {
        s->ws_bytes = dlpm_unet_workspace_bytes(cfg->unet, B);
        if (s->ws_bytes < 0) {
            dlpm_sampler_destroy(s);
            return DLPM_ERR_STATE;
        }
        if ((e = hipMalloc(&s->ws, (size_t)s->ws_bytes)) != hipSuccess) return fail(e);
        s->plan_version = dlpm_unet_plan_version(cfg->unet);
        int r = build_time_table(s);
        if (r != DLPM_OK) {
            dlpm_sampler_destroy(s);
            return r;
        }
    }
    *out = s;
    return DLPM_OK;
}

extern "C" int dlpm_sampler_reseed(dlpm_sampler *s, uint64_t seed, int64_t sample_offset) {
    DLPM_CHECK_ARG(s, "dlpm_sampler_reseed: null handle");
    if (seed == s->cfg.seed && sample_offset == s->cfg.sample_offset) return DLPM_OK;
    s->cfg.seed = seed;
    s->cfg.sample_offset = sample_offset;
    // The key is read from device memory by the captured update node: no recapture needed.  It is written by a
    // one-thread kernel on the stream of the next begin*() / set_state / steps / step_injected, whichever comes first --
    // stream order puts it behind every replay of the previous trajectory (dlpm_sampler_steps fences its private stream
    // back into the caller's), so nothing synchronises here.
    s->key_dirty = true;
    return DLPM_OK;
}

extern "C" int dlpm_sampler_begin(dlpm_sampler *s, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(s, "dlpm_sampler_begin: null handle");
    hipStream_t st = as_stream(stream);
    const dlpm_sampler_config &c = s->cfg;
    if (s->lim) {
        // per-step a (unclamped: gen_sas draws its own, Distributions.py:63-64) for steps i = 0..T-2, then x_0
        if (c.alpha != 2.0 && !(c.flags & DLPM_UPD_DLIM))
            TRY(dlpm_skewed_levy_philox_f32(s->A, c.T - 1, c.B, c.alpha, -1.0, c.seed, c.sample_offset, st));
        TRY(dlpm_init_state_philox_f32(s->x, c.B, s->D, c.alpha, c.clamp_eps, 1.0f, c.seed, c.sample_offset, st));
    } else if (s->elem) {
        TRY(dlpm_skewed_levy_elem_philox_f32(s->A, c.T, c.B, s->D, c.alpha, c.clamp_a, c.seed, c.sample_offset, st));
        TRY(dlpm_init_state_elem_philox_f32(s->x, c.B, s->D, c.alpha, c.clamp_eps, s->bs_last, c.seed, c.sample_offset, st));
    } else {
        TRY(dlpm_skewed_levy_philox_f32(s->A, c.T, c.B, c.alpha, c.clamp_a, c.seed, c.sample_offset, st));
        TRY(dlpm_init_state_philox_f32(s->x, c.B, s->D, c.alpha, c.clamp_eps, s->bs_last, c.seed, c.sample_offset, st));
    }
    return build_tables(s, st);
}

extern "C" int dlpm_sampler_begin_injected(dlpm_sampler *s, const float *A_dev, const float *xT_dev, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(s && A_dev && xT_dev, "dlpm_sampler_begin_injected: null argument");
    hipStream_t st = as_stream(stream);
    DLPM_HIP(hipMemcpyAsync(s->A, A_dev, (size_t)(s->lim ? s->cfg.T - 1 : s->cfg.T) * s->cols * sizeof(float),
                            hipMemcpyDeviceToDevice, st));
    DLPM_HIP(hipMemcpyAsync(s->x, xT_dev, (size_t)s->cfg.B * s->D * sizeof(float), hipMemcpyDeviceToDevice, st));
    return build_tables(s, st);
}

extern "C" int dlpm_sampler_set_state(dlpm_sampler *s, const float *x_dev, int32_t t, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(s, "dlpm_sampler_set_state: null handle");
    DLPM_CHECK_ARG(t >= 1 && t <= s->cfg.T - 1, "dlpm_sampler_set_state: t = %d outside 1..%d", t, s->cfg.T - 1);
    hipStream_t st = as_stream(stream);
    TRY(flush_key(s, st));   // reseed -> set_state -> steps (resuming a trajectory under a new key) must not draw with the old one
    if (x_dev)
        DLPM_HIP(hipMemcpyAsync(s->x, x_dev, (size_t)s->cfg.B * s->D * sizeof(float), hipMemcpyDeviceToDevice, st));
    k_set_t<<<1, 64, 0, st>>>(s->t_dev, t);
    DLPM_LAUNCH_CHECK();
    s->t_host = t;
    return DLPM_OK;
}

extern "C" int dlpm_sampler_step_injected(dlpm_sampler *s, const float *z_dev, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(s, "dlpm_sampler_step_injected: null handle");
    if (s->t_host < 1) return DLPM_OK;
    TRY(sync_plan(s));
    TRY(flush_key(s, as_stream(stream)));
    TRY(one_step(s, z_dev, true, as_stream(stream)));
    s->t_host -= 1;
    return DLPM_OK;
}

static int steps_on(dlpm_sampler *s, int32_t nsteps, hipStream_t st, bool graph) {
    if (graph && !s->exec) {
        // the first step runs eagerly (sets function attributes, pages code in), then `use_graph`
        // consecutive steps are captured into one graph
        TRY(one_step(s, nullptr, true, st));
        s->t_host -= 1;
        nsteps -= 1;
        const int gs = s->cfg.use_graph;
        if (nsteps >= gs) {
            DLPM_HIP(hipStreamSynchronize(st));
            DLPM_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int r = DLPM_OK;
            for (int i = 0; i < gs && r == DLPM_OK; i++) r = one_step(s, nullptr, true, st);
            hipError_t e = hipStreamEndCapture(st, &s->graph);
            if (r != DLPM_OK) {
                if (s->graph) (void)hipGraphDestroy(s->graph);
                s->graph = nullptr;
                return r;
            }
            DLPM_HIP(e);
            DLPM_HIP(hipGraphInstantiate(&s->exec, s->graph, nullptr, nullptr, 0));
            s->graph_steps = gs;
            // capture only records: nothing ran, so t is unchanged and the replays below do the work
        }
    }
    while (nsteps > 0) {
        if (graph && s->exec && nsteps >= s->graph_steps) {
            DLPM_HIP(hipGraphLaunch(s->exec, st));
            nsteps -= s->graph_steps;
            s->t_host -= s->graph_steps;
        } else {
            TRY(one_step(s, nullptr, true, st));
            nsteps -= 1;
            s->t_host -= 1;
        }
    }
    return DLPM_OK;
}

extern "C" int dlpm_sampler_steps(dlpm_sampler *s, int32_t nsteps, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(s && nsteps >= 0, "dlpm_sampler_steps: bad argument");
    hipStream_t st = as_stream(stream);
    if (nsteps > s->t_host) nsteps = s->t_host;
    if (nsteps == 0) return DLPM_OK;
    TRY(sync_plan(s));
    TRY(flush_key(s, st));   // one thread, stream-ordered before the (replayed) steps; free when the key is clean
    // toy net, plain stochastic DLPM steps: the whole run of steps is one launch (state in registers)
    // (one wave per sample: best while the batch is latency-bound; beyond ~16k samples the 4-samples-per-wave
    //  forward kernel + update kernel reuse the weights better)
    if (s->cfg.mlp && !(s->cfg.flags & (DLPM_UPD_DLIM | DLPM_UPD_CLIP | DLPM_SMP_NO_FUSED_MLP | DLPM_SMP_LIM)) && s->D <= 4 && s->cfg.mean_type == DLPM_MEAN_EPSILON &&
        s->cfg.B <= 16384 && !s->hist && !s->in_scale && !prof_enabled()) {
        TRY(dlpm_mlp_sample_steps_f32(s->cfg.mlp, s->x, s->c_eps, s->c_noise, s->g, s->cfg.T, s->cfg.B, s->t_host, nsteps,
                                      s->cfg.seed, s->cfg.sample_offset, s->key_dev, st));
        s->t_host -= nsteps;
        k_set_t<<<1, 64, 0, st>>>(s->t_dev, s->t_host);
        DLPM_LAUNCH_CHECK();
        return DLPM_OK;
    }
    const bool graph = s->cfg.use_graph && !prof_enabled();
    if (!graph) return steps_on(s, nsteps, st, false);
    if (!s->own) {
        DLPM_HIP(hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking));
        DLPM_HIP(hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming));
        DLPM_HIP(hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming));
    }
    DLPM_HIP(hipEventRecord(s->ev_in, st));            // everything the caller queued so far ...
    DLPM_HIP(hipStreamWaitEvent(s->own, s->ev_in, 0)); // ... precedes the replayed steps
    int r = steps_on(s, nsteps, s->own, true);
    DLPM_HIP(hipEventRecord(s->ev_out, s->own));
    DLPM_HIP(hipStreamWaitEvent(st, s->ev_out, 0));    // and the caller's later work follows them
    return r;
}

extern "C" int dlpm_sampler_set_history(dlpm_sampler *s, float *hist_dev, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(s, "dlpm_sampler_set_history: null handle");
    if (hist_dev == s->hist) return DLPM_OK;
    // the captured update node reads the base from a device cell, so the graph follows the new buffer; the cell is
    // written in stream order (earlier replays were fenced back into the caller's stream), no host synchronisation
    k_set_ptr<<<1, 64, 0, as_stream(stream)>>>(s->hist_cell, hist_dev);
    DLPM_LAUNCH_CHECK();
    s->hist = hist_dev;
    return DLPM_OK;
}

extern "C" int dlpm_sampler_copy_state(dlpm_sampler *s, float *out_dev, dlpm_stream_t stream) {
    DLPM_CHECK_ARG(s && out_dev, "dlpm_sampler_copy_state: null argument");
    DLPM_HIP(hipMemcpyAsync(out_dev, s->x, (size_t)s->cfg.B * s->D * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
    return DLPM_OK;
}

extern "C" float *dlpm_sampler_state(dlpm_sampler *s) { return s ? s->x : nullptr; }
extern "C" int32_t dlpm_sampler_t(const dlpm_sampler *s) { return s ? s->t_host : -1; }
extern "C" float *dlpm_sampler_table(dlpm_sampler *s, int which) {
    if (!s) return nullptr;
    switch (which) {
        case 0: return s->A;
        case 1: return s->c_eps;
        case 2: return s->c_noise;
        case 3: return s->eps;
        default: return nullptr;
    }
}

extern "C" void dlpm_sampler_destroy(dlpm_sampler *s) {
    if (!s) return;
    if (s->exec) (void)hipGraphExecDestroy(s->exec);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    if (s->ev_in) (void)hipEventDestroy(s->ev_in);
    if (s->ev_out) (void)hipEventDestroy(s->ev_out);
    if (s->own) (void)hipStreamDestroy(s->own);
    void *bufs[] = {s->g, s->bg, s->s, s->bs, s->A, s->c_eps == s->A ? nullptr : s->c_eps, s->c_noise, s->x, s->eps, s->tvec, s->emb_tab,
                    s->t_dev, s->key_dev, s->ws, s->hist_cell, s->lim_ts, s->lim_tmp, s->lim_cx, s->lim_cs, s->lim_cn,
                    s->in_scale, s->xin};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    delete s;
}
