/*
 * dlpm_amd.h -- C ABI of libdlpm_amd.so: the MI355X-native (gfx950) implementation of DLPM's
 * reverse-time sampling hot path.
 *
 * The reference (darioShar/DLPM) has no FFI layer: its boundary is Python duck typing
 * (`model(x, t)`, `method.sample(...)`).  This header is the boundary a
 * maintainer would bind (ctypes stubs in INTEGRATION.md); every entry point cites the
 * reference code it replaces.  Plain pointers and sizes only -- no torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative dlpm_status otherwise;
 *     dlpm_last_error() returns a thread-local message for the last failure.
 *   - `*_dev` pointers are device (HBM) pointers owned by the caller; `stream` is a
 *     hipStream_t passed as void* (NULL = the default stream).  Nothing here allocates on
 *     the caller's behalf except the opaque handles, which own their weights.
 *   - launch functions never synchronise and never allocate, so they may be captured into a
 *     hipGraph by the caller (dlpm_sampler_* does exactly that).
 *   - state tensors cross the boundary in the reference's layout: fp32, (B, C, H, W)
 *     contiguous (or (B, 1, 2) for the toy data), D = C*H*W elements per sample.
 */
#ifndef DLPM_AMD_H
#define DLPM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *dlpm_stream_t;

enum dlpm_status {
    DLPM_OK = 0,
    DLPM_ERR_ARG = -1,     /* bad argument (the reference would raise AssertionError/Exception) */
    DLPM_ERR_HIP = -2,     /* HIP runtime error; message carries hipGetErrorString */
    DLPM_ERR_UNSUPPORTED = -3,
    DLPM_ERR_STATE = -4,   /* call order (e.g. forward before finalize) */
    DLPM_ERR_NOMEM = -5,   /* workspace too small */
    DLPM_ERR_IO = -6       /* host file I/O (image dump) */
};

const char *dlpm_last_error(void);
int dlpm_abi_version(void);

/* Per-kernel-class timing with HIP events on the launch stream (eager launches only; the sampler
 * does not replay its graph while profiling is on).  dlpm_prof_report synchronises, then writes one
 * line per class: "name launches total_ms flops bytes" (flops/bytes = algorithmic totals). */
int dlpm_prof_enable(int on);
int dlpm_prof_report(char *buf, int64_t buf_bytes);

/* ------------------------------------------------------------------------------------------
 * Host side: schedule and the reference-compatible ("identical seeds") random streams
 * ------------------------------------------------------------------------------------------ */

/* Cosine schedule -> gammas, bargammas, sigmas, barsigmas, each [T] fp32 (host buffers).
 * Replaces DLPM.gen_noise_schedule('scale_preserving') + get_timesteps('linear'):
 * dlpm/methods/dlpm.py:103-156; also what rescale_diffusion (:176-185) recomputes. */
int dlpm_schedule_f32(int T, double alpha, float *g, float *bg, float *s, float *bs);

/* The 'scale_exploding' schedule (`--scale scale_exploding`, dlpm/methods/dlpm.py:134-149): gammas = bargammas = 1,
 * barsigmas = Karras grid (sigma_min 0.002, sigma_max 80, rho 7), sigmas^alpha = successive differences of
 * barsigmas^alpha (the reference accumulates them with an O(T^2) loop of running sums; this evaluates the same
 * recurrence in double). */
int dlpm_schedule_exploding_f32(int T, double alpha, float *g, float *bg, float *s, float *bs);

/* MT19937 stream with numpy/torch semantics.  `cached`/`has_cached` is the spare double normal
 * torch's CPU generator keeps for its scalar (n < 16) path. */
typedef struct dlpm_mt19937 {
    uint32_t key[624];
    int32_t pos;
    int32_t has_cached;
    double cached;
} dlpm_mt19937;

/* np.random.seed(int) / torch.manual_seed(int) (low 32 bits): init_genrand. */
int dlpm_mt19937_seed(dlpm_mt19937 *st, uint32_t seed);

/* Isotropic totally-skewed alpha/2-stable draws a[n] (fp32), consuming n uniforms THEN n
 * exponentials from a numpy-compatible stream, Chambers-Mallows-Stuck in fp64, cast to fp32,
 * optional clamp to [0, clamp_a] (clamp_a < 0: none); alpha == 2 returns the constant 2 and
 * consumes nothing.  Replaces gen_skewed_levy: bem/datasets/Distributions.py:33-51
 * (scipy.stats.levy_stable.rvs(alpha/2, 1, loc=0, scale=2cos(pi alpha/4)^(2/alpha))). */
int dlpm_skewed_levy_host_f32(dlpm_mt19937 *np_stream, double alpha, int64_t n, double clamp_a, float *out);

/* torch.randn(n) on the CPU generator (fp32): 24-bit uniforms + Box-Muller in blocks of 16 for
 * n >= 16, scalar double path with the cached spare for n < 16.  Replaces torch.randn /
 * randn_like at Distributions.py:65 and GenerativeLevyProcess.py:236 for CPU-seed parity. */
int dlpm_randn_host_f32(dlpm_mt19937 *torch_stream, int64_t n, float *out);

/* ------------------------------------------------------------------------------------------
 * Device side: noise, coefficient tables, fused update
 * ------------------------------------------------------------------------------------------ */

/* A_dev[T,B] <- skewed-Levy draws from Philox4x32-10 keyed by (seed, global sample index
 * sample_offset + b, row t): results do not depend on how the batch is sharded over GPUs.
 * CMS evaluated in fp64 like scipy, stored fp32, clamped to [0, clamp_a] if clamp_a >= 0.
 * Replaces DLPM.sample_A: dlpm/methods/dlpm.py:226-227 (keeps [T,B], not [T,B,C,H,W]). */
int dlpm_skewed_levy_philox_f32(float *A_dev, int T, int64_t B, double alpha, double clamp_a,
                                uint64_t seed, int64_t sample_offset, dlpm_stream_t stream);

/* Non-isotropic noise (`--non_iso`, script_utils.py:26-27): A_dev[T,B,D] with one independent draw per ELEMENT
 * (gen_skewed_levy with isotropic=False, bem/datasets/Distributions.py:47-48), Philox-keyed by (seed, global
 * sample index, element, row t).  The Sigma recursion is then per element: call dlpm_coeff_tables_f32 with
 * B*D in place of B and dlpm_update_f32 with DLPM_UPD_ELEMENTWISE. */
int dlpm_skewed_levy_elem_philox_f32(float *A_dev, int T, int64_t B, int64_t D, double alpha, double clamp_a,
                                     uint64_t seed, int64_t sample_offset, dlpm_stream_t stream);

/* Non-isotropic x_T: as dlpm_init_state_philox_f32 with an independent unclamped a0 per element
 * (gen_sas with isotropic=False, Distributions.py:57-73). */
int dlpm_init_state_elem_philox_f32(float *x_dev, int64_t B, int64_t D, double alpha, double clamp_eps,
                                    float barsigma_last, uint64_t seed, int64_t sample_offset, dlpm_stream_t stream);

/* x_dev[B,D] <- barsigma_last * clamp(sqrt(a0[b]) * z, +-clamp_eps), a0 an UNclamped skewed-Levy
 * draw, z ~ N(0,1) (Philox).  Replaces the x_T init: GenerativeLevyProcess.py:313 -> gen_sas,
 * Distributions.py:57-73. */
int dlpm_init_state_philox_f32(float *x_dev, int64_t B, int64_t D, double alpha, double clamp_eps,
                               float barsigma_last, uint64_t seed, int64_t sample_offset,
                               dlpm_stream_t stream);

/* Sigma recursion + per-step coefficients from A[T,B] and the schedule (device pointers, [T]):
 *   Sigma_t = s_t^2 A_t + g_t^2 Sigma_{t-1};  Gamma_t = 1 - g_t^2 Sigma_{t-1}/Sigma_t
 *   c_eps[t,b] = bs_t * Gamma_t ;  c_noise[t,b] = 1[t != 1] * sqrt(Gamma_t * Sigma_{t-1})
 * rows t = 1..T-1 are written (row 0 is zeroed).  sigmas_out_dev (nullable) receives Sigma[T,B].
 * Replaces compute_Sigmas / compute_Gamma_t / compute_Sigma_tilde_t_1: dlpm.py:230-257. */
int dlpm_coeff_tables_f32(const float *A_dev, const float *g_dev, const float *s_dev, const float *bs_dev,
                          int T, int64_t B, float *c_eps_dev, float *c_noise_dev, float *sigmas_out_dev,
                          dlpm_stream_t stream);

enum dlpm_update_flags {
    DLPM_UPD_DLIM = 1,        /* deterministic / DLIM step instead of the stochastic DLPM step */
    DLPM_UPD_CLIP = 2,        /* clip_denoised: eps <- predict_eps(clamp(predict_xstart)) first  */
    DLPM_UPD_ADVANCE = 4,     /* after the update, thread 0 decrements *t_dev (graph replay)     */
    DLPM_SMP_NO_FUSED_MLP = 8,/* sampler only: do not use the one-launch toy-net loop            */
    DLPM_UPD_ELEMENTWISE = 16,/* non-isotropic noise: c_eps / c_noise / A are [T,B,D] (also a sampler flag) */
    DLPM_SMP_LIM = 32,        /* sampler only: continuous-time LIM sampler (`method: lim`); DLPM_UPD_DLIM = its ODE */
    DLPM_UPD_HIST_ON = 64     /* accounting hint for dlpm_prof_*: the history cell currently holds a buffer, so the launch
                                 also writes 4 B/element (the kernel itself reads the cell; arithmetic is unaffected)  */
};

typedef struct dlpm_update_args {
    float *x_dev;             /* [B,D] state, updated in place                                   */
    const float *eps_dev;     /* [B,D] model output                                              */
    const float *z_dev;       /* [B,D] injected N(0,1) noise, or NULL = in-kernel Philox         */
    const int32_t *t_dev;     /* device scalar: current step index t in [1, T-1]                 */
    const float *g_dev, *bg_dev, *bs_dev;      /* schedule, [T]                                  */
    const float *c_eps_dev, *c_noise_dev;      /* [T,B] from dlpm_coeff_tables_f32               */
    const float *A_dev;       /* [T,B], only read for DLIM with eta > 0                          */
    int64_t B, D;
    int32_t T;
    int32_t flags;            /* dlpm_update_flags                                               */
    float dlim_eta;
    float alpha;
    uint64_t seed;            /* Philox key                                                      */
    int64_t sample_offset;    /* global index of sample 0 of this shard                          */
    const uint64_t *key_dev;  /* optional device pair {seed, sample_offset} overriding the two fields
                                 above: lets a captured graph be replayed under a new key          */
    float *const *hist_pp;    /* optional DEVICE cell holding the base of a [T,B,D] history buffer (or NULL in
                                 the cell = off): the output of step t is also stored at row T - t, the row the
                                 reference's `samples` list gives it (GenerativeLevyProcess.py:274-288).  A cell,
                                 not a pointer, so a captured graph follows a new buffer                        */
} dlpm_update_args;

/* One reverse step on the whole batch:
 *   DLPM: x <- (x - c_eps[t,b] eps)/g_t + c_noise[t,b] z     dlpm.py:272-278, GenerativeLevyProcess.py:225-239
 *   DLIM: x <- (x - bs_t eps)/g_t + bs_{t-1} eps  (eta = 0)   dlpm.py:281-297
 *   CLIP: eps <- (x - clamp((x - eps bs_t)/bg_t, -1, 1) bg_t)/bs_t first   GenerativeLevyProcess.py:186-207
 * DLPM_UPD_ELEMENTWISE (non-isotropic noise, `--non_iso`, Distributions.py:47-48): the three tables are
 * [T,B,D] and indexed per element instead of per sample.
 * HBM-bound: 12 B/element with Philox noise, 16 B/element with injected z; +8 B/element for elementwise
 * tables, +4 B/element when a history row is written. */
int dlpm_update_f32(const dlpm_update_args *args, dlpm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Model mean types (what the net predicts) -- p_mean_variance, GenerativeLevyProcess.py:182-207, ModelMeanType dlpm.py:10-18
 * ------------------------------------------------------------------------------------------ */
enum dlpm_mean_type {
    DLPM_MEAN_EPSILON = 0,    /* the net predicts eps (every shipped config)                                        */
    DLPM_MEAN_START_X = 1,    /* the net predicts x_0                                                               */
    DLPM_MEAN_Z = 2,          /* the net predicts z_t: eps = sqrt(A[t,b]) out  (:192-196)                           */
    DLPM_MEAN_PREVIOUS_X = 3  /* the net predicts the anterior mean: eps = (x - out g_t) / (bs_t Gamma_t)  (dlpm.py:204-209) */
};
enum dlpm_predict_flags {
    DLPM_PRED_TO_XSTART = 1,  /* model output -> x_0 per mean type (EPSILON: predict_xstart, dlpm.py:191-196)       */
    DLPM_PRED_CLIP = 2,       /* clamp to [-1, 1] (clip_denoised, process_xstart :162-167)                          */
    DLPM_PRED_TO_EPS = 4,     /* x_0 -> eps = (x - x_0 bg_t) / bs_t  (predict_eps, dlpm.py:198-202)                 */
    DLPM_PRED_ELEMENTWISE = 16/* non-isotropic: c_eps / A are [T,B,D] (same bit as DLPM_UPD_ELEMENTWISE)            */
};
typedef struct dlpm_predict_args {
    const float *x_dev;       /* [B,D] state x_t                                                                    */
    const float *in_dev;      /* [B,D] model output, or an x_0 (flags without DLPM_PRED_TO_XSTART)                  */
    float *out_dev;           /* [B,D] result; may alias in_dev                                                     */
    const int32_t *t_dev;     /* device scalar: step index t                                                        */
    const float *g_dev, *bg_dev, *bs_dev;   /* schedule, [T]                                                        */
    const float *c_eps_dev;   /* [T,B] bs_t Gamma_t from dlpm_coeff_tables_f32 (PREVIOUS_X only)                    */
    const float *A_dev;       /* [T,B] (Z only)                                                                     */
    int64_t B, D;
    int32_t T;
    int32_t mean_type;        /* dlpm_mean_type                                                                     */
    int32_t flags;            /* dlpm_predict_flags                                                                 */
} dlpm_predict_args;
/* The part of p_mean_variance between the model call and anterior_mean_variance_*: out = stages(in) in the order
 * TO_XSTART -> CLIP -> TO_EPS.  A caller with a `denoised_fn` runs TO_XSTART, applies its function, then CLIP | TO_EPS;
 * without one all three stages are one launch.  The reference indexes A[t] / Sigmas[t] with the [B] tensor t in the Z and
 * PREVIOUS_X branches, which only has the intended per-sample meaning for B = 1 (it raises for B > 1); here row t,
 * column b is used, i.e. the B = 1 result for every sample.  HBM-bound: 12 B/element. */
int dlpm_predict_f32(const dlpm_predict_args *args, dlpm_stream_t stream);

/* tvec_dev[b] = float(*t_dev) * (1/T): the `t/T` the reference feeds the net
 * (GenerativeLevyProcess._scale_timesteps, :92-96). */
int dlpm_fill_scaled_t_f32(float *tvec_dev, const int32_t *t_dev, int32_t T, int64_t B, dlpm_stream_t stream);

/* out_dev[i] = x_dev[i] * table_dev[*t_dev]: the `input_scaling` 1/(1 + barsigma_t) applied to the net input when
 * the schedule is scale_exploding (GenerativeLevyProcess.py:176-180). */
int dlpm_scale_by_table_f32(const float *x_dev, float *out_dev, int64_t n, const int32_t *t_dev, const float *table_dev,
                            dlpm_stream_t stream);

/* samples_dev <- clamp(x, -c, c) then (x+1)/2 for images: GenerationManager.generate post-processing,
 * bem/GenerationManager.py:50-63, bem/datasets/__init__.py:108-109. */
int dlpm_postprocess_f32(const float *x_dev, float *out_dev, int64_t n, float clamp, int affine, dlpm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LIM sampler (`method: lim`): same score net and noise, continuous-time coefficients.
 * Reference: dlpm/methods/LIM/functions/sde.py:5-49 (VPSDE, cosine schedule, T = 0.9946),
 * dlpm/methods/LIM/functions/sampler.py:85-181,217-258 (ode_score_update / sde_score_update, time grid),
 * GenerativeLevyProcess.lim_sample (dlpm/methods/GenerativeLevyProcess.py:454-507).
 * ------------------------------------------------------------------------------------------ */

/* [host] per-step scalars of a `steps`-step LIM run, step i going from time ts[i] to ts[i+1],
 * ts = linspace(0.9946, 1e-5, steps + 1):
 *   ts[steps+1]  the grid (ts[i] is also what the net is fed at step i)
 *   tmp[steps]   marginal_std(s)^-(alpha-1)        (alpha = 2: (marginal_std(s) + 1e-5)^-1)
 *   cx, cs, cn   x / score / noise coefficients of the SDE update, or of the ODE update when ode != 0 (cn = 0)
 * Evaluated in double and rounded to fp32 (the reference evaluates in fp32 torch ops; dlpm_amd's Python mirror
 * reproduces those bit for bit and passes its own tables -- this entry point serves non-Python hosts). */
int dlpm_lim_tables_f32(double alpha, int32_t steps, int32_t ode, float *ts, float *tmp, float *cx, float *cs, float *cn);

typedef struct dlpm_lim_update_args {
    float *x_dev;             /* [B,D] state, updated in place                                          */
    const float *eps_dev;     /* [B,D] model output                                                     */
    const float *z_dev;       /* [B,D] injected N(0,1), or NULL = in-kernel Philox                      */
    const int32_t *t_dev;     /* device countdown T-1 .. 1; the step index is i = (T-1) - *t_dev        */
    const float *tmp_dev, *cx_dev, *cs_dev, *cn_dev;   /* [T-1] from dlpm_lim_tables_f32 (device copies) */
    const float *A_dev;       /* [T-1,B] per-step, per-sample skewed-Levy a (gen_sas draws a fresh one each
                                 step, Distributions.py:63-65), or NULL = 1 (alpha = 2: plain Gaussian) */
    int64_t B, D;
    int32_t T;                /* steps + 1                                                              */
    int32_t flags;            /* DLPM_UPD_DLIM = ODE update (no noise); DLPM_UPD_ADVANCE                */
    float clamp_eps;          /* < 0: none; clamps sqrt(a) z (gen_sas), not applied when A_dev is NULL  */
    uint64_t seed;
    int64_t sample_offset;
    const uint64_t *key_dev;  /* as in dlpm_update_args                                                 */
    float *const *hist_pp;    /* as in dlpm_update_args: row T - t of a [T,B,D] buffer                  */
} dlpm_lim_update_args;

/* x <- cx x + cs (eps tmp) [+ cn clamp(sqrt(a) z)]   -- sde_score_update / ode_score_update. */
int dlpm_lim_update_f32(const dlpm_lim_update_args *args, dlpm_stream_t stream);

/* tvec_dev[b] = ts_dev[(T-1) - *t_dev]: the continuous time the LIM loop feeds the net (sampler.py:233). */
int dlpm_fill_table_t_f32(float *tvec_dev, const int32_t *t_dev, const float *ts_dev, int32_t T, int64_t B,
                          dlpm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Generated-image dump: what EvaluationManager does with each chunk of samples,
 * bem/evaluate/EvaluationManager.py:174-196 -- `tvu.save_image(samples[i], f"{i+total}.png")` per sample.
 * ------------------------------------------------------------------------------------------ */

/* [B,C,H,W] fp32 images in [0,1] (the output of dlpm_postprocess_f32) -> [B,H,W,3] bytes with
 * torchvision.utils.save_image's quantisation: mul(255).add_(0.5).clamp_(0,255).to(uint8), a 1-channel
 * image replicated to RGB as make_grid does.  C must be 1 or 3.  HBM-bound: 4*C + 3 bytes per pixel. */
int dlpm_images_to_rgb8(const float *x_dev, uint8_t *out_dev, int64_t B, int32_t C, int32_t H, int32_t W,
                        dlpm_stream_t stream);

/* [host] upper bound of one encoded H x W RGB PNG, in bytes (-1 on bad arguments). */
int64_t dlpm_png_bound(int32_t H, int32_t W);

/* [host] one H x W x 3 byte image -> PNG stream (8-bit truecolour, zlib `level` 0..9) in out[0..*len). */
int dlpm_png_encode_rgb8(const uint8_t *hwc, int32_t H, int32_t W, int32_t level, uint8_t *out, int64_t cap,
                         int64_t *len);

/* [host] B images -> files `<dir>/<first_index + i>.png` (the reference's naming, EvaluationManager.py:190),
 * encoded and written by `nthreads` host threads.  Blocks until all files are closed. */
int dlpm_png_write_rgb8(const uint8_t *hwc_batch, int64_t B, int32_t H, int32_t W, const char *dir,
                        int64_t first_index, int32_t level, int32_t nthreads);

/* ------------------------------------------------------------------------------------------
 * Score networks
 * ------------------------------------------------------------------------------------------ */

typedef struct dlpm_unet_config {
    int32_t in_channels, model_channels, out_channels, num_res_blocks;
    int32_t num_heads, image_size;
    int32_t n_mult, channel_mult[8];
    int32_t n_attn, attention_resolutions[8];
} dlpm_unet_config;                 /* = the arguments of _unet_model: dlpm/dlpm_experiment.py:38-56 */

typedef struct dlpm_unet dlpm_unet; /* opaque; owns device copies of the weights */

int dlpm_unet_create(const dlpm_unet_config *cfg, dlpm_unet **out);
/* Upload one tensor by its reference state_dict key (e.g. "input_blocks.7.1.qkv.weight"), from a
 * host fp32 buffer in the reference's layout (OIHW conv weights etc.); the library re-lays it out. */
int dlpm_unet_set_param(dlpm_unet *net, const char *key, const float *host_data, int64_t numel);
/* Number of parameter tensors the architecture expects / a key by index (for loaders). */
int dlpm_unet_num_params(const dlpm_unet *net);
const char *dlpm_unet_param_key(const dlpm_unet *net, int index, int64_t *numel_out);
int dlpm_unet_finalize(dlpm_unet *net);     /* all params set -> build the launch plan */
int64_t dlpm_unet_workspace_bytes(const dlpm_unet *net, int64_t B);
/* eps_dev[B,C,H,W] = UNet(x_dev[B,C,H,W], t_dev[B]) -- UNetModel.forward, dlpm/models/unet.py:463-492,
 * with t already scaled (i/T floats).  Activations live in workspace_dev (NHWC, fp32). */
int dlpm_unet_forward(dlpm_unet *net, const float *x_dev, const float *t_dev, float *eps_dev, int64_t B,
                      void *workspace_dev, int64_t workspace_bytes, dlpm_stream_t stream);
/* The UNet's time path -- timestep_embedding -> time_embed MLP -> the per-ResBlock emb linears, row-concatenated (unet.py:147-150,
 * 336-338, 470) -- depends on the timestep alone, and in the sampling loop the timestep is the step index (GenerativeLevyProcess.py
 * :319, _scale_timesteps :92-96): a sampler computes it ONCE for every step,
 *   dlpm_unet_time_embeddings: out_dev[M][dlpm_unet_time_embedding_width(net)] for t_dev[M], with
 *   dlpm_unet_time_embeddings_scratch_bytes(net, M) bytes of device scratch,
 * and binds the table with the device step counter that picks the row: while a table is bound, dlpm_unet_forward_uniform_t /
 * dlpm_unet_forward_update read row *row_index_dev instead of running the four time-path launches (same kernels produced the
 * rows: same bits).  Bind (NULL, NULL) to unbind.  The pointers are baked into a captured graph like every other argument.
 * ONE binding per CALLING HOST THREAD, which names the net: forward calls this thread enqueues for THAT net read the table; other
 * threads, and other nets, do not see it.  Binding a second net on the same thread REPLACES the first (no per-net map: bind, enqueue,
 * unbind -- as dlpm_sampler does); unbinding net A while net B is the one bound leaves B's binding alone.  The rows depend on the net's weights and GEMM policy: whoever caches
 * a table recomputes it when dlpm_unet_plan_version moves (dlpm_sampler does, before its next step). */
int64_t dlpm_unet_time_embedding_width(const dlpm_unet *net);
int64_t dlpm_unet_time_embeddings_scratch_bytes(const dlpm_unet *net, int64_t M);
int dlpm_unet_time_embeddings(dlpm_unet *net, const float *t_dev, int64_t M, float *out_dev, void *scratch_dev, int64_t scratch_bytes,
                              dlpm_stream_t stream);
int dlpm_unet_bind_time_table(dlpm_unet *net, const float *table_dev, const int32_t *row_index_dev);
/* The same forward for a batch that shares ONE timestep (what every step of the sampling loop is: t = [i] * B,
 * GenerativeLevyProcess.py:319): only t_dev[0] is read, and the time-embedding MLP and the per-ResBlock emb linears
 * (unet.py:147-150, 336-338) are evaluated for one row instead of B identical ones.  Same bits as dlpm_unet_forward. */
int dlpm_unet_forward_uniform_t(dlpm_unet *net, const float *x_dev, const float *t_dev, float *eps_dev, int64_t B,
                                void *workspace_dev, int64_t workspace_bytes, dlpm_stream_t stream);
/* One network evaluation AND the reverse update of the sampler in one call: x_in_dev is what the net reads (the state, or its
 * input-scaled copy), upd describes the update of upd->x_dev exactly as for dlpm_update_f32 except that upd->eps_dev is ignored --
 * eps is what this forward computes.  For the stochastic DLPM step without clipping / DLIM / element-wise tables, on nets whose
 * head convolution runs on the VALU head kernel (Cout <= 4, 16 | model_channels, 8 <= W <= 64), the update is applied in that
 * kernel's epilogue and eps never reaches HBM; bit-identical to dlpm_unet_forward_uniform_t + dlpm_update_f32.  Every other
 * variant runs exactly that pair, through eps_scratch_dev ([B, D]; may be NULL when the fused form applies).  t_dev: B floats,
 * all equal (as for dlpm_unet_forward_uniform_t).  Replaces p_mean_variance + p_sample, GenerativeLevyProcess.py:170-239. */
int dlpm_unet_forward_update(dlpm_unet *net, const float *x_in_dev, const float *t_dev, const dlpm_update_args *upd,
                             float *eps_scratch_dev, int64_t B, void *workspace_dev, int64_t workspace_bytes,
                             dlpm_stream_t stream);
/* After a forward: copy block output `index` (0..n_in-1 down, then middle, then up blocks) to a
 * device buffer as NCHW for bisecting against UNetModel.get_feature_vectors (unet.py:494-524). */
/* Block outputs are only kept when asked for: by default the activation arena recycles each buffer after its last
 * consumer (the workspace is the PEAK, not the sum).  on = 1: no recycling, features stay readable after a forward. */
int dlpm_unet_keep_features(dlpm_unet *net, int on);
int dlpm_unet_num_features(const dlpm_unet *net);
int dlpm_unet_feature_shape(const dlpm_unet *net, int index, int32_t *C, int32_t *H, int32_t *W);
int dlpm_unet_get_feature(dlpm_unet *net, int index, float *out_nchw_dev, int64_t B, dlpm_stream_t stream);
int64_t dlpm_unet_flops_per_sample(const dlpm_unet *net);   /* 2*MAC, conv/linear/attention */

/* Which kernel generation the 3x3 stride-1 convolutions take.  The generations (Winograd F(4x4,3x3), Winograd
 * F(2x2,3x3), implicit GEMM) round differently, so the choice is a function of the layer's geometry and of this
 * per-net policy ONLY -- never of the batch a call happens to carry: a sample's value must not depend on how its batch
 * was sharded over GPUs or cut into chunks.
 *   DLPM_CONV_AUTO   the fastest generation each layer's geometry admits (default);
 *   DLPM_CONV_F4 / _F2 / _IGEMM   that generation wherever the geometry admits it (tests, A/B measurements).
 * dispatch_batch > 0 (AUTO only): additionally weigh grid occupancy for THAT batch -- a property the caller declares
 * for the whole configuration (e.g. the per-GPU shard of the global batch), identical on every rank and chunk; 0 = off.
 * May be called any time after dlpm_unet_finalize; samplers re-capture their graph on the next step. */
enum dlpm_conv_generation { DLPM_CONV_AUTO = 0, DLPM_CONV_F4 = 1, DLPM_CONV_F2 = 2, DLPM_CONV_IGEMM = 3 };
int dlpm_unet_set_conv_policy(dlpm_unet *net, int32_t generation, int64_t dispatch_batch);
/* Which matrix pipe the 1x1 convolutions (qkv, proj, skip connections) and the stride-2 downsampling convolutions run on.  DLPM_GEMM_BF16X3: every fp32 operand is
 * cut exactly into three bf16 planes and a product is formed from the six partial products above 2^-16 of it, accumulated
 * in fp32 (v_mfma_f32_32x32x16_bf16) -- fp32-grade results (measured against float64 beside the fp32 MFMA in the tests)
 * at 6/16 of the fp32 pipe's cost.  DLPM_GEMM_F32: v_mfma_f32_32x32x2_f32 everywhere.  DLPM_GEMM_AUTO = BF16X3 where the
 * shape admits it (Cout % 128 == 0, Cin % 32 == 0, whole 128-pixel tiles).  Like the conv policy: per net, never a
 * function of the batch. */
enum dlpm_gemm_mode { DLPM_GEMM_AUTO = 0, DLPM_GEMM_F32 = 1, DLPM_GEMM_BF16X3 = 2 };
int dlpm_unet_set_gemm_policy(dlpm_unet *net, int32_t mode);
/* Bumped by every call that changes the launch plan (dlpm_unet_set_conv_policy, dlpm_unet_finalize). */
int64_t dlpm_unet_plan_version(const dlpm_unet *net);
void dlpm_unet_destroy(dlpm_unet *net);

typedef struct dlpm_mlp dlpm_mlp;   /* MLPModel of 2d_data.yml: dlpm/models/Model.py:17-211 */
int dlpm_mlp_create(int32_t nfeatures, int32_t nunits, int32_t nblocks, int32_t time_emb_size, dlpm_mlp **out);
int dlpm_mlp_set_param(dlpm_mlp *net, const char *key, const float *host_data, int64_t numel);
int dlpm_mlp_finalize(dlpm_mlp *net);
/* eps_dev[B,1,F] = MLP(x_dev[B,1,F], t_dev[B]) -- MLPModel.forward, Model.py:148-211. */
int dlpm_mlp_forward(dlpm_mlp *net, const float *x_dev, const float *t_dev, float *eps_dev, int64_t B,
                     dlpm_stream_t stream);
/* `nsteps` whole reverse steps t = t_start, t_start-1, ... for the toy net in ONE launch: model forward
 * + DLPM update (Philox noise) with the state held in registers (nfeatures <= 4).  x_dev[B,1,F] is
 * updated in place; tables as produced by dlpm_coeff_tables_f32.  key_dev (nullable) overrides
 * {seed, sample_offset} from device memory.  Replaces the loop body of p_sample_loop_progressive
 * (GenerativeLevyProcess.py:317-330) for BASELINE configs[0], which is otherwise launch-bound. */
int dlpm_mlp_sample_steps_f32(dlpm_mlp *net, float *x_dev, const float *c_eps_dev, const float *c_noise_dev,
                              const float *g_dev, int32_t T, int64_t B, int32_t t_start, int32_t nsteps,
                              uint64_t seed, int64_t sample_offset, const uint64_t *key_dev, dlpm_stream_t stream);
void dlpm_mlp_destroy(dlpm_mlp *net);

/* ------------------------------------------------------------------------------------------
 * Building-block launches (exported so each kernel can be parity-tested through the C ABI)
 * ------------------------------------------------------------------------------------------ */

typedef struct dlpm_conv_args {
    const float *src0, *src1;   /* NHWC inputs [B,Hin,Win,C0] and [B,Hin,Win,C1] (virtual channel concat; src1 may be NULL) */
    int32_t C0, C1;
    int32_t B, Hin, Win, Hout, Wout;
    int32_t ksize;              /* 1 or 3 (padding = ksize/2) */
    int32_t stride;             /* 1 or 2 */
    int32_t upsample;           /* 1: nearest x2 upsample fused into the loads (Upsample, unet.py:73) */
    const float *weight;        /* OIHW, as in the reference state_dict, on the DEVICE */
    const float *bias;          /* [Cout] or NULL */
    const float *coefA, *coefB; /* [B, C0+C1] fused normalisation y = x*A + B (NULL: none) */
    int32_t act_silu;           /* apply SiLU after the affine (GN -> SiLU -> conv) */
    const float *res0, *res1;   /* residual added in the epilogue, NHWC at output size (virtual concat) */
    int32_t R0;
    float *out;                 /* NHWC [B,Hout,Wout,Cout] */
    int32_t Cout;
    int32_t in_nchw, out_nchw;  /* boundary layouts (direct kernel only) */
    int32_t force_direct;       /* bit 0: use the direct (non-MFMA) kernel regardless of shape; bit 1: no Winograd;
                                   bit 2: 1x1 through the weight-streaming kernel; bit 3: 3x3 through the
                                   Winograd F(4x4,3x3) kernel where the shape qualifies (needs scratch for it);
                                   bit 4: 1x1, or 3x3 as an implicit GEMM, through the bf16-split kernel where the shape
                                   qualifies (scratch: + 1.5x weight); bit 5: the head (Cout <= 3) as a 1x1 GEMM onto its tap
                                   channels + gather; bit 6: the head as the one-pass kernel (tap channels stay in LDS; scratch >= 80 Cin
                                   floats), on the bf16 matrix pipe with the exact three-plane split; bit 7: with bit 6, on the
                                   fp32 MFMA */
    int64_t scratch_floats;     /* size of scratch_dev in floats; room for a second, fragment-ordered copy of a
                                   3x3 weight (+1 KB per 32 output channels) enables the weight-streaming kernel */
} dlpm_conv_args;

/* conv2d / conv1d(k=1) / Linear as implicit GEMM on the fp32 MFMA (or the direct kernel for
 * channel counts the MFMA tiling does not cover).  Takes OIHW weights and re-lays them out into a
 * scratch buffer (scratch_dev, >= 2*weight bytes) -- test/bring-up entry point; the UNet handle
 * pre-transforms its weights once.  Replaces F.conv2d / conv1d / linear: unet.py:64,96,143,157,168,213,215. */
int dlpm_conv2d_f32(const dlpm_conv_args *args, float *scratch_dev, dlpm_stream_t stream);

/* GroupNorm statistics -> per-(sample, channel) affine coefficients, optionally folding the
 * ResBlock scale/shift: y = GN(x)*(1+scale)+shift == x*A + B.  x is NHWC (virtual concat).
 * ss_dev: [B, ss_stride] rows holding [scale(C) | shift(C)] at column ss_offset, or NULL.
 * Replaces GroupNorm32 + the scale-shift of ResBlock._forward: nn.py:17-19, unet.py:187-191. */
int dlpm_groupnorm_coeffs_f32(const float *src0, const float *src1, int32_t C0, int32_t C1, int32_t B, int32_t HW,
                              int32_t groups, const float *gamma_dev, const float *beta_dev,
                              const float *ss_dev, int64_t ss_stride, int64_t ss_offset,
                              float *coefA_dev, float *coefB_dev, dlpm_stream_t stream);

/* QKVAttention over qkv[B,T,3C] (NHWC; channel layout head-major [head][q|k|v][C/heads] as produced
 * by the reference's reshape) -> out[B,T,C].  QK^T and PV on the fp32 MFMA, softmax in fp32.
 * Replaces QKVAttention.forward: unet.py:236-250. */
int dlpm_attention_f32(const float *qkv_dev, float *out_dev, int32_t B, int32_t T, int32_t C, int32_t heads,
                       dlpm_stream_t stream);

/* Whole blocks of SMALL images in one launch, activations resident in LDS, one workgroup per image (round 4; the UNet handle
 * takes them for its 64-channel blocks on 8x8 / 4x4 images under DLPM_CONV_AUTO).  Weights arrive in the reference's layouts
 * (OIHW / [O][I][1]) on the device and are re-laid out into scratch_dev on every call (test / bring-up entry points). */
typedef struct dlpm_resblock_args {
    const float *x0, *x1;        /* NHWC [B,H,W,C0], [B,H,W,C1] (virtual concat; x1 may be NULL); C0 + C1 = 64 or 128 */
    int32_t C0, C1, B, H, W;     /* H = W = 8 or 4 */
    const float *gn1_w, *gn1_b;  /* in_layers.0 */
    const float *conv1_w, *conv1_b;   /* in_layers.2: [64][C0+C1][3][3], [64] */
    const float *ss;             /* emb_layers output rows [B][ss_stride]: scale (64) | shift (64) -- unet.py:187-191 */
    int64_t ss_stride;           /* 0: one row for the whole batch */
    const float *gn2_w, *gn2_b;  /* out_layers.0 */
    const float *conv2_w, *conv2_b;   /* out_layers.3: [64][64][3][3], [64] */
    const float *skip_w, *skip_b;     /* skip_connection [64][C0+C1][1][1] -- required for 128 input channels, NULL for 64 */
    float *out;                  /* NHWC [B,H,W,64] */
    float *stats_out;            /* optional [B][64][2]: per image and channel (mean, centred sum of squares) of the output */
} dlpm_resblock_args;
/* ResBlock._forward with use_scale_shift_norm (unet.py:176-195): out = skip(x) + conv(silu(GN(conv(silu(GN(x)))) (1 + scale) + shift)).
 * scratch_floats >= 64 (C0+C1) 9 + 64 64 9 + 64 (C0+C1). */
int dlpm_resblock_small_f32(const dlpm_resblock_args *args, float *scratch_dev, int64_t scratch_floats, dlpm_stream_t stream);

/* Round 6: the same block with 32 output channels on 32x32 images (the first level of the MNIST-sized nets) in ONE launch, Winograd F(4x4,3x3)
 * inside (conv_wino4.hip: k_resblock_wino4_img).  Same argument struct: C0 + C1 in {32, 64, 96} (multiples of 8), H = W = 32, conv1_w [32][C0+C1][3][3],
 * conv2_w [32][32][3][3], ss rows scale (32) | shift (32), skip_w [32][C0+C1][1][1] required unless C0 + C1 = 32, out NHWC [B,32,32,32],
 * stats_out optional [B][4][32][2]: (mean, centred sum of squares) per 256-pixel quadrant and channel, the layout the convolution kernels emit.
 * H = W = 16 selects the second shape: 64 output channels on 16x16 images (k_resblock_wino4_img16; C0 + C1 in {32, 64, 96, 128}, multiples of 16,
 * conv1_w [64][C0+C1][3][3], conv2_w [64][64][3][3], ss rows scale (64) | shift (64), skip_w required unless C0 + C1 = 64, out [B,16,16,64],
 * stats_out optional [B][64][2] per image); its intermediate stays in LDS and it rounds differently from the separate launches (K split in two).
 * GroupNorm-1's coefficients are computed from the activations by a launch in front (the UNet plan hands the kernel its producers' statistics instead).
 * scratch_floats >= dlpm_resblock_img_scratch_floats(B, C0 + C1).  Bit-identical to the separate launches (convolution, coefficients, convolution). */
int64_t dlpm_resblock_img_scratch_floats(int64_t B, int32_t Cin);
int dlpm_resblock_img_f32(const dlpm_resblock_args *args, float *scratch_dev, int64_t scratch_floats, dlpm_stream_t stream);

typedef struct dlpm_attnblock_args {
    const float *x;              /* NHWC [B,H,W,64] */
    int32_t C, heads, B, H, W;   /* C = 64, heads = 4, H = W = 16 (round 6: k_attnblock16, one head at a time), 8 or 4 */
    const float *gn_w, *gn_b;    /* norm */
    const float *qkv_w, *qkv_b;  /* qkv: [192][64][1], [192] (head-major channel order, unet.py:224,243-244) */
    const float *proj_w, *proj_b;/* proj_out: [64][64][1], [64] */
    float *out;                  /* NHWC [B,H,W,64] */
    float *stats_out;            /* optional, as above */
} dlpm_attnblock_args;
/* AttentionBlock._forward (unet.py:220-228) + QKVAttention (:236-250): out = x + proj(softmax(q k^T / sqrt(ch)) v).
 * scratch_floats >= 192 64 + 64 64. */
int dlpm_attnblock_small_f32(const dlpm_attnblock_args *args, float *scratch_dev, int64_t scratch_floats, dlpm_stream_t stream);

/* emb_dev[B,dim] = [cos(t f_i) | sin(t f_i)]: timestep_embedding, nn.py:103-121. */
int dlpm_timestep_embedding_f32(const float *t_dev, float *emb_dev, int64_t B, int32_t dim, dlpm_stream_t stream);

/* layout helpers */
int dlpm_nchw_to_nhwc_f32(const float *src, float *dst, int32_t B, int32_t C, int32_t H, int32_t W, dlpm_stream_t stream);
int dlpm_nhwc_to_nchw_f32(const float *src, float *dst, int32_t B, int32_t C, int32_t H, int32_t W, dlpm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sampler: the T-step loop as a replayed hipGraph
 * ------------------------------------------------------------------------------------------ */

typedef struct dlpm_sampler_config {
    dlpm_unet *unet;            /* exactly one of unet / mlp is non-NULL */
    dlpm_mlp *mlp;
    int64_t B;                  /* samples on this GPU */
    int32_t C, H, W;            /* per-sample shape (toy data: C=1, H=1, W=nfeatures) */
    int32_t T;                  /* reverse steps */
    double alpha;
    double clamp_a, clamp_eps;  /* < 0: none */
    int32_t flags;              /* DLPM_UPD_DLIM | DLPM_UPD_CLIP | DLPM_SMP_NO_FUSED_MLP | DLPM_UPD_ELEMENTWISE | DLPM_SMP_LIM */
    float dlim_eta;
    uint64_t seed;
    int64_t sample_offset;      /* global index of this shard's first sample */
    int32_t use_graph;          /* > 0: capture this many consecutive reverse steps into one hipGraph and
                                   replay it (1 suits the UNets; launch-bound nets want tens)          */
    const float *g, *bg, *s, *bs; /* host schedule [T] each, or all NULL = dlpm_schedule_f32(T, alpha) */
    /* DLPM_SMP_LIM: T = steps + 1 and the schedule above is unused; host tables of dlpm_lim_tables_f32
     * (ts[T], the others [T-1]), or all NULL = computed by it */
    const float *lim_ts, *lim_tmp, *lim_cx, *lim_cs, *lim_cn;
    const float *in_scale;      /* optional host table [T]: the net sees x * in_scale[t] (input_scaling), NULL = x */
    int32_t mean_type;          /* dlpm_mean_type: what the net predicts (0 = eps); other types run dlpm_predict_f32 between
                                   the net and the update inside the captured step                                  */
} dlpm_sampler_config;

typedef struct dlpm_sampler dlpm_sampler;

int dlpm_sampler_create(const dlpm_sampler_config *cfg, dlpm_sampler **out);
/* New Philox key / shard offset for the next begin(); the captured graph stays valid (the key lives
 * in device memory). */
int dlpm_sampler_reseed(dlpm_sampler *s, uint64_t seed, int64_t sample_offset);
/* Draw A (Philox), build the tables, draw x_T; sets t = T-1.  p_sample_loop_progressive prologue:
 * GenerativeLevyProcess.py:306-315. */
int dlpm_sampler_begin(dlpm_sampler *s, dlpm_stream_t stream);
/* Same prologue with caller-provided noise (parity with the CPU reference on identical seeds):
 * A_dev[T,B] and xT_dev[B,C,H,W] are copied in; the tables are built from A.  (Non-isotropic: A_dev[T,B,D].
 * LIM: A_dev[T-1,B] holds the per-step a of gen_sas, row i for step i; xT_dev is x_0 = gen_eps.generate.) */
int dlpm_sampler_begin_injected(dlpm_sampler *s, const float *A_dev, const float *xT_dev, dlpm_stream_t stream);
/* Jump to step t (1 <= t <= T-1) of the trajectory with state x_dev[B,C,H,W] (NULL keeps the current state): the next
 * step maps x_t to x_{t-1} with the tables of the last begin*().  For teacher-forced parity checks against single
 * reference p_sample calls (GenerativeLevyProcess.py:225-239) and for resuming a trajectory. */
int dlpm_sampler_set_state(dlpm_sampler *s, const float *x_dev, int32_t t, dlpm_stream_t stream);
/* One reverse step with caller-provided N(0,1) noise z_dev[B,C,H,W] (eager, no graph). */
int dlpm_sampler_step_injected(dlpm_sampler *s, const float *z_dev, dlpm_stream_t stream);
/* Run `nsteps` reverse steps (model forward + fused update), stopping at t == 0.  Loop body of
 * p_sample_loop_progressive: GenerativeLevyProcess.py:317-330. */
int dlpm_sampler_steps(dlpm_sampler *s, int32_t nsteps, dlpm_stream_t stream);
/* Record every intermediate state on the device: hist_dev is a caller-owned [T,B,D] buffer (or NULL to stop).
 * dlpm_sampler_begin* stores x_T in row 0 and each later step stores its output in row T - t, i.e. the
 * `torch.stack(x_hist)` the reference returns with get_sample_history (GenerativeLevyProcess.py:274-288) --
 * written by the update kernel inside the captured graph, no per-step host round trip.  Call before begin. */
int dlpm_sampler_set_history(dlpm_sampler *s, float *hist_dev, dlpm_stream_t stream);

/* Copy the current state x[B,C,H,W] into a caller buffer (device to device, on `stream`). */
int dlpm_sampler_copy_state(dlpm_sampler *s, float *out_dev, dlpm_stream_t stream);
/* Device pointer to the current state x[B,C,H,W] and the current t (host copy). */
float *dlpm_sampler_state(dlpm_sampler *s);
int32_t dlpm_sampler_t(const dlpm_sampler *s);
/* Device pointers to the sampler's tables (for tests): which = 0 A, 1 c_eps, 2 c_noise, 3 eps. */
float *dlpm_sampler_table(dlpm_sampler *s, int which);
void dlpm_sampler_destroy(dlpm_sampler *s);

#ifdef __cplusplus
}
#endif
#endif /* DLPM_AMD_H */
