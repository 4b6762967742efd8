// host.cpp -- host-side pieces of the sampling path: error channel, noise schedule and the
// reference-compatible MT19937 streams used for "identical seeds" parity with the CPU reference.
#include <mutex>
#include <set>
#include <utility>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace dlpm {
static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
}  // namespace dlpm

namespace dlpm {
namespace {
struct ProfRec {
    std::string name;
    double flops, bytes;
    hipEvent_t e0, e1;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
}  // namespace
bool prof_enabled() { return g_prof_on; }

int ensure_dynamic_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    DLPM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(std::make_pair(dev, kernel))) return DLPM_OK;
    DLPM_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert(std::make_pair(dev, kernel));
    return DLPM_OK;
}
#ifdef DLPM_PHASE_TIMING
unsigned long long *phase_buffer() {
    static unsigned long long *buf = nullptr;
    if (!buf && hipMalloc(&buf, 32 * sizeof(unsigned long long)) == hipSuccess) hipMemset(buf, 0, 32 * sizeof(unsigned long long));
    return buf;
}
#endif
bool prof_detail() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("DLPM_PROF_DETAIL"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
ProfScope::ProfScope(const char *n, double fl, double by, hipStream_t s) : name(n), flops(fl), bytes(by), st(s) {
    if (!g_prof_on) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
    (void)hipEventRecord(e0, st);
}
ProfScope::~ProfScope() {
    if (!e0 || !e1) return;
    (void)hipEventRecord(e1, st);
    g_prof.push_back({name, flops, bytes, e0, e1});
}
}  // namespace dlpm

using namespace dlpm;

extern "C" int dlpm_prof_enable(int on) {
    g_prof_on = on != 0;
    return DLPM_OK;
}

extern "C" int dlpm_prof_report(char *buf, int64_t n) {
    DLPM_CHECK_ARG(buf && n > 0, "dlpm_prof_report: bad buffer");
    DLPM_HIP(hipDeviceSynchronize());
    struct Agg { std::string name; int64_t launches = 0; double ms = 0, flops = 0, bytes = 0; };
    std::vector<Agg> aggs;
    for (auto &r : g_prof) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
        Agg *a = nullptr;
        for (auto &x : aggs) if (x.name == r.name) a = &x;
        if (!a) { aggs.push_back(Agg()); a = &aggs.back(); a->name = r.name; }
        a->launches++; a->ms += ms; a->flops += r.flops; a->bytes += r.bytes;
    }
    g_prof.clear();
    std::string out;
    char line[256];
    for (auto &a : aggs) {
        snprintf(line, sizeof(line), "%s %lld %.6f %.6e %.6e\n", a.name.c_str(), (long long)a.launches, a.ms, a.flops, a.bytes);
        out += line;
    }
    if ((int64_t)out.size() + 1 > n) {
        set_error("dlpm_prof_report: buffer too small (%zu needed)", out.size() + 1);
        return DLPM_ERR_NOMEM;
    }
    memcpy(buf, out.c_str(), out.size() + 1);
    return DLPM_OK;
}

extern "C" const char *dlpm_last_error(void) { return g_err.c_str(); }
extern "C" int dlpm_abi_version(void) { return 6; }

// ---------------------------------------------------------------------------------------------
// Schedule.  The reference builds it with fp32 torch ops (dlpm.py:114-156); several entries are
// differences of nearly equal fp32 numbers (beta_t = 1 - abar_t/abar_{t-1}), so the fp32 op
// ORDER is part of the contract: every intermediate below is rounded to fp32 exactly where the
// reference rounds.  Transcendentals are evaluated in double and rounded once (correctly
// rounded results, which is what torch's vector kernels return for all but a few inputs).
// ---------------------------------------------------------------------------------------------
static inline float f32_pow(float x, float p) { return (float)std::pow((double)x, (double)p); }

extern "C" int dlpm_schedule_f32(int T, double alpha, float *g, float *bg, float *s, float *bs) {
    DLPM_CHECK_ARG(T >= 2, "dlpm_schedule_f32: T must be >= 2, got %d", T);
    DLPM_CHECK_ARG(alpha > 0.0 && alpha <= 2.0, "dlpm_schedule_f32: alpha must be in (0,2], got %g", alpha);
    DLPM_CHECK_ARG(g && bg && s && bs, "dlpm_schedule_f32: null output");
    const float fT = (float)T;
    const float half_pi_num = 3.14159274101257324f;  // float(torch.pi)
    std::vector<float> abar(T);
    float f0 = 0.f;
    for (int i = 0; i < T; i++) {
        float u = (float)i / fT;         // timesteps / diffusion_steps
        u = u + 0.008f;                  // + s
        u = u / 1.008f;                  // / (1 + s)
        u = u * half_pi_num;             // * torch.pi
        u = u / 2.0f;                    // / 2
        float c = (float)std::cos((double)u);
        float f = c * c;                 // ** 2
        if (i == 0) f0 = f;
        abar[i] = f / f0;
    }
    const float inv_a = (float)(1.0 / alpha), fa = (float)alpha;
    float run = 1.f;
    for (int i = 0; i < T; i++) {
        float prev = abar[i == 0 ? 0 : i - 1];
        float beta = 1.0f - abar[i] / prev;
        float al = 1.0f - beta;
        g[i] = f32_pow(al, inv_a);
        run = (i == 0) ? g[0] : run * g[i];  // cumprod
        bg[i] = run;
        s[i] = f32_pow(1.0f - f32_pow(g[i], fa), inv_a);
        bs[i] = f32_pow(1.0f - f32_pow(bg[i], fa), inv_a);
    }
    return DLPM_OK;
}

extern "C" int dlpm_schedule_exploding_f32(int T, double alpha, float *g, float *bg, float *s, float *bs) {
    DLPM_CHECK_ARG(T >= 2, "dlpm_schedule_exploding_f32: T must be >= 2, got %d", T);
    DLPM_CHECK_ARG(alpha > 0.0 && alpha <= 2.0, "dlpm_schedule_exploding_f32: alpha must be in (0,2], got %g", alpha);
    DLPM_CHECK_ARG(g && bg && s && bs, "dlpm_schedule_exploding_f32: null output");
    const double smin = 0.002, smax = 80.0, rho = 7.0;
    const double lo = std::pow(smin, 1.0 / rho), hi = std::pow(smax, 1.0 / rho);
    double run = 0.0;   // sum of sigmas^alpha so far
    for (int i = 0; i < T; i++) {
        g[i] = 1.0f;
        bg[i] = 1.0f;
        const double b = std::pow(lo + ((double)i / (double)(T - 1)) * (hi - lo), rho);
        bs[i] = (float)b;
        const double ba = std::pow((double)bs[i], alpha);
        const double sa = i == 0 ? ba : ba - run;
        run += sa;
        s[i] = (float)std::pow(sa, 1.0 / alpha);
    }
    return DLPM_OK;
}

// ---------------------------------------------------------------------------------------------
// MT19937 with numpy / torch draw semantics
// ---------------------------------------------------------------------------------------------
namespace {
struct Mt {
    dlpm_mt19937 *st;
    explicit Mt(dlpm_mt19937 *s) : st(s) {}
    void twist() {
        uint32_t *k = st->key;
        for (int i = 0; i < 624; i++) {
            uint32_t y = (k[i] & 0x80000000u) | (k[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = k[(i + 397) % 624] ^ (y >> 1);
            k[i] = (y & 1u) ? (v ^ 0x9908b0dfu) : v;
        }
        st->pos = 0;
    }
    uint32_t u32() {
        if (st->pos >= 624) twist();
        uint32_t y = st->key[st->pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        return y ^ (y >> 18);
    }
    // numpy legacy random_sample: 53 bits from two draws (27 | 26)
    double np_uniform() {
        uint32_t a = u32() >> 5, b = u32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // torch uniform_real_distribution<double>: low 53 bits of (hi << 32 | lo)
    double torch_uniform53() {
        uint64_t hi = u32(), lo = u32();
        return (double)(((hi << 32) | lo) & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
    }
    float torch_uniform24() { return (float)(u32() & 0xffffffu) * (1.0f / 16777216.0f); }
};
}  // namespace

extern "C" int dlpm_mt19937_seed(dlpm_mt19937 *st, uint32_t seed) {
    DLPM_CHECK_ARG(st, "dlpm_mt19937_seed: null state");
    for (int i = 0; i < 624; i++) {
        st->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)(i + 1);
    }
    st->pos = 624;
    st->has_cached = 0;
    st->cached = 0.0;
    return DLPM_OK;
}

extern "C" int dlpm_skewed_levy_host_f32(dlpm_mt19937 *st, double alpha, int64_t n, double clamp_a, float *out) {
    DLPM_CHECK_ARG(st && out, "dlpm_skewed_levy_host_f32: null pointer");
    DLPM_CHECK_ARG(alpha > 0.0 && alpha <= 2.0, "Wrong value of alpha (%g) for skewed levy r.v generation", alpha);
    DLPM_CHECK_ARG(n >= 0, "dlpm_skewed_levy_host_f32: negative n");
    if (alpha == 2.0) {  // Distributions.py:40-42: constant, no draws, no clamp
        for (int64_t i = 0; i < n; i++) out[i] = 2.0f;
        return DLPM_OK;
    }
    Mt mt(st);
    std::vector<double> U(n), W(n);
    for (int64_t i = 0; i < n; i++) U[i] = mt.np_uniform();
    for (int64_t i = 0; i < n; i++) W[i] = -std::log(1.0 - mt.np_uniform());
    const double pi = 3.141592653589793;
    const double a = alpha / 2.0;
    const double zeta = std::tan(pi * a / 2.0);       // beta * tan(pi a / 2), beta = 1
    const double th0 = std::atan(zeta) / a;
    const double scale = 2.0 * std::pow(std::cos(pi * alpha / 4.0), 2.0 / alpha);
    for (int64_t i = 0; i < n; i++) {
        double th = U[i] * pi + (-pi / 2.0);
        double ath = a * th, c = std::cos(th), tg = std::tan(th);
        double lead = W[i] / (c / std::tan(a * (th0 + th)) + std::sin(th));
        double core = (std::cos(ath) + std::sin(ath) * tg - zeta * (std::sin(ath) - std::cos(ath) * tg)) / W[i];
        float v = (float)(lead * std::pow(core, 1.0 / a) * scale + 0.0);
        if (clamp_a >= 0.0) v = std::fmin(std::fmax(v, 0.0f), (float)clamp_a);
        out[i] = v;
    }
    return DLPM_OK;
}

static void box_muller_block16(float *d) {
    for (int j = 0; j < 8; j++) {
        float u1 = 1.0f - d[j], u2 = d[j + 8];
        float r = std::sqrt(-2.0f * std::log(u1));
        float th = 6.28318530717958647692f * u2;  // fp32 product, as torch's AVX2 kernel
        d[j] = r * std::cos(th);
        d[j + 8] = r * std::sin(th);
    }
}

extern "C" int dlpm_randn_host_f32(dlpm_mt19937 *st, int64_t n, float *out) {
    DLPM_CHECK_ARG(st && out, "dlpm_randn_host_f32: null pointer");
    DLPM_CHECK_ARG(n >= 0, "dlpm_randn_host_f32: negative n");
    Mt mt(st);
    if (n >= 16) {
        for (int64_t i = 0; i < n; i++) out[i] = mt.torch_uniform24();
        for (int64_t i = 0; i + 16 <= n; i += 16) box_muller_block16(out + i);
        if (n % 16) {  // the ragged tail is redrawn as one full block ending at n
            float *d = out + n - 16;
            for (int i = 0; i < 16; i++) d[i] = mt.torch_uniform24();
            box_muller_block16(d);
        }
        return DLPM_OK;
    }
    for (int64_t i = 0; i < n; i++) {
        if (st->has_cached) {
            out[i] = (float)st->cached;
            st->has_cached = 0;
            continue;
        }
        double u1 = mt.torch_uniform53(), u2 = mt.torch_uniform53();
        double r = std::sqrt(-2.0 * std::log1p(-u2)), th = 2.0 * 3.14159265358979323846 * u1;
        st->cached = r * std::sin(th);
        st->has_cached = 1;
        out[i] = (float)(r * std::cos(th));
    }
    return DLPM_OK;
}

// ---------------------------------------------------------------------------------------------- LIM tables
// VPSDE, cosine schedule (dlpm/methods/LIM/functions/sde.py:5-49) and the per-step coefficients of
// ode_score_update / sde_score_update (dlpm/methods/LIM/functions/sampler.py:85-152).
extern "C" int dlpm_lim_tables_f32(double alpha, int32_t steps, int32_t ode, float *ts, float *tmp, float *cx, float *cs,
                                   float *cn) {
    DLPM_CHECK_ARG(ts && tmp && cx && cs && cn, "dlpm_lim_tables_f32: null pointer");
    DLPM_CHECK_ARG(steps >= 1, "dlpm_lim_tables_f32: steps must be >= 1, got %d", steps);
    DLPM_CHECK_ARG(alpha > 0.0 && alpha <= 2.0, "Wrong value of alpha (%g) for skewed levy r.v generation", alpha);
    const double pi = 3.141592653589793, cs_ = 0.008, Tend = 0.9946, eps = 1e-5;
    const double log_a0 = std::log(std::cos(cs_ / (1.0 + cs_) * pi / 2.0));
    auto logmean = [&](double t) { return std::log(std::cos((t + cs_) / (1.0 + cs_) * pi / 2.0)) - log_a0; };
    auto stdv = [&](double t) { return std::pow(1.0 - std::exp(logmean(t) * alpha), 1.0 / alpha); };
    auto beta = [&](double t) { return pi / 2.0 * alpha / (cs_ + 1.0) * std::tan((t + cs_) / (1.0 + cs_) * pi / 2.0); };
    // torch.linspace(T, eps, steps + 1) in fp32: start + i*step for the first half, end - (n-1-i)*step for the second
    const int n = steps + 1;
    const float fstart = (float)Tend, fend = (float)eps;
    const float fstep = (fend - fstart) / (float)(n - 1);
    for (int i = 0; i < n; i++) ts[i] = i < n / 2 ? fstart + fstep * (float)i : fend - fstep * (float)(n - 1 - i);
    for (int i = 0; i < steps; i++) {
        const double s = ts[i], t = ts[i + 1];
        const double beta_step = beta(s) * (s - t);
        if (alpha == 2.0) {
            tmp[i] = (float)std::pow(stdv(s) + 1e-5, -(alpha - 1.0));
            cx[i] = (float)(1.0 + beta_step / alpha);
            cs[i] = (float)(ode ? beta_step / 2.0 : beta_step);
            cn[i] = ode ? 0.0f : (float)std::pow(beta_step, 1.0 / alpha);
        } else {
            tmp[i] = (float)std::pow(stdv(s), -(alpha - 1.0));
            const double a = std::exp(logmean(t) - logmean(s));
            cx[i] = (float)a;
            cs[i] = (float)(ode ? -alpha * (1.0 - a) : alpha * alpha * (a - 1.0));
            cn[i] = ode ? 0.0f : (float)std::pow(std::pow(a, alpha) - 1.0, 1.0 / alpha);
        }
    }
    return DLPM_OK;
}

#ifdef DLPM_PHASE_TIMING
// developer builds only (not in include/dlpm_amd.h): read and clear the phase counters
extern "C" int dlpm_debug_phases(unsigned long long *out32) {
    unsigned long long *b = dlpm::phase_buffer();
    if (!b) return DLPM_ERR_HIP;
    DLPM_HIP(hipDeviceSynchronize());
    DLPM_HIP(hipMemcpy(out32, b, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    DLPM_HIP(hipMemset(b, 0, 32 * sizeof(unsigned long long)));
    return DLPM_OK;
}
#endif
