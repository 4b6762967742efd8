/*
 * oracle/rng.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * Plain-C restatement of the two CPU random streams the reference's sampling
 * loop consumes:
 *
 *   N: numpy's global RandomState (MT19937), reached through
 *      scipy.stats.levy_stable.rvs at bem/datasets/Distributions.py:45,48.
 *      scipy is a third-party dependency that is NOT under /root/reference; the
 *      reference pins no version (bem/requirements.txt:10), the container has
 *      scipy 1.15.3 / numpy 2.2.6.  Restated here from the published algorithm:
 *      Chambers-Mallows-Stuck in Nolan's S1 form (scipy _rvs_Z1, beta != 0,
 *      alpha != 1 branch), uniforms = legacy random_sample, exponentials =
 *      legacy standard_exponential (-log(1-U)); all n uniforms are drawn before
 *      all n exponentials.
 *   P: torch's default CPU generator (MT19937), reached through torch.randn /
 *      randn_like at bem/datasets/Distributions.py:65 and
 *      dlpm/methods/GenerativeLevyProcess.py:236.  torch 2.10: n >= 16 fills
 *      24-bit uniforms then Box-Muller over blocks of 16 (tail block redrawn);
 *      n < 16 uses the scalar double path with a cached second normal.
 *
 * Pinned by tests/golden/f2_skewed_levy.npz and f2_randn.npz, which were
 * produced by the imported reference / torch (tools/make_fixtures.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t key[MT_N];
    int32_t pos;
    /* torch CPUGeneratorImpl keeps one cached double normal for the scalar path */
    int32_t has_cached;
    double cached;
} orc_mt;

void orc_mt_seed(orc_mt *st, uint32_t seed)
{
    for (int i = 0; i < MT_N; i++) {
        st->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)(i + 1);
    }
    st->pos = MT_N;
    st->has_cached = 0;
    st->cached = 0.0;
}

static void mt_refill(orc_mt *st)
{
    uint32_t *k = st->key;
    for (int i = 0; i < MT_N; i++) {
        uint32_t y = (k[i] & 0x80000000u) | (k[(i + 1) % MT_N] & 0x7fffffffu);
        uint32_t v = k[(i + MT_M) % MT_N] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        k[i] = v;
    }
    st->pos = 0;
}

uint32_t orc_mt_next(orc_mt *st)
{
    if (st->pos >= MT_N) mt_refill(st);
    uint32_t y = st->key[st->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* numpy legacy double: 27 + 26 bits from two draws */
static double np_double(orc_mt *st)
{
    uint32_t a = orc_mt_next(st) >> 5, b = orc_mt_next(st) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

void orc_np_random_sample(orc_mt *st, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = np_double(st);
}

void orc_np_standard_exponential(orc_mt *st, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = -log(1.0 - np_double(st));
}

/* one CMS transform; a = alpha/2 (stability index of the subordinator), beta = 1 */
static double cms_skewed(double a, double U, double W)
{
    const double pi = 3.141592653589793;
    double TH = U * pi + (-pi / 2.0);
    double val0 = tan(pi * a / 2.0); /* beta = 1 */
    double th0 = atan(val0) / a;
    double aTH = a * TH;
    double cosTH = cos(TH), tanTH = tan(TH);
    double val3 = W / (cosTH / tan(a * (th0 + TH)) + sin(TH));
    double inner = (cos(aTH) + sin(aTH) * tanTH - val0 * (sin(aTH) - cos(aTH) * tanTH)) / W;
    return val3 * pow(inner, 1.0 / a);
}

/* scipy.stats.levy_stable.rvs(alpha/2, 1, loc=0, scale=2*cos(pi*alpha/4)**(2/alpha), size=n)
 * as called by gen_skewed_levy (Distributions.py:45).  scratch: 2*n doubles. */
void orc_skewed_levy(orc_mt *st, double alpha, int64_t n, double *out, double *scratch)
{
    const double pi = 3.141592653589793;
    double *U = scratch, *W = scratch + n;
    orc_np_random_sample(st, n, U);
    orc_np_standard_exponential(st, n, W);
    double scale = 2.0 * pow(cos(pi * alpha / 4.0), 2.0 / alpha);
    for (int64_t i = 0; i < n; i++) out[i] = cms_skewed(alpha / 2.0, U[i], W[i]) * scale + 0.0;
}

void orc_cms_from_uw(double alpha, int64_t n, const double *U, const double *W, double *out)
{
    const double pi = 3.141592653589793;
    double scale = 2.0 * pow(cos(pi * alpha / 4.0), 2.0 / alpha);
    for (int64_t i = 0; i < n; i++) out[i] = cms_skewed(alpha / 2.0, U[i], W[i]) * scale + 0.0;
}

/* ---- torch CPU normal_ (float32, contiguous) -------------------------------- */
static void box_muller_16(float *d)
{
    for (int j = 0; j < 8; j++) {
        float u1 = 1.0f - d[j];
        float u2 = d[j + 8];
        float radius = sqrtf(-2.0f * logf(u1));
        /* torch's AVX2 kernel (normal_fill_16_AVX2) multiplies by an fp32 2*pi; the scalar
         * fallback would form the product in double -- the fixtures pin the fp32 form */
        float theta = 6.28318530717958647692f * u2;
        d[j] = radius * cosf(theta);
        d[j + 8] = radius * sinf(theta);
    }
}

static double torch_u53(orc_mt *st)
{
    uint64_t hi = orc_mt_next(st), lo = orc_mt_next(st);
    uint64_t r = (hi << 32) | lo;
    return (double)(r & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
}

void orc_torch_randn(orc_mt *st, int64_t n, float *out)
{
    if (n >= 16) {
        for (int64_t i = 0; i < n; i++)
            out[i] = (float)(orc_mt_next(st) & 0xffffffu) * (1.0f / 16777216.0f);
        for (int64_t i = 0; i + 16 <= n; i += 16) box_muller_16(out + i);
        if (n % 16) {
            float *d = out + n - 16;
            for (int i = 0; i < 16; i++)
                d[i] = (float)(orc_mt_next(st) & 0xffffffu) * (1.0f / 16777216.0f);
            box_muller_16(d);
        }
        return;
    }
    for (int64_t i = 0; i < n; i++) {
        if (st->has_cached) {
            out[i] = (float)st->cached;
            st->has_cached = 0;
            continue;
        }
        double u1 = torch_u53(st), u2 = torch_u53(st);
        double r = sqrt(-2.0 * log1p(-u2));
        double theta = 2.0 * 3.14159265358979323846 * u1;
        st->cached = r * sin(theta);
        st->has_cached = 1;
        out[i] = (float)(r * cos(theta));
    }
}
