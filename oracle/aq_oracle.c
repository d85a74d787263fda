/*
 * CPU oracle for the adaptive-quantisation hot path.  TEST INFRASTRUCTURE ONLY:
 * nothing under quantized-cnn_amd/ may link, load or call this file; it is the
 * checker used by tests/, by __graft_entry__.smoke() and by bench.py's
 * cpu_baseline leg.
 *
 * It restates, in plain C, the algorithm of the reference's quantized layers:
 *   weight synthesis   adp_qtz_fan.py:563-571  (fix:668-673, hint:665-670,
 *                      hard codes: adp_qtz_fan_test.py:304-305)
 *   regulariser        adp_qtz_fan.py:737-754  (sum of W^2 per layer)
 *   9-tap convolution  adp_qtz_fan.py:574-601
 *   linear             adp_qtz_fan.py:633-648
 * and the gradients autograd derives from them (adp_qtz_fan.py:290).
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against the .npz fixtures under tests/golden, which tests/golden/make_golden.py produced
 * by running the reference's own classes.
 *
 * Row-local arithmetic (one weight = one row of L primary coefficients) is a
 * fully specified sequence of IEEE-754 binary32 operations (fmaf, *, +, /,
 * sqrtf, rintf, ldexpf, no contraction, no reassociation), so that the HIP
 * kernels, which perform the same sequence, can be compared BIT FOR BIT on W,
 * on the codes and on dp_c.  The sequence ("8x4 layout"):
 *   - a row is padded with zeros to LP = 4*G elements, G = 8, 16 or 32 lanes
 *     (the smallest with LP >= L); lane j owns elements 4j..4j+3;
 *   - a per-lane partial is a left-to-right chain over the lane's 4 elements;
 *   - lane partials are combined by an xor butterfly (masks 1,2,4,..,G/2),
 *     i.e. a balanced binary tree in natural lane order;
 *   - exp() is aqo_expf below (Cody-Waite reduction + degree-5 polynomial);
 *   - W = (sum_l e_l*q_l) * (1/sum_l e_l), s_l = e_l * (1/sum_l e_l), 1/||p|| = 1/sqrtf(sum p^2).
 * Sums that run ACROSS rows (sum W^2, dq_level, conv accumulations) are kept
 * in double here; the kernels reduce them in fp32 in a fixed, documented order
 * and are compared within a tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define AQO_MAX_LP 128

/* Hot entry points are compiled twice (hardware-FMA clone + baseline clone) and
 * picked at load time, so one .so runs on any x86-64 host; fmaf() is exact in
 * both, only its speed differs. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define AQO_CLONES __attribute__((target_clones("arch=haswell", "default")))
#else
#define AQO_CLONES
#endif
#define AQO_INLINE static inline __attribute__((always_inline))

/* ------------------------------------------------------------------ exp --- */
/* exp(x) for binary32, ~1 ulp: x = q*ln2 + s, |s| <= ln2/2, two-constant
 * reduction, polynomial in s, scale by 2^q.  Same operation sequence as the
 * device function aq::expf_spec in quantized-cnn_amd/csrc/aq_arith.h. */
AQO_INLINE float expf_spec(float x)
{
    const float R_LN2 = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2U = 0.693145751953125f;
    const float L2L = 1.428606765330187045e-06f;
    float qf = rintf(x * R_LN2);
    float s = fmaf(qf, -L2U, x);
    s = fmaf(qf, -L2L, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    /* Domain |x| <= 64 (callers bound |beta|; |p/||p||| <= 1): 2^q*u neither overflows nor
     * underflows there, so no range checks are part of the sequence.  NaN in -> NaN out. */
    const int qi = (qf == qf) ? (int)qf : 0;
    return ldexpf(u, qi);
}

float aqo_expf(float x) { return expf_spec(x); }

AQO_INLINE int lanes_for(int L) { return L <= 32 ? 8 : (L <= 64 ? 16 : 32); }

AQO_INLINE float tree(float *v, int G)
{
    float t[32];
    for (int m = 1; m < G; m <<= 1) {
        for (int j = 0; j < G; ++j) t[j] = v[j] + v[j ^ m];
        memcpy(v, t, sizeof(float) * G);
    }
    return v[0];
}

typedef struct {
    float pn[AQO_MAX_LP], s[AQO_MAX_LP];
    float rn, W;
    int idx;
} row_state;

/* forward of one row; shared by forward and backward */
AQO_INLINE void row_forward(const float *p, const float *q, int L, float beta, row_state *st)
{
    const int G = lanes_for(L), LP = 4 * G;
    float pp[AQO_MAX_LP], qq[AQO_MAX_LP], e[AQO_MAX_LP], v[32];
    for (int i = 0; i < LP; ++i) { pp[i] = i < L ? p[i] : 0.0f; qq[i] = i < L ? q[i] : 0.0f; }
    for (int j = 0; j < G; ++j) {
        const float *a = pp + 4 * j;
        float t = a[0] * a[0];
        t = fmaf(a[1], a[1], t); t = fmaf(a[2], a[2], t); t = fmaf(a[3], a[3], t);
        v[j] = t;
    }
    const float ss = tree(v, G);
    const float rn = 1.0f / sqrtf(ss);
    for (int i = 0; i < LP; ++i) {
        st->pn[i] = pp[i] * rn;
        e[i] = i < L ? expf_spec(beta * st->pn[i]) : 0.0f;
    }
    for (int j = 0; j < G; ++j) v[j] = ((e[4 * j] + e[4 * j + 1]) + e[4 * j + 2]) + e[4 * j + 3];
    const float den = tree(v, G);
    const float rden = 1.0f / den;
    for (int i = 0; i < LP; ++i) st->s[i] = e[i] * rden;
    for (int j = 0; j < G; ++j) {
        const float *a = e + 4 * j, *b = qq + 4 * j;
        float t = a[0] * b[0];
        t = fmaf(a[1], b[1], t); t = fmaf(a[2], b[2], t); t = fmaf(a[3], b[3], t);
        v[j] = t;
    }
    st->W = tree(v, G) * rden;
    st->rn = rn;
    /* code = arg-max over the softmax coefficients (adp_qtz_fan_test.py:304).  e -> e/den is
     * monotone, so the arg-max is taken over e (ties -> lowest index, as torch.max); when the
     * denominator is NaN every coefficient is NaN and torch.max returns index 0. */
    int bi = 0;
    if (den == den) {
        float best = e[0];
        for (int i = 1; i < L; ++i)
            if (e[i] > best) { best = e[i]; bi = i; }
    }
    st->idx = bi;
}

/* W[r] (soft), idx[r], W_hard[r] = q[idx[r]], *sumsq = sum_r W[r]^2 (double).
 * Any output pointer may be NULL. */
AQO_CLONES void aqo_synth_fwd(const float *p_c, const float *q, int64_t rows, int L, float beta,
                   float *W, int32_t *idx, float *W_hard, double *sumsq)
{
    double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        row_state st;
        row_forward(p_c + r * (int64_t)L, q, L, beta, &st);
        if (W) W[r] = st.W;
        if (idx) idx[r] = st.idx;
        if (W_hard) W_hard[r] = q[st.idx];
        acc += (double)st.W * (double)st.W;
    }
    if (sumsq) *sumsq = acc;
}

/* dp_c[r,:] and dq[:] for upstream dW[r]; wd = 2*l2_reg*(d loss / d reg) folds the
 * regulariser's gradient in: g = dW + wd*W.  dq accumulates in double. */
AQO_CLONES void aqo_synth_bwd(const float *p_c, const float *q, int64_t rows, int L, float beta,
                   const float *dW, float wd, float *dp_c, double *dq)
{
    const int G = lanes_for(L), LP = 4 * G;
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    double *part = (double *)calloc((size_t)nt * AQO_MAX_LP, sizeof(double));
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *mine = part + (size_t)tid * AQO_MAX_LP;
#pragma omp for schedule(static)
        for (int64_t r = 0; r < rows; ++r) {
            row_state st;
            float qq[AQO_MAX_LP], gpn[AQO_MAX_LP], v[32];
            row_forward(p_c + r * (int64_t)L, q, L, beta, &st);
            for (int i = 0; i < LP; ++i) qq[i] = i < L ? q[i] : 0.0f;
            const float g = fmaf(wd, st.W, dW ? dW[r] : 0.0f);
            const float dot = g * st.W;
            for (int i = 0; i < LP; ++i) {
                const float gs = g * qq[i];
                const float gz = st.s[i] * (gs - dot);
                gpn[i] = beta * gz;
            }
            for (int j = 0; j < G; ++j) {
                const float *a = st.pn + 4 * j, *b = gpn + 4 * j;
                float t = a[0] * b[0];
                t = fmaf(a[1], b[1], t); t = fmaf(a[2], b[2], t); t = fmaf(a[3], b[3], t);
                v[j] = t;
            }
            const float dpn = tree(v, G);
            for (int i = 0; i < L; ++i) {
                if (dp_c) dp_c[r * (int64_t)L + i] = fmaf(-st.pn[i], dpn, gpn[i]) * st.rn;
                mine[i] += (double)st.s[i] * (double)g;
            }
        }
    }
    if (dq)
        for (int i = 0; i < L; ++i) {
            double a = 0.0;
            for (int t = 0; t < nt; ++t) a += part[(size_t)t * AQO_MAX_LP + i];
            dq[i] = a;
        }
    free(part);
}

/* ------------------------------------------------------------ convolution --- */
/* y[n,co,oy,ox] = sum_{ci,dy,dx} W[co,ci,dy,dx] * x[n,ci,oy*s+dy-pad,ox*s+dx-pad]
 * (cross-correlation, zero padding, no bias), accumulated tap by tap in the
 * reference's loop order: width tap outer, height tap inner (adp_qtz_fan.py:581-599). */
void aqo_conv_fwd(const float *x, const float *W, int N, int Cin, int H, int Wd, int Cout,
                  int ks, int stride, int pad, float *y)
{
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (Wd + 2 * pad - ks) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co) {
            float *yo = y + ((int64_t)n * Cout + co) * Ho * Wo;
            for (int i = 0; i < Ho * Wo; ++i) yo[i] = 0.0f;
            for (int dx = 0; dx < ks; ++dx)
                for (int dy = 0; dy < ks; ++dy)
                    for (int oy = 0; oy < Ho; ++oy) {
                        const int iy = oy * stride + dy - pad;
                        if (iy < 0 || iy >= H) continue;
                        for (int ox = 0; ox < Wo; ++ox) {
                            const int ix = ox * stride + dx - pad;
                            if (ix < 0 || ix >= Wd) continue;
                            double a = 0.0;
                            for (int ci = 0; ci < Cin; ++ci)
                                a += (double)W[(((int64_t)co * Cin + ci) * ks + dy) * ks + dx] *
                                     (double)x[(((int64_t)n * Cin + ci) * H + iy) * Wd + ix];
                            yo[oy * Wo + ox] += (float)a;
                        }
                    }
        }
}

/* dx = d loss / d x  given dy */
void aqo_conv_dgrad(const float *dy, const float *W, int N, int Cin, int H, int Wd, int Cout,
                    int ks, int stride, int pad, float *dx)
{
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (Wd + 2 * pad - ks) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Cin; ++ci)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < Wd; ++ix) {
                    double a = 0.0;
                    for (int ky = 0; ky < ks; ++ky) {
                        const int ty = iy + pad - ky;
                        if (ty < 0 || ty % stride) continue;
                        const int oy = ty / stride;
                        if (oy >= Ho) continue;
                        for (int kx = 0; kx < ks; ++kx) {
                            const int tx = ix + pad - kx;
                            if (tx < 0 || tx % stride) continue;
                            const int ox = tx / stride;
                            if (ox >= Wo) continue;
                            for (int co = 0; co < Cout; ++co)
                                a += (double)W[(((int64_t)co * Cin + ci) * ks + ky) * ks + kx] *
                                     (double)dy[(((int64_t)n * Cout + co) * Ho + oy) * Wo + ox];
                        }
                    }
                    dx[(((int64_t)n * Cin + ci) * H + iy) * Wd + ix] = (float)a;
                }
}

/* dW = d loss / d W  given dy */
void aqo_conv_wgrad(const float *x, const float *dy, int N, int Cin, int H, int Wd, int Cout,
                    int ks, int stride, int pad, float *dW)
{
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (Wd + 2 * pad - ks) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < ks; ++ky)
                for (int kx = 0; kx < ks; ++kx) {
                    double a = 0.0;
                    for (int n = 0; n < N; ++n)
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * stride + ky - pad;
                            if (iy < 0 || iy >= H) continue;
                            const float *xr = x + (((int64_t)n * Cin + ci) * H + iy) * Wd;
                            const float *gr = dy + (((int64_t)n * Cout + co) * Ho + oy) * Wo;
                            for (int ox = 0; ox < Wo; ++ox) {
                                const int ix = ox * stride + kx - pad;
                                if (ix < 0 || ix >= Wd) continue;
                                a += (double)xr[ix] * (double)gr[ox];
                            }
                        }
                    dW[(((int64_t)co * Cin + ci) * ks + ky) * ks + kx] = (float)a;
                }
}

/* ----------------------------------------------------------------- linear --- */
/* y[n,o] = sum_i x[n,i] * W[i,o] + b[o]; W is (in, out) as in adp_qtz_fan.py:609,647 */
void aqo_linear_fwd(const float *x, const float *W, const float *b, int N, int In, int Out, float *y)
{
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < Out; ++o) {
            double a = 0.0;
            for (int i = 0; i < In; ++i) a += (double)x[n * In + i] * (double)W[i * Out + o];
            y[n * Out + o] = (float)a + (b ? b[o] : 0.0f);
        }
}

void aqo_linear_bwd(const float *x, const float *W, const float *dy, int N, int In, int Out,
                    float *dx, float *dW, float *db)
{
    if (dx)
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < In; ++i) {
                double a = 0.0;
                for (int o = 0; o < Out; ++o) a += (double)dy[n * Out + o] * (double)W[i * Out + o];
                dx[n * In + i] = (float)a;
            }
    if (dW)
        for (int i = 0; i < In; ++i)
            for (int o = 0; o < Out; ++o) {
                double a = 0.0;
                for (int n = 0; n < N; ++n) a += (double)x[n * In + i] * (double)dy[n * Out + o];
                dW[i * Out + o] = (float)a;
            }
    if (db)
        for (int o = 0; o < Out; ++o) {
            double a = 0.0;
            for (int n = 0; n < N; ++n) a += (double)dy[n * Out + o];
            db[o] = (float)a;
        }
}

/* ---- batch normalisation (+ReLU) ------------------------------------------------------------
 * `F.relu(self.bnX(x))` with nn.BatchNorm2d in training mode (adp_qtz_fan.py:661-663, 674-675, 690,
 * 726, 730): per channel c over (N, HW): mean, biased variance, y = gamma*(x-mean)/sqrt(var+eps)+beta,
 * then max(0, .).  Statistics are accumulated in double.  stats[c] = {mean, rstd, unbiased var}. */
void aqo_bn_relu_fwd(const float *x, const float *gamma, const float *beta, int N, int Cn, int HW,
                     float eps, int relu, float *y, double *stats)
{
    for (int c = 0; c < Cn; ++c) {
        double s = 0.0, m2 = 0.0;
        const double cnt = (double)N * HW;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) s += x[((size_t)n * Cn + c) * HW + i];
        const double mean = s / cnt;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                const double d = x[((size_t)n * Cn + c) * HW + i] - mean;
                m2 += d * d;
            }
        const double rstd = 1.0 / sqrt(m2 / cnt + (double)eps);
        stats[3 * c] = mean;
        stats[3 * c + 1] = rstd;
        stats[3 * c + 2] = cnt > 1 ? m2 / (cnt - 1) : m2;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                const size_t o = ((size_t)n * Cn + c) * HW + i;
                const double t = (x[o] - mean) * rstd * gamma[c] + beta[c];
                y[o] = (float)((relu && t < 0.0) ? 0.0 : t);
            }
    }
}

/* gradient of the above (autograd of batch_norm + relu): g = dy*[y>0],
 * dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) (+res), dgamma = sum g*xhat, dbeta = sum g */
void aqo_bn_relu_bwd(const float *dy, const float *x, const float *gamma, const float *beta, const float *res,
                     int N, int Cn, int HW, float eps, int relu, float *dx, float *dgamma, float *dbeta)
{
    for (int c = 0; c < Cn; ++c) {
        double s = 0.0, m2 = 0.0, s1 = 0.0, s2 = 0.0;
        const double cnt = (double)N * HW;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) s += x[((size_t)n * Cn + c) * HW + i];
        const double mean = s / cnt;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                const double d = x[((size_t)n * Cn + c) * HW + i] - mean;
                m2 += d * d;
            }
        const double rstd = 1.0 / sqrt(m2 / cnt + (double)eps);
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                const size_t o = ((size_t)n * Cn + c) * HW + i;
                const double xh = (x[o] - mean) * rstd;
                const double g = (!relu || xh * gamma[c] + beta[c] > 0.0) ? dy[o] : 0.0;
                s1 += g;
                s2 += g * xh;
            }
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                const size_t o = ((size_t)n * Cn + c) * HW + i;
                const double xh = (x[o] - mean) * rstd;
                const double g = (!relu || xh * gamma[c] + beta[c] > 0.0) ? dy[o] : 0.0;
                dx[o] = (float)(gamma[c] * rstd * (g - s1 / cnt - xh * s2 / cnt) + (res ? res[o] : 0.0));
            }
    }
}

/* torch.optim.Adam update (defaults: no weight decay, no amsgrad) as the reference applies it to every
 * parameter group (adp_qtz_fan.py:208-213, 291-293), for step number t (1-based), fp32 state:
 *   m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps) */
void aqo_adam_step(float *p, const float *g, float *m, float *v, long long n, int t, double lr, double b1, double b2, double eps)
{
    const double bc1 = 1.0 - pow(b1, t), bc2s = sqrt(1.0 - pow(b2, t));
    for (long long i = 0; i < n; ++i) {
        const double mm = m[i] + (1.0 - b1) * ((double)g[i] - m[i]);
        const double vv = b2 * v[i] + (1.0 - b2) * (double)g[i] * g[i];
        m[i] = (float)mm;
        v[i] = (float)vv;
        p[i] = (float)(p[i] - lr / bc1 * ((double)m[i] / (sqrt((double)v[i]) / bc2s + eps)));
    }
}

int aqo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void aqo_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------ ternary / power-of-two quantisers ---
 * PARITY UNPINNED - no reference implementation.  The reference cites these methods (README.md:3-29) but its
 * scripts implement none of them, and the linked repositories are not available here; the functions below
 * restate the published algorithms:
 *   TWN  Li, Zhang, Liu, "Ternary Weight Networks" (2016), eq. 4-6: delta = 0.7 * E|w|, alpha = mean of |w| above delta
 *   TTQ  Zhu, Han, Mao, Dally, "Trained Ternary Quantization" (2017), eq. 6-8 (scale gradients as autograd of the forward
 *        formula gives them: dWn carries the minus sign of w~ = -Wn)
 *   INQ  Zhou et al., "Incremental Network Quantization" (2017), eq. 3-4 (level rule) and eq. 8 (masked update) */
void aqo_tern_stats(const float *w, int64_t n, double *mean_abs, double *max_abs)
{
    double s = 0.0, m = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double a = fabs((double)w[i]);
        s += a;
        if (a > m) m = a;
    }
    *mean_abs = n ? s / (double)n : 0.0;
    *max_abs = m;
}

/* codes for a given threshold: 0 where |w| <= delta, 1 where w > delta, 2 where w < -delta; sums for alpha */
void aqo_tern_codes(const float *w, int64_t n, float delta, uint8_t *codes, double *sum_abs_nz, int64_t *npos, int64_t *nneg)
{
    double s = 0.0;
    int64_t p = 0, q = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float x = w[i];
        const uint8_t c = x > delta ? 1 : (x < -delta ? 2 : 0);
        if (codes) codes[i] = c;
        if (c) s += fabs((double)x);
        p += c == 1;
        q += c == 2;
    }
    *sum_abs_nz = s; *npos = p; *nneg = q;
}

/* mode 0 TWN: dw = dwt.  mode 1 TTQ: dw scaled by region, dwp / dwn.  mode 2 INQ: dw = mask ? dwt : 0 */
void aqo_tern_bwd(const float *w, const float *dwt, int64_t n, int mode, float delta, float pos, float neg, const uint8_t *mask,
                  float *dw, double *dwp, double *dwn)
{
    double sp = 0.0, sn = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float d = dwt[i];
        if (mode == 1) {
            if (w[i] > delta) { sp += (double)d; d = d * pos; }
            else if (w[i] < -delta) { sn += (double)d; d = d * neg; }
        } else if (mode == 2) {
            d = mask[i] ? d : 0.0f;
        }
        dw[i] = d;
    }
    if (dwp) *dwp = sp;
    if (dwn) *dwn = -sn;
}

/* INQ level of a magnitude a > 0 for exponents n2..n1: the power of two beta = 2^k with (alpha + beta) / 2 <= a < 3 beta / 2,
 * alpha = beta / 2 (alpha = 0 for k = n2); 0 below 2^(n2-1); 2^n1 at and above 1.5 * 2^n1.  Returns k, or INT_MIN for 0. */
static int inq_level(float a, int n1, int n2)
{
    if (!(a > 0.0f)) return -2147483647 - 1;
    int e;
    const float m = frexpf(a, &e);  /* a = m * 2^e, m in [0.5, 1) */
    int k = (m >= 0.75f) ? e : e - 1;   /* a in [0.75 * 2^k, 1.5 * 2^k) */
    if (k < n2) k = (a >= ldexpf(1.0f, n2 - 1)) ? n2 : (-2147483647 - 1);
    if (k > n1) k = n1;
    return k;
}

void aqo_inq_partition(float *w, uint8_t *mask, uint8_t *codes, int64_t n, float tau, int n1, int n2)
{
    for (int64_t i = 0; i < n; ++i) {
        float x = w[i];
        if (mask[i] && fabsf(x) >= tau) {
            const int k = inq_level(fabsf(x), n1, n2);
            const float qv = k == (-2147483647 - 1) ? 0.0f : ldexpf(1.0f, k);
            x = x < 0.0f ? -qv : qv;
            w[i] = x;
            mask[i] = 0;
        }
        if (codes) {
            unsigned c = 255;
            if (!mask[i]) {
                if (x == 0.0f) c = 0;
                else {
                    int e;
                    const float m = frexpf(fabsf(x), &e);
                    const int k = e - 1;
                    c = (m == 0.5f && k >= n2 && k <= n1) ? 1u + 2u * (unsigned)(k - n2) + (x < 0.0f ? 1u : 0u) : 254u;
                }
            }
            codes[i] = (uint8_t)c;
        }
    }
}
