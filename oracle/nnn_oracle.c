/*
 * nnn_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see nnn_oracle.h).
 *
 * Scalar f32 restatement of jneem/nnnoiseless v0.5.1 `DenoiseState::process_frame`
 * (src/denoise.rs:95-116) and everything it calls.  Build with
 *     gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math
 * so that every a*b+c rounds twice exactly like the (non-contracting) Rust reference;
 * integer results (pitch indices) depend on it.
 *
 * Citations: "ref: <file>:<lines>" are into the reference tree.
 */
#include "nnn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define FRAME_SIZE 480
#define WINDOW_SIZE 960
#define FREQ_SIZE 481
#define NB_BANDS 22
#define CEPS_MEM 8
#define NB_DELTA_CEPS 6
#define NB_FEATURES 42
#define PITCH_MIN_PERIOD 60
#define PITCH_MAX_PERIOD 768
#define PITCH_FRAME_SIZE 960
#define PITCH_BUF_SIZE 1728
#define MAX_NEURONS 128

/* FFT working precision.  Default double: the oracle is then the best available statement
 * of "the" spectrum, and any f32 FFT (rustfft's or the GPU's) sits within f32 rounding of it.
 * -DNNNO_FFT_F32 gives an all-f32 FFT like the reference's, used for CPU-baseline timing. */
#ifdef NNNO_FFT_F32
typedef float fftr;
#else
typedef double fftr;
#endif
typedef struct { fftr re, im; } cpx;

/* ref: src/lib.rs:55-58 */
static const int EBAND_5MS[NB_BANDS] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12,
                                        14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};

/* ------------------------------------------------------------------------------------ */
/* Tables                                                                                 */
/* ------------------------------------------------------------------------------------ */
static struct {
    int ready;
    float window[WINDOW_SIZE];
    float dct_table[NB_BANDS * NB_BANDS];
    float wnorm;
    float tansig[201];
    cpx tw480[480]; /* exp(-2 pi i k / 480) */
    cpx tw960[481]; /* exp(-2 pi i k / 960), k = 0..480 */
} T;

static void init_tables(void)
{
    if (T.ready) return;
    const double pi = 3.14159265358979323846;
    /* ref: src/lib.rs:107-116 -- window in f64, stored f32; wnorm = 1 / sum(w^2) in f32 */
    for (int i = 0; i < FRAME_SIZE; i++) {
        double s = sin(0.5 * pi * ((double)i + 0.5) / (double)FRAME_SIZE);
        float w = (float)sin(0.5 * pi * s * s);
        T.window[i] = w;
        T.window[WINDOW_SIZE - i - 1] = w;
    }
    float acc = 0.0f;
    for (int i = 0; i < WINDOW_SIZE; i++) acc += T.window[i] * T.window[i];
    T.wnorm = 1.0f / acc;
    /* ref: src/lib.rs:118-127 */
    for (int i = 0; i < NB_BANDS; i++)
        for (int j = 0; j < NB_BANDS; j++) {
            float v = (float)cos(((double)i + 0.5) * (double)j * pi / (double)NB_BANDS);
            if (j == 0) v *= sqrtf(0.5f);
            T.dct_table[i * NB_BANDS + j] = v;
        }
    /* ref: src/util.rs:3-27 -- tanh(0.04 i) printed to six decimals.  Three upstream entries
     * are not the nearest six-decimal value of tanh; they are kept as upstream has them. */
    for (int i = 0; i <= 200; i++)
        T.tansig[i] = (float)(floor(tanh(0.04 * (double)i) * 1e6 + 0.5) / 1e6);
    T.tansig[70] = 0.992631f;
    T.tansig[170] = 0.999997f;
    T.tansig[190] = 1.000000f;
    for (int k = 0; k < 480; k++) {
        T.tw480[k].re = (fftr)cos(-2.0 * pi * k / 480.0);
        T.tw480[k].im = (fftr)sin(-2.0 * pi * k / 480.0);
    }
    for (int k = 0; k <= 480; k++) {
        T.tw960[k].re = (fftr)cos(-2.0 * pi * k / 960.0);
        T.tw960[k].im = (fftr)sin(-2.0 * pi * k / 960.0);
    }
    T.ready = 1;
}

void nnno_get_tables(float *window960, float *dct22x22, float *wnorm, float *tansig201)
{
    init_tables();
    if (window960) memcpy(window960, T.window, sizeof(T.window));
    if (dct22x22) memcpy(dct22x22, T.dct_table, sizeof(T.dct_table));
    if (wnorm) *wnorm = T.wnorm;
    if (tansig201) memcpy(tansig201, T.tansig, sizeof(T.tansig));
}

/* ------------------------------------------------------------------------------------ */
/* FFT (third-party in the reference: easyfft -> realfft -> rustfft; un-normalised both   */
/* ways; call sites src/features.rs:264,290).  Mixed radix 4/2/3/5 decimation in time.    */
/* ------------------------------------------------------------------------------------ */
static inline cpx cmul(cpx a, cpx b)
{
    cpx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    return r;
}

/* twiddle exp(sign * 2 pi i t / 480), t taken mod 480 */
static inline cpx tw(int t, int inverse)
{
    cpx w = T.tw480[t % 480];
    if (inverse) w.im = -w.im;
    return w;
}

static void cfft_rec(int n, const cpx *in, int stride, cpx *out, int inverse)
{
    if (n == 1) {
        out[0] = in[0];
        return;
    }
    int p = (n % 4 == 0) ? 4 : (n % 2 == 0) ? 2 : (n % 3 == 0) ? 3 : 5;
    int m = n / p;
    for (int q = 0; q < p; q++) cfft_rec(m, in + q * stride, stride * p, out + q * m, inverse);
    int tstep = 480 / n; /* w_n^k = w_480^(k*tstep) */
    for (int k = 0; k < m; k++) {
        cpx t[5], y[5];
        for (int q = 0; q < p; q++) t[q] = q ? cmul(out[q * m + k], tw(q * k * tstep, inverse)) : out[k];
        if (p == 2) {
            y[0].re = t[0].re + t[1].re; y[0].im = t[0].im + t[1].im;
            y[1].re = t[0].re - t[1].re; y[1].im = t[0].im - t[1].im;
        } else if (p == 4) {
            cpx a = {t[0].re + t[2].re, t[0].im + t[2].im};
            cpx b = {t[0].re - t[2].re, t[0].im - t[2].im};
            cpx c = {t[1].re + t[3].re, t[1].im + t[3].im};
            cpx d = {t[1].re - t[3].re, t[1].im - t[3].im};
            /* forward: -i*d = (d.im, -d.re); inverse: +i*d = (-d.im, d.re) */
            cpx jd;
            if (!inverse) { jd.re = d.im; jd.im = -d.re; } else { jd.re = -d.im; jd.im = d.re; }
            y[0].re = a.re + c.re; y[0].im = a.im + c.im;
            y[1].re = b.re + jd.re; y[1].im = b.im + jd.im;
            y[2].re = a.re - c.re; y[2].im = a.im - c.im;
            y[3].re = b.re - jd.re; y[3].im = b.im - jd.im;
        } else {
            int pstep = 480 / p;
            for (int j = 0; j < p; j++) {
                cpx s = t[0];
                for (int q = 1; q < p; q++) {
                    cpx w = tw(((j * q) % p) * pstep, inverse);
                    cpx v = cmul(t[q], w);
                    s.re += v.re; s.im += v.im;
                }
                y[j] = s;
            }
        }
        for (int j = 0; j < p; j++) out[k + j * m] = y[j];
    }
}

/* forward real FFT of 960 points -> 481 bins, un-normalised */
static void rfft960(const float *x, cpx *X /*481*/)
{
    cpx z[480], Z[480];
    for (int n = 0; n < 480; n++) { z[n].re = (fftr)x[2 * n]; z[n].im = (fftr)x[2 * n + 1]; }
    cfft_rec(480, z, 1, Z, 0);
    for (int k = 0; k <= 480; k++) {
        cpx a = Z[k % 480];
        cpx b = Z[(480 - k) % 480]; /* conj(b) used below */
        cpx e = {(a.re + b.re) * (fftr)0.5, (a.im - b.im) * (fftr)0.5};
        /* o = (a - conj(b)) / (2i) */
        cpx d = {a.re - b.re, a.im + b.im};
        cpx o = {d.im * (fftr)0.5, -d.re * (fftr)0.5};
        cpx wo = cmul(o, T.tw960[k]);
        X[k].re = e.re + wo.re;
        X[k].im = e.im + wo.im;
    }
}

/* inverse (complex-to-real) FFT of 481 bins -> 960 points, un-normalised
 * (x[n] = sum over the Hermitian-extended spectrum of X[k] e^{+2 pi i k n/960}). */
static void irfft960(const cpx *X /*481*/, float *x)
{
    cpx Zin[480], z[480];
    for (int k = 0; k < 480; k++) {
        cpx a = X[k];
        cpx b = X[480 - k];
        cpx e2 = {a.re + b.re, a.im - b.im};        /* 2E = X[k] + conj(X[480-k]) */
        cpx d = {a.re - b.re, a.im + b.im};         /* X[k] - conj(X[480-k])      */
        cpx w = T.tw960[k];
        w.im = -w.im;                               /* e^{+2 pi i k/960}          */
        cpx o2 = cmul(d, w);                        /* 2O                          */
        Zin[k].re = e2.re - o2.im;                  /* 2E + i 2O                   */
        Zin[k].im = e2.im + o2.re;
    }
    cfft_rec(480, Zin, 1, z, 1);
    for (int n = 0; n < 480; n++) { x[2 * n] = (float)z[n].re; x[2 * n + 1] = (float)z[n].im; }
}

void nnno_rfft960(const float *in960, float *out)
{
    init_tables();
    cpx X[FREQ_SIZE];
    rfft960(in960, X);
    for (int k = 0; k < FREQ_SIZE; k++) { out[2 * k] = (float)X[k].re; out[2 * k + 1] = (float)X[k].im; }
}

void nnno_irfft960(const float *in, float *out960)
{
    init_tables();
    cpx X[FREQ_SIZE];
    for (int k = 0; k < FREQ_SIZE; k++) { X[k].re = (fftr)in[2 * k]; X[k].im = (fftr)in[2 * k + 1]; }
    irfft960(X, out960);
}

/* ------------------------------------------------------------------------------------ */
/* Model (.rnn container)                                                                 */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int nb_inputs, nb_neurons, activation;
    const int8_t *weights, *bias;
} dense_t;
typedef struct {
    int nb_inputs, nb_neurons, activation;
    const int8_t *weights, *rec, *bias;
} gru_t;

struct nnno_model {
    int8_t *blob;
    dense_t input_dense, denoise_output, vad_output;
    gru_t vad_gru, noise_gru, denoise_gru;
};

/* ref: src/rnn.rs:145-164 */
static int read_dense(const int8_t **p, size_t *left, dense_t *l)
{
    if (*left < 3) return 0;
    const int8_t *b = *p;
    if (b[0] < 0 || b[1] < 0 || b[2] < 0 || b[2] > 2) return 0;
    l->nb_inputs = b[0]; l->nb_neurons = b[1]; l->activation = b[2];
    size_t nw = (size_t)l->nb_inputs * (size_t)l->nb_neurons, nb = (size_t)l->nb_neurons;
    if (*left - 3 < nw + nb) return 0;
    l->weights = b + 3; l->bias = b + 3 + nw;
    *p = b + 3 + nw + nb; *left -= 3 + nw + nb;
    return 1;
}

/* ref: src/rnn.rs:166-187 */
static int read_gru(const int8_t **p, size_t *left, gru_t *l)
{
    if (*left < 3) return 0;
    const int8_t *b = *p;
    if (b[0] < 0 || b[1] < 0 || b[2] < 0 || b[2] > 2) return 0;
    l->nb_inputs = b[0]; l->nb_neurons = b[1]; l->activation = b[2];
    size_t n = (size_t)l->nb_neurons;
    size_t nw = 3 * n * (size_t)l->nb_inputs, nr = 3 * n * n, nb = 3 * n;
    if (*left - 3 < nw + nr + nb) return 0;
    l->weights = b + 3; l->rec = b + 3 + nw; l->bias = b + 3 + nw + nr;
    *p = b + 3 + nw + nr + nb; *left -= 3 + nw + nr + nb;
    return 1;
}

/* ref: src/rnn.rs:116-232 */
nnno_model *nnno_model_from_bytes(const uint8_t *bytes, size_t len)
{
    init_tables();
    nnno_model *m = (nnno_model *)calloc(1, sizeof(*m));
    if (!m) return NULL;
    m->blob = (int8_t *)malloc(len ? len : 1);
    memcpy(m->blob, bytes, len);
    const int8_t *p = m->blob;
    size_t left = len;
    int ok = read_dense(&p, &left, &m->input_dense) && read_gru(&p, &left, &m->vad_gru) &&
             read_gru(&p, &left, &m->noise_gru) && read_gru(&p, &left, &m->denoise_gru) &&
             read_dense(&p, &left, &m->denoise_output) && read_dense(&p, &left, &m->vad_output);
    if (ok && left != 0) ok = 0;                                                  /* :196-198 */
    if (ok && (m->input_dense.nb_inputs != 42 || m->denoise_output.nb_neurons != 22 ||
               m->vad_output.nb_neurons != 1)) ok = 0;                            /* :204-209 */
    if (ok && (m->input_dense.nb_neurons != m->vad_gru.nb_inputs ||
               m->vad_gru.nb_neurons != m->vad_output.nb_inputs)) ok = 0;         /* :210-213 */
    if (ok && 42 + m->input_dense.nb_neurons + m->vad_gru.nb_neurons != m->noise_gru.nb_inputs) ok = 0;
    if (ok && 42 + m->vad_gru.nb_neurons + m->noise_gru.nb_neurons != m->denoise_gru.nb_inputs) ok = 0;
    if (ok && m->denoise_gru.nb_neurons != m->denoise_output.nb_inputs) ok = 0;
    if (!ok) { nnno_model_free(m); return NULL; }
    return m;
}

void nnno_model_free(nnno_model *m)
{
    if (!m) return;
    free(m->blob);
    free(m);
}

void nnno_model_shape(const nnno_model *m, int32_t s[12])
{
    s[0] = m->input_dense.nb_inputs; s[1] = m->input_dense.nb_neurons; s[2] = m->vad_gru.nb_neurons;
    s[3] = m->noise_gru.nb_neurons; s[4] = m->denoise_gru.nb_neurons; s[5] = m->denoise_output.nb_neurons;
    s[6] = m->input_dense.activation; s[7] = m->vad_gru.activation; s[8] = m->noise_gru.activation;
    s[9] = m->denoise_gru.activation; s[10] = m->denoise_output.activation; s[11] = m->vad_output.activation;
}

/* ------------------------------------------------------------------------------------ */
/* State                                                                                  */
/* ------------------------------------------------------------------------------------ */
struct nnno_state {
    const nnno_model *model;
    /* ref: src/features.rs:18-46 */
    float input_mem[PITCH_BUF_SIZE];
    float cepstral_mem[CEPS_MEM][NB_BANDS];
    int mem_id;
    float mem_hp_x[2];
    float synthesis_mem[FRAME_SIZE];
    /* ref: src/denoise.rs:37-42 */
    float lastg[NB_BANDS];
    /* ref: src/pitch.rs:4-17 */
    int last_period;
    float last_gain;
    /* ref: src/rnn.rs:65-70 */
    float vad_gru_state[MAX_NEURONS], noise_gru_state[MAX_NEURONS], denoise_gru_state[MAX_NEURONS];
    /* scratch that survives the call only as taps */
    cpx x[FREQ_SIZE], p[FREQ_SIZE];
    float ex[NB_BANDS], ep[NB_BANDS], exp_[NB_BANDS];
    float features[NB_FEATURES];
    float pitch_buf[PITCH_BUF_SIZE / 2];
    nnno_taps taps;
};

nnno_state *nnno_create(const nnno_model *m)
{
    init_tables();
    nnno_state *st = (nnno_state *)calloc(1, sizeof(*st)); /* all-zero init, ref: features.rs:58-74 */
    if (st) st->model = m;
    return st;
}

void nnno_destroy(nnno_state *st) { free(st); }
void nnno_get_taps(const nnno_state *st, nnno_taps *t) { *t = st->taps; }

/* ------------------------------------------------------------------------------------ */
/* util.rs                                                                                */
/* ------------------------------------------------------------------------------------ */
/* ref: src/util.rs:29-45 */
static float tansig_approx(float x)
{
    if (!(x < 8.0f)) return 1.0f;
    if (!(x > -8.0f)) return -1.0f;
    float sign = 1.0f;
    if (x < 0.0f) { x = -x; sign = -1.0f; }
    float fi = floorf(0.5f + 25.0f * x);
    x -= 0.04f * fi;
    float y = T.tansig[(int)fi];
    float dy = 1.0f - y * y;
    y = y + x * dy * (1.0f - y * x);
    return sign * y;
}
/* ref: src/util.rs:47-53 */
static float sigmoid_approx(float x) { return 0.5f + 0.5f * tansig_approx(0.5f * x); }
static float relu(float x) { return x > 0.0f ? x : 0.0f; } /* f32::max(x, 0) */

static inline float fmax_rs(float a, float b) { return (a > b || b != b) ? a : b; } /* f32::max */
static inline float fmin_rs(float a, float b) { return (a < b || b != b) ? a : b; } /* f32::min */

/* ref: src/util.rs:95-107 -- DF2T biquad, f64 arithmetic, f32 state; a=[-1.99599,0.996] b=[-2,1] */
static void biquad_hp(float *out, float mem[2], const float *in, int n)
{
    const double a0 = (double)-1.99599f, a1 = (double)0.99600f, b0 = (double)-2.0f, b1 = (double)1.0f;
    for (int i = 0; i < n; i++) {
        double x64 = (double)in[i];
        double y64 = x64 + (double)mem[0];
        mem[0] = (float)((double)mem[1] + (b0 * x64 - a0 * y64));
        mem[1] = (float)(b1 * x64 - a1 * y64);
        out[i] = (float)y64;
    }
}

/* ------------------------------------------------------------------------------------ */
/* lib.rs band ops                                                                        */
/* ------------------------------------------------------------------------------------ */
/* ref: src/lib.rs:65-82 */
static void compute_band_corr(float *out, const cpx *x, const cpx *p)
{
    for (int i = 0; i < NB_BANDS; i++) out[i] = 0.0f;
    for (int i = 0; i < NB_BANDS - 1; i++) {
        int band_size = (EBAND_5MS[i + 1] - EBAND_5MS[i]) << 2;
        for (int j = 0; j < band_size; j++) {
            float frac = (float)j / (float)band_size;
            int idx = (EBAND_5MS[i] << 2) + j;
            float xr = (float)x[idx].re, xi = (float)x[idx].im, pr = (float)p[idx].re, pi_ = (float)p[idx].im;
            float corr = xr * pr + xi * pi_;
            out[i] += (1.0f - frac) * corr;
            out[i + 1] += frac * corr;
        }
    }
    out[0] *= 2.0f;
    out[NB_BANDS - 1] *= 2.0f;
}

/* ref: src/lib.rs:84-97 */
static void interp_band_gain(float *out /*481*/, const float *band_e)
{
    for (int i = 0; i < FREQ_SIZE; i++) out[i] = 0.0f;
    for (int i = 0; i < NB_BANDS - 1; i++) {
        int band_size = (EBAND_5MS[i + 1] - EBAND_5MS[i]) << 2;
        for (int j = 0; j < band_size; j++) {
            float frac = (float)j / (float)band_size;
            int idx = (EBAND_5MS[i] << 2) + j;
            out[idx] = (1.0f - frac) * band_e[i] + frac * band_e[i + 1];
        }
    }
}

/* ref: src/lib.rs:139-148 */
static void dct(float *out, const float *x)
{
    for (int i = 0; i < NB_BANDS; i++) {
        float sum = 0.0f;
        for (int j = 0; j < NB_BANDS; j++) sum += x[j] * T.dct_table[j * NB_BANDS + i];
        out[i] = (float)((double)sum * sqrt(2.0 / (double)NB_BANDS));
    }
}

/* ------------------------------------------------------------------------------------ */
/* pitch.rs                                                                               */
/* ------------------------------------------------------------------------------------ */
/* ref: src/pitch.rs:225-244 */
static float inner_prod(const float *xs, const float *ys, int n)
{
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int n4 = n - n % 4;
    for (int i = 0; i < n4; i += 4) {
        s0 += xs[i] * ys[i];
        s1 += xs[i + 1] * ys[i + 1];
        s2 += xs[i + 2] * ys[i + 2];
        s3 += xs[i + 3] * ys[i + 3];
    }
    float sum = s0 + s1 + s2 + s3;
    for (int i = n4; i < n; i++) sum += xs[i] * ys[i];
    return sum;
}

/* ref: src/pitch.rs:296-363.  The 4x4 unrolled reference loop adds, for every lag i, the
 * products xs[j]*ys[i+j] in increasing j into one accumulator that starts at 0 -- i.e. a
 * strictly sequential per-lag sum -- which is what is restated here. */
static void pitch_xcorr(const float *xs, int xlen, const float *ys, float *xcorr, int nlags)
{
    for (int i = 0; i < nlags; i++) {
        float c = 0.0f;
        for (int j = 0; j < xlen; j++) c += xs[j] * ys[i + j];
        xcorr[i] = c;
    }
}

/* ref: src/pitch.rs:372-405 */
static void find_best_pitch(const float *xcorr, int nlags, const float *ys, int len, int *best, int *second)
{
    float best_num = -1.0f, second_best_num = -1.0f, best_den = 0.0f, second_best_den = 0.0f;
    int best_pitch = 0, second_best_pitch = 1;
    float y_sq_norm = 1.0f;
    for (int j = 0; j < len; j++) y_sq_norm += ys[j] * ys[j];
    for (int i = 0; i < nlags; i++) {
        float corr = xcorr[i];
        if (corr > 0.0f) {
            float num = corr * corr;
            if (num * second_best_den > second_best_num * y_sq_norm) {
                if (num * best_den > best_num * y_sq_norm) {
                    second_best_num = best_num; second_best_den = best_den; second_best_pitch = best_pitch;
                    best_num = num; best_den = y_sq_norm; best_pitch = i;
                } else {
                    second_best_num = num; second_best_den = y_sq_norm; second_best_pitch = i;
                }
            }
        }
        y_sq_norm += ys[i + len] * ys[i + len] - ys[i] * ys[i];
        y_sq_norm = fmax_rs(y_sq_norm, 1.0f);
    }
    *best = best_pitch;
    *second = second_best_pitch;
}

/* ref: src/pitch.rs:257-292 */
static void lpc4(float *lpc, const float *ac)
{
    const int p = 4;
    float error = ac[0];
    for (int i = 0; i < p; i++) lpc[i] = 0.0f;
    if (ac[0] == 0.0f) return;
    for (int i = 0; i < p; i++) {
        float rr = 0.0f;
        for (int j = 0; j < i; j++) rr += lpc[j] * ac[i - j];
        rr += ac[i + 1];
        float r = -rr / error;
        lpc[i] = r;
        for (int j = 0; j < (i + 1) / 2; j++) {
            float tmp1 = lpc[j], tmp2 = lpc[i - 1 - j];
            lpc[j] = tmp1 + r * tmp2;
            lpc[i - 1 - j] = tmp2 + r * tmp1;
        }
        error = error - r * r * error;
        if (error < 0.001f * ac[0]) return;
    }
}

/* ref: src/pitch.rs:448-483 (celt_autocorr :433-446, fir5_in_place :407-429) */
static void pitch_downsample(const float *x /*1728*/, float *x_lp /*864*/, nnno_taps *taps)
{
    const int n = PITCH_BUF_SIZE / 2;
    float ac[5], lpc[4], lpc2[5];
    for (int i = 1; i < n; i++) x_lp[i] = ((x[2 * i - 1] + x[2 * i + 1]) / 2.0f + x[2 * i]) / 2.0f;
    x_lp[0] = (x[1] / 2.0f + x[0]) / 2.0f;

    /* celt_autocorr: lag = 4, fast_n = n - 4 */
    const int fast_n = n - 4;
    pitch_xcorr(x_lp, fast_n, x_lp, ac, 5);
    for (int k = 0; k < 5; k++) {
        float d = 0.0f;
        for (int i = k + fast_n; i < n; i++) d += x_lp[i] * x_lp[i - k];
        ac[k] += d;
    }
    ac[0] *= 1.0001f;
    for (int i = 1; i < 5; i++) ac[i] -= ac[i] * (0.008f * (float)i) * (0.008f * (float)i);
    lpc4(lpc, ac);
    float tmp = 1.0f;
    for (int i = 0; i < 4; i++) { tmp *= 0.9f; lpc[i] *= tmp; }
    lpc2[0] = lpc[0] + 0.8f;
    lpc2[1] = lpc[1] + 0.8f * lpc[0];
    lpc2[2] = lpc[2] + 0.8f * lpc[1];
    lpc2[3] = lpc[3] + 0.8f * lpc[2];
    lpc2[4] = 0.8f * lpc[3];
    memcpy(taps->ac, ac, sizeof(ac));
    memcpy(taps->lpc2, lpc2, sizeof(lpc2));
    /* fir5_in_place with zero initial memory */
    float m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    for (int i = 0; i < n; i++) {
        float xi = x_lp[i];
        float o = xi + lpc2[0] * m0 + lpc2[1] * m1 + lpc2[2] * m2 + lpc2[3] * m3 + lpc2[4] * m4;
        m4 = m3; m3 = m2; m2 = m1; m1 = m0; m0 = xi;
        x_lp[i] = o;
    }
}

/* ref: src/pitch.rs:63-115 */
static int pitch_search(const float *pitch_buf, nnno_taps *taps)
{
    const float *x_lp = pitch_buf + PITCH_MAX_PERIOD / 2;
    const float *y = pitch_buf;
    const int len = PITCH_FRAME_SIZE;                               /* 960 */
    const int max_pitch = PITCH_MAX_PERIOD - 3 * PITCH_MIN_PERIOD;  /* 588 */
    float x_lp4[240], y_lp4[240 + 147], xcorr[294];
    for (int j = 0; j < len / 4; j++) x_lp4[j] = x_lp[2 * j];
    for (int j = 0; j < len / 4 + max_pitch / 4; j++) y_lp4[j] = y[2 * j];
    pitch_xcorr(x_lp4, len / 4, y_lp4, xcorr, max_pitch / 4);
    memcpy(taps->xcorr1, xcorr, 147 * sizeof(float));
    int best, second;
    find_best_pitch(xcorr, max_pitch / 4, y_lp4, len / 4, &best, &second);
    taps->best1[0] = best; taps->best1[1] = second;
    for (int i = 0; i < max_pitch / 2; i++) {
        xcorr[i] = 0.0f;
        if (abs(i - 2 * best) > 2 && abs(i - 2 * second) > 2) continue;
        xcorr[i] = fmax_rs(inner_prod(x_lp, y + i, len / 2), -1.0f);
    }
    memcpy(taps->xcorr2, xcorr, 294 * sizeof(float));
    find_best_pitch(xcorr, max_pitch / 2, y, len / 2, &best, &second);
    int offset = 0;
    if (best > 0 && best < max_pitch / 2 - 1) {
        float a = xcorr[best - 1], b = xcorr[best], c = xcorr[best + 1];
        if (c - a > 0.7f * (b - a)) offset = 1;
        else if (a - c > 0.7f * (b - c)) offset = -1;
    }
    return 2 * best - offset;
}

static inline float pitch_gain(float xy, float xx, float yy) { return xy / sqrtf(1.0f + xx * yy); } /* :485-487 */

/* ref: src/pitch.rs:118-221 */
static int remove_doubling(const float *x /*pitch_buf*/, int pitch_idx, int last_period, float last_gain, float *gain_out)
{
    static const int SECOND_CHECK[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};
    const int min_period = PITCH_MIN_PERIOD / 2, max_period = PITCH_MAX_PERIOD / 2, n = PITCH_FRAME_SIZE / 2;
    int t0 = pitch_idx / 2;
    if (t0 > max_period - 1) t0 = max_period - 1;
    const int prev_period = last_period / 2;
    float yy_lookup[PITCH_MAX_PERIOD / 2 + 1];
    int t = t0;
    float xx = inner_prod(x + max_period, x + max_period, n);
    float xy = inner_prod(x + max_period, x + max_period - t0, n);
    yy_lookup[0] = xx;
    float yy = xx;
    for (int i = 1; i <= max_period; i++) {
        yy += x[max_period - i] * x[max_period - i] - x[max_period + n - i] * x[max_period + n - i];
        yy_lookup[i] = fmax_rs(yy, 0.0f);
    }
    yy = yy_lookup[t0];
    float best_xy = xy, best_yy = yy;
    float g0 = pitch_gain(xy, xx, yy);
    float g = g0;
    for (int k = 2; k <= 15; k++) {
        int t1 = (2 * t0 + k) / (2 * k);
        if (t1 < min_period) break;
        int t1b;
        if (k == 2) t1b = (t1 + t0 > max_period) ? t0 : t0 + t1;
        else t1b = (2 * SECOND_CHECK[k] * t0 + k) / (2 * k);
        xy = inner_prod(x + max_period, x + max_period - t1, n);
        float xy2 = inner_prod(x + max_period, x + max_period - t1b, n);
        xy = (xy + xy2) / 2.0f;
        yy = (yy_lookup[t1] + yy_lookup[t1b]) / 2.0f;
        float g1 = pitch_gain(xy, xx, yy);
        float cont;
        if (abs(t1 - prev_period) <= 1) cont = last_gain;
        else if (abs(t1 - prev_period) <= 2 && 5 * k * k < t0) cont = last_gain / 2.0f;
        else cont = 0.0f;
        float thresh;
        if (t1 < 3 * min_period) thresh = fmax_rs(0.85f * g0 - cont, 0.4f);
        else if (t1 < 2 * min_period) thresh = fmax_rs(0.9f * g0 - cont, 0.5f); /* unreachable, kept as in ref */
        else thresh = fmax_rs(0.7f * g0 - cont, 0.3f);
        if (g1 > thresh) { best_xy = xy; best_yy = yy; t = t1; g = g1; }
    }
    best_xy = fmax_rs(best_xy, 0.0f);
    float pg = (best_yy <= best_xy) ? 1.0f : best_xy / (best_yy + 1.0f);
    float xc[3];
    for (int k = 0; k < 3; k++) xc[k] = inner_prod(x + max_period, x + max_period - (t + k - 1), n);
    int offset = 0;
    if (xc[2] - xc[0] > 0.7f * (xc[1] - xc[0])) offset = 1;
    else if (xc[0] - xc[2] > 0.7f * (xc[1] - xc[2])) offset = -1;
    pg = fmin_rs(pg, g);
    int res = 2 * t + offset;
    if (res < PITCH_MIN_PERIOD) res = PITCH_MIN_PERIOD;
    *gain_out = pg;
    return res;
}

/* ------------------------------------------------------------------------------------ */
/* features.rs                                                                            */
/* ------------------------------------------------------------------------------------ */
/* ref: src/features.rs:281-298 */
static void transform_input(const float *input_mem, int lag, cpx *x, float *ex)
{
    float buf[WINDOW_SIZE];
    const float *in = input_mem + (PITCH_BUF_SIZE - WINDOW_SIZE - lag);
    for (int i = 0; i < WINDOW_SIZE; i++) buf[i] = in[i] * T.window[i];
    rfft960(buf, x);
    for (int k = 0; k < FREQ_SIZE; k++) {
        /* `*x *= norm` on f32 bins; the product is rounded to f32 like the reference's storage */
        x[k].re = (fftr)((float)x[k].re * T.wnorm);
        x[k].im = (fftr)((float)x[k].im * T.wnorm);
    }
    compute_band_corr(ex, x, x);
}

/* ref: src/features.rs:115-219; returns 1 on silence */
static int compute_frame_features(nnno_state *st)
{
    float ly[NB_BANDS], tmp[NB_BANDS];
    nnno_taps *tp = &st->taps;
    transform_input(st->input_mem, 0, st->x, st->ex);
    /* find_pitch, ref: src/features.rs:106-110 and src/pitch.rs:45-54 */
    pitch_downsample(st->input_mem, st->pitch_buf, tp);
    memcpy(tp->xlp, st->pitch_buf, sizeof(tp->xlp));
    int ps = pitch_search(st->pitch_buf, tp);
    tp->pitch_search = ps;
    float pgain;
    int pitch_idx = remove_doubling(st->pitch_buf, PITCH_MAX_PERIOD - ps, st->last_period, st->last_gain, &pgain);
    st->last_period = pitch_idx;
    st->last_gain = pgain;
    tp->pitch_idx = pitch_idx;
    tp->pitch_gain = pgain;

    transform_input(st->input_mem, pitch_idx, st->p, st->ep);
    compute_band_corr(st->exp_, st->x, st->p);
    for (int i = 0; i < NB_BANDS; i++) st->exp_[i] /= sqrtf(0.001f + st->ex[i] * st->ep[i]);
    dct(tmp, st->exp_);
    float *f = st->features;
    for (int i = 0; i < NB_DELTA_CEPS; i++) f[NB_BANDS + 2 * NB_DELTA_CEPS + i] = tmp[i];
    f[NB_BANDS + 2 * NB_DELTA_CEPS] -= 1.3f;
    f[NB_BANDS + 2 * NB_DELTA_CEPS + 1] -= 0.9f;
    f[NB_BANDS + 3 * NB_DELTA_CEPS] = 0.01f * ((float)pitch_idx - 300.0f);
    float log_max = -2.0f, follow = -2.0f, e = 0.0f;
    for (int i = 0; i < NB_BANDS; i++) {
        ly[i] = fmax_rs(fmax_rs(log10f(1e-2f + st->ex[i]), log_max - 7.0f), follow - 1.5f);
        log_max = fmax_rs(log_max, ly[i]);
        follow = fmax_rs(follow - 1.5f, ly[i]);
        e += st->ex[i];
    }
    if (e < 0.04f) {
        for (int i = 0; i < NB_FEATURES; i++) f[i] = 0.0f;
        return 1;
    }
    dct(f, ly);
    f[0] -= 12.0f;
    f[1] -= 4.0f;
    int c0 = st->mem_id;
    int c1 = st->mem_id < 1 ? CEPS_MEM + st->mem_id - 1 : st->mem_id - 1;
    int c2 = st->mem_id < 2 ? CEPS_MEM + st->mem_id - 2 : st->mem_id - 2;
    for (int i = 0; i < NB_BANDS; i++) st->cepstral_mem[c0][i] = f[i];
    st->mem_id += 1;
    const float *ceps0 = st->cepstral_mem[c0], *ceps1 = st->cepstral_mem[c1], *ceps2 = st->cepstral_mem[c2];
    for (int i = 0; i < NB_DELTA_CEPS; i++) {
        f[i] = ceps0[i] + ceps1[i] + ceps2[i];
        f[NB_BANDS + i] = ceps0[i] - ceps2[i];
        f[NB_BANDS + NB_DELTA_CEPS + i] = ceps0[i] - 2.0f * ceps1[i] + ceps2[i];
    }
    float spec_variability = 0.0f;
    if (st->mem_id == CEPS_MEM) st->mem_id = 0;
    for (int i = 0; i < CEPS_MEM; i++) {
        float min_dist = 1e15f;
        for (int j = 0; j < CEPS_MEM; j++) {
            float dist = 0.0f;
            for (int k = 0; k < NB_BANDS; k++) {
                float d = st->cepstral_mem[i][k] - st->cepstral_mem[j][k];
                dist += d * d;
            }
            if (j != i) min_dist = fmin_rs(min_dist, dist);
        }
        spec_variability += min_dist;
    }
    f[NB_BANDS + 3 * NB_DELTA_CEPS + 1] = spec_variability / (float)CEPS_MEM - 2.1f;
    return 0;
}

static inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

/* ref: src/features.rs:223-257 */
static void pitch_filter(nnno_state *st, const float *gain)
{
    float r[NB_BANDS], rf[FREQ_SIZE], new_e[NB_BANDS];
    for (int i = 0; i < NB_BANDS; i++) {
        float v;
        if (st->exp_[i] > gain[i]) v = 1.0f;
        else {
            float exp_sq = st->exp_[i] * st->exp_[i], g_sq = gain[i] * gain[i];
            v = exp_sq * (1.0f - g_sq) / (0.001f + g_sq * (1.0f - exp_sq));
        }
        v = sqrtf(clamp01(v));
        v *= sqrtf(st->ex[i] / (1e-8f + st->ep[i]));
        r[i] = v;
    }
    interp_band_gain(rf, r);
    for (int k = 0; k < FREQ_SIZE; k++) {
        /* DC goes through get_offset (real only); bins 1..480 are complex.  DC imag is 0. */
        st->x[k].re = (fftr)((float)st->x[k].re + (float)st->p[k].re * rf[k]);
        st->x[k].im = (fftr)((float)st->x[k].im + (float)st->p[k].im * rf[k]);
    }
    compute_band_corr(new_e, st->x, st->x);
    for (int i = 0; i < NB_BANDS; i++) r[i] = sqrtf(st->ex[i] / (1e-8f + new_e[i]));
    interp_band_gain(rf, r);
    for (int k = 0; k < FREQ_SIZE; k++) {
        st->x[k].re = (fftr)((float)st->x[k].re * rf[k]);
        st->x[k].im = (fftr)((float)st->x[k].im * rf[k]);
    }
}

/* ref: src/features.rs:263-275 */
static void frame_synthesis(nnno_state *st, float *out)
{
    float buf[WINDOW_SIZE];
    irfft960(st->x, buf);
    for (int i = 0; i < WINDOW_SIZE; i++) buf[i] /= 2.0f;
    for (int i = 0; i < WINDOW_SIZE; i++) buf[i] *= T.window[i];
    for (int i = 0; i < FRAME_SIZE; i++) {
        out[i] = buf[i] + st->synthesis_mem[i];
        st->synthesis_mem[i] = buf[FRAME_SIZE + i];
    }
}

/* ------------------------------------------------------------------------------------ */
/* rnn.rs                                                                                 */
/* ------------------------------------------------------------------------------------ */
static float activate(int act, float x)
{
    switch (act) {
    case 0: return tansig_approx(x);
    case 1: return sigmoid_approx(x);
    default: return relu(x);
    }
}

/* ref: src/rnn.rs:402-410 -- input-major rows of `stride`, column window at `offset` */
static void mul_add(const int8_t *data, int stride, int offset, float *out, int n_out, const float *in, int n_in)
{
    for (int k = 0; k < n_in; k++) {
        const int8_t *row = data + (size_t)k * stride + offset;
        float v = in[k];
        for (int j = 0; j < n_out; j++) out[j] += (float)row[j] * v;
    }
}

/* ref: src/rnn.rs:251-272 */
static void dense_compute(const dense_t *l, float *out, const float *in)
{
    const float scale = 1.0f / 256.0f;
    for (int j = 0; j < l->nb_neurons; j++) out[j] = (float)l->bias[j];
    mul_add(l->weights, l->nb_neurons, 0, out, l->nb_neurons, in, l->nb_inputs);
    for (int j = 0; j < l->nb_neurons; j++) out[j] = activate(l->activation, out[j] * scale);
}

/* ref: src/rnn.rs:292-327 */
static void gru_compute(const gru_t *l, float *state, const float *in)
{
    const float scale = 1.0f / 256.0f;
    float z[MAX_NEURONS], r[MAX_NEURONS], h[MAX_NEURONS];
    const int n = l->nb_neurons, m = l->nb_inputs, stride = 3 * n;
    for (int j = 0; j < n; j++) z[j] = (float)l->bias[j];
    mul_add(l->weights, stride, 0, z, n, in, m);
    mul_add(l->rec, stride, 0, z, n, state, n);
    for (int j = 0; j < n; j++) z[j] = sigmoid_approx(scale * z[j]);
    for (int j = 0; j < n; j++) r[j] = (float)l->bias[n + j];
    mul_add(l->weights, stride, n, r, n, in, m);
    mul_add(l->rec, stride, n, r, n, state, n);
    for (int j = 0; j < n; j++) r[j] = state[j] * sigmoid_approx(scale * r[j]);
    for (int j = 0; j < n; j++) h[j] = (float)l->bias[2 * n + j];
    mul_add(l->weights, stride, 2 * n, h, n, in, m);
    mul_add(l->rec, stride, 2 * n, h, n, r, n);
    for (int j = 0; j < n; j++) {
        float hh = activate(l->activation, scale * h[j]);
        state[j] = z[j] * state[j] + (1.0f - z[j]) * hh;
    }
}

/* ref: src/rnn.rs:343-379 */
static void rnn_compute(nnno_state *st, float *gains, float *vad, const float *input)
{
    const nnno_model *m = st->model;
    float buf[MAX_NEURONS * 3], dbuf[MAX_NEURONS * 3];
    memset(buf, 0, sizeof(buf));
    memset(dbuf, 0, sizeof(dbuf));
    const int nd = m->input_dense.nb_neurons, nv = m->vad_gru.nb_neurons, nn = m->noise_gru.nb_neurons;
    dense_compute(&m->input_dense, buf, input);
    gru_compute(&m->vad_gru, st->vad_gru_state, buf);
    dense_compute(&m->vad_output, vad, st->vad_gru_state);
    memcpy(buf + nd, st->vad_gru_state, nv * sizeof(float));
    memcpy(buf + nd + nv, input, NB_FEATURES * sizeof(float));
    gru_compute(&m->noise_gru, st->noise_gru_state, buf);
    memcpy(dbuf, st->vad_gru_state, nv * sizeof(float));
    memcpy(dbuf + nv, st->noise_gru_state, nn * sizeof(float));
    memcpy(dbuf + nv + nn, input, NB_FEATURES * sizeof(float));
    gru_compute(&m->denoise_gru, st->denoise_gru_state, dbuf);
    dense_compute(&m->denoise_output, gains, st->denoise_gru_state);
}

/* ------------------------------------------------------------------------------------ */
/* denoise.rs                                                                             */
/* ------------------------------------------------------------------------------------ */
/* ref: src/denoise.rs:95-116 */
float nnno_process_frame(nnno_state *st, float *out, const float *in)
{
    float g[NB_BANDS], gf[FREQ_SIZE], vad = 0.0f;
    nnno_taps *tp = &st->taps;
    for (int i = 0; i < NB_BANDS; i++) g[i] = 0.0f;
    /* shift_and_filter_input, ref: src/features.rs:97-104 */
    memmove(st->input_mem, st->input_mem + FRAME_SIZE, (PITCH_BUF_SIZE - FRAME_SIZE) * sizeof(float));
    biquad_hp(st->input_mem + PITCH_BUF_SIZE - FRAME_SIZE, st->mem_hp_x, in, FRAME_SIZE);
    memcpy(tp->filtered, st->input_mem + PITCH_BUF_SIZE - FRAME_SIZE, sizeof(tp->filtered));

    int silence = compute_frame_features(st);
    tp->silence = silence;
    for (int k = 0; k < FREQ_SIZE; k++) {
        tp->X[2 * k] = (float)st->x[k].re; tp->X[2 * k + 1] = (float)st->x[k].im;
        tp->P[2 * k] = (float)st->p[k].re; tp->P[2 * k + 1] = (float)st->p[k].im;
    }
    memcpy(tp->ex, st->ex, sizeof(tp->ex));
    memcpy(tp->ep, st->ep, sizeof(tp->ep));
    memcpy(tp->exp_, st->exp_, sizeof(tp->exp_));
    memcpy(tp->features, st->features, sizeof(tp->features));
    memset(tp->g_raw, 0, sizeof(tp->g_raw));
    if (!silence) {
        rnn_compute(st, g, &vad, st->features);
        memcpy(tp->g_raw, g, sizeof(tp->g_raw));
        pitch_filter(st, g);
        for (int i = 0; i < NB_BANDS; i++) {
            g[i] = fmax_rs(g[i], 0.6f * st->lastg[i]);
            st->lastg[i] = g[i];
        }
        interp_band_gain(gf, g);
        for (int k = 0; k < FREQ_SIZE; k++) {
            st->x[k].re = (fftr)((float)st->x[k].re * gf[k]);
            st->x[k].im = (fftr)((float)st->x[k].im * gf[k]);
        }
    }
    memcpy(tp->g, g, sizeof(tp->g));
    tp->vad = vad;
    float tmp[FRAME_SIZE];
    frame_synthesis(st, tmp);
    memcpy(out, tmp, sizeof(tmp));
    memcpy(tp->out, tmp, sizeof(tmp));
    return vad;
}

/* Conditioning of the most recent frame: pitch_filter (src/features.rs:223-257) picks r = 1 when exp > g and a closed
 * form otherwise; where the closed form is far from 1 at exp == g (both near 1, or g near 0) the branch is a jump, and
 * rounding noise in the FFT decides the frame's output.  Returns the smallest |exp - g| over the bands whose jump exceeds
 * 0.01 (a large number if none, or if the frame is silent): parity tests excuse frames where this is tiny. */
float nnno_frame_condition(const nnno_state *st)
{
    float best = 1e30f;
    if (st->taps.silence) return best;
    for (int i = 0; i < NB_BANDS; i++) {
        double e = st->taps.exp_[i], g = st->taps.g_raw[i];
        double jump = 1.0 - e * e * (1.0 - g * g) / (0.001 + g * g * (1.0 - e * e));
        float d = (float)fabs(e - g);
        if (jump > 0.01 && d < best) best = d;
    }
    return best;
}

/* Discrete decisions of the most recent frame: bit i (< 22) = pitch_filter took its `exp > g` branch in band i
 * (src/features.rs:227), bit 22 = the frame was silent (src/features.rs:160-166, src/denoise.rs:100).  Parity tests
 * excuse the audio comparison only on frames where the device's mask differs from this one. */
int32_t nnno_frame_branch(const nnno_state *st)
{
    int32_t m = 0;
    if (st->taps.silence) return 1 << NB_BANDS;
    for (int i = 0; i < NB_BANDS; i++)
        if (st->taps.exp_[i] > st->taps.g_raw[i]) m |= 1 << i;
    return m;
}

/* the activation functions on their own (src/util.rs:29-53), for a direct known-answer sweep against the device's */
float nnno_tansig(float x) { init_tables(); return tansig_approx(x); }
float nnno_sigmoid(float x) { init_tables(); return sigmoid_approx(x); }

int nnno_run_streams_full(const nnno_model *m, int n_streams, int n_frames, const float *in, float *out, float *vad,
                          int32_t *pitch, float *gains, float *feats, float *cond, int32_t *branch, float *g_raw,
                          float *exp_, int n_threads)
{
    init_tables();
    int used = 1;
#ifdef _OPENMP
    if (n_threads > 1) used = n_threads;
#pragma omp parallel for schedule(static) num_threads(used)
#endif
    for (int s = 0; s < n_streams; s++) {
        nnno_state *st = nnno_create(m);
        float o[FRAME_SIZE];
        for (int t = 0; t < n_frames; t++) {
            size_t ft = (size_t)s * n_frames + t;
            float v = nnno_process_frame(st, o, in + ft * FRAME_SIZE);
            if (out) memcpy(out + ft * FRAME_SIZE, o, sizeof(o));
            if (vad) vad[ft] = v;
            if (pitch) pitch[ft] = st->taps.pitch_idx;
            if (gains) memcpy(gains + ft * NB_BANDS, st->taps.g, NB_BANDS * sizeof(float));
            if (feats) memcpy(feats + ft * NB_FEATURES, st->features, NB_FEATURES * sizeof(float));
            if (cond) cond[ft] = nnno_frame_condition(st);
            if (branch) branch[ft] = nnno_frame_branch(st);
            if (g_raw) memcpy(g_raw + ft * NB_BANDS, st->taps.g_raw, NB_BANDS * sizeof(float));
            if (exp_) memcpy(exp_ + ft * NB_BANDS, st->taps.exp_, NB_BANDS * sizeof(float));
        }
        nnno_destroy(st);
    }
    return used;
}

int nnno_run_streams_cond(const nnno_model *m, int n_streams, int n_frames, const float *in, float *out,
                          float *vad, int32_t *pitch, float *gains, float *feats, float *cond, int n_threads)
{
    return nnno_run_streams_full(m, n_streams, n_frames, in, out, vad, pitch, gains, feats, cond, NULL, NULL, NULL, n_threads);
}

/* ---- CPU baseline timing (bench.py cpu_baseline) ---------------------------------------------------
 * Times process_frame on `n_threads` host threads with nothing but process_frame in the timed region: every thread
 * owns one state and one 100-frame input buffer made before the clock starts (no allocation, no shared output).
 *   kind 0: SURVEY 8(d) synthetic mix -- per thread a sine (80..1000 Hz) plus uniform noise, `iters` x 100 frames
 *           on ONE continuing state per thread;
 *   kind 1: the reference's own bench shape, benches/sin.rs:9-20 -- 100 frames of a 440 Hz sine at amplitude
 *           i16::MAX, FRESH state every iteration (the state reset is inside the timed region there too).
 * secs[t] = wall seconds of thread t; returns the wall seconds of the whole parallel region. */
double nnno_bench(const nnno_model *m, int n_threads, int iters, int kind, double *secs)
{
    init_tables();
    if (n_threads < 1) n_threads = 1;
    double wall = 0.0;
#ifdef _OPENMP
    double t_begin = 0.0;
#pragma omp parallel num_threads(n_threads)
    {
        const int tid = omp_get_thread_num();
#else
    {
        const int tid = 0;
#endif
        float *buf = (float *)malloc(sizeof(float) * 100 * FRAME_SIZE);
        float o[FRAME_SIZE];
        unsigned rng = 0x9E3779B9u * (unsigned)(tid + 1);
        const double f = kind ? 440.0 : 80.0 + 920.0 * ((double)(tid % 97) / 97.0);
        const double amp = kind ? 32767.0 : 3000.0 + 50.0 * (tid % 64);
        for (int i = 0; i < 100 * FRAME_SIZE; i++) {
            double v = amp * sin(2.0 * 3.14159265358979323846 * f * (double)i / 48000.0);
            if (!kind) {
                rng = rng * 1664525u + 1013904223u;
                v += 600.0 * ((double)(rng >> 8) / 8388608.0 - 1.0);
            }
            buf[i] = (float)floor(v + 0.5);
        }
        nnno_state *st = nnno_create(m);
        for (int t = 0; t < 4; t++) nnno_process_frame(st, o, buf + t * FRAME_SIZE);   /* warm the caches */
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
        t_begin = omp_get_wtime();
#pragma omp barrier
        const double t0 = omp_get_wtime();
#else
        const double t0 = 0.0;
#endif
        for (int it = 0; it < iters; it++) {
            if (kind) { const nnno_model *mm = st->model; memset(st, 0, sizeof(*st)); st->model = mm; }   /* DenoiseState::new() */
            for (int t = 0; t < 100; t++) nnno_process_frame(st, o, buf + t * FRAME_SIZE);
        }
#ifdef _OPENMP
        if (secs) secs[tid] = omp_get_wtime() - t0;
#pragma omp barrier
#pragma omp master
        wall = omp_get_wtime() - t_begin;
#endif
        nnno_destroy(st);
        free(buf);
    }
    return wall;
}

int nnno_run_streams(const nnno_model *m, int n_streams, int n_frames, const float *in, float *out,
                     float *vad, int32_t *pitch, float *gains, float *feats, int n_threads)
{
    return nnno_run_streams_cond(m, n_streams, n_frames, in, out, vad, pitch, gains, feats, NULL, n_threads);
}

/* ---- training-feature rows ------------------------------------------------------------------------ */

/* shift_and_filter_input (src/features.rs:97-104) + compute_frame_features (:115-219) on one DenoiseFeatures */
static int features_frame(nnno_state *st, const float *in)
{
    memmove(st->input_mem, st->input_mem + FRAME_SIZE, (PITCH_BUF_SIZE - FRAME_SIZE) * sizeof(float));
    biquad_hp(st->input_mem + PITCH_BUF_SIZE - FRAME_SIZE, st->mem_hp_x, in, FRAME_SIZE);
    return compute_frame_features(st);
}

/* ref: src/training.rs:113-160, the per-frame body: three DenoiseFeatures (clean, noise, mix) -> one 87-column row.
 * The NoiseSimulator that produces the three signals, the band cutoff and the vad label (:263-422) is the caller's. */
void nnno_training_rows(const nnno_model *m, int n_streams, int n_frames, const float *signal, const float *noise,
                        const float *combined, const int32_t *cutoff, const float *vad, float *rows, int n_threads)
{
    init_tables();
    int used = n_threads > 1 ? n_threads : 1;
    (void)used;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(used)
#endif
    for (int s = 0; s < n_streams; s++) {
        nnno_state *clean = nnno_create(m), *nz = nnno_create(m), *comb = nnno_create(m);
        for (int t = 0; t < n_frames; t++) {
            const size_t ft = ((size_t)s * n_frames + t) * FRAME_SIZE, ts = (size_t)t * n_streams + s;
            float *row = rows + ts * (NB_FEATURES + 2 * NB_BANDS + 1);
            features_frame(clean, signal + ft);                               /* :125,:131 */
            features_frame(nz, noise + ft);                                   /* :126,:132 */
            const int silence = features_frame(comb, combined + ft);          /* :127,:134 */
            const int cut = silence ? 0 : cutoff[ts];                         /* :135 */
            memcpy(row, comb->features, NB_FEATURES * sizeof(float));         /* :153 */
            for (int i = 0; i < NB_BANDS; i++) {
                float g = -1.0f;                                              /* :146-148 */
                if (i < cut && !(clean->ex[i] < 5e-2f && comb->ex[i] < 5e-2f)) {   /* :136-145 */
                    g = sqrtf((clean->ex[i] + 1e-3f) / (comb->ex[i] + 1e-3f));
                    g = g < 1.0f ? g : 1.0f;
                    if (g != g) g = 1.0f;                                     /* f32::min drops a NaN */
                }
                row[NB_FEATURES + i] = g;
                row[NB_FEATURES + NB_BANDS + i] = log10f(nz->ex[i] + 1e-2f);  /* :150-152 */
            }
            row[NB_FEATURES + 2 * NB_BANDS] = vad[ts];                        /* :156 */
        }
        nnno_destroy(clean); nnno_destroy(nz); nnno_destroy(comb);
    }
}

/* ---- the reference's multi-channel callers ------------------------------------------------------- */

/* ref: src/nnnoiseless.rs:301-331 (frame loop: `break 'outer` on a short read drops the partial frame; `first`
 * suppresses the first write) and :147-160 (RawFrameWriter: max(i16::MIN).min(i16::MAX).round() as i16). */
long nnno_cli_raw_i16(const nnno_model *m, const int16_t *in, long n, int channels, int16_t *out)
{
    nnno_state **st = (nnno_state **)malloc(sizeof(*st) * (size_t)channels);
    float ibuf[FRAME_SIZE], obuf[FRAME_SIZE];
    long written = 0;
    for (int c = 0; c < channels; c++) st[c] = nnno_create(m);
    for (long t = 0; (t + 1) * FRAME_SIZE <= n; t++) {
        for (int c = 0; c < channels; c++) {
            for (int i = 0; i < FRAME_SIZE; i++) ibuf[i] = (float)in[((size_t)t * FRAME_SIZE + i) * channels + c];
            nnno_process_frame(st[c], obuf, ibuf);
            if (t == 0) continue;
            for (int i = 0; i < FRAME_SIZE; i++) {
                float v = obuf[i];
                v = v > -32768.0f ? v : -32768.0f;   /* f32::max / f32::min return the non-NaN operand */
                v = v < 32767.0f ? v : 32767.0f;
                out[((size_t)(t - 1) * FRAME_SIZE + i) * channels + c] = (int16_t)roundf(v);
            }
        }
        if (t > 0) written += FRAME_SIZE;
    }
    for (int c = 0; c < channels; c++) nnno_destroy(st[c]);
    free(st);
    return written;
}

/* ref: src/signal.rs:83-137.  refill_out_bufs (:90-106) returns false without touching anything when the input is
 * already exhausted, otherwise reads 480 sample frames (equilibrium = 0 past the end), scales by 32768 (:98), runs
 * every channel's state and returns !exhausted.  The constructor refills twice (:83-87); next() (:116-137) hands out
 * out_bufs / 32768 clamped to [-1, 1] and refills after the 480th sample, rewinding only if that refill said true. */
/* one refill_out_bufs (src/signal.rs:90-106); returns its bool */
static int signal_refill(nnno_state **st, float *ob, const float *in, long n, int channels, long *consumed)
{
    float ibuf[FRAME_SIZE];
    if (*consumed >= n) return 0;
    for (int c = 0; c < channels; c++) {
        for (int i = 0; i < FRAME_SIZE; i++) {
            long k = *consumed + i;
            ibuf[i] = (k < n ? in[(size_t)k * channels + c] : 0.0f) * 32768.0f;
        }
        nnno_process_frame(st[c], ob + (size_t)c * FRAME_SIZE, ibuf);
    }
    *consumed += FRAME_SIZE;
    return *consumed < n;
}

long nnno_denoise_signal(const nnno_model *m, const float *in, long n, int channels, float *out)
{
    nnno_state **st = (nnno_state **)malloc(sizeof(*st) * (size_t)channels);
    float *ob = (float *)calloc((size_t)channels * FRAME_SIZE, sizeof(float));
    long consumed = 0, written = 0;
    for (int c = 0; c < channels; c++) st[c] = nnno_create(m);
    signal_refill(st, ob, in, n, channels, &consumed);   /* discard_first_frame, :83-87 */
    signal_refill(st, ob, in, n, channels, &consumed);
    do {                                                  /* next() until is_exhausted(), :108-137 */
        for (int i = 0; i < FRAME_SIZE; i++)
            for (int c = 0; c < channels; c++) {
                float v = ob[(size_t)c * FRAME_SIZE + i] / 32768.0f;
                if (v < -1.0f) v = -1.0f;
                if (v > 1.0f) v = 1.0f;
                out[(size_t)(written + i) * channels + c] = v;
            }
        written += FRAME_SIZE;
    } while (signal_refill(st, ob, in, n, channels, &consumed));
    for (int c = 0; c < channels; c++) nnno_destroy(st[c]);
    free(st);
    free(ob);
    return written;
}

/* ---- 16-tap sinc resampler of the CLI ------------------------------------------------ */

/* dasp_ring_buffer 0.11.0 Fixed<[f32; 16]>: push overwrites the oldest element, index i counts from the oldest and wraps
 * modulo the length (get() does not bounds-check the logical index).  dasp_interpolate 0.11.0 sinc::Sinc: idx counts
 * the frames pushed, up to depth = len / 2.  Neither crate is part of the reference tree: restated from their published
 * sources, UNPINNED by any reference test. */
typedef struct { float data[16]; int first; int idx; } sinc16;

static void sinc16_push(sinc16 *s, float x)            /* Sinc::next_source_frame */
{
    s->data[s->first] = x;
    s->first = (s->first + 1) & 15;
    if (s->idx < 8) s->idx++;
}
static float sinc16_at(const sinc16 *s, int i) { return s->data[(s->first + i) & 15]; }

static float sinc16_interpolate(const sinc16 *s, double x)   /* Sinc::interpolate */
{
    const double pi = 3.14159265358979323846;
    const double phil = x, phir = 1.0 - x;
    const int depth = 8, nl = s->idx, nr = s->idx + 1;
    const int rightmost = nl + depth, leftmost = nr - depth;
    const int max_depth = rightmost >= 16 ? 16 - depth : (leftmost < 0 ? depth + leftmost : depth);
    float v = 0.0f;
    for (int n = 0; n < max_depth; n++) {
        double a = pi * (phil + (double)n);
        double first = a == 0.0 ? 1.0 : sin(a) / a, second = 0.5 + 0.5 * cos(a / (double)depth);
        v += (float)(first * second * (double)sinc16_at(s, nl - n));
        a = pi * (phir + (double)n);
        first = a == 0.0 ? 1.0 : sin(a) / a;
        second = 0.5 + 0.5 * cos(a / (double)depth);
        v += (float)(first * second * (double)sinc16_at(s, nr + n));
    }
    return v;
}

/* ref: src/nnnoiseless.rs:19-32 (resampled), :106-131 (Resample::next_sample): `in` = n sample frames of `channels` interleaved
 * floats at the source rate, ratio = source_rate / 48000; writes up to cap output sample frames and returns their number
 * (the loop ends when an output needs a source sample that is not there). */
long nnno_resample(const float *in, long n, int channels, double ratio, float *out, long cap)
{
    sinc16 *st = (sinc16 *)calloc((size_t)channels, sizeof(sinc16));
    double pos = 0.0;
    long consumed = 0, written = 0;
    while (written < cap) {
        int ended = 0;
        pos += ratio;
        while (pos >= 1.0) {
            pos -= 1.0;
            if (consumed >= n) { ended = 1; break; }
            for (int c = 0; c < channels; c++) sinc16_push(&st[c], in[(size_t)consumed * channels + c]);
            consumed++;
        }
        if (ended) break;
        for (int c = 0; c < channels; c++) out[(size_t)written * channels + c] = sinc16_interpolate(&st[c], pos);
        written++;
    }
    free(st);
    return written;
}
