/*
 * nnn_oracle.h -- CPU oracle for the nnnoiseless `DenoiseState::process_frame` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nnnoiseless_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the timed CPU baseline -- never as the thing shipped.
 *
 * It is a plain-C restatement (not a copy) of the reference algorithm, scalar f32, with the
 * reference's summation orders, compiled with -ffp-contract=off.  Each function in
 * nnn_oracle.c cites the reference file:line it follows (paths relative to the
 * jneem/nnnoiseless tree, crate v0.5.1).
 *
 * Parity pin: tests/test_oracle_golden.py runs test_data/testing.raw through this oracle and
 * checks reference_output.raw with the reference's own metric (src/lib.rs:184-213, < 1e-4).
 * The FFT arithmetic lives in un-vendored crates (easyfft 0.4.2 -> realfft 3.5.0 ->
 * rustfft 6.4.1); it is restated as a mixed-radix FFT here, un-normalised in both
 * directions like the crates, and pinned only end-to-end by that golden test.
 * The genuine Rust reference cannot be built here (no cargo/rustc), so there is no
 * oracle/_ref.
 */
#ifndef NNN_ORACLE_H
#define NNN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNNO_FRAME_SIZE 480
#define NNNO_WINDOW_SIZE 960
#define NNNO_FREQ_SIZE 481
#define NNNO_NB_BANDS 22
#define NNNO_NB_FEATURES 42
#define NNNO_PITCH_BUF 1728
#define NNNO_XLP 864
#define NNNO_MAX_NEURONS 128

typedef struct nnno_model nnno_model;
typedef struct nnno_state nnno_state;

/* Intermediate quantities of the most recent process_frame call (per-stage known answers). */
typedef struct nnno_taps {
    float filtered[NNNO_FRAME_SIZE];   /* high-passed input appended to the history        */
    float xlp[NNNO_XLP];               /* pitch_buf after pitch_downsample                  */
    float ac[5];                       /* autocorrelation after noise floor + lag window    */
    float lpc2[5];                     /* 5-tap FIR coefficients                            */
    float xcorr1[147];                 /* coarse (4x decimated) cross-correlation           */
    int32_t best1[2];                  /* best / second-best coarse lag                     */
    float xcorr2[294];                 /* fine (2x decimated) cross-correlation, sparse     */
    int32_t pitch_search;              /* return value of pitch_search                      */
    int32_t pitch_idx;                 /* final pitch period after remove_doubling          */
    float pitch_gain;
    float X[2 * NNNO_FREQ_SIZE];       /* spectrum (re,im) after transform_input, lag 0     */
    float P[2 * NNNO_FREQ_SIZE];       /* spectrum of the pitch-lagged window               */
    float ex[NNNO_NB_BANDS], ep[NNNO_NB_BANDS], exp_[NNNO_NB_BANDS];
    float features[NNNO_NB_FEATURES];
    int32_t silence;
    float g_raw[NNNO_NB_BANDS];        /* RNN gains before smoothing                        */
    float g[NNNO_NB_BANDS];            /* gains after max(g, 0.6 lastg)                     */
    float vad;
    float out[NNNO_FRAME_SIZE];
} nnno_taps;

/* .rnn parser; NULL on malformed / wrongly-shaped input (same rules as the reference). */
nnno_model *nnno_model_from_bytes(const uint8_t *bytes, size_t len);
void nnno_model_free(nnno_model *m);
/* layer sizes: [dense_in, dense_out, vad_n, noise_n, denoise_n, out_n] + 6 activations */
void nnno_model_shape(const nnno_model *m, int32_t shape[12]);

nnno_state *nnno_create(const nnno_model *m); /* model is borrowed and must outlive the state */
void nnno_destroy(nnno_state *st);
float nnno_process_frame(nnno_state *st, float *out, const float *in); /* out may alias in */
void nnno_get_taps(const nnno_state *st, nnno_taps *taps);

/*
 * Run n_streams independent states for n_frames frames each (fresh state per stream).
 *   in   [n_streams][n_frames][480]
 *   out  [n_streams][n_frames][480]      (may be NULL)
 *   vad  [n_streams][n_frames]           (may be NULL)
 *   pitch[n_streams][n_frames] int32     (may be NULL)
 *   gains[n_streams][n_frames][22]       (may be NULL)  smoothed gains, zeros on silent frames
 *   feats[n_streams][n_frames][42]       (may be NULL)
 * n_threads <= 1 runs single-threaded; otherwise streams are statically partitioned (OpenMP).
 * Returns the number of threads actually used.
 */
int nnno_run_streams(const nnno_model *m, int n_streams, int n_frames, const float *in,
                     float *out, float *vad, int32_t *pitch, float *gains, float *feats,
                     int n_threads);

/* The same, plus cond[n_streams][n_frames] = nnno_frame_condition of every frame (may be NULL). */
int nnno_run_streams_cond(const nnno_model *m, int n_streams, int n_frames, const float *in, float *out,
                          float *vad, int32_t *pitch, float *gains, float *feats, float *cond, int n_threads);
/* Everything run_streams_cond returns plus, per frame: branch = nnno_frame_branch, g_raw[22] (RNN gains before
 * smoothing), exp_[22] (normalised pitch correlation).  Any output pointer may be NULL. */
int nnno_run_streams_full(const nnno_model *m, int n_streams, int n_frames, const float *in, float *out, float *vad,
                          int32_t *pitch, float *gains, float *feats, float *cond, int32_t *branch, float *g_raw,
                          float *exp_, int n_threads);
/* bit i < 22: pitch_filter's `exp > g` branch taken in band i (src/features.rs:227); bit 22: silent frame. */
int32_t nnno_frame_branch(const nnno_state *st);
/* tansig_approx / sigmoid_approx on their own (src/util.rs:29-53) */
float nnno_tansig(float x);
float nnno_sigmoid(float x);
/* CPU baseline timing: see nnn_oracle.c */
double nnno_bench(const nnno_model *m, int n_threads, int iters, int kind, double *secs);
/* Smallest |exp - g| over the bands where pitch_filter's branch (src/features.rs:229-236) is a jump discontinuity
 * (large if none): frames where it is tiny are decided by FFT rounding noise in the REFERENCE itself. */
float nnno_frame_condition(const nnno_state *st);

/*
 * The reference's two multi-channel callers of process_frame, restated around the oracle state.
 * Both return the number of sample frames (one sample per channel) written to `out`.
 *
 * nnno_cli_raw_i16: the CLI's raw-PCM loop, src/nnnoiseless.rs:301-331 with RawFrameWriter :147-160 --
 *   `in` = n sample frames of `channels` interleaved int16; a trailing partial frame is dropped, the first processed
 *   frame is not written, every sample is clamped to the i16 range and rounded half away from zero.
 * nnno_denoise_signal: the dasp adapter DenoiseSignal, src/signal.rs:83-137, fed by an n-sample-frame signal of
 *   unit-range floats that reports exhaustion once all n were consumed (dasp_signal 0.11.0 `from_iter`, a crate that
 *   is not part of the reference tree: its end-of-signal behaviour is restated from its documentation, unpinned).
 *   `out` needs room for (n / 480 + 2) * 480 sample frames.
 */
long nnno_cli_raw_i16(const nnno_model *m, const int16_t *in, long n, int channels, int16_t *out);
long nnno_denoise_signal(const nnno_model *m, const float *in, long n, int channels, float *out);

/*
 * The per-frame body of the reference's training-data generator, src/training.rs:113-160: per stream three
 * DenoiseFeatures states (clean, noise, mix), one row [42 features of the mix | 22 gains | 22 noise levels | vad].
 *   signal / noise / combined [n_streams][n_frames][480];  cutoff, vad [n_frames][n_streams];  rows [n_frames][n_streams][87]
 */
void nnno_training_rows(const nnno_model *m, int n_streams, int n_frames, const float *signal, const float *noise,
                        const float *combined, const int32_t *cutoff, const float *vad, float *rows, int n_threads);

/* The CLI's resampler to 48 kHz (src/nnnoiseless.rs:19-32, 106-131; dasp_interpolate 0.11.0 Sinc<[f32; 16]>, a crate outside
 * the reference tree: restated from its published source, unpinned).  See nnn_oracle.c. */
long nnno_resample(const float *in, long n, int channels, double ratio, float *out, long cap);

/* Stand-alone FFT entry points so tests can pin the restated FFT against a naive DFT. */
void nnno_rfft960(const float *in960, float *out_re_im_481x2);   /* un-normalised forward  */
void nnno_irfft960(const float *in_re_im_481x2, float *out960);  /* un-normalised inverse  */

/* Tables (for cross-checking the device-side tables in tests). */
void nnno_get_tables(float *window960, float *dct22x22, float *wnorm, float *tansig201);

#ifdef __cplusplus
}
#endif
#endif
