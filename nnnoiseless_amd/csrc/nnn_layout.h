// nnn_layout.h -- sizes, HBM layouts and the kernel argument block of the batched
// process_frame path (reference: jneem/nnnoiseless src/denoise.rs:95-116).
//
// Two canonical layouts (S_pad = streams rounded up to a multiple of 64):
//   SM ("stream-major")     a[s * LEN + i]                      one stream contiguous; used by the
//                            wave-per-stream kernels (FFT, data-dependent-lag inner products)
//   TI ("tile-interleaved")  a[((s / 64) * LEN + i) * 64 + s % 64]   64 streams of one tile side by
//                            side; lane = stream kernels read/write one 256-byte row per wave
//                            instruction (fully coalesced) and every lane runs the reference's scalar
//                            recurrence in the reference's order.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nnn {

constexpr int FRAME = 480;        // src/lib.rs:38
constexpr int WINDOW = 960;       // src/lib.rs:39
constexpr int FREQ = 481;         // src/lib.rs:41
constexpr int NB = 22;            // src/lib.rs:49
constexpr int NFEAT = 42;         // src/lib.rs:53
constexpr int CEPS_MEM = 8;       // src/lib.rs:50
constexpr int HIST = 1728;        // PITCH_BUF_SIZE, src/lib.rs:46
#ifndef NNN_GROUP
#define NNN_GROUP 24
#endif
#ifndef NNN_DEPTH
#define NNN_DEPTH 2
#endif
constexpr int GROUP = NNN_GROUP;  // frames per launch: every kernel is launched once per group of up to GROUP consecutive frames (kernels
                                  // with a frame-to-frame recurrence loop over the group's frames inside the launch).  24 since round 3
                                  // (16 before): a pipelined call is cut into an even number of near-equal groups, so the bench's
                                  // 48-frame call is 2 x 24 instead of 3 x 16 -- one group per lane, fewer pipeline ticks and
                                  // prologues per frame: 55.5 -> 57.2 M frames/s at 4096 streams, same at 65536; 64-frame calls stay
                                  // 4 x 16.  The price is memory: 48 scratch sets and 76 ring slots per stream instead of 32 / 52.
constexpr int DEPTH = NNN_DEPTH;  // groups in flight (blocks of GROUP scratch sets in rotation)
// A batch is sized for groups of up to `gmax` <= GROUP frames (GROUP unless the host says its calls are shorter: a real-time host
// that ticks one frame per call asks for gmax = 1 and pays a seventeenth of the memory, nnn_batch_create_opts):
//   scratch sets   depth * gmax, depth = the groups a batch keeps in flight behind the high-pass: 1 (one lane, the default since the
//                  end of round 3) or DEPTH = 2 (NNN_LANES >= 2 / NNN_SCHED=stages at creation)
//   ring slots     (depth + 1) * gmax + 4: the high-pass may run a group ahead of the depth groups in flight, whose oldest frame
//                  still reads a 1728-sample history (3 slots behind it).  History ring instead of the reference's memmove.
constexpr int NSET = DEPTH * GROUP;              // the most scratch sets a batch can have
__host__ __device__ inline int slots_for(int gmax, int depth = DEPTH) { return (depth + 1) * gmax + 4; }
__host__ __device__ inline int ring_len(int nslot) { return nslot * 480; }
__host__ __device__ inline int hist_stride(int nslot) { return nslot * 480 + 32; }   // a stream's stride in the history array (whole 128-byte lines):
                                  // hist[ring_len] repeats hist[0], so a sample pair that starts on the ring's last sample is still one 8-byte read
constexpr int XLP = 864;          // HIST / 2
constexpr int PITCH_MIN = 60, PITCH_MAX = 768;
constexpr int NLAG1 = 147;        // coarse lags  (PITCH_MAX - 3*PITCH_MIN) / 4
constexpr int NLAG2 = 294;        // fine lags
constexpr int TILE = 64;
constexpr int FSTR = 512;         // row stride (float2) of the spectra in memory.  A spectrum is stored as the transforms hold it: lane j of the
                                  // stream's wave owns bins k = j + 64 u (u < 4, k <= 240) and their mirror images 480 - k, and element pair
                                  // 2 (64 u + j), + 1 holds (bin k, bin 480 - k) -- one 16-byte store per pair in k_fft_xp, one 16-byte load in
                                  // k_synth, lanes contiguous (until round 4: natural bin order, 8-byte accesses at 0.54-0.70 of the 16-byte rate).
                                  // Bin 240 pairs with itself, pairs of k > 240 do not exist (slots left alone).
constexpr int MAXN = 127;         // layer sizes are non-negative i8 (src/rnn.rs:128-134)

struct ModelDims {
    int nd, nv, nn, ndn;                       // input_dense, vad_gru, noise_gru, denoise_gru neurons
    int act_d, act_v, act_n, act_dn, act_o, act_vo;
};

// RNN as batched GEMMs on the matrix cores (k_rnn, k_rnn_wf).  Activations of a 64-stream tile live in LDS as
// [stream][column] matrices in three bf16 planes (x = hi + mid + lo exactly), weights are small integers
// (exact in bf16) pre-packed on the host in MFMA B-fragment order [neuron block][gate][k-step][lane][8].
struct GemmDesc {
    int wofs;     // uint4 index of neuron block 0 in the packed weight buffer
    int ksteps;   // k-steps of 32 columns
    int kbase;    // first LDS column
    int ngates;   // gates packed per neuron block (3 for a GRU, 1 for a dense layer)
};
struct LayerDesc {
    GemmDesc in, rec;
    int n;            // neurons
    int nb;           // neuron blocks of 16
    int act;
    int bias;         // float index into the f32 parameter buffer ([gate][n])
    int out_col;      // LDS column of the input matrix that receives this layer's output
};
struct RnnPlan {
    int in_w, rec_w;  // row strides (bf16 elements) of the input and recurrent LDS matrices
    int cF, cV;       // LDS columns of the features and of the vad-GRU state
    LayerDesc dense, vad, noise, dn, out;
    int vo_w, vo_b, act_vo;   // vad output layer (1 x nv), f32 parameters
};

struct Buffers {
    // ---- persistent per-stream state (src/denoise.rs:37-42, features.rs:18-46, pitch.rs:4-17, rnn.rs:65-70)
    int nslot;           // history ring slots of this batch (slots_for(gmax))
    float *hist;         // SM [hist_stride(nslot)]   high-passed input history, ring of nslot frame slots (+ the wrap-around sample)
    float *hp_mem;       // TI [2]      biquad state
    float *hp_last;      // TI [1]      last filtered sample of the previous frame
    float *dec;          // TI [dec_len(nslot)]  2:1 decimated history: ring of nslot x 240 values whose first 960 are mirrored behind its end,
                         //             so that the 864-value window of any frame is one contiguous run (240 values are new per frame)
    float *lpc_head;     // [NT][5][64]  one-frame calls: the five autocorrelation sums over the part of the window older than the frame (k_hp2 -> k_pitch)
    float *xlp0;         // TI [nslot]  per ring slot: pitch_downsample's special first element (x[1]/2 + x[0])/2 of that frame
    float *lpc;          // TI [nslot * 10]  per ring slot: that frame's windowed autocorrelation ac[5] and FIR taps lpc2[5], k_lpc -> k_pitch.
                         //             Kept by ring slot, not by scratch set: k_lpc rides on the high-pass stream, which runs ahead of the
                         //             groups in flight, and the ring's slots are what that stream already waits for
    float *ceps_mem;     // TI [8*22]
    int *mem_id;         // TI [1]
    float *synth_mem;    // SM [480]
    float *lastg;        // TI [22]
    int *last_period;    // TI [1]
    float *last_gain;    // TI [1]
    float *gru_v, *gru_n, *gru_dn;  // SM, tile t at t * 64 * gru_*_w (the widest resident model), rows of the tile's own width
    int gru_v_w, gru_n_w, gru_dn_w;
    // ---- per-frame scratch (doubles as the parity taps)
    float *xlp_ti;       // TI [864]    pitch_buf
    float *xc1;          // TI [147]    coarse cross-correlation (stored only while taps are on: it lives in LDS otherwise)
    int *best1;          // TI [2]
    float *xc2;          // TI [10]     fine xcorr at 2*best-2..+2, 2*second-2..+2
    int *psearch;        // TI [1]
    int *pitch;          // TI [1]
    int *pflag;          // TI [1]      per quarter tile (its first stream's entry): number of the frame whose pitch and gain are in memory
    float *pgain;        // TI [1]
    float2 *X, *P;       // SM [FSTR]   spectra as (bin k, bin 480 - k) pairs in the transforms' lane order (see FSTR; spectrum_index); the
                         //             fused back end (k_back) keeps them in registers and writes them for the parity taps only
    float *ex, *ep, *exp_;  // TI [22]
    float *cn;           // TI [28]     the frame's own cepstrum (22) and pitch-correlation DCT (6), made at the end of k_fft_p
    float *feat;         // TI [42]
    int *silence;        // TI [1]
    int *branch;         // TI [1]      bit i: pitch_filter took `exp > g` in band i (ref: src/features.rs:227); bit 22: silent frame
    float *g_raw, *g;    // TI [22]
    float *vad;          // TI [1]
    // ---- read-only tables
    const float *window;     // [960]
    const float *window_a;   // [960]  window / 2 for the analysis transforms: the real-transform split step's 1/2 folded (exactly) into the
                             //        multiply every sample gets anyway
    const float *window_s;   // [960]  window / 2 for the synthesis (ref: src/features.rs:263-275 halves the inverse transform)
    const float *dct;        // [22*22]
    const float2 *tw960;     // [960]  exp(-2 pi i k / 960)
    const float *tansig;     // [201]
    const float *bin_frac;   // [400]  j / band_size
    const int *bin_band;     // [400]
    long long *stamps;       // [64] optional phase time stamps of block 0 (built with -DNNN_STAMPS)
    int *fault;              // [1]  set by a kernel that gave up waiting for another workgroup's flag (k_pitch); page-locked host memory
                             //      mapped into the device, so the host sees it without a copy (checked at every call, sticky until reset)
    unsigned *ticket;        // [1]  chained k_pitch launches: workgroups take their work item in the order they START (see k_pitch)
    int dbg_withhold;        // test hook: the frame number whose hand-off flag is never published (0 = none)
    long long handoff_ticks; // how long a pitch workgroup waits for its predecessor's flag before it raises the fault: ticks of the 100 MHz
                             // constant clock (s_memrealtime), i.e. wall time -- a slow but live predecessor (a shared GPU, a debugger
                             // stall) is not a lost hand-off; 10 s unless NNN_HANDOFF_TIMEOUT_MS says otherwise, 0.2 s under the test hook
    const int *seg;          // [192] band-sum segments: k0[64], count[64], first segment[32], segments[32] per interval
    const void *fft_img;     // the transform kernels' LDS tables (twiddles, band weights, bands, segments) in their LDS layout (FftLds)
    float wnorm;
    int S, S_pad, NT;
    int taps;                // != 0: kernels also store the quantities only parity tests look at (xc1, xc2, P from bin 400 up)
};

// Per-frame scratch of set f lies f * S_pad * LEN elements after set 0 in every scratch array, so a launch that covers
// several consecutive frames (block index = frame * blocks_per_frame + block) reaches its frame's set by offsetting.
// (the first six are written only while the parity taps are on -- everything they hold stays in LDS otherwise -- and are
// allocated when the taps are first switched on)
#define NNN_TAP_FIELDS(F) F(xlp_ti, XLP) F(xc1, NLAG1) F(best1, 2) F(xc2, 10) F(psearch, 1)
#define NNN_WORK_FIELDS(F)                                                                                           \
    F(pitch, 1) F(pflag, 1) F(pgain, 1) F(X, FSTR) F(P, FSTR) F(ex, NB) F(ep, NB) F(exp_, NB) F(cn, 28) F(feat, NFEAT)      \
    F(silence, 1) F(branch, 1) F(g_raw, NB) F(g, NB) F(vad, 1)
#define NNN_SCRATCH_FIELDS(F) NNN_TAP_FIELDS(F) NNN_WORK_FIELDS(F)
__host__ __device__ inline Buffers frame_view(Buffers b, int f)
{
    const size_t sp = (size_t)b.S_pad * (size_t)f;
#define NNN_F(name, len) b.name += sp * (size_t)(len);
    NNN_SCRATCH_FIELDS(NNN_F)
#undef NNN_F
    return b;
}

// Per-frame launch parameters: a call's table (one entry per frame, filled on the device by k_fill_params) lives in device
// memory, and a group's launches get a pointer to their first frame's entry.
// PCM sample formats at the batch boundary (values match enum nnn_pcm_format in include/nnn_batch.h)
enum { PCM_F32 = 0, PCM_I16 = 1, PCM_F32_UNIT = 2 };
__host__ __device__ inline int pcm_elem_bytes(int fmt) { return fmt == PCM_I16 ? 2 : 4; }

struct StepParams {
    // sample i of stream s (channel s % channels of group s / channels), all in BYTES:
    //   in + (s / channels) * group_stride + ((s % channels) + i * channels) * elem
    const char *in;
    char *out;
    float *vad;        // [n_streams] for this frame, may be null
    long long group_stride, frame_stride;   // bytes
    int fmt, channels;
    int discard;       // frame tables: != 0 = this frame's audio is not written; call parameters: frames to drop
    int slot;          // history ring slot that receives this frame (frame index mod nslot)
    int n_streams;
    // optional per-frame record for parity tests (nnn_batch_set_frame_log): [n_streams][FRAME_LOG_WORDS] words, or null.
    // Frame tables: this frame's record; call parameters: the first frame's record and the frames that still have room.
    unsigned *log;
    int log_frames;
};
constexpr int FRAME_LOG_WORDS = 2 + NB;   // pitch index, branch mask, the 22 smoothed band gains (f32 bits)

// where bin k (0 .. 480) of a spectrum sits in its FSTR-long row (float2 index)
__host__ __device__ inline int spectrum_index(int k)
{
    const int kk = k <= 240 ? k : 480 - k;   // the pair's lower bin: lane kk % 64, slot kk / 64
    return 2 * kk + (k <= 240 ? 0 : 1);      // (64 u + j = kk)
}
// the same for the pitch-lagged spectrum P (spectrum_store_p: slot 0's bins alone, then pairs; bins 417 .. 480 behind them, taps only)
__host__ __device__ inline int spectrum_index_p(int k)
{
    const int kk = k <= 240 ? k : 480 - k;
    if (kk < 64) return k <= 240 ? kk : 64 + 2 * 177 + kk;
    return 64 + 2 * (kk - 64) + (k <= 240 ? 0 : 1);
}
// ring position of logical input_mem[0] when the newest frame sits in slot `slot`
__host__ __device__ inline int ring_base(int slot, int nslot)
{
    const int x = FRAME * slot - (HIST - FRAME);
    return x < 0 ? x + ring_len(nslot) : x;
}
// ring position of logical decimated index 0 (864 logical values, the newest 240 in slot `slot`)
constexpr int DEC_MIRROR = 4;               // frames whose 240 values are stored twice (slots 0..3 cover the 864-value overhang)
__host__ __device__ inline int dec_ring_len(int nslot) { return nslot * 240; }
__host__ __device__ inline int dec_len(int nslot) { return (nslot + DEC_MIRROR) * 240; }
__host__ __device__ inline int dec_base(int slot, int nslot)
{
    const int x = 240 * slot - (XLP - 240);
    return x < 0 ? x + dec_ring_len(nslot) : x;
}

}  // namespace nnn
