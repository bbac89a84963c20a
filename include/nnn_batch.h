// SPDX-License-Identifier: BSD-3-Clause
/*
 * nnn_batch.h -- batched entry points of the MI355X process_frame backend.
 *
 * The reference has no batched API: its only "N independent states in lock-step" call sites are
 * the per-channel loops `for ch { states[ch].process_frame(out[ch], in[ch]) }` at
 * src/signal.rs:102-104 and src/nnnoiseless.rs:318-320.  nnn_batch_process_* replaces exactly
 * that loop: n_streams independent DenoiseState's (src/denoise.rs:37-42) advanced by n_frames
 * calls of DenoiseState::process_frame (src/denoise.rs:95-116) each.
 *
 * Plain C ABI: pointers and sizes only.  All functions return 0 on success, non-zero on error
 * (nnn_last_error() has the text); nothing falls back to a CPU path.
 */
#ifndef NNN_BATCH_H
#define NNN_BATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nnn_batch nnn_batch;
#ifndef RNNOISE_H
typedef struct RNNModel RNNModel;
#endif

#define NNN_FRAME_SIZE 480 /* DenoiseState::FRAME_SIZE, src/denoise.rs:46 */

/* RnnModel::from_bytes (src/rnn.rs:75-77, validation :116-232): NULL where the reference returns None. */
RNNModel *nnn_model_from_bytes(const uint8_t *bytes, size_t len);
/* RnnModel::default (src/rnn.rs:235-240): the built-in weights.rnn. */
RNNModel *nnn_model_default(void);
/* RNNoise / rnnoise-nu TEXT model ("rnnoise-nu model file version 1" + whitespace-separated integers) -> the binary
 * .rnn format: what the reference's train/convert_rnnoise.py:18-29 does (drop the header line, every integer modulo
 * 256 as one byte).  nnn_convert_rnnoise_text writes the bytes to out (cap bytes of room) and returns their count, or
 * -1 on a wrong header / a token that is not an integer / too little room (pass out = NULL to get the count only).
 * nnn_model_from_rnnoise_text = convert + nnn_model_from_bytes. */
long nnn_convert_rnnoise_text(const char *text, size_t len, uint8_t *out, size_t cap);
RNNModel *nnn_model_from_rnnoise_text(const char *text, size_t len);
void nnn_model_free(RNNModel *m);
/* RnnModel is Clone (#[derive(Clone)], src/rnn.rs:54): an independent copy of the parameters, freed with nnn_model_free.
 * NULL on a NULL model. */
RNNModel *nnn_model_clone(const RNNModel *m);
/* shape[0..5] = input_dense in/out, vad/noise/denoise GRU neurons, gains out; shape[6..11] = activations */
void nnn_model_shape(const RNNModel *m, int32_t shape[12]);

/* n_streams x DenoiseState::with_model(model) (src/denoise.rs:72-74); model NULL = DenoiseState::new().
 * The model is copied to the device; it need not outlive the batch.  device = HIP device ordinal. */
nnn_batch *nnn_batch_create(const RNNModel *model, int n_streams, int device);
/* Several models resident at once (e.g. the rnnoise-nu model zoo, one model per audience): streams
 * [0, group_streams[0]) run models[0], the next group_streams[1] run models[1], ...  Every group but the last must be
 * a multiple of 64 streams (the kernels' tile).  models == NULL or models[i] == NULL selects the built-in model.
 * Everything except the RNN kernel is model-independent; the RNN runs as one launch per group. */
nnn_batch *nnn_batch_create_grouped(const RNNModel *const *models, const int *group_streams, int n_groups, int device);
/* The same with options (NULL = defaults; zero-initialise the struct).
 * max_group_frames: the kernels work on groups of consecutive frames -- up to 24 by default, which sizes the per-stream scratch
 * and history rings for it (360 KB per stream).  A host whose calls are short says so here and gets a batch sized for groups of
 * that many frames: a real-time host that ticks ONE 10 ms frame per call passes 1 and pays 33 KB per stream (the reference's
 * DenoiseState is 10.3 KB, src/features.rs:18-46), i.e. 8.7 million live streams' worth of HBM instead of 800 thousand.  Longer
 * calls still work on such a batch, cut into groups of at most this many frames (slower, same results).  0 = default;
 * values above 24 (the kernels' longest group) are rejected.  How many groups a batch keeps in flight (1, or 2 under
 * NNN_LANES >= 2 / NNN_SCHED=stages) is fixed when it is created -- nnn_batch_set_schedule on an existing batch works with
 * the scratch sets and ring slots that are there -- and is part of what a state snapshot must match. */
typedef struct nnn_batch_opts {
    int max_group_frames;
    int reserved[7];                     /* must be zero */
} nnn_batch_opts;
nnn_batch *nnn_batch_create_opts(const RNNModel *const *models, const int *group_streams, int n_groups, int device,
                                 const nnn_batch_opts *opts);
int nnn_batch_max_group_frames(const nnn_batch *b);
/* Device memory the batch holds (state, scratch, tables, weights), bytes. */
size_t nnn_batch_device_bytes(const nnn_batch *b);
void nnn_batch_destroy(nnn_batch *b);
int nnn_batch_num_streams(const nnn_batch *b);
/* Back to freshly-created state (all zeros, src/features.rs:58-74). */
int nnn_batch_reset(nnn_batch *b);

/* DenoiseState is Clone (src/denoise.rs:36): a second batch with the same models and a copy of every stream's state as of
 * the calls made so far (device-to-device copy; waits for them).  NULL on failure.  The two continue independently and, fed
 * the same input, bit-identically. */
nnn_batch *nnn_batch_clone(nnn_batch *b);
/* The same state as host bytes: nnn_batch_state_bytes() of them.  A snapshot only loads into a batch of the same stream
 * count and models made by the same build of the library (it is a raw image of the state slab, not an interchange
 * format). */
size_t nnn_batch_state_bytes(const nnn_batch *b);
int nnn_batch_save_state(nnn_batch *b, void *host_dst, size_t dst_bytes);
int nnn_batch_load_state(nnn_batch *b, const void *host_src, size_t src_bytes);

/*
 * n_frames x process_frame for every stream, buffers resident in device memory.
 *   sample i of frame t of stream s:  d_in [s * stream_stride + t * frame_stride + i]   (floats)
 *                                     d_out[s * stream_stride + t * frame_stride + i]   (may alias d_in)
 *   VAD probability (return value of process_frame): d_vad[t * n_streams + s]           (NULL to skip)
 * hip_stream: a hipStream_t to enqueue on.  NULL does NOT mean HIP's null stream: it means the batch's OWN non-blocking stream, which
 * is not ordered with anything the caller enqueued elsewhere -- a caller that produces d_in on a stream of its own passes that
 * stream (recommended) or synchronises it first.  Asynchronous: the call returns when the work is enqueued.
 */
int nnn_batch_process_device(nnn_batch *b, const float *d_in, float *d_out, float *d_vad, int n_frames,
                             size_t stream_stride, size_t frame_stride, void *hip_stream);
/* Same with host buffers (copies over PCIe, synchronous).  This is the shape of the reference's own interface -- process_frame
 * takes host slices (src/denoise.rs:95) -- and its rate is the bus's, not the kernels': a call of many gap-free frames
 * (frame_stride == 480 * channels) is cut into about sixteen chunks of 1 to 16 frames, chunk i + 1 going up and chunk i - 1 coming back
 * while chunk i is processed (same bits as one piece).  Buffers from nnn_host_alloc (page-locked) are transferred by DMA, both directions at once;
 * any other host memory works through the runtime's staging copies at a fraction of that rate. */
int nnn_batch_process_host(nnn_batch *b, const float *in, float *out, float *vad, int n_frames,
                           size_t stream_stride, size_t frame_stride);
/* Page-locked host memory for the *_host entry points (hipHostMalloc / hipHostFree underneath).  NULL on failure. */
void *nnn_host_alloc(size_t bytes);
void nnn_host_free(void *p);

/*
 * The same with the sample formats and channel interleave of the reference's callers fused into the first and last
 * kernels, so a CLI / DenoiseSignal style per-channel loop is ONE batched call on packed PCM:
 *   NNN_PCM_F32       floats in i16 range, what process_frame itself takes (src/denoise.rs:86-90)
 *   NNN_PCM_I16       int16 in; out = round-half-away(clamp(x, -32768, 32767)) as the CLI's frame writers do
 *                     (src/nnnoiseless.rs:147-177)
 *   NNN_PCM_F32_UNIT  floats in [-1, 1]: in * 32768, out = clamp(x / 32768, -1, 1)  (DenoiseSignal,
 *                     src/signal.rs:95-100 and :123-127)
 * Streams are `channels`-interleaved groups: stream s is channel s % channels of group s / channels, and
 *   sample i of frame t of stream s = buf[(s / channels) * group_stride + t * frame_stride + i * channels + s % channels]
 * in ELEMENTS of the format (a C-channel 16-bit file is one group with frame_stride = 480 * C).  n_streams must be a
 * multiple of channels.  discard_first != 0 reproduces the callers' dropped first frame (src/nnnoiseless.rs:322-330,
 * src/signal.rs:83-87): if the batch has processed no frame since create/reset, frame 0 produces no audio and the
 * call writes n_frames - 1 frames starting at frame position 0 of d_out.  VAD values are written for every frame.
 */
enum nnn_pcm_format { NNN_PCM_F32 = 0, NNN_PCM_I16 = 1, NNN_PCM_F32_UNIT = 2 };
typedef struct nnn_pcm_layout {
    int format;          /* enum nnn_pcm_format */
    int channels;
    int discard_first;
    int reserved;        /* 0 */
    size_t group_stride; /* elements */
    size_t frame_stride; /* elements, >= 480 * channels */
} nnn_pcm_layout;
int nnn_batch_process_pcm_device(nnn_batch *b, const void *d_in, void *d_out, float *d_vad, int n_frames,
                                 const nnn_pcm_layout *layout, void *hip_stream);
int nnn_batch_process_pcm_host(nnn_batch *b, const void *in, void *out, float *vad, int n_frames,
                               const nnn_pcm_layout *layout);
int nnn_batch_synchronize(nnn_batch *b);
/* 1 if a pitch workgroup of an earlier call ran out of patience waiting for the previous frame's result (the frames of a group
 * run side by side below 16 384 streams and hand the last pitch from workgroup to workgroup): the state of the affected streams
 * is invalid from that frame on.  Work items are handed out in the order workgroups start, so the wait cannot deadlock whatever
 * order the hardware dispatches them in; the condition exists as a safety net, and what trips it is wall time (10 s without the
 * predecessor's flag), not a spin count: a predecessor slowed by a shared GPU is waited for.  It is sticky: every later process call and
 * nnn_batch_synchronize fail with it until nnn_batch_reset or nnn_batch_load_state.  Cheap (a read of page-locked host memory the
 * device writes into): callers that synchronise their own stream instead of calling nnn_batch_synchronize can poll it. */
int nnn_batch_fault(const nnn_batch *b);
/* Test hook for the above: withhold the hand-off flag of the frame `frames_ahead` frames from now (negative: off). */
int nnn_batch_debug_withhold_flag(nnn_batch *b, int frames_ahead);

/* Parity taps: intermediate quantities of the most recent frame, copied to the host as
 * [n_streams][len] (float32 or int32, see nnn_tap_info).  Test/diagnostic interface.  Everything inside the pitch kernel
 * (XLP, XCORR1, BEST1, XCORR2C, PITCH_SEARCH), X, P and FEATURES are quantities the kernels keep on
 * chip: they are stored to device memory only after nnn_batch_set_taps(batch, 1) (which also allocates their arrays), and
 * reading them without it is an error.  With taps at 1 the coarse pitch search computes all 147 cross-correlations exactly (the full
 * search, so that XCORR1 is complete); nnn_batch_set_taps(batch, 2) stores the same taps from the certified search production runs take --
 * XCORR1 then holds NaN at every lag the search ruled out and the exact sum at the lags it kept (ref: src/pitch.rs:83-84, 372-405). */
enum nnn_tap {
    NNN_TAP_FILTERED = 0, /* [480] f32  high-passed input (features.rs:97-104)           */
    NNN_TAP_XLP,          /* [864] f32  pitch_buf after pitch_downsample (pitch.rs:448)   */
    NNN_TAP_AC,           /* [5]   f32  windowed autocorrelation                          */
    NNN_TAP_LPC2,         /* [5]   f32  FIR taps                                          */
    NNN_TAP_XCORR1,       /* [147] f32  coarse cross-correlation                          */
    NNN_TAP_BEST1,        /* [2]   i32  best / second best coarse lag                     */
    NNN_TAP_XCORR2C,      /* [10]  f32  fine cross-correlation at 2*best-2..+2, 2*second-2..+2 */
    NNN_TAP_PITCH_SEARCH, /* [1]   i32                                                    */
    NNN_TAP_PITCH,        /* [1]   i32  pitch period after remove_doubling                */
    NNN_TAP_PITCH_GAIN,   /* [1]   f32                                                    */
    NNN_TAP_X,            /* [962] f32  (re,im) x 481, signal spectrum before filtering   */
    NNN_TAP_P,            /* [962] f32  pitch-lagged spectrum                             */
    NNN_TAP_EX, NNN_TAP_EP, NNN_TAP_EXP, /* [22] f32                                      */
    NNN_TAP_FEATURES,     /* [42]  f32                                                    */
    NNN_TAP_SILENCE,      /* [1]   i32                                                    */
    NNN_TAP_G_RAW,        /* [22]  f32  RNN gains                                         */
    NNN_TAP_G,            /* [22]  f32  smoothed gains                                    */
    NNN_TAP_VAD,          /* [1]   f32                                                    */
    NNN_TAP_BRANCH,       /* [1]   i32  bit i < 22: pitch_filter took `exp > g` in band i (src/features.rs:227); bit 22: silent frame */
    NNN_TAP_COUNT
};
int nnn_tap_info(int tap, int *len, int *is_int);
int nnn_batch_set_taps(nnn_batch *b, int on);
int nnn_batch_read_tap(nnn_batch *b, int tap, void *host_dst, size_t dst_bytes);

/* Parity-test record of whole calls (the taps above hold the most recent frame only): from this call on, every processed frame
 * t = 0, 1, ... writes d_log[(t * n_streams + s) * NNN_FRAME_LOG_WORDS + i] for stream s -- word 0: pitch index (int32), word 1: the
 * BRANCH tap (int32), words 2..23: the 22 smoothed band gains (float32; zero on silent frames) -- until `frames` frames have
 * been recorded.  d_log is device memory owned by the caller; NULL or 0 frames switches the record off. */
#define NNN_FRAME_LOG_WORDS 24
int nnn_batch_set_frame_log(nnn_batch *b, void *d_log, size_t frames);

/* Per-kernel timing with HIP events on the launch stream (off by default: it adds two event
 * records per launch).  Times accumulate until read; reading resets them. */
int nnn_batch_set_profiling(nnn_batch *b, int on);
int nnn_batch_num_kernels(void);
const char *nnn_batch_kernel_name(int k);
int nnn_batch_read_kernel_times(nnn_batch *b, double *total_ms, int64_t *launches, int n);

/* Developer instrumentation: 64 shader-clock stamps of block 0 (zeros unless the library was built with
 * -DNNN_STAMPS). */
int nnn_batch_read_stamps(nnn_batch *b, long long *dst64);

/* Retained from the round-1 ABI, no effect: a frame group is five kernel launches now and they are always eager. */
int nnn_batch_set_graph(nnn_batch *b, int on);
/* 1 = calls of 32 frames or more spread their frame groups over the batch's internal HIP streams so that independent stages
 * overlap (default), 0 = every call runs its groups back to back on the caller's stream.  Results are bit-identical. */
int nnn_batch_set_pipeline(nnn_batch *b, int on);
/* The caller's promise about INPUT buffers of the device-pointer entry points: 1 = a call's input is final when the call is
 * made (uploaded and synchronised, or produced by work that has completed) -- not produced by work the caller enqueued on
 * the call's stream after the previous call.  Consecutive pipelined calls on one stream may then overlap at the boundary:
 * the next call's high-pass chain (the only stage that reads the input) starts while the previous call is still draining.
 * Outputs stay ordered on the caller's stream exactly as without it; results are bit-identical.  0 (default): a call's
 * input is read only after everything enqueued on its stream before the call. */
int nnn_batch_set_inputs_ready(nnn_batch *b, int on);
/* Which kernels run the part of a frame behind the pitch analysis.  0: transforms (k_fft_xp) -> RNN (k_rnn / k_rnn_wf) -> synthesis
 * (k_synth), the spectra crossing device memory in between.  1: groups of ONE frame -- the real-time host ticking 10 ms per call, the
 * reference's primary use (src/capi.rs:75-85, src/signal.rs:102-104) -- take the fused back end instead (k_back: one launch, both
 * spectra in registers from their transforms to the inverse transform); 2: every group does.  3 / 4: the fused kernel's RNN stretch
 * alone (16 waves, one layer at a time) replaces the RNN kernels for one-frame / all groups.  -1 (default): by batch size, as
 * measured -- one-frame groups take the fused kernel up to 8192 streams and the RNN stretch alone above that or while other batches
 * tick beside this one ON THE SAME DEVICE; longer groups stay with the layer-pipelined RNN between k_fft_xp and k_synth.
 * Every choice gives the same bits: a stream may change back end from call to call. */
int nnn_batch_set_back_end(nnn_batch *b, int mode);
/* How a pipelined call uses the internal streams: mode 0 = not at all (as set_pipeline(0)); 1 = "lanes": the high-pass
 * chain on its own stream, the other four stages of group k on lane k mod `lanes` (1..4; default 2; lane 0 is the caller's stream);
 * 2 = "stages": one stream per stage, every stream a chain of groups.  Environment: NNN_SCHED=seq|lanes|stages,
 * NNN_LANES=n.  Nobody choosing, the library does: calls of 32 frames or more are pipelined with one lane on batches of up to
 * 16 384 streams (from 8192 streams the high-pass chain of a group waits for the previous group's pitch kernel); bigger batches keep two
 * groups in flight (650 instead of 360 KB of device memory per stream) and run such calls with one stream per stage when they are two groups
 * long, on two lanes when longer (+2-3 % over one stream in order: every kernel of a big batch ends in a tail of half-empty compute
 * units, which another stage's blocks fill).  Shorter calls run in order on the caller's stream. */
int nnn_batch_set_schedule(nnn_batch *b, int mode, int lanes);
/*
 * Environment.  The library reads exactly these variables (the first eight when a batch is created, NNN_NODE_THREADS when a node is,
 * NNN_DEVICE when rnnoise_create / rnnoise_init make their batch of one), every setting gives the same bits, and each is exercised by
 * a test (named on the right).  It sets none: in particular GPU_MAX_HW_QUEUES (real-time hosts ticking several batches side by side
 * want 8) is the host's to export before its first HIP call -- the library only LOOKS whether it is set, to print
 * one note on stderr per process when batches are driven side by side on a device without it (they largely serialise on 4 queues).
 *   NNN_SCHED=seq|lanes|stages   how a call of 32 frames or more uses the batch's internal streams (nnn_batch_set_schedule)   test_hostsim_knobs
 *   NNN_LANES=1..4               lanes of the "lanes" schedule                                                                  test_hostsim_knobs
 *   NNN_HOST_CHUNK=n             frames per chunk of a host-buffer call (0 = one piece; default by call length)               test_gpu_parity / test_hostsim_pcm
 *   NNN_RNN_ROWS=16|32           stream rows per RNN workgroup (default by model and batch size)                              test_gpu_parity / test_hostsim_parity
 *   NNN_RNN_WF_MIN_G=n           shortest frame group the layer-pipelined RNN kernel takes                                    test_gpu_parity / test_hostsim_parity
 *   NNN_HP_SPLIT=0|1             the high-pass on one wave per 64 streams or two (default: two for launches of <= 256 tiles)  test_gpu_back_end / test_hostsim_parity
 *   NNN_LPC_HEAD=0|1             one-frame calls: the old part of the LPC sums in the high-pass launch                        test_gpu_back_end / test_hostsim_parity
 *   NNN_PITCH_CHAIN=0|1|2        frames of a group side by side in k_pitch with a flag hand-off, or looped (default by size)  test_gpu_parity / test_hostsim_parity
 *   NNN_NODE_THREADS=0           a node's shards one after the other on the caller's thread                                   test_hostsim_node
 *   NNN_DEVICE=n                 HIP device of the rnnoise_* single-stream surface (default 0)                                 test_gpu_node
 * Earlier rounds' A/B probe knobs (NNN_BACK, NNN_HP_TPB, NNN_X_RIDES, NNN_LPC_IN_PITCH, NNN_HP_AFTER, NNN_PIPE_MAX, ...) exist only in
 * builds with -DNNN_DEV_KNOBS (the tests' interpreter build, scripts/build_variant*.sh); the product does not read them.
 */

/* Diagnostic: the device's activation functions on their own, y[i] = act(x[i]) for n host floats; act 0 = tansig_approx,
 * 1 = sigmoid_approx, 2 = relu (src/util.rs:29-53). */
int nnn_debug_activations(int device, int act, const float *x, float *y, int n);

/* The CPUs local to a HIP device's PCI function in the kernel's cpulist syntax ("0-63,128-191"; /sys/bus/pci/devices/<id>/local_cpulist):
 * where a host thread that feeds that device should run (the node object pins its workers there).  0 and the text, or non-zero. */
int nnn_device_local_cpulist(int device, char *buf, size_t cap);

const char *nnn_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NNN_BATCH_H */
