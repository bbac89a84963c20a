// SPDX-License-Identifier: BSD-3-Clause
/*
 * nnn_train.h -- batched training-feature rows.
 *
 * The reference's training-data generator (src/training.rs) produces one 87-column row per frame:
 * it keeps three DenoiseFeatures states -- clean speech, noise, and their mix -- calls
 * shift_and_filter_input + compute_frame_features on each (src/training.rs:125-134), and writes
 *     [ 42 features of the mix | 22 ideal band gains | 22 noise levels | vad ]      (src/training.rs:136-158)
 * nnn_train_process_* is that per-frame body for n_streams independent (clean, noise, mix) triples at once, on the
 * same HIP kernels as the denoiser (high-pass, pitch search, spectra, band energies, features; no RNN, no synthesis).
 * The simulator that produces the three signals, the band cutoff and the VAD label (file reading, random gains and
 * filters, src/training.rs:263-422) stays with the caller.
 *
 * Plain C ABI; all functions return 0 on success (nnn_last_error() of nnn_batch.h has the text); no CPU fallback.
 */
#ifndef NNN_TRAIN_H
#define NNN_TRAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nnn_train nnn_train;

#define NNN_TRAIN_COLS 87 /* NB_FEATURES + 2 * NB_BANDS + 1, src/training.rs:89 */

/* n_streams x three DenoiseFeatures::new() (src/training.rs:113-115). */
nnn_train *nnn_train_create(int n_streams, int device);
void nnn_train_destroy(nnn_train *t);
int nnn_train_reset(nnn_train *t);

/*
 * n_frames rows per stream, buffers resident in device memory.
 *   sample i of frame f of stream s:  d_signal / d_noise / d_combined [s * stream_stride + f * frame_stride + i]
 *                                     (floats in i16 range; combined = signal + noise as the caller mixed them)
 *   d_cutoff[f * n_streams + s]       NoisyFrame::band_gain_cutoff (bands from here up get gain -1)
 *   d_vad   [f * n_streams + s]       NoisyFrame::vad, copied into column 86
 *   d_rows  [(f * n_streams + s) * 87 ...]
 * hip_stream: a hipStream_t to enqueue on (NULL = the object's own stream).  Asynchronous.
 */
int nnn_train_process_device(nnn_train *t, const float *d_signal, const float *d_noise, const float *d_combined,
                             const int32_t *d_cutoff, const float *d_vad, float *d_rows, int n_frames,
                             size_t stream_stride, size_t frame_stride, void *hip_stream);
/* Same with host buffers, dense layout [n_streams][n_frames][480] (copies over PCIe, synchronous). */
int nnn_train_process_host(nnn_train *t, const float *signal, const float *noise, const float *combined,
                           const int32_t *cutoff, const float *vad, float *rows, int n_frames);

#ifdef __cplusplus
}
#endif
#endif /* NNN_TRAIN_H */
