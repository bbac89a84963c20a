// SPDX-License-Identifier: BSD-3-Clause
/*
 * nnn_resample.h -- batched 16-tap windowed-sinc resampler to 48 kHz.
 *
 * The reference CLI resamples any input that is not at 48 kHz before denoising it: `Resample<RS>` around
 * dasp_interpolate 0.11's `Sinc<[f32; 16]>` (src/nnnoiseless.rs:15-32, 40-46, 106-131; call sites :179-227), one
 * interpolator per channel, driven sample by sample: pos += ratio; while pos >= 1 { pos -= 1; push a source sample };
 * out = sinc.interpolate(pos), with ratio = source_rate / 48000.  nnn_resampler_* is that loop for n_streams independent
 * mono streams of one common source rate at once: the position sequence and the tap weights (functions of the
 * position only, f64) are computed once per call on the host, the 16-tap sums per output sample on the GPU.
 *
 * dasp_interpolate / dasp_ring_buffer are not part of the reference tree (crates.io, Cargo.lock pins 0.11.0): the
 * interpolator is restated from the published source of that version -- Hann-windowed sinc, depth 8, the tap products
 * in f64 and the running sum in f32, ring indices wrapping modulo 16 -- and is NOT pinned by any reference test.
 *
 * Plain C ABI; 0 on success (nnn_last_error() of nnn_batch.h has the text); no CPU fallback.
 */
#ifndef NNN_RESAMPLE_H
#define NNN_RESAMPLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nnn_resampler nnn_resampler;

/* n_streams x Sinc::new(Fixed::from([0.0; 16])) with pos = 0 (src/nnnoiseless.rs:19-32); ratio = source_rate / 48000. */
nnn_resampler *nnn_resampler_create(int n_streams, double ratio, int device);
void nnn_resampler_destroy(nnn_resampler *r);
int nnn_resampler_reset(nnn_resampler *r);
/* Upper bound of the output samples n_in more source samples can produce. */
long nnn_resampler_max_output(const nnn_resampler *r, long n_in);

/*
 * Feed n_in source samples per stream (d_in[s * in_stride + i]) and collect the output samples they complete
 * (d_out[s * out_stride + m], m < *n_out <= cap_out; the same count for every stream).  An output sample whose
 * source samples have not all arrived is produced by the next call: any chunking of the input gives the same output
 * as one call.  Buffers resident in device memory; asynchronous on hip_stream (NULL = the resampler's own stream).
 * cap_out must be at least nnn_resampler_max_output(r, n_in): every source sample of a call is consumed by it, so a
 * smaller buffer would drop outputs -- such a call is refused (non-zero return, nothing consumed).
 */
int nnn_resampler_process_device(nnn_resampler *r, const float *d_in, long n_in, size_t in_stride, float *d_out, long cap_out,
                                 size_t out_stride, long *n_out, void *hip_stream);
/* The same with host buffers, dense layouts [n_streams][n_in] -> [n_streams][cap_out] (copies over PCIe, synchronous). */
int nnn_resampler_process_host(nnn_resampler *r, const float *in, long n_in, float *out, long cap_out, long *n_out);

#ifdef __cplusplus
}
#endif
#endif /* NNN_RESAMPLE_H */
