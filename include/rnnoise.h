// SPDX-License-Identifier: BSD-3-Clause
/*
 * rnnoise.h -- RNNoise-compatible C ABI, served by the MI355X batched backend.
 *
 * Drop-in for the header cbindgen generates from the reference's src/capi.rs (cbindgen.toml:
 * include guard RNNOISE_H, sys include stdio.h, C++ guards; header name `rnnoise`, Cargo.toml:63-64).
 * Each prototype cites the reference entry point it replaces.  test_data/rnnoise_demo.c from the
 * reference compiles against this header unmodified.
 *
 * Every DenoiseState here is a batch of ONE stream on the GPU (functionally identical to the
 * reference, never fast).  Throughput lives behind include/nnn_batch.h.
 */
#ifndef RNNOISE_H
#define RNNOISE_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct DenoiseState DenoiseState; /* opaque; src/capi.rs:9  */
typedef struct RNNModel RNNModel;         /* opaque; src/capi.rs:11 */

/* Number of samples processed per call: 480.  (src/capi.rs:16-19) */
int rnnoise_get_frame_size(void);

/* Size in bytes of a DenoiseState for callers that allocate it themselves.  (src/capi.rs:24-27) */
int rnnoise_get_size(void);

/* Initialise caller-allocated storage of rnnoise_get_size() bytes; model NULL = built-in model.
 * Returns 0 (-1 if no MI355X device / kernel library is usable).  (src/capi.rs:32-43)
 * OWNERSHIP, where this differs from the reference: the reference's state is plain memory with no heap behind it, so storage that
 * was rnnoise_init'ed is simply abandoned.  Here the storage holds a handle to a GPU batch of one stream (device memory, streams,
 * events) that only rnnoise_destroy releases -- and rnnoise_destroy also free()s the storage, as the reference's does
 * (src/capi.rs:62-65), so it is only for states from rnnoise_create.  A state made by rnnoise_init in caller storage therefore
 * keeps its GPU resources until the process exits: hosts that cycle states use rnnoise_create / rnnoise_destroy. */
int rnnoise_init(DenoiseState *st, RNNModel *model);

/* Allocate and initialise a state; model NULL = built-in.  A non-NULL model is borrowed and must
 * outlive the state.  Returns NULL if no MI355X device/kernel library is usable.  (src/capi.rs:48-57) */
DenoiseState *rnnoise_create(RNNModel *model);

/* Free a state returned by rnnoise_create.  (src/capi.rs:62-65) */
void rnnoise_destroy(DenoiseState *st);

/* Denoise 480 samples (f32 in i16 range); `out` may alias `in`.  Returns the voice-activity
 * probability.  Aborts on a NULL state like the reference panics.  (src/capi.rs:75-85) */
float rnnoise_process_frame(DenoiseState *st, float *out, float *in);

/* Load a binary .rnn model.  Like the reference this takes ownership of `f` and fclose()s it.
 * NULL on read or parse error.  (src/capi.rs:88-105) */
RNNModel *rnnoise_model_from_file(FILE *f);

/* Free a model returned by rnnoise_model_from_file.  (src/capi.rs:110-113) */
void rnnoise_model_free(RNNModel *model);

#ifdef __cplusplus
}
#endif
#endif /* RNNOISE_H */
