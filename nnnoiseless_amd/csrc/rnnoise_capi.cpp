// rnnoise_capi.cpp -- the RNNoise-compatible C ABI (include/rnnoise.h) on top of the batched
// backend: every DenoiseState is a batch of one stream on the GPU.  Mirrors the reference's
// src/capi.rs entry point by entry point; there is no CPU fallback.
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <vector>

#include "../../include/nnn_batch.h"
#include "../../include/rnnoise.h"

struct DenoiseState {
    nnn_batch *batch;
};

static int pick_device()
{
    const char *e = getenv("NNN_DEVICE");
    return e ? atoi(e) : 0;
}

extern "C" int rnnoise_get_frame_size(void) { return NNN_FRAME_SIZE; }               // src/capi.rs:16-19
extern "C" int rnnoise_get_size(void) { return (int)sizeof(DenoiseState); }           // src/capi.rs:24-27

extern "C" int rnnoise_init(DenoiseState *st, RNNModel *model)                        // src/capi.rs:32-43
{
    // process_frame takes one frame per call: a batch sized for one-frame groups (3 MB for the 64-stream tile instead of 41)
    nnn_batch_opts opts = {};
    opts.max_group_frames = 1;
    const RNNModel *m = model;
    const int one = 1;
    st->batch = nnn_batch_create_opts(&m, &one, 1, pick_device(), &opts);
    if (!st->batch) {
        fprintf(stderr, "rnnoise_init: %s\n", nnn_last_error());
        return -1;  // the reference cannot fail here; a missing GPU can
    }
    return 0;
}

extern "C" DenoiseState *rnnoise_create(RNNModel *model)                              // src/capi.rs:48-57
{
    DenoiseState *st = (DenoiseState *)malloc(sizeof(DenoiseState));
    if (!st) return NULL;
    if (rnnoise_init(st, model) != 0) {
        free(st);
        return NULL;
    }
    return st;
}

extern "C" void rnnoise_destroy(DenoiseState *st)                                     // src/capi.rs:62-65
{
    if (!st) return;
    nnn_batch_destroy(st->batch);
    free(st);
}

extern "C" float rnnoise_process_frame(DenoiseState *st, float *out, float *in)       // src/capi.rs:75-85
{
    if (!st || !st->batch) {  // the reference panics with "Invalid pointer"
        fprintf(stderr, "rnnoise_process_frame: Invalid pointer\n");
        abort();
    }
    float vad = 0.0f;
    if (nnn_batch_process_host(st->batch, in, out, &vad, 1, NNN_FRAME_SIZE, NNN_FRAME_SIZE) != 0) {
        fprintf(stderr, "rnnoise_process_frame: %s\n", nnn_last_error());
        abort();
    }
    return vad;
}

extern "C" RNNModel *rnnoise_model_from_file(FILE *f)                                 // src/capi.rs:88-105
{
    if (!f) return NULL;
    // like the reference: dup the descriptor, close the caller's FILE, read the rest of the file
    int fd = dup(fileno(f));
    fclose(f);
    if (fd < 0) return NULL;
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    for (;;) {
        ssize_t n = read(fd, buf, sizeof(buf));
        if (n < 0) { close(fd); return NULL; }
        if (n == 0) break;
        data.insert(data.end(), buf, buf + n);
    }
    close(fd);
    return nnn_model_from_bytes(data.data(), data.size());
}

extern "C" void rnnoise_model_free(RNNModel *model) { nnn_model_free(model); }        // src/capi.rs:110-113
