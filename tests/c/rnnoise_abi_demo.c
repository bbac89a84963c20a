/* ABI acceptance program for include/rnnoise.h (our own; mirrors what the reference's CI does with its
 * C demo: .github/workflows/rust.yml:27-33): raw i16 in -> rnnoise_process_frame in place -> raw i16 out,
 * first frame skipped.  Optional 3rd argument: a binary .rnn model loaded with rnnoise_model_from_file. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rnnoise.h"

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s in.raw out.raw [model.rnn]\n", argv[0]);
        return 2;
    }
    const int n = rnnoise_get_frame_size();
    RNNModel *model = NULL;
    if (argc > 3) {
        FILE *mf = fopen(argv[3], "rb");
        if (!mf || !(model = rnnoise_model_from_file(mf))) { /* from_file owns and closes mf */
            fprintf(stderr, "cannot load model %s\n", argv[3]);
            return 3;
        }
    }
    DenoiseState *st = rnnoise_create(model);
    if (!st) {
        fprintf(stderr, "rnnoise_create failed\n");
        return 4;
    }
    FILE *fi = fopen(argv[1], "rb"), *fo = fopen(argv[2], "wb");
    if (!fi || !fo) return 5;
    short *pcm = malloc(sizeof(short) * n);
    float *x = malloc(sizeof(float) * n);
    int frame = 0;
    double vad_sum = 0.0;
    while (fread(pcm, sizeof(short), n, fi) == (size_t)n) {
        for (int i = 0; i < n; i++) x[i] = pcm[i];
        vad_sum += rnnoise_process_frame(st, x, x);
        for (int i = 0; i < n; i++) {
            float v = roundf(x[i]);
            pcm[i] = (short)(v > 32767.f ? 32767.f : (v < -32768.f ? -32768.f : v));
        }
        if (frame++ > 0) fwrite(pcm, sizeof(short), n, fo);
    }
    fclose(fi);
    fclose(fo);
    rnnoise_destroy(st);
    if (model) rnnoise_model_free(model);
    printf("frames %d mean_vad %.4f\n", frame, frame ? vad_sum / frame : 0.0);
    free(pcm);
    free(x);
    return 0;
}
