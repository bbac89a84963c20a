// nnn_model.h -- host-side RnnModel container (mirror of the reference's src/rnn.rs:14-62).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "nnn_layout.h"

struct NnnDense {
    int nb_inputs = 0, nb_neurons = 0, activation = 0;
    size_t weights = 0, bias = 0;  // byte offsets into RNNModel::blob
};
struct NnnGru {
    int nb_inputs = 0, nb_neurons = 0, activation = 0;
    size_t weights = 0, rec = 0, bias = 0;
};

// Opaque `RNNModel` of the C ABI (reference: src/capi.rs:11 wraps rnn::RnnModel).
struct RNNModel {
    std::vector<int8_t> blob;
    NnnDense input_dense, denoise_output, vad_output;
    NnnGru vad_gru, noise_gru, denoise_gru;
};

// RnnModel::from_bytes, ref: src/rnn.rs:116-232.  Returns nullptr where the reference returns None.
RNNModel *nnn_model_parse(const uint8_t *bytes, size_t len);
const uint8_t *nnn_builtin_weights(size_t *len);

// Pack the i8 weights for the MFMA RNN kernel: bf16 (exact: |w| <= 128) in B-fragment order, biases
// and the vad output layer as f32, plus the LDS column plan.  Returns the dynamic LDS bytes the kernel needs.
size_t nnn_model_pack(const RNNModel &m, std::vector<uint16_t> &wq, std::vector<float> &fpar, nnn::RnnPlan &plan,
                      nnn::ModelDims &md);
