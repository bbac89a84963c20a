// nnn_model.cpp -- .rnn model container: parser with the reference's validation rules, the
// built-in weights, and the i8 -> f32 expansion the RNN kernel consumes.
#include "nnn_model.h"

#include <string.h>

// The built-in model is the reference's src/weights.rnn (BSD-3-Clause, (c) Mozilla / Xiph / J. Neeman),
// shipped as data in nnnoiseless_amd/data/weights.rnn and linked in verbatim
// (reference: include_bytes!("weights.rnn"), src/rnn.rs:237).
#ifndef NNN_WEIGHTS_PATH
#error "build with -DNNN_WEIGHTS_PATH=\"/abs/path/to/weights.rnn\""
#endif
__asm__(".section .rodata\n"
        ".global nnn_builtin_weights_begin\n"
        ".balign 16\n"
        "nnn_builtin_weights_begin:\n"
        ".incbin \"" NNN_WEIGHTS_PATH "\"\n"
        ".global nnn_builtin_weights_end\n"
        "nnn_builtin_weights_end:\n"
        ".byte 0\n"
        ".text\n");
extern "C" const uint8_t nnn_builtin_weights_begin[];
extern "C" const uint8_t nnn_builtin_weights_end[];

const uint8_t *nnn_builtin_weights(size_t *len)
{
    *len = (size_t)(nnn_builtin_weights_end - nnn_builtin_weights_begin);
    return nnn_builtin_weights_begin;
}

namespace {
struct Cursor {
    const int8_t *base;
    size_t pos, len;
    size_t left() const { return len - pos; }
};

// three header bytes: nb_inputs, nb_neurons (non-negative i8), activation in {0,1,2}; ref: src/rnn.rs:128-152
bool read_header(Cursor &c, int &nin, int &nout, int &act)
{
    if (c.left() < 3) return false;
    const int8_t *b = c.base + c.pos;
    if (b[0] < 0 || b[1] < 0 || b[2] < 0 || b[2] > 2) return false;
    nin = b[0]; nout = b[1]; act = b[2];
    c.pos += 3;
    return true;
}
bool take(Cursor &c, size_t n, size_t &ofs)
{
    if (c.left() < n) return false;
    ofs = c.pos;
    c.pos += n;
    return true;
}
bool read_dense(Cursor &c, NnnDense &l)  // ref: src/rnn.rs:145-164
{
    return read_header(c, l.nb_inputs, l.nb_neurons, l.activation) &&
           take(c, (size_t)l.nb_inputs * l.nb_neurons, l.weights) && take(c, (size_t)l.nb_neurons, l.bias);
}
bool read_gru(Cursor &c, NnnGru &l)  // ref: src/rnn.rs:166-187
{
    if (!read_header(c, l.nb_inputs, l.nb_neurons, l.activation)) return false;
    size_t n = (size_t)l.nb_neurons;
    return take(c, 3 * n * (size_t)l.nb_inputs, l.weights) && take(c, 3 * n * n, l.rec) && take(c, 3 * n, l.bias);
}
}  // namespace

RNNModel *nnn_model_parse(const uint8_t *bytes, size_t len)
{
    RNNModel *m = new RNNModel();
    m->blob.assign((const int8_t *)bytes, (const int8_t *)bytes + len);
    Cursor c{m->blob.data(), 0, len};
    bool ok = read_dense(c, m->input_dense) && read_gru(c, m->vad_gru) && read_gru(c, m->noise_gru) &&
              read_gru(c, m->denoise_gru) && read_dense(c, m->denoise_output) && read_dense(c, m->vad_output);
    ok = ok && c.left() == 0;                                                                   // :196-198
    ok = ok && m->input_dense.nb_inputs == 42 && m->denoise_output.nb_neurons == 22 &&
         m->vad_output.nb_neurons == 1;                                                         // :204-209
    ok = ok && m->input_dense.nb_neurons == m->vad_gru.nb_inputs &&
         m->vad_gru.nb_neurons == m->vad_output.nb_inputs;                                      // :210-213
    ok = ok && 42 + m->input_dense.nb_neurons + m->vad_gru.nb_neurons == m->noise_gru.nb_inputs;   // :214-216
    ok = ok && 42 + m->vad_gru.nb_neurons + m->noise_gru.nb_neurons == m->denoise_gru.nb_inputs;   // :217-219
    ok = ok && m->denoise_gru.nb_neurons == m->denoise_output.nb_inputs;                        // :220-222
    if (!ok) {
        delete m;
        return nullptr;
    }
    return m;
}

void nnn_model_expand(const RNNModel &m, std::vector<float> &w, nnn::ModelDims &md)
{
    w.clear();
    auto push = [&](size_t ofs, size_t n) {
        int at = (int)w.size();
        for (size_t i = 0; i < n; i++) w.push_back((float)m.blob[ofs + i]);
        while (w.size() % 4) w.push_back(0.0f);  // keep every array 16-byte aligned for wide scalar loads
        return at;
    };
    md.nd = m.input_dense.nb_neurons;
    md.nv = m.vad_gru.nb_neurons;
    md.nn = m.noise_gru.nb_neurons;
    md.ndn = m.denoise_gru.nb_neurons;
    md.act_d = m.input_dense.activation;
    md.act_v = m.vad_gru.activation;
    md.act_n = m.noise_gru.activation;
    md.act_dn = m.denoise_gru.activation;
    md.act_o = m.denoise_output.activation;
    md.act_vo = m.vad_output.activation;
    md.w_d = push(m.input_dense.weights, (size_t)42 * md.nd);
    md.b_d = push(m.input_dense.bias, md.nd);
    auto gru = [&](const NnnGru &g, int &wo, int &ro, int &bo) {
        size_t n = g.nb_neurons;
        wo = push(g.weights, 3 * n * g.nb_inputs);
        ro = push(g.rec, 3 * n * n);
        bo = push(g.bias, 3 * n);
    };
    gru(m.vad_gru, md.w_v, md.r_v, md.b_v);
    gru(m.noise_gru, md.w_n, md.r_n, md.b_n);
    gru(m.denoise_gru, md.w_dn, md.r_dn, md.b_dn);
    md.w_o = push(m.denoise_output.weights, (size_t)md.ndn * 22);
    md.b_o = push(m.denoise_output.bias, 22);
    md.w_vo = push(m.vad_output.weights, md.nv);
    md.b_vo = push(m.vad_output.bias, 1);
}
