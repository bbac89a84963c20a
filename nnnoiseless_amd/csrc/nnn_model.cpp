// nnn_model.cpp -- .rnn model container: parser with the reference's validation rules, the
// built-in weights, and the packing the RNN kernels consume (bf16 weights in MFMA B-fragment order, f32 biases).
#include "nnn_model.h"

#include <string.h>

#include <algorithm>

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

namespace {
inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }
inline uint16_t bf16_of_int(int v)  // |v| <= 128: exact
{
    float f = (float)v;
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)(u >> 16);
}

// Appends one GEMM operand in fragment order [neuron block][gate][k-step][lane][8]:
// element = W[colmap(k)][gate * n + neuron], k = kbase + 32 ks + 8 (lane >> 4) + e, neuron = 16 nb + (lane & 15).
template <class ColMap>
nnn::GemmDesc pack_gemm(std::vector<uint16_t> &wq, const int8_t *W, int row_stride, int n, int ngates, int ksteps, int kbase,
                        ColMap colmap)
{
    nnn::GemmDesc g;
    g.wofs = (int)(wq.size() / 8);
    g.ksteps = ksteps;
    g.kbase = kbase;
    g.ngates = ngates;
    const int nb = pad_to(n, 16) / 16;
    for (int b = 0; b < nb; b++)
        for (int gate = 0; gate < ngates; gate++)
            for (int ks = 0; ks < ksteps; ks++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        int k = kbase + 32 * ks + 8 * (lane >> 4) + e, neuron = 16 * b + (lane & 15);
                        int row = colmap(k);
                        int v = (row >= 0 && neuron < n) ? W[(size_t)row * row_stride + gate * n + neuron] : 0;
                        wq.push_back(bf16_of_int(v));
                    }
    return g;
}
}  // namespace

size_t nnn_model_pack(const RNNModel &m, std::vector<uint16_t> &wq, std::vector<float> &fpar, nnn::RnnPlan &plan,
                      nnn::ModelDims &md)
{
    using nnn::LayerDesc;
    wq.clear();
    fpar.clear();
    const int nd = m.input_dense.nb_neurons, nv = m.vad_gru.nb_neurons, nn = m.noise_gru.nb_neurons,
              ndn = m.denoise_gru.nb_neurons;
    md.nd = nd; md.nv = nv; md.nn = nn; md.ndn = ndn;
    md.act_d = m.input_dense.activation; md.act_v = m.vad_gru.activation; md.act_n = m.noise_gru.activation;
    md.act_dn = m.denoise_gru.activation; md.act_o = m.denoise_output.activation; md.act_vo = m.vad_output.activation;
    const int8_t *blob = m.blob.data();
    // LDS columns of the input matrix: [ noise state | vad state | features (48) | dense out ]
    const int cN = 0, cV = pad_to(nn, 8), cF = cV + pad_to(nv, 8), cD = cF + 48;
    const int NF = 42;
    auto ks_of = [](int cols) { return (cols + 31) / 32; };
    auto fbias = [&](size_t ofs, int count) {
        int at = (int)fpar.size();
        for (int i = 0; i < count; i++) fpar.push_back((float)blob[ofs + i]);
        return at;
    };
    auto finish = [&](LayerDesc &L, int n, int act, int bias, int out_col) {
        L.n = n;
        L.nb = pad_to(n, 16) / 16;
        L.act = act;
        L.bias = bias;
        L.out_col = out_col;
    };
    // input dense: features -> D                                  (ref: src/rnn.rs:353-355)
    plan.dense.in = pack_gemm(wq, blob + m.input_dense.weights, nd, nd, 1, 2, cF,
                              [&](int k) { return (k >= cF && k < cF + NF) ? k - cF : -1; });
    plan.dense.rec = nnn::GemmDesc{0, 0, 0, 0};
    finish(plan.dense, nd, m.input_dense.activation, fbias(m.input_dense.bias, nd), cD);
    // vad GRU: input D                                             (ref: src/rnn.rs:356-358)
    plan.vad.in = pack_gemm(wq, blob + m.vad_gru.weights, 3 * nv, nv, 3, ks_of(nd), cD,
                            [&](int k) { return (k >= cD && k < cD + nd) ? k - cD : -1; });
    plan.vad.rec = pack_gemm(wq, blob + m.vad_gru.rec, 3 * nv, nv, 3, ks_of(nv), 0, [&](int k) { return k < nv ? k : -1; });
    finish(plan.vad, nv, m.vad_gru.activation, fbias(m.vad_gru.bias, 3 * nv), cV);
    // noise GRU: reference input order [D | V | F]                 (ref: src/rnn.rs:361-366)
    plan.noise.in = pack_gemm(wq, blob + m.noise_gru.weights, 3 * nn, nn, 3, ks_of(cD + nd - cV), cV, [&](int k) {
        if (k >= cV && k < cV + nv) return nd + (k - cV);
        if (k >= cF && k < cF + NF) return nd + nv + (k - cF);
        if (k >= cD && k < cD + nd) return k - cD;
        return -1;
    });
    plan.noise.rec = pack_gemm(wq, blob + m.noise_gru.rec, 3 * nn, nn, 3, ks_of(nn), 0, [&](int k) { return k < nn ? k : -1; });
    finish(plan.noise, nn, m.noise_gru.activation, fbias(m.noise_gru.bias, 3 * nn), cN);
    // denoise GRU: reference input order [V | N | F]               (ref: src/rnn.rs:368-377)
    plan.dn.in = pack_gemm(wq, blob + m.denoise_gru.weights, 3 * ndn, ndn, 3, ks_of(cF + NF), 0, [&](int k) {
        if (k < nn) return nv + k;
        if (k >= cV && k < cV + nv) return k - cV;
        if (k >= cF && k < cF + NF) return nv + nn + (k - cF);
        return -1;
    });
    plan.dn.rec = pack_gemm(wq, blob + m.denoise_gru.rec, 3 * ndn, ndn, 3, ks_of(ndn), 0, [&](int k) { return k < ndn ? k : -1; });
    finish(plan.dn, ndn, m.denoise_gru.activation, fbias(m.denoise_gru.bias, 3 * ndn), 0);
    // gains: denoise state (written to columns 0..ndn) -> 22       (ref: src/rnn.rs:378)
    plan.out.in = pack_gemm(wq, blob + m.denoise_output.weights, 22, 22, 1, ks_of(ndn), 0, [&](int k) { return k < ndn ? k : -1; });
    plan.out.rec = nnn::GemmDesc{0, 0, 0, 0};
    finish(plan.out, 22, m.denoise_output.activation, fbias(m.denoise_output.bias, 22), 0);
    // vad output, 1 x nv, stays on the vector ALU                  (ref: src/rnn.rs:359)
    plan.vo_w = fbias(m.vad_output.weights, nv);
    plan.vo_b = fbias(m.vad_output.bias, 1);
    plan.act_vo = m.vad_output.activation;
    plan.cF = cF;
    plan.cV = cV;
    int width = 0;
    auto need = [&](const nnn::GemmDesc &g) { width = std::max(width, g.kbase + 32 * g.ksteps); };
    need(plan.dense.in); need(plan.vad.in); need(plan.noise.in); need(plan.dn.in); need(plan.out.in);
    width = std::max(width, pad_to(ndn, 8));
    plan.in_w = pad_to(width, 16) + 8;  // row stride = 16 bytes (mod 32): 16-byte fragment reads of 16 rows spread over all banks
    plan.rec_w = pad_to(32 * ks_of(std::max(nv, std::max(nn, ndn))), 16) + 8;
    (void)cN;
    // dynamic LDS: tanh table (256 floats) + live flags (64 ints) + 3 planes of both matrices + the staged
    // cepstral ring of the feature stage (8 x 22 rows of 64 floats) and its 28 pair distances
    return 256 * 4 + 64 * 4 + (size_t)3 * 64 * (plan.in_w + plan.rec_w) * 2 + (size_t)(8 * 22 + 28) * 64 * 4;
}
