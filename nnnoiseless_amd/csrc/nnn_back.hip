// nnn_back.hip -- the fused back end of the batched process_frame path: everything of a frame behind the pitch analysis in ONE
// launch per 16-stream block -- transform_input x 2, band energies and correlation, the 42 features, the RNN, pitch filter, gains,
// inverse transform and overlap-add (ref: src/denoise.rs:100-114, src/features.rs:119-275, src/rnn.rs:343-379) -- with the two spectra
// X and P held in the registers of the stream's wave from their transforms to the synthesis.  In the unfused pipeline (k_fft_xp ->
// k_rnn_wf -> k_synth) they cross HBM once each way around the RNN: 14 of the path's 28 KB per stream-frame, and two launch
// boundaries on the critical path of a one-frame (real-time) call.
//
// Block = 16 consecutive streams of a tile = 16 waves.  A frame is three stretches:
//   wave = stream   the two windowed transforms, band sums, feature head (the code of k_fft_xp, FUSED variant of transform_inputs),
//                   then the stream's own feature stage on all 64 lanes of its wave (cepstral ring and pair distances in LDS)
//   16 waves, one layer at a time
//                   the RNN on the 16 rows (one MFMA M-tile): every GRU is split into (neuron block, gate) units dealt over the
//                   16 waves -- z and r gates with their recurrent and input products, the candidate's input part -- then, behind
//                   a barrier, the candidate's recurrent part on r * state and the update.  Per gate the MFMAs run in the order
//                   k_rnn and k_rnn_wf use (bias, recurrent k-steps, input k-steps; planes hi, mid, lo inside a k-step), so the
//                   three kernels give the same bits.  k_rnn runs a (neuron block, all three gates) unit per wave: at 16 rows six
//                   of its eight waves carry 63 dependent MFMAs and 32 activations each per phase; here a wave carries at most 33
//                   and 8.
//   wave = stream   pitch filter, band renormalisation, gains, inverse transform, overlap-add (the code of k_synth on the
//                   transforms' own bin order), output conversion.
// The launch loops over the g frames of its group: GRU states (LDS planes), cepstral ring and pair distances (LDS), overlap memory
// (registers) stay on chip from frame to frame.  FUSED = false is the RNN stretch alone between k_fft_xp and k_synth (features from
// and gains to the scratch arrays): the stand-alone measurement of the 16-wave RNN step, and a one-frame RNN kernel in its own right.
#pragma once
#include "nnn_kernels.hip"

namespace nnn {

constexpr int BK_ROWS = 16, BK_WAVES = 16, BK_T = 64 * BK_WAVES;
constexpr int BK_ZW = 136;       // row stride (floats) of the z-gate buffer: up to 128 neuron columns + 8
constexpr int BK_CW = 32;        // row stride of the per-stream cepstrum staging and pair distances (28 used)
constexpr int BK_GW = 48;        // per-stream gains handed to the synthesis: raw [0, 22), smoothed [24, 46)

// The kernel is compiled for a SHAPE CLASS -- the layer sizes (input dense, vad / noise / denoise GRU neurons) as compile-time constants,
// so that every LDS offset, row stride, k-step count and unit count is an immediate and every k-step loop unrolls; with the plan as
// run-time values (k_rnn's way) a 16-wave kernel that also holds two spectra per wave runs out of scalar registers (500 scalar spills
// into vector registers in the first build).  bk_make_plan restates nnn_model_pack's column plan and packing offsets (nnn_model.cpp)
// for given sizes; the host compares it with the model's own plan field by field and takes the unfused kernels for any model outside
// the compiled classes.  Activation kinds stay run-time values (the models of a class differ in them).
constexpr int bk_pad(int v, int m) { return (v + m - 1) / m * m; }
constexpr int bk_ks(int cols) { return (cols + 31) / 32; }
constexpr GemmDesc bk_gd(int &at, int n, int ngates, int ksteps, int kbase)
{
    GemmDesc g{at, ksteps, kbase, ngates};
    at += bk_pad(n, 16) / 16 * ngates * ksteps * 64;
    return g;
}
constexpr LayerDesc bk_ld(GemmDesc in, GemmDesc rec, int n, int &bias_at, int nbias, int out_col)
{
    LayerDesc L{in, rec, n, bk_pad(n, 16) / 16, 0, bias_at, out_col};
    bias_at += nbias;
    return L;
}
constexpr RnnPlan bk_make_plan(int nd, int nv, int nn, int ndn)
{
    const int cN = 0, cV = bk_pad(nn, 8), cF = cV + bk_pad(nv, 8), cD = cF + 48, NF = 42;
    int at = 0, bias = 0;
    RnnPlan p{};
    const GemmDesc none{0, 0, 0, 0};
    const GemmDesc d_in = bk_gd(at, nd, 1, 2, cF);
    p.dense = bk_ld(d_in, none, nd, bias, nd, cD);
    const GemmDesc v_in = bk_gd(at, nv, 3, bk_ks(nd), cD), v_rec = bk_gd(at, nv, 3, bk_ks(nv), 0);
    p.vad = bk_ld(v_in, v_rec, nv, bias, 3 * nv, cV);
    const GemmDesc n_in = bk_gd(at, nn, 3, bk_ks(cD + nd - cV), cV), n_rec = bk_gd(at, nn, 3, bk_ks(nn), 0);
    p.noise = bk_ld(n_in, n_rec, nn, bias, 3 * nn, cN);
    const GemmDesc dn_in = bk_gd(at, ndn, 3, bk_ks(cF + NF), 0), dn_rec = bk_gd(at, ndn, 3, bk_ks(ndn), 0);
    p.dn = bk_ld(dn_in, dn_rec, ndn, bias, 3 * ndn, 0);
    const GemmDesc o_in = bk_gd(at, 22, 1, bk_ks(ndn), 0);
    p.out = bk_ld(o_in, none, 22, bias, 22, 0);
    p.vo_w = bias;
    p.vo_b = bias + nv;
    p.act_vo = 0;
    p.cF = cF;
    p.cV = cV;
    int width = bk_pad(ndn, 8);
    const GemmDesc all[5] = {d_in, v_in, n_in, dn_in, o_in};
    for (int i = 0; i < 5; i++) width = width > all[i].kbase + 32 * all[i].ksteps ? width : all[i].kbase + 32 * all[i].ksteps;
    p.in_w = bk_pad(width, 16) + 8;
    const int widest = nv > nn ? (nv > ndn ? nv : ndn) : (nn > ndn ? nn : ndn);
    p.rec_w = bk_pad(32 * bk_ks(widest), 16) + 8;
    return p;
}
// everything of two plans but the activation kinds
inline bool bk_same_shape(const RnnPlan &a, const RnnPlan &b)
{
    auto gd = [](const GemmDesc &x, const GemmDesc &y) { return x.wofs == y.wofs && x.ksteps == y.ksteps && x.kbase == y.kbase && x.ngates == y.ngates; };
    auto ld = [&](const LayerDesc &x, const LayerDesc &y) {
        return gd(x.in, y.in) && gd(x.rec, y.rec) && x.n == y.n && x.nb == y.nb && x.bias == y.bias && x.out_col == y.out_col;
    };
    return a.in_w == b.in_w && a.rec_w == b.rec_w && a.cF == b.cF && a.cV == b.cV && ld(a.dense, b.dense) && ld(a.vad, b.vad) && ld(a.noise, b.noise) &&
           ld(a.dn, b.dn) && ld(a.out, b.out) && a.vo_w == b.vo_w && a.vo_b == b.vo_b;
}
// the built-in model's class (src/weights.rnn; GregorR's rnnoise-models share it): 24 / 24 / 48 / 96
struct BkShapeBuiltin { static constexpr RnnPlan plan() { return bk_make_plan(24, 24, 48, 96); } };

// byte offsets into the kernel's dynamic LDS
struct BackLds {
    int tab, live, flag, vadl, gout, crs, dcw, cnw, FS, SPv, SPn, SPdn, tbl, U, IN, RS, ZB, Z, part, total;
    int sw_v, sw_n, sw_dn;
};
constexpr int bk_take(int &at, int bytes) { const int r = at; at += (bytes + 15) & ~15; return r; }
constexpr BackLds back_lds(const RnnPlan &pl, bool fused)
{
    BackLds o{};
    o.sw_v = 32 * pl.vad.rec.ksteps + 8; o.sw_n = 32 * pl.noise.rec.ksteps + 8; o.sw_dn = 32 * pl.dn.rec.ksteps + 8;
    int at = 0;
    o.tab = bk_take(at, 256 * 4);
    o.live = bk_take(at, BK_ROWS * 4);
    o.flag = bk_take(at, BK_ROWS * 4);
    o.vadl = bk_take(at, BK_ROWS * 4);
    o.gout = bk_take(at, BK_ROWS * BK_GW * 4);
    o.crs = bk_take(at, BK_ROWS * CEPS_MEM * NB * 4);
    o.dcw = bk_take(at, BK_ROWS * BK_CW * 4);
    o.cnw = bk_take(at, BK_ROWS * BK_CW * 4);
    o.FS = bk_take(at, 3 * BK_ROWS * FS_W * 2);
    o.SPv = bk_take(at, 3 * BK_ROWS * o.sw_v * 2);
    o.SPn = bk_take(at, 3 * BK_ROWS * o.sw_n * 2);
    o.SPdn = bk_take(at, 3 * BK_ROWS * o.sw_dn * 2);
    o.tbl = fused ? bk_take(at, (int)sizeof(FftLds)) : at;
    o.U = at;
    // the RNN's per-frame operands ...
    o.IN = bk_take(at, 3 * BK_ROWS * pl.in_w * 2);
    o.RS = bk_take(at, 3 * BK_ROWS * pl.rec_w * 2);
    o.ZB = bk_take(at, BK_ROWS * BK_ZW * 4);
    const int end_rnn = at;
    // ... share their space with the transforms' buffers: neither outlives its stretch of the frame
    at = o.U;
    o.Z = bk_take(at, fused ? BK_ROWS * NFFT_BUF * 8 : 0);
    o.part = bk_take(at, fused ? BK_ROWS * 3 * 64 * 4 : 0);
    o.total = at > end_rnn ? at : end_rnn;
    return o;
}

// ---- the feature stage of one frame for one stream on its own wave (ref: src/features.rs:170-219): ring update, the 7 pair distances
//      the new cepstrum takes part in (lane = partner row), the 42 outputs (lane = output).  Same operations in the same order as
//      features_row / wf_features; `cnw` holds the frame's cepstrum (22) and pitch-correlation DCT (6).
__device__ __forceinline__ void bk_features(const Buffers &b, int f, int tile, int trow, int row, int lane, int pitch, bool silent,
                                            float *crs, float *dcw, const float *cnw, unsigned short *FS, int *live, int &mem_id)
{
    float fr = 0.0f;
    if (!silent) {   // (wave-uniform) "if there's no audio, avoid messing up the state" (ref: src/features.rs:160-166)
        const int c0 = mem_id, c1 = mem_id < 1 ? CEPS_MEM + mem_id - 1 : mem_id - 1;
        const int c2 = mem_id < 2 ? CEPS_MEM + mem_id - 2 : mem_id - 2;
        if (lane < NB) {
            const float v = cnw[lane];
            crs[c0 * NB + lane] = v;
            NNN_TI(b.ceps_mem, CEPS_MEM * NB, tile, trow)[(size_t)(c0 * NB + lane) * TILE] = v;
        }
        mem_id = mem_id + 1 == CEPS_MEM ? 0 : mem_id + 1;
        if (lane < CEPS_MEM && lane != c0) {
            // squared distance of the new row to ring row `lane`, summed over the 22 bands in order (ref: src/features.rs:203-208)
            const float *rj = crs + lane * NB;
            float dist = 0.0f;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const float d = cnw[k] - rj[k];
                dist += d * d;
            }
            dcw[pair_index(lane < c0 ? lane : c0, lane < c0 ? c0 : lane)] = dist;
        }
        wave_lds_sync();
        if (lane < NFEAT) {
            const int k = lane;
            const int i = k < 6 ? k : (k >= 28 ? k - 28 : (k >= NB ? k - NB : 0));   // band of the delta features
            const int ic = i < 6 ? i : 0;
            const float v0 = cnw[k < NB ? k : (k < 34 ? ic : (k < 40 ? NB + (k - 34) : 0))];
            const float v1 = crs[c1 * NB + ic], v2 = crs[c2 * NB + ic];
            if (k < 6) fr = v0 + v1 + v2;
            else if (k < NB) fr = v0;
            else if (k < NB + 6) fr = v0 - v2;
            else if (k < NB + 12) fr = v0 - 2.0f * v1 + v2;
            else if (k < 40) fr = v0;
            else if (k == 40) fr = 0.01f * ((float)pitch - 300.0f);
            else fr = spectral_variability(dcw, 0, 1);
        }
    }
    if (lane == 0) live[row] = silent ? 0 : 1;
    if (lane < NFEAT) {
        if (b.taps) NNN_TIF(b, feat, NFEAT, f, tile, trow)[(size_t)lane * TILE] = fr;
        store_split(FS, BK_ROWS * FS_W, row * FS_W + lane, fr);
    }
}

// ---- the RNN on 16 rows, 16 waves ---------------------------------------------------------------------------------------------
struct BkRnn {
    const float *tab;
    const int *live;
    unsigned short *IN, *RS;
    float *ZB;
    int in_ps, in_w, rs_ps, rec_w;
};

// acc += A[16 rows][kbase ..] * B(one gate of one neuron block) over all k-steps and the three activation planes, in gemm_acc's order
// (k-step major; planes hi, mid, lo inside a k-step): the same chain of MFMAs per output element as k_rnn / k_rnn_wf issue.  `Bg` points
// at the gate's fragments [k-step][lane].  Written for small code (the kernel holds eight of these per GRU layer): the first KSMAX
// k-steps' fragments are requested together, each k-step is a uniform branch; the next k-step's operand planes are read before the
// current one's MFMAs issue.
struct BkFrags { uint4 f[KSMAX]; };
__device__ __forceinline__ void bk_frags_load(BkFrags &fr, const GemmDesc &g, const uint4 *__restrict__ Bg, int lane)
{
#pragma unroll
    for (int ks = 0; ks < KSMAX; ks++) fr.f[ks] = ks < g.ksteps ? Bg[ks * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ f32x4 bk_gemm(f32x4 acc, const unsigned short *A, int plane_stride, int row_w, const GemmDesc &g,
                                         const uint4 *__restrict__ Bg, int lane, const BkFrags &fr)
{
    const unsigned short *a0 = A + (size_t)(lane & 15) * row_w + g.kbase + 8 * (lane >> 4);
    uint4 cur[3], nxt[3];
#pragma unroll
    for (int pl = 0; pl < GPL; pl++) cur[pl] = *(const uint4 *)(a0 + (size_t)pl * plane_stride);
#pragma unroll
    for (int ks = 0; ks < KSMAX; ks++) {
        if (ks < g.ksteps) {
            if (ks + 1 < g.ksteps) {
#pragma unroll
                for (int pl = 0; pl < GPL; pl++) nxt[pl] = *(const uint4 *)(a0 + (size_t)pl * plane_stride + (ks + 1) * 32);
            }
#pragma unroll
            for (int pl = 0; pl < GPL; pl++) acc = mfma_16x16x32_bf16(cur[pl], fr.f[ks], acc);
#pragma unroll
            for (int pl = 0; pl < GPL; pl++) cur[pl] = nxt[pl];
        }
    }
#pragma nounroll
    for (int ks = KSMAX; ks < g.ksteps; ks++) {   // (layers wider than 4 k-steps: fetch as we go)
        const uint4 bf = Bg[ks * 64 + lane];
        if (ks + 1 < g.ksteps) {
#pragma unroll
            for (int pl = 0; pl < GPL; pl++) nxt[pl] = *(const uint4 *)(a0 + (size_t)pl * plane_stride + (ks + 1) * 32);
        }
#pragma unroll
        for (int pl = 0; pl < GPL; pl++) acc = mfma_16x16x32_bf16(cur[pl], bf, acc);
#pragma unroll
        for (int pl = 0; pl < GPL; pl++) cur[pl] = nxt[pl];
    }
    return acc;
}

// what a candidate unit keeps from the first phase of its layer to the second
struct BkH {
    f32x4 acc;
    int nbi;
    bool on;
};

// One GRU layer on the block's 16 waves (ref: src/rnn.rs:292-327).  Units 0 .. 2 nb - 1 are the (neuron block, z | r) pairs: bias,
// recurrent product on the state planes, input product; z -> ZB, r * state -> RS.  Units 2 nb .. 3 nb - 1 are the candidates: bias +
// input product in the first phase; behind the barrier the recurrent product on r * state, the activation and the state update, the
// new state going to the layer's state planes and to its columns of the input matrix.  Wave w takes units w and w + 16 (layers of
// up to 8 neuron blocks; at most one of a wave's two units is a candidate).
// The weight fragments of a wave's first unit of a phase (bias included) are requested BEFORE the barrier that opens the phase
// (bk_gru_load_a / bk_gru_load_b, called at the end of the phase before): a phase then starts with its operands in registers
// instead of with a trip to the L2 -- the trip is 1 of the 2 us a phase took without it.
struct BkW {
    BkFrags rec, in;
    float bias;
};
__device__ __forceinline__ void bk_unit_of(const LayerDesc &L, int u, bool &cand, int &nbi, int &gate)
{
    const int nzr = 2 * L.nb;
    cand = u >= nzr;
    nbi = cand ? u - nzr : u >> 1;
    gate = cand ? 2 : (u & 1);
}
__device__ __forceinline__ void bk_gru_load_a(const LayerDesc &L, const uint4 *__restrict__ Wq, const float *__restrict__ fpar, int u, int lane, BkW &w)
{
    if (u >= 3 * L.nb) return;
    bool cand;
    int nbi, gate;
    bk_unit_of(L, u, cand, nbi, gate);
    const int neuron = nbi * 16 + (lane & 15);
    const uint4 *Bin = Wq + L.in.wofs + ((size_t)nbi * 3 + gate) * L.in.ksteps * 64;
    const uint4 *Brec = Wq + L.rec.wofs + ((size_t)nbi * 3 + gate) * L.rec.ksteps * 64;
    if (!cand) bk_frags_load(w.rec, L.rec, Brec, lane);
    bk_frags_load(w.in, L.in, Bin, lane);
    w.bias = neuron < L.n ? fpar[L.bias + gate * L.n + neuron] : 0.0f;
}
__device__ __forceinline__ void bk_gru_phase_a(const LayerDesc &L, const BkRnn &R, const unsigned short *SP, int sw, const uint4 *__restrict__ Wq,
                                               const float *__restrict__ fpar, int wave, int lane, BkW &w, BkH &h)
{
    const float scale = 1.0f / 256.0f;
    const int units = 3 * L.nb, ps = BK_ROWS * sw, kcols = 32 * L.rec.ksteps;
    h.on = false;
    h.nbi = 0;
    h.acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma nounroll
    for (int rd = 0; rd < 2; rd++) {
        const int u = wave + BK_WAVES * rd;
        if (u >= units) break;
        if (rd) bk_gru_load_a(L, Wq, fpar, u, lane, w);   // (the second unit of a wave: fetched on the spot)
        bool cand;
        int nbi, gate;
        bk_unit_of(L, u, cand, nbi, gate);
        const int neuron = nbi * 16 + (lane & 15);
        const bool nvalid = neuron < L.n;
        const uint4 *Bin = Wq + L.in.wofs + ((size_t)nbi * 3 + gate) * L.in.ksteps * 64;
        const uint4 *Brec = Wq + L.rec.wofs + ((size_t)nbi * 3 + gate) * L.rec.ksteps * 64;
        f32x4 acc = f32x4{w.bias, w.bias, w.bias, w.bias};
        if (!cand) acc = bk_gemm(acc, SP, ps, sw, L.rec, Brec, lane, w.rec);
        acc = bk_gemm(acc, R.IN, R.in_ps, R.in_w, L.in, Bin, lane, w.in);
        if (cand) {
            h.acc = acc;
            h.nbi = nbi;
            h.on = true;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = 4 * (lane >> 4) + q;
                const float sg = sigmoid_approx(scale * acc[q], R.tab);
                if (gate == 0) {
                    R.ZB[row * BK_ZW + neuron] = sg;
                } else {
                    const float so = nvalid ? load_split(SP, ps, row * sw + neuron) : 0.0f;   // the three planes hold the state exactly
                    const float rs = so * sg;
                    if (neuron < kcols) store_split(R.RS, R.rs_ps, row * R.rec_w + neuron, rs);
                    if (nbi == L.nb - 1 && neuron + 16 < kcols) store_split(R.RS, R.rs_ps, row * R.rec_w + neuron + 16, 0.0f);
                }
            }
        }
    }
}
// the candidate unit's recurrent fragments, requested ahead of the barrier between the two phases
__device__ __forceinline__ void bk_gru_load_b(const LayerDesc &L, const uint4 *__restrict__ Wq, int lane, const BkH &h, BkW &w)
{
    if (!h.on) return;
    const uint4 *Brec = Wq + L.rec.wofs + ((size_t)h.nbi * 3 + 2) * L.rec.ksteps * 64;
    bk_frags_load(w.rec, L.rec, Brec, lane);
}
__device__ __forceinline__ void bk_gru_phase_b(const LayerDesc &L, const BkRnn &R, unsigned short *SP, int sw, const uint4 *__restrict__ Wq, int lane,
                                               const BkW &w, const BkH &h)
{
    if (!h.on) return;
    const float scale = 1.0f / 256.0f;
    const int ps = BK_ROWS * sw;
    const int nbi = h.nbi, neuron = nbi * 16 + (lane & 15);
    const uint4 *Brec = Wq + L.rec.wofs + ((size_t)nbi * 3 + 2) * L.rec.ksteps * 64;
    const f32x4 acc = bk_gemm(h.acc, R.RS, R.rs_ps, R.rec_w, L.rec, Brec, lane, w.rec);
    if (neuron < L.n) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = 4 * (lane >> 4) + q;
            const float hh = activate(L.act, scale * acc[q], R.tab);
            const float z = R.ZB[row * BK_ZW + neuron], so = load_split(SP, ps, row * sw + neuron);
            float snew = z * so + (1.0f - z) * hh;
            snew = R.live[row] ? snew : so;   // silent frames leave the state alone (ref: src/denoise.rs:100)
            store_split(R.IN, R.in_ps, row * R.in_w + L.out_col + neuron, snew);
            store_split(SP, ps, row * sw + neuron, snew);
        }
    }
}

// dense layer: neuron block w on wave w (its fragments and bias requested ahead by bk_dense_load); sink(row, neuron, value, q) with
// row = 4 (lane >> 4) + q
__device__ __forceinline__ void bk_dense_load(const LayerDesc &L, const uint4 *__restrict__ Wq, const float *__restrict__ fpar, int nbi, int lane, BkW &w)
{
    if (nbi >= L.nb) return;
    const int neuron = nbi * 16 + (lane & 15);
    bk_frags_load(w.in, L.in, Wq + L.in.wofs + (size_t)nbi * L.in.ksteps * 64, lane);
    w.bias = neuron < L.n ? fpar[L.bias + neuron] : 0.0f;
}
template <class Sink>
__device__ __forceinline__ void bk_dense(const LayerDesc &L, const BkRnn &R, const uint4 *__restrict__ Wq, const float *__restrict__ fpar, int wave,
                                         int lane, BkW &w, Sink &&sink)
{
#pragma nounroll
    for (int nbi = wave; nbi < L.nb; nbi += BK_WAVES) {
        if (nbi != wave) bk_dense_load(L, Wq, fpar, nbi, lane, w);
        const int neuron = nbi * 16 + (lane & 15);
        const uint4 *Bnb = Wq + L.in.wofs + (size_t)nbi * L.in.ksteps * 64;
        const f32x4 acc = bk_gemm(f32x4{w.bias, w.bias, w.bias, w.bias}, R.IN, R.in_ps, R.in_w, L.in, Bnb, lane, w.in);
        if (neuron < L.n) {
#pragma unroll
            for (int q = 0; q < 4; q++) sink(4 * (lane >> 4) + q, neuron, activate(L.act, acc[q] * (1.0f / 256.0f), R.tab), q);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_back<true>: the fused back end.  k_back<false>: its RNN stretch alone (features in from k_fft_xp's scratch, gains out to k_synth's).
// ---------------------------------------------------------------------------------------------------------------------------------
struct BkActs { int dense, vad, noise, dn, out, vo; };   // activation kinds of the model at hand (ref: src/rnn.rs:242-250)
// XR: the fused kernel of a one-frame call whose X transform rode in k_pitch's launch (transform_inputs, xt_rider).
template <bool FUSED, class SH, bool XR = false>
__global__ void __launch_bounds__(BK_T) k_back(Buffers b, const StepParams *sp0, BkActs acts, const uint4 *__restrict__ Wq,
                                              const float *__restrict__ fpar, int tile0, int g)
{
    HIP_DYNAMIC_SHARED(float, lds_raw)
    char *lds = (char *)lds_raw;
    constexpr RnnPlan plan0 = SH::plan();
    constexpr BackLds o = back_lds(plan0, FUSED);
    RnnPlan pl = plan0;   // (compile-time values; the activation kinds come with the launch)
    pl.dense.act = acts.dense; pl.vad.act = acts.vad; pl.noise.act = acts.noise; pl.dn.act = acts.dn; pl.out.act = acts.out; pl.act_vo = acts.vo;
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane0 = threadIdx.x & 63;
    int wave = wave0, lane = lane0;
    constexpr int per = TILE / BK_ROWS;
    int tile, sub;
    xcd_tile_block((int)blockIdx.x, (tile0 & 7) ? 1 : (int)gridDim.x / per, per, tile, sub);
    tile += tile0;                                 // tile0: first tile of this model's run
    const int r0 = sub * BK_ROWS;                  // first row of the tile handled here
    if (tile * TILE + r0 >= b.S) return;   // (a block whose streams are all padding -- the last tile of a batch that is not a multiple of 64 -- has nothing to do)
    const int sl = r0 + wave0, s = tile * TILE + sl;   // this wave's stream: its row in the tile, its index in the batch
    // ---- LDS
    float *tab = (float *)(lds + o.tab);
    int *live = (int *)(lds + o.live), *flagw = (int *)(lds + o.flag) + wave0;
    float *vadl = (float *)(lds + o.vadl), *gout = (float *)(lds + o.gout);
    float *crs = (float *)(lds + o.crs) + wave0 * (CEPS_MEM * NB), *dcw = (float *)(lds + o.dcw) + wave0 * BK_CW,
          *cnw = (float *)(lds + o.cnw) + wave0 * BK_CW;
    unsigned short *FS = (unsigned short *)(lds + o.FS);
    unsigned short *SPv = (unsigned short *)(lds + o.SPv), *SPn = (unsigned short *)(lds + o.SPn), *SPdn = (unsigned short *)(lds + o.SPdn);
    unsigned short *IN = (unsigned short *)(lds + o.IN), *RS = (unsigned short *)(lds + o.RS);
    float *ZB = (float *)(lds + o.ZB);
    FftLds &t = *(FftLds *)(lds + o.tbl);
    float2 *Z = (float2 *)(lds + o.Z) + wave0 * NFFT_BUF;
    float *part = (float *)(lds + o.part) + wave0 * (3 * 64);
    const int in_ps = BK_ROWS * pl.in_w, rs_ps = BK_ROWS * pl.rec_w;
    const BkRnn R{tab, live, IN, RS, ZB, in_ps, pl.in_w, rs_ps, pl.rec_w};
    float *sv = b.gru_v + ((size_t)tile * TILE * b.gru_v_w + (size_t)r0 * pl.vad.n),
          *sn = b.gru_n + ((size_t)tile * TILE * b.gru_n_w + (size_t)r0 * pl.noise.n),
          *sdn = b.gru_dn + ((size_t)tile * TILE * b.gru_dn_w + (size_t)r0 * pl.dn.n);
    // ---- once per launch: zero the persistent operand planes (padding columns must read as 0), tables, ring, states.  Everything that
    //      comes from global memory is requested first and lands while the planes are zeroed; only the transforms' tables have to be
    //      in place before the first frame's wave = stream stretch, which needs none of the rest.
    NNN_STAMP(b, 0);
    float *sm = b.synth_mem + (size_t)s * FRAME;
    int mem_id;
    {
        constexpr int NV = BK_ROWS * plan0.vad.n, NN_ = BK_ROWS * plan0.noise.n, ND = BK_ROWS * plan0.dn.n;
        constexpr int PV = (NV + BK_T - 1) / BK_T, PN = (NN_ + BK_T - 1) / BK_T, PD = (ND + BK_T - 1) / BK_T;
        float stg[3], gv[PV], gn[PN], gd[PD];
        const float *cm = NNN_TI(b.ceps_mem, CEPS_MEM * NB, tile, sl);
#pragma unroll
        for (int i = 0; i < 3; i++) stg[i] = lane + 64 * i < CEPS_MEM * NB ? cm[(size_t)(lane + 64 * i) * TILE] : 0.0f;
#pragma unroll
        for (int i = 0; i < PV; i++) gv[i] = (int)threadIdx.x + BK_T * i < NV ? sv[threadIdx.x + BK_T * i] : 0.0f;
#pragma unroll
        for (int i = 0; i < PN; i++) gn[i] = (int)threadIdx.x + BK_T * i < NN_ ? sn[threadIdx.x + BK_T * i] : 0.0f;
#pragma unroll
        for (int i = 0; i < PD; i++) gd[i] = (int)threadIdx.x + BK_T * i < ND ? sdn[threadIdx.x + BK_T * i] : 0.0f;
        mem_id = NNN_TI(b.mem_id, 1, tile, sl)[0];   // (wave-uniform)
        if (FUSED) fft_tables_load(t, b);
        for (int i = (int)threadIdx.x; i < 201; i += BK_T) tab[i] = b.tansig[i];
        uint4 *z = (uint4 *)FS;
        constexpr int n16 = (o.SPdn + 3 * BK_ROWS * o.sw_dn * 2 - o.FS) / 16;
        for (int i = (int)threadIdx.x; i < n16; i += BK_T) z[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (lane + 64 * i < CEPS_MEM * NB) crs[lane + 64 * i] = stg[i];
        lds_barrier();   // planes zeroed, tables and ring in place
        auto put = [&](unsigned short *SP, int sw, int n, int e, float v) {
            const int row = e / n, col = e - row * n;
            store_split(SP, BK_ROWS * sw, row * sw + col, v);
        };
#pragma unroll
        for (int i = 0; i < PV; i++)
            if ((int)threadIdx.x + BK_T * i < NV) put(SPv, o.sw_v, plan0.vad.n, threadIdx.x + BK_T * i, gv[i]);
#pragma unroll
        for (int i = 0; i < PN; i++)
            if ((int)threadIdx.x + BK_T * i < NN_) put(SPn, o.sw_n, plan0.noise.n, threadIdx.x + BK_T * i, gn[i]);
#pragma unroll
        for (int i = 0; i < PD; i++)
            if ((int)threadIdx.x + BK_T * i < ND) put(SPdn, o.sw_dn, plan0.dn.n, threadIdx.x + BK_T * i, gd[i]);
        if (lane < 28) dcw[lane] = pair_dist(crs, lane, 0, 1);
        // (the state planes are first read behind the barrier that closes the first frame's feature stage)
    }
    NNN_STAMP(b, 1);
    // (XR: a one-frame call by construction -- x_rides() -- so the frame loop is no loop: nothing is kept across it)
    for (int f = 0; f < (XR ? 1 : g); f++) {
        lane = XR ? lane0 : launder_v(lane0);   // keep the frame loop's addresses inside the loop (see launder_v)
        wave = XR ? wave0 : launder_s(wave0);
        const StepParams *sp = sp0 + f;
        const Buffers bf = frame_view(b, f);
        NNN_STAMP(b, 2);
        // ---------------- wave = stream: transforms, band quantities, feature head; then the stream's feature stage
        XpKeep K;
        int pitch;
        bool silent;
        if (FUSED) {
            K.sl = sl;
            K.cnw = cnw;
            K.flag = flagw;
            transform_inputs<true, true, XR>(bf, sp, tile, 0, t, Z, part, &K);
            pitch = NNN_TI(bf.pitch, 1, tile, sl)[0];
            silent = __builtin_amdgcn_readfirstlane(K.silent) != 0;
#ifdef NNN_PROBE_BACK_NOHOLD   // (developer probe, wrong audio, timing only: what the stretch costs when the spectra need not be kept)
#pragma unroll
            for (int u = 0; u < 8; u++) { K.X[u] = make_float2((float)u, 1.0f); K.P[u] = make_float2(1.0f, (float)u); }
#endif
        } else {
            const float cv = lane < 28 ? NNN_TI(bf.cn, 28, tile, sl)[(size_t)lane * TILE] : 0.0f;
            pitch = NNN_TI(bf.pitch, 1, tile, sl)[0];
            silent = __builtin_amdgcn_readfirstlane(NNN_TI(bf.silence, 1, tile, sl)[0]) != 0;
            if (lane < 28) cnw[lane] = cv;
        }
        wave_lds_sync();
        NNN_STAMP(b, 3);
        bk_features(b, f, tile, sl, wave, lane, pitch, silent, crs, dcw, cnw, FS, live, mem_id);
        NNN_STAMP(b, 4);
        // the last gains of the rows and bands this lane smooths (the output layer's units), needed at the end of the RNN stretch
        float lastg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (wave < pl.out.nb) {
            const int band = wave * 16 + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (band < NB) lastg[q] = NNN_TI(b.lastg, NB, tile, r0 + 4 * (lane >> 4) + q)[(size_t)band * TILE];
        }
        lds_barrier();   // every stream's features staged; the transforms' buffers are free: the RNN's operands take their space
        NNN_STAMP(b, 5);
        // ---------------- 16 waves, one layer at a time
        {   // the input matrix of this frame: zeros (padding columns must read as 0) with the staged features in their columns
            const int w8 = pl.in_w / 8, n16 = 3 * BK_ROWS * w8, c0 = pl.cF / 8;
            for (int i = (int)threadIdx.x; i < n16; i += BK_T) {
                const int plx = i / (BK_ROWS * w8), rem = i - plx * (BK_ROWS * w8), row = rem / w8, c8 = rem - row * w8;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (c8 >= c0 && c8 < c0 + 6) v = *(const uint4 *)(FS + (size_t)plx * BK_ROWS * FS_W + row * FS_W + 8 * (c8 - c0));
                *(uint4 *)(IN + (size_t)plx * in_ps + row * pl.in_w + 8 * c8) = v;
            }
        }
        // the weight fragments of a phase are requested before the barrier that opens it (see BkW); NNN_BK_PREFETCH_FUSED=0 builds the fused
        // kernel without (measured: 61.3 against 59.2 us per one-frame launch at 4096 streams -- the registers the prefetch takes are not
        // what makes the fused kernel spill its spectra)
#ifndef NNN_BK_PREFETCH_FUSED
#define NNN_BK_PREFETCH_FUSED 1
#endif
        constexpr bool PF = !FUSED || NNN_BK_PREFETCH_FUSED;
#define BK_EDGE(LOAD) do { if (PF) { LOAD; lds_barrier(); } else { lds_barrier(); LOAD; } } while (0)
        BkW W;
        BkH H;
        BK_EDGE(bk_dense_load(pl.dense, Wq, fpar, wave, lane, W));
        NNN_STAMP(b, 6);
        // input dense (ref: src/rnn.rs:353-355)
        bk_dense(pl.dense, R, Wq, fpar, wave, lane, W, [&](int row, int neuron, float v, int) {
            store_split(IN, in_ps, row * pl.in_w + pl.dense.out_col + neuron, v);
        });
        BK_EDGE(bk_gru_load_a(pl.vad, Wq, fpar, wave, lane, W));
        NNN_STAMP(b, 7);
        // vad GRU (ref: src/rnn.rs:356-358)
        bk_gru_phase_a(pl.vad, R, SPv, o.sw_v, Wq, fpar, wave, lane, W, H);
        BK_EDGE(bk_gru_load_b(pl.vad, Wq, lane, H, W));   // z and r * state complete; every wave is done reading the old state planes and the layer's inputs
        bk_gru_phase_b(pl.vad, R, SPv, o.sw_v, Wq, lane, W, H);
        BK_EDGE(bk_gru_load_a(pl.noise, Wq, fpar, wave, lane, W));
        NNN_STAMP(b, 8);
        // noise GRU (ref: src/rnn.rs:361-366); beside its first phase, on a wave it leaves idle, the vad output
        bk_gru_phase_a(pl.noise, R, SPn, o.sw_n, Wq, fpar, wave, lane, W, H);
        if (wave == BK_WAVES - 1 && lane < BK_ROWS) {   // 1 x nv, lane = stream (ref: src/rnn.rs:359)
            float acc = fpar[pl.vo_b];
            for (int k = 0; k < pl.vad.n; k++) acc = fmaf(fpar[pl.vo_w + k], load_split(IN, in_ps, lane * pl.in_w + pl.cV + k), acc);
            const float v = live[lane] ? activate(pl.act_vo, acc * (1.0f / 256.0f), tab) : 0.0f;
            vadl[lane] = v;
            NNN_TIF(b, vad, 1, f, tile, r0 + lane)[0] = v;
        }
        BK_EDGE(bk_gru_load_b(pl.noise, Wq, lane, H, W));
        bk_gru_phase_b(pl.noise, R, SPn, o.sw_n, Wq, lane, W, H);
        BK_EDGE(bk_gru_load_a(pl.dn, Wq, fpar, wave, lane, W));
        NNN_STAMP(b, 9);
        // denoise GRU (ref: src/rnn.rs:368-377)
        bk_gru_phase_a(pl.dn, R, SPdn, o.sw_dn, Wq, fpar, wave, lane, W, H);
        BK_EDGE(bk_gru_load_b(pl.dn, Wq, lane, H, W));
        bk_gru_phase_b(pl.dn, R, SPdn, o.sw_dn, Wq, lane, W, H);
        BK_EDGE(bk_dense_load(pl.out, Wq, fpar, wave, lane, W));
        NNN_STAMP(b, 10);
        // gains (ref: src/rnn.rs:378) and smoothing g = max(g, 0.6 lastg) (ref: src/denoise.rs:106-109)
        bk_dense(pl.out, R, Wq, fpar, wave, lane, W, [&](int lrow, int band, float v, int q) {
            const int row = r0 + lrow;
            const bool lv = live[lrow] != 0;
            const float gr = lv ? v : 0.0f;
            NNN_TIF(b, g_raw, NB, f, tile, row)[(size_t)band * TILE] = gr;
            float gs = 0.0f;
            if (lv) {
                const float lg = q == 0 ? lastg[0] : (q == 1 ? lastg[1] : (q == 2 ? lastg[2] : lastg[3]));
                gs = fmaxf(gr, 0.6f * lg);
                NNN_TI(b.lastg, NB, tile, row)[(size_t)band * TILE] = gs;
            }
            NNN_TIF(b, g, NB, f, tile, row)[(size_t)band * TILE] = gs;
            gout[lrow * BK_GW + band] = gr;
            gout[lrow * BK_GW + 24 + band] = gs;
        });
#undef BK_EDGE
        if (!FUSED) {
            lds_barrier();   // (the next frame's features may be staged)
            NNN_STAMP(b, 11);
            continue;
        }
        lds_barrier();   // gains and vad of every row in place; the RNN's operands are dead: the synthesis takes their space
        NNN_STAMP(b, 11);
        // ---------------- wave = stream: pitch filter, gains, inverse transform, overlap-add
        {
            const float b_graw = lane < NB ? gout[wave * BK_GW + lane] : 0.0f, b_g = lane < NB ? gout[wave * BK_GW + 24 + lane] : 0.0f;
            float4 smq[2];   // (the overlap memory: read and written inside, per frame)
            synth_frame<true>(b, sp, f, tile, sl, s, lane, t, Z, part, K.X, K.P, K.ex, K.ep, K.xn, b_graw, b_g, vadl[wave], !silent, sm, smq);
        }
        NNN_STAMP(b, 12);
    }
    // ---- states back to HBM
    {
        auto save_state = [&](const LayerDesc &L, float *state, const unsigned short *SP, int sw) {
            for (int e = (int)threadIdx.x; e < BK_ROWS * L.n; e += BK_T) {
                const int row = e / L.n, col = e - row * L.n;
                state[e] = load_split(SP, BK_ROWS * sw, row * sw + col);
            }
        };
        save_state(pl.vad, sv, SPv, o.sw_v);
        save_state(pl.noise, sn, SPn, o.sw_n);
        save_state(pl.dn, sdn, SPdn, o.sw_dn);
    }
    if (lane0 == 0) NNN_TI(b.mem_id, 1, tile, sl)[0] = mem_id;
    NNN_STAMP(b, 13);
}

}  // namespace nnn
