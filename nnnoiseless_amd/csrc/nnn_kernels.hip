// nnn_kernels.hip -- CDNA4 (gfx950) kernels for the batched nnnoiseless process_frame path.
//
// Compiled with -ffp-contract=off: everything upstream of the integer pitch index must round
// exactly like the scalar reference (which never fuses a*b+c); fmaf() is written explicitly
// where fusing is allowed (FFT, RNN mat-vecs, band sums: tolerance-only quantities).
//
// Mapping rule: a stage whose result feeds the pitch index runs lane = stream on
// the tile-interleaved (TI) layout, so every lane executes the reference's scalar recurrence in the
// reference's summation order at full lane utilisation, with extra parallelism (lag chunks, sample
// chunks, neuron blocks) spread over waves.  Stages with data-dependent addressing or a 960-point
// transform run wave = stream on the stream-major (SM) layout with the stream's data staged in LDS.
//
// Reference citations are into jneem/nnnoiseless v0.5.1.
#pragma once
#include "nnn_layout.h"
#include <nnn_mfma.h>

namespace nnn {

#define NNN_TI(ptr, len, tile, lane) ((ptr) + ((size_t)(tile) * (len)) * TILE + (lane))
// the same for a per-frame scratch array of frame `f` of a group (set f lies f * S_pad * len after set 0, see frame_view):
// kernels that loop over a group's frames address single fields this way instead of re-basing the whole argument block
#define NNN_TIF(b, field, len, f, tile, lane) ((b).field + ((size_t)(b).S_pad * (size_t)(f) + (size_t)(tile) * TILE) * (size_t)(len) + (lane))

// A launch of a kernel without cross-frame recurrence covers several consecutive frames: block index = frame * PER +
// block-of-frame.  Re-bases the (by-value) Buffers `b` on that frame's scratch set; `frame` and `bx` are left in scope.
#define NNN_FRAME_SPLIT(PER)                              \
    const int frame = (int)blockIdx.x / (int)(PER);      \
    const int bx = (int)blockIdx.x - frame * (int)(PER); \
    b = frame_view(b, frame);

// Block index -> (tile, block of the tile) for kernels that give a tile's 64 streams to `bpt` consecutive blocks.  Workgroup i runs on
// XCD i mod 8 (observed dispatch order; a speed matter only), so consecutive blocks would spread a tile over several XCDs -- and the
// tile-interleaved per-stream scalars (band energies, gains, cepstrum, pitch: 64 streams to a 256-byte row) would be fetched into,
// and written back from, as many L2s.  Tile t goes to XCD t mod 8 instead: where k_hp's block t and k_pitch's blocks of tile t ran.
__device__ __forceinline__ void xcd_tile_block(int blk, int ntiles, int bpt, int &tile, int &sub)
{
    int m8 = ntiles & ~7;   // the tiles that come in eights are dealt to the XCDs; the last few keep block order
#ifdef NNN_NO_XCD_MAP
    m8 = 0;
#endif
    if (blk < m8 * bpt) {
        const int xcd = blk & 7, j = blk >> 3;
        tile = xcd + 8 * (j / bpt);
        sub = j % bpt;
    } else {
        const int r = blk - m8 * bpt;
        tile = m8 + r / bpt;
        sub = r % bpt;
    }
}
// the same for launches that cover `n` units (frames, chunks of frames) per tile-block, the units of a tile-block on consecutive
// blocks of its XCD: blk -> (unit, tile, sub)
__device__ __forceinline__ void xcd_tile_block_units(int blk, int ntiles, int bpt, int n, int &unit, int &tile, int &sub)
{
    int m8 = ntiles & ~7;
#ifdef NNN_NO_XCD_MAP
    m8 = 0;
#endif
    if (blk < m8 * bpt * n) {
        const int xcd = blk & 7, j = blk >> 3, tb = j / n;
        unit = j - tb * n;
        tile = xcd + 8 * (tb / bpt);
        sub = tb % bpt;
    } else {
        // (what is left keeps the order the kernels had before: unit-major)
        const int r = blk - m8 * bpt * n, per = (ntiles - m8) * bpt;
        unit = r / per;
        const int q = r - unit * per;
        tile = m8 + q / bpt;
        sub = q % bpt;
    }
}

// Optional phase stamps (developer instrumentation, off in the shipped build): block 0 / thread 0 records the
// shader clock at labelled points so a phase breakdown can be read back through nnn_batch_read_stamps.
#ifdef NNN_STAMPS
#define NNN_STAMP(b, i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) (b).stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
// the same from lane 0 of any wave of block 0, when `cond` holds (role-by-role breakdowns)
#define NNN_STAMPW(b, i, cond) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (cond)) (b).stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#ifndef NNN_WFSTAMP_PHASE
#define NNN_WFSTAMP_PHASE 1
#endif
#else
#define NNN_STAMP(b, i) do { } while (0)
#define NNN_STAMPW(b, i, cond) do { } while (0)
#endif

// Bark-ish band edges in units of 4 bins (ref: src/lib.rs:55-58) and SECOND_CHECK (ref: src/pitch.rs:489)
// (internal linkage: the library is two translation units since round 6 -- nnn_hp.hip -- and each carries its own copy)
#ifdef __HIPCC__
#define NNN_CONSTANT static __constant__
#else
#define NNN_CONSTANT __constant__   // (the tests' interpreter build defines __constant__ as static)
#endif
NNN_CONSTANT int kEband[NB] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};
NNN_CONSTANT int kSecondCheck[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};

// ---- PCM formats at the boundary ------------------------------------------------------------------
// Input conversions of the reference's callers: i16 samples are used as they are (src/nnnoiseless.rs:179-228 hands
// i16-range floats to process_frame), unit-range floats are scaled by 32768 (src/signal.rs:95-100).
template <int FMT> __device__ __forceinline__ float pcm_load(const char *p)
{
    if (FMT == PCM_I16) return (float)ld_global<short>(p);
    const float v = ld_global<float>(p);
    return FMT == PCM_F32_UNIT ? v * 32768.0f : v;
}
// Output conversions: round-half-away + clamp to i16 (RawFrameWriter / WavFrameWriter, src/nnnoiseless.rs:147-177),
// /32768 then clamp to [-1, 1] (DenoiseSignal::next, src/signal.rs:123-127).
__device__ __forceinline__ short pcm_to_i16(float v) { return (short)roundf(fminf(fmaxf(v, -32768.0f), 32767.0f)); }
__device__ __forceinline__ float pcm_to_unit(float v)
{
    v = v / 32768.0f;
    if (v < -1.0f) v = -1.0f;
    if (v > 1.0f) v = 1.0f;
    return v;
}
__device__ __forceinline__ void pcm_store(char *p, int fmt, float v)
{
    if (fmt == PCM_I16) *(short *)p = pcm_to_i16(v);
    else *(float *)p = fmt == PCM_F32_UNIT ? pcm_to_unit(v) : v;
}
// One lane's next HP_CH samples of its own stream, kept in raw form until the recurrence of the previous HP_CH is done (so
// the loads stay in flight behind it).  VEC: mono stream with 16-byte aligned rows, 16 bytes per load.
// (HP_CH = 16 at 168 registers, tried so that the 64 lone waves of a 4096-stream launch would slip in beside the group's other
// kernels: 20 -> 26 us per frame on its own, 70 -> 90 at 65536 streams, and no gain in the pipelined run.)
#ifndef NNN_HP_CH
#define NNN_HP_CH 32
#endif
constexpr int HP_CH = NNN_HP_CH;
template <int FMT, bool VEC> struct HpChunk {
    static constexpr int NV = FMT == PCM_I16 ? HP_CH / 8 : HP_CH / 4;
    uint4 v[VEC ? NV : 1];
    unsigned w[VEC ? 1 : HP_CH];
    __device__ __forceinline__ void load(const char *p, int sstride)
    {
        if (VEC) {
#pragma unroll
            for (int q = 0; q < NV; q++) v[q] = ld_global_u4(p + 16 * q);
        } else {
#pragma unroll
            for (int j = 0; j < HP_CH; j++)
                w[j] = FMT == PCM_I16 ? (unsigned)(int)ld_global<short>(p + (long long)j * sstride)
                                      : ld_global<unsigned>(p + (long long)j * sstride);
        }
    }
    __device__ __forceinline__ void get(float (&x)[HP_CH]) const
    {
        if (VEC && FMT == PCM_I16) {
#pragma unroll
            for (int q = 0; q < NV; q++) {
                const unsigned u[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    x[8 * q + 2 * e] = (float)(short)(u[e] & 0xffffu);
                    x[8 * q + 2 * e + 1] = (float)((int)u[e] >> 16);
                }
            }
        } else if (VEC) {
#pragma unroll
            for (int q = 0; q < NV; q++) {
                x[4 * q] = __uint_as_float(v[q].x); x[4 * q + 1] = __uint_as_float(v[q].y);
                x[4 * q + 2] = __uint_as_float(v[q].z); x[4 * q + 3] = __uint_as_float(v[q].w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < HP_CH; j++) x[j] = FMT == PCM_I16 ? (float)(int)w[j] : __uint_as_float(w[j]);
        }
        if (FMT == PCM_F32_UNIT) {
#pragma unroll
            for (int j = 0; j < HP_CH; j++) x[j] *= 32768.0f;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// K1  hp_filter: high-pass biquad (f64 arithmetic, f32 state) + append to the history ring
//     (ref: src/features.rs:97-104, src/util.rs:95-107), and the 2:1 decimation of pitch_downsample
//     (ref: src/pitch.rs:455-458) done incrementally: decimated sample d = ((s[2d-1] + s[2d+1])/2 + s[2d])/2
//     depends only on absolute samples, so each frame adds 240 values to a persistent ring instead of
//     recomputing all 864; only the reference's special first element is per frame.  The first 960 values of the
//     ring are mirrored behind its end: every frame's 864-value window is then one contiguous run for its reader.
//     lane = stream; the 480-step recurrence is inherently serial per stream and a lone wave is bound by
//     instruction issue, so each lane moves its own stream's samples with 16-byte accesses (a full 128-byte
//     line per 32 samples) instead of transposing tiles through LDS for coalescing.  One launch covers the `g`
//     consecutive frames of a group: the biquad state stays in registers from frame to frame.
// ---------------------------------------------------------------------------------------------
struct HpState { float m0, m1, prev; };

// The history ring is stored in rows: a lane's own 32 results are 128 contiguous bytes of ITS stream, so storing them directly
// makes every store instruction touch 64 cache lines with 16 bytes each -- measured (same box, stores left out) at 5 % of the whole
// pipeline's throughput at 4096 streams, more than the kernel's share of anything.  The results of a chunk therefore cross LDS
// (row stride 33 floats: conflict-free both ways) and leave as 8 stores of 8 streams x 128 contiguous bytes.
constexpr int HP_LD = HP_CH + 1;

// HP_CH steps of the biquad (ref: src/util.rs:95-107, coefficients :68-71): y = x + m0; m0 = f32(m1 + (b0 x - a0 y)); m1 = f32(b1 x - a1 y), in
// f64 with the state rounded to f32 each step.  b0 = -2 and b1 = 1: their products are exact, so b0 x - a0 y is ONE rounding of
// -2 x - fl(a0 y) -- what fma(x, -2, -fl(a0 y)) returns -- and b1 x is x.  Twelve f64 instructions per step on a chain of six.
__device__ __forceinline__ void hp_recurrence(const float (&xs)[HP_CH], float (&ys)[HP_CH], float &m0, float &m1)
{
    const double a0 = (double)-1.99599f, a1 = (double)0.99600f;
#pragma unroll
    for (int j = 0; j < HP_CH; j++) {
        const double x64 = (double)xs[j];
        const double y64 = x64 + (double)m0;
        m0 = (float)((double)m1 + fma(x64, -2.0, -(a0 * y64)));
        m1 = (float)(x64 - a1 * y64);
        ys[j] = (float)y64;
    }
}

template <int FMT, bool VEC>
__device__ __forceinline__ void hp_frame(const Buffers &b, const char *sp_in, long long sp_group_stride, int slot, int ch, int tile, int lane, HpState &st, float *Ly)
{
    const int elem = pcm_elem_bytes(FMT), sstride = ch * elem;
    const int s = tile * TILE + lane;
    // padding lanes of the last tile re-read the last real stream: their state is never looked at
    const int sc = s < b.S ? s : b.S - 1, grp = sc / ch;
    const char *in = sp_in + (long long)grp * sp_group_stride + (long long)(sc - grp * ch) * elem;
    // (two chunks in flight -- chunk c + 2 requested when chunk c leaves its registers -- was measured in round 4 for the lone waves of a
    // one-frame call: 378 registers, the launch 27 -> 34 us.  One chunk ahead it stays.)
    HpChunk<FMT, VEC> nxt;
    nxt.load(in, sstride);
    float m0 = st.m0, m1 = st.m1, prev = st.prev;
    NNN_STAMP(b, 24);
    const int nslot = b.nslot, hstr = hist_stride(nslot);
    float *ring = NNN_TI(b.dec, dec_len(nslot), tile, lane);
    float *h = b.hist + (size_t)s * hstr;
    {   // x_lp[0] = (x[1] / 2 + x[0]) / 2 on the oldest two samples of this frame's 1728-sample history: kept beside
        // the ring, per slot (the ring position it replaces is still a regular value for the previous frame)
        const int rb = ring_base(slot, nslot);
        const float x0 = h[rb], x1 = h[rb + 1];   // (rb + 1 = the ring's length reads the copy of sample 0 kept there)
        NNN_TI(b.xlp0, nslot, tile, lane)[(size_t)slot * TILE] = (x1 / 2.0f + x0) / 2.0f;
    }
    float *dec = ring + (size_t)(240 * slot) * TILE;
    const bool mirror = slot < DEC_MIRROR;
    float4 *hw = (float4 *)(h + slot * FRAME);   // the stride * 4 and FRAME * 4 are multiples of 16
    // Software pipeline over HP_CH-sample chunks.  Loads and stores share one in-order counter (vmcnt), so waiting for
    // chunk c's samples also waits for every store issued before: the stores of chunk c - 1 are therefore issued right
    // after that wait, and both they and the loads of chunk c + 1 travel behind the ~0.7 us recurrence of chunk c.
    float ys[HP_CH], dvs[HP_CH / 2];
    for (int c = 0; c <= FRAME / HP_CH; c++) {
        float xs[HP_CH];
        if (c < FRAME / HP_CH) nxt.get(xs);
        if (c > 0) {
#pragma unroll
            for (int t = 0; t < HP_CH / 2; t++) dec[(size_t)(HP_CH / 2 * (c - 1) + t) * TILE] = dvs[t];
            if (mirror) {
#pragma unroll
                for (int t = 0; t < HP_CH / 2; t++) dec[(size_t)(dec_ring_len(nslot) + HP_CH / 2 * (c - 1) + t) * TILE] = dvs[t];
            }
            if (HP_CH == 32) {
                wave_lds_sync();   // (the previous chunk's rows have been read)
#pragma unroll
                for (int j = 0; j < HP_CH; j++) Ly[lane * HP_LD + j] = ys[j];
                wave_lds_sync();
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int r = 8 * it + (lane >> 3);
                    const float *y = Ly + r * HP_LD + 4 * (lane & 7);
                    float4 *hr = (float4 *)(b.hist + (size_t)(tile * TILE + r) * hstr + slot * FRAME + (c - 1) * HP_CH) + (lane & 7);
                    *hr = make_float4(y[0], y[1], y[2], y[3]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < HP_CH / 4; q++) hw[HP_CH / 4 * (c - 1) + q] = make_float4(ys[4 * q], ys[4 * q + 1], ys[4 * q + 2], ys[4 * q + 3]);
            }
            if (slot == 0 && c == 1) h[ring_len(nslot)] = ys[0];   // the ring's first sample again behind its end (8-byte reads across the wrap)
        }
        if (c == FRAME / HP_CH) break;
        if (c + 1 < FRAME / HP_CH) nxt.load(in + (long long)(c + 1) * HP_CH * sstride, sstride);
        hp_recurrence(xs, ys, m0, m1);
#pragma unroll
        for (int t = 0; t < HP_CH / 2; t++) {
            const float a = t == 0 ? prev : ys[2 * t - 1], m = ys[2 * t], n = ys[2 * t + 1];
            dvs[t] = ((a + n) / 2.0f + m) / 2.0f;
        }
        prev = ys[HP_CH - 1];
    }
    st.m0 = m0; st.m1 = m1; st.prev = prev;
    NNN_STAMP(b, 25);
}

// Entry t of a call's per-frame parameter table from the call's own parameters `v` (frame 0): what k_fill_params writes, and what
// k_hp -- the first kernel of a call -- works out for itself when the table is its to fill.
__device__ __forceinline__ StepParams step_params_at(const StepParams &v, int t, int nslot)
{
    StepParams p = v;
    p.in = v.in + (long long)t * v.frame_stride;
    p.out = v.out + (long long)(t - v.discard) * v.frame_stride;   // dropped frames take no room in the output
    p.discard = t < v.discard;
    p.vad = v.vad ? v.vad + (size_t)t * v.n_streams : nullptr;
    p.slot = (v.slot + t) % nslot;
    p.log = (v.log && t < v.log_frames) ? v.log + (size_t)t * v.n_streams * FRAME_LOG_WORDS : nullptr;
    return p;
}

// `fill` > 0: this launch is the first of a call of `fill` frames whose table nobody has filled -- a launch of its own for that costs
// a one-frame call 6 of its 150 us -- so block 0 writes it (for the kernels behind this one, which start when this one is done) and
// every block takes its own frames' entries from the call's parameters `v0`; the group's first frame is entry `t0` of the call.
template <int FMT, bool VEC>
__device__ __forceinline__ void hp_group(const Buffers &b, const StepParams *sp, int g, int tile, int lane, float *Ly, const StepParams &v0, int fill)
{
    float *hp = NNN_TI(b.hp_mem, 2, tile, lane);
    float *hl = NNN_TI(b.hp_last, 1, tile, lane);
    HpState st{hp[0], hp[TILE], hl[0]};
    for (int f = 0; f < g; f++) {
        // (what a frame needs of its table entry: where its input starts and which ring slot takes it)
        const char *in = fill > 0 ? v0.in + (long long)f * v0.frame_stride : sp[f].in;
        const int slot = fill > 0 ? (v0.slot + f) % b.nslot : sp[f].slot;
        hp_frame<FMT, VEC>(b, in, fill > 0 ? v0.group_stride : sp[f].group_stride, slot, fill > 0 ? v0.channels : sp[f].channels, tile, lane, st, Ly);
    }
    hp[0] = st.m0;
    hp[TILE] = st.m1;
    hl[0] = st.prev;
}

// The first 16 LPC_HEAD_BLK steps of lag K's sum -- i + K < 624: rows older than the frame being filtered -- for k_hp2's head waves
// (lane = stream on the tile-interleaved ring, like k_lpc); k_pitch's pk_autocorr carries on from there in the same order.
constexpr int LPC_HEAD_BLK = 38;
static_assert(16 * LPC_HEAD_BLK + 4 <= XLP - 240, "");
template <int K>
__device__ __forceinline__ float lpc_head_chain(const float *base, float x0)
{
    constexpr int CH = 16;
    float cur[CH + 4], nxt[CH];
#pragma unroll
    for (int i = 0; i < CH + 4; i++) cur[i] = base[(size_t)i * TILE];
    cur[0] = x0;
    float c = 0.0f;
#pragma nounroll
    for (int ch = 0; ch < LPC_HEAD_BLK; ch++) {
        const float *nb = base + (size_t)((ch + 1 < LPC_HEAD_BLK ? ch + 1 : ch) * CH + 4) * TILE;
#pragma unroll
        for (int i = 0; i < CH; i++) nxt[i] = nb[(size_t)i * TILE];
#pragma unroll
        for (int j = 0; j < CH; j++) c += cur[j] * cur[j + K];
#pragma unroll
        for (int i = 0; i < 4; i++) cur[i] = cur[CH + i];
#pragma unroll
        for (int i = 0; i < CH; i++) cur[4 + i] = nxt[i];
    }
    return c;
}

// The same frame on TWO waves (k_hp2, launches that leave most of the GPU empty: a one-frame call of 4096 streams is 64 lone waves).  The
// recurrence issues 12 f64 instructions per step whatever else the wave does, and everything else -- the results' trip through LDS, 40
// stores per chunk, the decimation -- used to stand between one chunk's recurrence and the next (0.5 of every 1.3 us).  Here wave 0
// runs loads and recurrence only and leaves each chunk's results in one of two LDS buffers; wave 1 (another SIMD) takes them from there
// behind one block barrier per chunk and does the rest while wave 0 is a chunk further.  Same arithmetic, same bits.
template <int FMT, bool VEC>
__device__ __forceinline__ void hp_chain_frame(const Buffers &b, const char *sp_in, long long sp_group_stride, int ch, int tile, int lane, float &m0, float &m1, float *Ly2, int &k)
{
    const int elem = pcm_elem_bytes(FMT), sstride = ch * elem;
    const int s = tile * TILE + lane;
    const int sc = s < b.S ? s : b.S - 1, grp = sc / ch;
    const char *in = sp_in + (long long)grp * sp_group_stride + (long long)(sc - grp * ch) * elem;
    HpChunk<FMT, VEC> nxt;
    nxt.load(in, sstride);
    for (int c = 0; c < FRAME / HP_CH; c++, k++) {
        float xs[HP_CH], ys[HP_CH];
        nxt.get(xs);
        if (c + 1 < FRAME / HP_CH) nxt.load(in + (long long)(c + 1) * HP_CH * sstride, sstride);
        hp_recurrence(xs, ys, m0, m1);
        float *L = Ly2 + (k & 1) * (TILE * HP_LD) + lane * HP_LD;
#pragma unroll
        for (int j = 0; j < HP_CH; j++) L[j] = ys[j];
        __syncthreads();   // chunk k is in its buffer (and wave 1 is done with chunk k - 1: this buffer's turn again at k + 2)
    }
}
__device__ __forceinline__ void hp_store_frame(const Buffers &b, int slot, int tile, int lane, float &prev, const float *Ly2, int &k)
{
    const int s = tile * TILE + lane;
    const int nslot = b.nslot, hstr = hist_stride(nslot);
    float *ring = NNN_TI(b.dec, dec_len(nslot), tile, lane);
    float *h = b.hist + (size_t)s * hstr;
    {   // x_lp[0], as in hp_frame
        const int rb = ring_base(slot, nslot);
        const float x0 = h[rb], x1 = h[rb + 1];
        NNN_TI(b.xlp0, nslot, tile, lane)[(size_t)slot * TILE] = (x1 / 2.0f + x0) / 2.0f;
    }
    float *dec = ring + (size_t)(240 * slot) * TILE;
    const bool mirror = slot < DEC_MIRROR;
    for (int c = 0; c < FRAME / HP_CH; c++, k++) {
        __syncthreads();
        const float *L = Ly2 + (k & 1) * (TILE * HP_LD);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int r = 8 * it + (lane >> 3);
            const float *y = L + r * HP_LD + 4 * (lane & 7);
            float4 *hr = (float4 *)(b.hist + (size_t)(tile * TILE + r) * hstr + slot * FRAME + c * HP_CH) + (lane & 7);
            *hr = make_float4(y[0], y[1], y[2], y[3]);
        }
        float ys[HP_CH];
#pragma unroll
        for (int j = 0; j < HP_CH; j++) ys[j] = L[lane * HP_LD + j];
#pragma unroll
        for (int t = 0; t < HP_CH / 2; t++) {
            const float a = t == 0 ? prev : ys[2 * t - 1], m = ys[2 * t], n = ys[2 * t + 1];
            const float dv = ((a + n) / 2.0f + m) / 2.0f;
            dec[(size_t)(HP_CH / 2 * c + t) * TILE] = dv;
            if (mirror) dec[(size_t)(dec_ring_len(nslot) + HP_CH / 2 * c + t) * TILE] = dv;
        }
        prev = ys[HP_CH - 1];
        if (slot == 0 && c == 0) h[ring_len(nslot)] = ys[0];
    }
}
template <int FMT, bool VEC>
__device__ __forceinline__ void hp_chain_group(const Buffers &b, const StepParams *sp, int g, int tile, int lane, float *Ly2, const StepParams &v0, int fill)
{
    float *hp = NNN_TI(b.hp_mem, 2, tile, lane);
    float m0 = hp[0], m1 = hp[TILE];
    int k = 0;
    for (int f = 0; f < g; f++) {
        const char *in = fill > 0 ? v0.in + (long long)f * v0.frame_stride : sp[f].in;
        hp_chain_frame<FMT, VEC>(b, in, fill > 0 ? v0.group_stride : sp[f].group_stride, fill > 0 ? v0.channels : sp[f].channels, tile, lane, m0, m1, Ly2, k);
    }
    hp[0] = m0;
    hp[TILE] = m1;
}
// `head` (a one-frame launch whose LPC analysis runs inside k_pitch): blocks NT .. are not the high-pass at all -- each of their waves takes
// one (tile, lag) of the five autocorrelation sums through the rows of the frame's window that are older than the frame (608 of the 860
// steps: they end before the first decimated value this launch produces), while the recurrence above runs its 20 us; k_pitch then
// starts every sum there instead of at zero: 11 -> 3.5 us of its critical path.
// TPB tiles per block (waves 0 .. TPB - 1 the recurrences, waves TPB .. the helpers; TPB <= 2: every wave a SIMD of its own -- with four
// tiles a helper shares the SIMD of a recurrence and the kernel takes twice as long).  Groups run two tiles per block: half as many
// compute units carry a wave that takes most of its SIMD's issue slots from the pipelined call's other kernels (4096 x 48 +1 %, 8192 x 48
// +2 %; keeping k_pitch's blocks off those units altogether by padding the block's LDS: measured, no gain).
// The two high-pass kernels are compiled in a translation unit of their own (nnn_hp.hip) WITH the compiler's SLP pairing, the rest of the
// library without it (round 6): the recurrence is one wave's serial chain of f64 instructions, bound by that wave's own issue rate, and
// where the loads, address arithmetic and stores land between the chain's instructions decides its pace -- with the pairing pass on the
// same source is 14 % faster per frame on small launches (k_hp2 21.5 against 25.0 us per frame at 4096 streams, 26 against 29 us in a
// one-frame tick), while k_pitch, the transforms and the synthesis are 2-5 % faster without it.  Same instructions on the same values either
// way.  NNN_HP_EXTERN: this unit only declares them; NNN_ONLY_HP: this unit is nnn_hp.hip and defines nothing else.  A build that
// defines neither (the tests' interpreter, scripts/build_variant.sh, the stamp builds) holds everything in one unit, as before.
#ifdef NNN_HP_EXTERN
template <int TPB> __global__ void k_hp2(Buffers b, const StepParams *sp, int g, StepParams v0, int fill, int head);
extern template __global__ void k_hp2<1>(Buffers b, const StepParams *sp, int g, StepParams v0, int fill, int head);
extern template __global__ void k_hp2<2>(Buffers b, const StepParams *sp, int g, StepParams v0, int fill, int head);
__global__ void k_hp(Buffers b, const StepParams *sp, int g, StepParams v0, int fill);
#else
template <int TPB>
__global__ void __launch_bounds__(128 * TPB) k_hp2(Buffers b, const StepParams *sp, int g, StepParams v0, int fill, int head)
{
    static_assert(HP_CH == 32, "");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int role = wave / TPB, nblk = (b.NT + TPB - 1) / TPB;
    if ((int)blockIdx.x >= nblk) {
        const int item = 2 * TPB * ((int)blockIdx.x - nblk) + wave;
        if (!head || item >= 5 * b.NT) return;
        const int tile = item / 5, lag = item - 5 * tile, nslot = b.nslot;
        const int slot = fill > 0 ? v0.slot : sp->slot;
        const float *h = b.hist + (size_t)(tile * TILE + lane) * hist_stride(nslot);
        const int rb = ring_base(slot, nslot);
        const float x0 = (h[rb + 1] / 2.0f + h[rb]) / 2.0f;   // x_lp[0] is special (ref: src/pitch.rs:458); see hp_frame
        const float *base = b.dec + ((size_t)tile * dec_len(nslot) + (size_t)dec_base(slot, nslot)) * TILE + lane;
        float c;
        if (lag == 0) c = lpc_head_chain<0>(base, x0);
        else if (lag == 1) c = lpc_head_chain<1>(base, x0);
        else if (lag == 2) c = lpc_head_chain<2>(base, x0);
        else if (lag == 3) c = lpc_head_chain<3>(base, x0);
        else c = lpc_head_chain<4>(base, x0);
        b.lpc_head[(size_t)item * TILE + lane] = c;
        return;
    }
    const int tile = (int)blockIdx.x * TPB + (wave - role * TPB);
    __shared__ float Ly2s[TPB][2 * TILE * HP_LD];
    float *Ly2 = Ly2s[wave - role * TPB];
    if (tile >= b.NT) {   // (a ragged last block: its spare waves only keep the barrier count)
        for (int i = 0; i < g * (FRAME / HP_CH); i++) __syncthreads();
        return;
    }
    if (role == 1) {
        if (fill > 0 && tile == 0)
            for (int t = lane; t < fill; t += 64) ((StepParams *)sp)[t] = step_params_at(v0, t, b.nslot);
        float *hl = NNN_TI(b.hp_last, 1, tile, lane);
        float prev = hl[0];
        int k = 0;
        for (int f = 0; f < g; f++) hp_store_frame(b, fill > 0 ? (v0.slot + f) % b.nslot : sp[f].slot, tile, lane, prev, Ly2, k);
        hl[0] = prev;
        return;
    }
    const int fmt = fill > 0 ? v0.fmt : sp->fmt;
    wf_setprio_high();
    const StepParams &lay = fill > 0 ? v0 : *sp;
    const bool vec = lay.channels == 1 && ((((size_t)lay.in) | (size_t)lay.group_stride | (size_t)lay.frame_stride) & 15) == 0;
    if (fmt == PCM_F32) { if (vec) hp_chain_group<PCM_F32, true>(b, sp, g, tile, lane, Ly2, v0, fill); else hp_chain_group<PCM_F32, false>(b, sp, g, tile, lane, Ly2, v0, fill); }
    else if (fmt == PCM_I16) { if (vec) hp_chain_group<PCM_I16, true>(b, sp, g, tile, lane, Ly2, v0, fill); else hp_chain_group<PCM_I16, false>(b, sp, g, tile, lane, Ly2, v0, fill); }
    else { if (vec) hp_chain_group<PCM_F32_UNIT, true>(b, sp, g, tile, lane, Ly2, v0, fill); else hp_chain_group<PCM_F32_UNIT, false>(b, sp, g, tile, lane, Ly2, v0, fill); }
}

__global__ void __launch_bounds__(64, HP_CH <= 16 ? 3 : 1) k_hp(Buffers b, const StepParams *sp, int g, StepParams v0, int fill)
{
    const int lane = threadIdx.x, tile = blockIdx.x;
    if (fill > 0 && tile == 0)
        for (int t = lane; t < fill; t += 64) ((StepParams *)sp)[t] = step_params_at(v0, t, b.nslot);
    const int fmt = fill > 0 ? v0.fmt : sp->fmt;
    wf_setprio_high();   // a lone wave on a serial chain that shares its SIMD with another kernel's wave (+1 % at 4096 streams)
    const StepParams &lay = fill > 0 ? v0 : *sp;
    const bool vec = lay.channels == 1 && ((((size_t)lay.in) | (size_t)lay.group_stride | (size_t)lay.frame_stride) & 15) == 0;
    __shared__ float Ly[TILE * HP_LD];
    if (fmt == PCM_F32) { if (vec) hp_group<PCM_F32, true>(b, sp, g, tile, lane, Ly, v0, fill); else hp_group<PCM_F32, false>(b, sp, g, tile, lane, Ly, v0, fill); }
    else if (fmt == PCM_I16) { if (vec) hp_group<PCM_I16, true>(b, sp, g, tile, lane, Ly, v0, fill); else hp_group<PCM_I16, false>(b, sp, g, tile, lane, Ly, v0, fill); }
    else { if (vec) hp_group<PCM_F32_UNIT, true>(b, sp, g, tile, lane, Ly, v0, fill); else hp_group<PCM_F32_UNIT, false>(b, sp, g, tile, lane, Ly, v0, fill); }
}

#ifdef NNN_ONLY_HP
template __global__ void k_hp2<1>(Buffers b, const StepParams *sp, int g, StepParams v0, int fill, int head);
template __global__ void k_hp2<2>(Buffers b, const StepParams *sp, int g, StepParams v0, int fill, int head);
#endif
#endif   // NNN_HP_EXTERN

#ifndef NNN_ONLY_HP   // (everything from here to the end of the file)
// ---------------------------------------------------------------------------------------------
// K2  lpc: the part of pitch_downsample between the decimation and the FIR -- 5-lag autocorrelation of the 864-value window,
//     lag window, order-4 Levinson recursion, bandwidth expansion and the extra zero (ref: src/pitch.rs:433-446, 460-480,
//     257-292) -- lane = stream on the tile-interleaved decimated ring, one wave per (tile, up to eight consecutive frames).
//     Each of the five sums is a serial chain of 860 steps in the reference's order; inside k_pitch (one block per 16
//     streams) they occupied two waves for 8.7 of the block's 44 us with the other six waiting behind a barrier.  Here every lane
//     carries its own stream's five chains (independent of each other: five-way instruction-level parallelism on full waves), the
//     frames of a group run side by side (nothing here carries over from frame to frame; a wave takes up to four consecutive
//     frames and reads the rows their windows share once), every load is a whole 256-byte row, and the launch rides on the high-pass stream, ahead of the pitch stage.  Output: ac[5], FIR taps[5] per stream-frame.
// ---------------------------------------------------------------------------------------------
constexpr int LPC_CH = 20;    // rows per unrolled chunk: 860 = 43 x 20
constexpr int LPC_CHW = 43;   // k_lpc_wide: 20 chunks.  A lone one-frame launch has the GPU to itself and is paced by the trips to memory it
                              // makes one after the other, not by the rows in flight: 23.5 -> 17 us for a frame of 4096 streams (of which some
                              // 8 us are the launch; 86 rows on four waves per block, with the overflow in AGPRs, measured no better)
static_assert((XLP - 4) % LPC_CH == 0 && (XLP - 4) % LPC_CHW == 0, "");
// lags K0 .. K0 + NK - 1 of the autocorrelation of the 864 rows at base[i * TILE] (row 0 replaced by x0): the reference's
// sequential sum per lag, then its tail (ref: src/pitch.rs:433-446)
template <int K0, int NK, int CH>
__device__ __forceinline__ void lpc_chains(const float *base, float x0, float (&ac)[NK])
{
    float cur[CH + 4], nxt[CH];
#pragma unroll
    for (int i = 0; i < CH + 4; i++) cur[i] = base[(size_t)i * TILE];
    cur[0] = x0;
    float c[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) c[k] = 0.0f;
    constexpr int NCH = (XLP - 4) / CH;
#pragma nounroll
    for (int ch = 0; ch < NCH; ch++) {
        // rows CH (ch + 1) + 4 .. + CH + 3 travel while this chunk is summed (the last chunk re-reads its own rows: in range, unused)
        const float *nb = base + (size_t)((ch + 1 < NCH ? ch + 1 : ch) * CH + 4) * TILE;
#pragma unroll
        for (int i = 0; i < CH; i++) nxt[i] = nb[(size_t)i * TILE];
        // ac[k] += x[i] * x[i + k], i ascending: the reference's sequential sum per lag (pitch_xcorr's unrolling keeps that order)
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int k = 0; k < NK; k++) c[k] += cur[j] * cur[j + K0 + k];
        if (ch + 1 < NCH) {
#pragma unroll
            for (int i = 0; i < 4; i++) cur[i] = cur[CH + i];
#pragma unroll
            for (int i = 0; i < CH; i++) cur[4 + i] = nxt[i];
        }
    }
    // tail d_k = sum_{i = k + 860}^{863} x[i] x[i - k], added after the main sum; cur[] holds the last CH + 4 rows
#pragma unroll
    for (int kk = 0; kk < NK; kk++) {
        constexpr int O = XLP - CH - 4;
        const int k = K0 + kk;
        float d = 0.0f;
#pragma unroll
        for (int i = k + XLP - 4; i < XLP; i++) d += cur[i - O] * cur[i - k - O];
        ac[kk] = c[kk] + d;
    }
}

// lag window, Levinson recursion, bandwidth expansion, extra zero (ref: src/pitch.rs:460-480, 257-292) on a frame's five sums; the
// windowed autocorrelation and the FIR taps go to the frame's ring slot
__device__ __forceinline__ void lpc_finish(const Buffers &b, int tile, int lane, int slot, float (&ac)[5], float *taps = nullptr)
{
    ac[0] *= 1.0001f;
#pragma unroll
    for (int i = 1; i < 5; i++) ac[i] -= ac[i] * (0.008f * (float)i) * (0.008f * (float)i);
    float lpc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ac[0] != 0.0f) {
        float error = ac[0];
        bool done = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (!done) {
                float rr = 0.0f;
#pragma unroll
                for (int j = 0; j < i; j++) rr += lpc[j] * ac[i - j];
                rr += ac[i + 1];
                float r = -rr / error;
                lpc[i] = r;
#pragma unroll
                for (int j = 0; j < (i + 1) / 2; j++) {
                    float t1 = lpc[j], t2 = lpc[i - 1 - j];
                    lpc[j] = t1 + r * t2;
                    lpc[i - 1 - j] = t2 + r * t1;
                }
                error = error - r * r * error;
                if (error < 0.001f * ac[0]) done = true;
            }
        }
    }
    float tmp = 1.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) { tmp *= 0.9f; lpc[i] *= tmp; }
    float l2[5];
    l2[0] = lpc[0] + 0.8f;
    l2[1] = lpc[1] + 0.8f * lpc[0];
    l2[2] = lpc[2] + 0.8f * lpc[1];
    l2[3] = lpc[3] + 0.8f * lpc[2];
    l2[4] = 0.8f * lpc[3];
    float *o = NNN_TI(b.lpc, b.nslot * 10, tile, lane) + (size_t)(slot * 10) * TILE;
#pragma unroll
    for (int i = 0; i < 5; i++) { o[(size_t)i * TILE] = ac[i]; o[(size_t)(5 + i) * TILE] = l2[i]; }
    if (taps) {
#pragma unroll
        for (int i = 0; i < 5; i++) taps[i] = l2[i];
    }
}

// k_lpc's packed form.  One lag sum of a frame pair takes a product: MODE 1 the pair's first frame only, 2 the second only, 3 both;
// H = the half of `p` that holds the product.
template <int MODE, int H>
__device__ __forceinline__ void lpc_add(v2f &a, v2f p)
{
    if (MODE == 3) a = H ? pk_add_by(a, p) : pk_add_bx(a, p);
    else if (MODE == 1) a.x = sadd(a.x, H ? p.y : p.x);
    else a.y = sadd(a.y, H ? p.y : p.x);
}
// the chunk's LPC_CH rows (x[i] in aligned pairs, `first` in row 0's place) against rows i .. i + 4, into the five lag sums: ac[k] +=
// x[i] * x[i + k], i ascending (ref: src/pitch.rs:433-446)
template <int MODE, int NP>
__device__ __forceinline__ void lpc_rows(v2f (&acc)[5], const v2f (&rows)[NP], float first)
{
    v2f cur[LPC_CH / 2 + 2];
#pragma unroll
    for (int i = 0; i < LPC_CH / 2 + 2; i++) cur[i] = rows[i];
    cur[0].x = first;
#pragma unroll
    for (int m = 0; m < LPC_CH / 2; m++) {
        {   // row 2m: (p0, p1) (p2, p3) (p4, -)
            const v2f a = pk_mul_bx(cur[m], cur[m]), b = pk_mul_bx(cur[m], cur[m + 1]), c = pk_mul_bx(cur[m], cur[m + 2]);
            lpc_add<MODE, 0>(acc[0], a); lpc_add<MODE, 1>(acc[1], a); lpc_add<MODE, 0>(acc[2], b); lpc_add<MODE, 1>(acc[3], b); lpc_add<MODE, 0>(acc[4], c);
        }
        {   // row 2m + 1: (-, p0) (p1, p2) (p3, p4)
            const v2f a = pk_mul_by(cur[m], cur[m]), b = pk_mul_by(cur[m], cur[m + 1]), c = pk_mul_by(cur[m], cur[m + 2]);
            lpc_add<MODE, 1>(acc[0], a); lpc_add<MODE, 0>(acc[1], b); lpc_add<MODE, 1>(acc[2], b); lpc_add<MODE, 0>(acc[3], c); lpc_add<MODE, 1>(acc[4], c);
        }
    }
}
template <int MODE, int NP>
__device__ __forceinline__ void lpc_rows(v2f (&acc)[5], const v2f (&rows)[NP]) { lpc_rows<MODE>(acc, rows, rows[0].x); }
template <int NP>
__device__ __forceinline__ float lpc_cur(const v2f (&cur)[NP], int n) { return n & 1 ? cur[n >> 1].y : cur[n >> 1].x; }

// k_lpc: one wave per (tile, LPC_FC consecutive frames).  Consecutive frames' windows overlap by 624 of 864 rows, and frames of a
// tile running as separate waves do not find each other's rows in L2 (the few hundred waves an XCD has in flight read 16 MB of
// rows between two uses of a line: 3.4 KB per stream-frame from HBM).  Here a wave walks the union of its frames' windows once
// -- 864 + 240 per further frame rows -- and every row serves each frame whose window holds it: the same sums in the same order
// per frame, 2.7x fewer rows with eight frames per wave.
#ifndef NNN_LPC_FC_MAX
#define NNN_LPC_FC_MAX 8
#endif
constexpr int LPC_FC = NNN_LPC_FC_MAX;
static_assert(LPC_FC % 2 == 0, "k_lpc sums frames in pairs");
static_assert(240 % LPC_CH == 0, "");
// `fc` <= LPC_FC frames per wave: the host gives small launches fewer (more, shorter waves)
__global__ void __launch_bounds__(64) k_lpc(Buffers b, const StepParams *sp0, int g, int fc)
{
    const int lane = threadIdx.x;
    const int nch = (g + fc - 1) / fc;
    // block -> (tile, chunk of frames).  Workgroup i runs on XCD i mod 8 (observed; a speed matter only): tile t's chunks go to XCD
    // t mod 8, where k_hp's block t wrote the ring.
    int tile, chunk, sub_;
    xcd_tile_block_units((int)blockIdx.x, b.NT, 1, nch, chunk, tile, sub_);
    const int f0 = chunk * fc, nf = g - f0 < fc ? g - f0 : fc;
    const int nslot = b.nslot, ring = dec_ring_len(nslot);
    int slot[LPC_FC];
    float x0[LPC_FC];   // x_lp[0] of each frame is special (ref: src/pitch.rs:458)
#pragma unroll
    for (int c = 0; c < LPC_FC; c++) {
        slot[c] = sp0[f0 + (c < nf ? c : 0)].slot;
        x0[c] = NNN_TI(b.xlp0, nslot, tile, lane)[(size_t)slot[c] * TILE];
    }
    // union row r of the chunk sits at ring position (base0 + r) mod ring (consecutive frames' windows start 240 apart)
    const float *rows = b.dec + (size_t)tile * dec_len(nslot) * TILE + lane;
    const int base0 = dec_base(slot[0], nslot);
    auto row = [&](int r) {
        int p = base0 + r;
        p = p >= ring ? p - ring : p;
        return rows[(size_t)p * TILE];
    };
    constexpr int NCH = (XLP - 4) / LPC_CH, STEP = 240 / LPC_CH;   // chunks of a window (43), chunks between two windows (12)
    const int J = NCH + STEP * (nf - 1);
    // Rows in aligned register pairs, frames in pairs (acc[q][k] = lag k of frames 2q and 2q + 1): a row's five products are three packed
    // multiplies, and while both frames of a pair hold the row -- 31 of a window's 43 chunks -- one packed add serves both (the product
    // in both halves by operand selection).  Same products, same sums in the same order as lpc_chains above.  (The arithmetic is not what
    // paces this kernel -- the rows are: packed or not, 22.7 us per frame at 65536 streams with four frames per wave; the packed form's
    // registers let a wave take eight: 19.6, profiles/r5_experiments_ab.txt O.)
    constexpr int NP = (LPC_CH + 4) / 2;
    v2f cur[NP], nxt[LPC_CH / 2];
#pragma unroll
    for (int i = 0; i < NP; i++) cur[i] = v2f{row(2 * i), row(2 * i + 1)};
    v2f acc[LPC_FC / 2][5];
#pragma unroll
    for (int q = 0; q < LPC_FC / 2; q++)
#pragma unroll
        for (int k = 0; k < 5; k++) acc[q][k] = v2f{0.0f, 0.0f};
#pragma nounroll
    for (int j = 0; j < J; j++) {
        const int jn = j + 1 < J ? j + 1 : j;
#pragma unroll
        for (int i = 0; i < LPC_CH / 2; i++) nxt[i] = v2f{row(jn * LPC_CH + 4 + 2 * i), row(jn * LPC_CH + 5 + 2 * i)};
#pragma unroll
        for (int q = 0; q < LPC_FC / 2; q++) {
            const int jj0 = j - STEP * 2 * q, jj1 = jj0 - STEP;   // this chunk's place in the two frames' windows
            const bool a0 = 2 * q < nf && jj0 >= 0 && jj0 < NCH, a1 = 2 * q + 1 < nf && jj1 >= 0 && jj1 < NCH;
            if (a0 && a1 && jj1 != 0) lpc_rows<3>(acc[q], cur);
            else {
                // a window's first row is the frame's own x_lp[0] (ref: src/pitch.rs:458): that chunk on its own
                if (a0) lpc_rows<1>(acc[q], cur, jj0 == 0 ? x0[2 * q] : cur[0].x);
                if (a1) lpc_rows<2>(acc[q], cur, jj1 == 0 ? x0[2 * q + 1] : cur[0].x);
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
                if ((h ? a1 : a0) && (h ? jj1 : jj0) == NCH - 1) {
                    // tail d_k = sum_{i = k + 860}^{863} x[i] x[i - k], added after the main sum; cur[] holds rows 840 .. 863 of the window
                    float ac[5];
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        constexpr int O = XLP - LPC_CH - 4;
                        float d = 0.0f;
#pragma unroll
                        for (int i = k + XLP - 4; i < XLP; i++) d += lpc_cur(cur, i - O) * lpc_cur(cur, i - k - O);
                        ac[k] = (h ? acc[q][k].y : acc[q][k].x) + d;
                    }
                    lpc_finish(b, tile, lane, slot[2 * q + h], ac);
                }
        }
        if (j + 1 < J) {
            cur[0] = cur[LPC_CH / 2];
            cur[1] = cur[LPC_CH / 2 + 1];
#pragma unroll
            for (int i = 0; i < LPC_CH / 2; i++) cur[2 + i] = nxt[i];
        }
    }
}

// k_lpc_wide, for launches too small to fill the GPU (a one-frame call on a few thousand streams is 64 waves walking five 860-step
// chains each): five waves per (tile, frame), one lag each, the five sums meeting in LDS.
__global__ void __launch_bounds__(320) k_lpc_wide(Buffers b, const StepParams *sp0, int g)
{
    __shared__ float acs[5][TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tile, f, sub_;
    xcd_tile_block_units((int)blockIdx.x, b.NT, 1, g, f, tile, sub_);
    const int slot = sp0[f].slot;
    const float *base = b.dec + ((size_t)tile * dec_len(b.nslot) + (size_t)dec_base(slot, b.nslot)) * TILE + lane;
    const float x0 = NNN_TI(b.xlp0, b.nslot, tile, lane)[(size_t)slot * TILE];   // x_lp[0] is special (ref: src/pitch.rs:458)
    float a1[1];
    if (wave == 0) lpc_chains<0, 1, LPC_CHW>(base, x0, a1);
    else if (wave == 1) lpc_chains<1, 1, LPC_CHW>(base, x0, a1);
    else if (wave == 2) lpc_chains<2, 1, LPC_CHW>(base, x0, a1);
    else if (wave == 3) lpc_chains<3, 1, LPC_CHW>(base, x0, a1);
    else lpc_chains<4, 1, LPC_CHW>(base, x0, a1);
    acs[wave][lane] = a1[0];
    __syncthreads();
    if (wave != 0) return;
    float ac[5];
#pragma unroll
    for (int k = 0; k < 5; k++) ac[k] = acs[k][lane];
    lpc_finish(b, tile, lane, slot, ac);
}

// ---------------------------------------------------------------------------------------------
// K3  pitch: the pitch analysis of a frame from the FIR on, one block per 16 consecutive streams (a quarter tile):
//       pitch_downsample's last step FIR5 with k_lpc's taps -> pitch_buf (ref: src/pitch.rs:407-429)
//       pitch_search                 coarse cross-correlation 147 lags x 240 taps on the 4x-decimated signal, find_best_pitch,
//                                    fine cross-correlation within +-2 of 2*best / 2*second, find_best_pitch, pseudo-
//                                    interpolation (ref: src/pitch.rs:63-115, 296-405)
//       remove_doubling              (ref: src/pitch.rs:118-221)
//     pitch_buf (864 values per stream) lives in LDS from the FIR that makes it to the last inner product that reads it and
//     never travels to HBM (round 2a: written once and read by two more launches, 17 KB per stream-frame); so do the coarse
//     cross-correlation, the coarse-lag energies and the check points of the two long energy scans (fine-lag energies,
//     yy_lookup) that are looked up at data-dependent lags later in the frame.
//
//     Every sum that feeds the integer pitch index keeps the reference's order (sequential per lag / per partial), so the
//     parallelism is over streams x independent chains, lane = (stream, chain):
//       FIR                  (stream, 32-row chunk)         elementwise
//       coarse xcorr         (stream, group of 13 lags)     12 groups on three waves, packed accumulators (see the phase)
//       energy scans         (stream) on three waves        serial running sums with their clamps
//       inner products       (stream, partial q of 4), the wave picks the candidate lags: 2 (fine) or 4 (remove_doubling)
//                            candidates share every read of the fixed operand
//       decisions            (stream) on wave 0             find_best_pitch x 2, the k = 2..15 loop of remove_doubling
//     LDS layout of pitch_buf: even rows and odd rows apart, element (r, s) at (r & 1 ? ODD : 0) + (r >> 1) * 16 + s with ODD =
//     432 * 16 + 16.  ds_read_b32 / ds_read2_b32 bank modulo 32, a 32-lane group is 16 streams x 2 chains: chains reading rows r
//     and r + 2 (the inner-product partials, dealt to the lanes as q = 0, 2 | 1, 3) sit on neighbouring rows of one half,
//     chains reading rows r and r + 1 on the same row index of the two halves, which the pad
//     of 16 puts on different banks; the coarse cross-correlation reads the even half only -- the 4x-decimated signal,
//     compact -- and its two lag groups per 32 lanes start 13 rows apart: all conflict-free (round 2a: 31 % of k_pitch2's LDS
//     cycles were conflicts).  Rows 2 apart (consecutive taps of one partial, of the decimated signal, of the window pairs)
//     are 16 or 32 floats apart at any parity: one ds_read2_b32 fetches two of them into a register pair, which is what
//     v_pk_mul_f32 wants.
//     remove_doubling carries last_period / last_gain from frame to frame: the launch loops over the `g` frames of its
//     group; the next frame's decimated window is requested a frame ahead and waits in registers.
// ---------------------------------------------------------------------------------------------
constexpr int PK_SPB = 16;                       // streams per block
constexpr int PK_WAVES = 8;
constexpr int PK_T = 64 * PK_WAVES;
#ifndef NNN_PK_LC
#define NNN_PK_LC 13
#endif
constexpr int PK_LC = NNN_PK_LC;                 // coarse lags per lane: 12 groups of 13 (the last one 4 lags wide) on three waves
constexpr int PK_NG = (NLAG1 + PK_LC - 1) / PK_LC;
static_assert((PK_LC & 1) && PK_NG <= 4 * 8, "odd group size: neighbouring groups start on different row parities");
constexpr int PK_XW = (PK_NG + 3) / 4;           // waves of the coarse cross-correlation
constexpr int PK_JB = 8;                         // taps per unrolled step of the coarse cross-correlation (240 = 30 x 8)
constexpr int PK_NP = PK_LC / 2;                 // packed accumulators per lane (+ one single: PK_LC is odd)
constexpr int PK_NT = PK_JB / 2 + PK_NP;         // window pairs per alignment
static_assert((PK_NG - 1) * PK_LC + 239 + PK_JB + PK_LC < XLP / 2, "the window stays inside the even rows");
constexpr int PK_KMAX = 12;                      // largest divisor of remove_doubling that can pass its `t1 >= min_period` test
constexpr int PK_NE = 1 + 2 * (PK_KMAX - 1);     // candidate periods of the decision loop: t0, then two per divisor k = 2 .. 12
constexpr int PK_NSLOT = PK_NE + 2;              // + the two neighbours of t0
constexpr int PK_NC = 36;                        // inner-product slots: 10 fine lags | 25 candidates of remove_doubling, 32 .. 34 the refinement

constexpr int PK_HALF = (XLP / 2) * PK_SPB;      // floats of the even rows
constexpr int PK_ODD = PK_HALF + 16;             // first odd row
__device__ __forceinline__ int pk_at(int r, int s) { return ((r & 1) ? PK_ODD : 0) + (r >> 1) * PK_SPB + s; }


// running best / second-best update of find_best_pitch, ref: src/pitch.rs:383-400
struct BestPitch {
    float best_num, second_num, best_den, second_den;
    int best, second;
    __device__ void init() { best_num = -1.0f; second_num = -1.0f; best_den = 0.0f; second_den = 0.0f; best = 0; second = 1; }
    // the reference's nested ifs as selects (the same comparisons on the same values: a NaN fails them either way); this runs
    // on one wave with the rest of the block waiting, where a taken branch costs more than the selects
    __device__ __forceinline__ void update(int i, float corr, float y_sq_norm) {
        const float num = corr * corr;
        const bool in = corr > 0.0f && num * second_den > second_num * y_sq_norm;
        const bool top = in && num * best_den > best_num * y_sq_norm;
        const bool mid = in && !top;
        second_num = top ? best_num : (mid ? num : second_num);
        second_den = top ? best_den : (mid ? y_sq_norm : second_den);
        second = top ? best : (mid ? i : second);
        best_num = top ? num : best_num;
        best_den = top ? y_sq_norm : best_den;
        best = top ? i : best;
    }
    // the same with corr * corr handed in, NaN standing for "corr > 0 failed" (every comparison with it fails): the coarse search
    // squares its 147 correlations on the lanes that made them
    __device__ __forceinline__ void update_sq(int i, float num, float y_sq_norm) {
        const bool in = num * second_den > second_num * y_sq_norm;
        if (!wave_any(in)) return;   // none of the wave's streams takes this lag: nothing changes
        const bool top = in && num * best_den > best_num * y_sq_norm;
        const bool mid = in && !top;
        second_num = top ? best_num : (mid ? num : second_num);
        second_den = top ? best_den : (mid ? y_sq_norm : second_den);
        second = top ? best : (mid ? i : second);
        best_num = top ? num : best_num;
        best_den = top ? y_sq_norm : best_den;
        best = top ? i : best;
    }
};

struct Xc2 {   // xcorr[] of the fine search: zero except within 2 of 2*best / 2*second (ref: src/pitch.rs:88-96)
    float v[10];
    int lo1, lo2;
    __device__ __forceinline__ float at(int i) const
    {
        float r = 0.0f;
        if (i >= 0 && i < NLAG2) {
            const int d1 = i - lo1, d2 = i - lo2;
#pragma unroll
            for (int u = 4; u >= 0; u--) if (d2 == u) r = v[5 + u];
#pragma unroll
            for (int u = 4; u >= 0; u--) if (d1 == u) r = v[u];   // first window wins where they overlap (same value)
        }
        return r;
    }
};

__device__ __forceinline__ float pitch_gain(float xy, float xx, float yy) { return xy / sqrtf(1.0f + xx * yy); }

// The two long energy scans run once per frame, serially, and are looked up at a few data-dependent lags later in the frame.
// They leave check points in LDS (every 8th fine lag, every 5th step of yy); a lookup replays the few steps from the check
// point below it -- the same additions in the same order.  (Whole tables would be 43 KB per block; through global scratch the
// scans were bound by the depth of a wave's store queue.)
constexpr int PK_CKF = 8, PK_NCKF = (NLAG2 + PK_CKF - 1) / PK_CKF;    // 37
constexpr int PK_CKY = 5, PK_NCKY = 384 / PK_CKY + 1;                  // 77
// The certified coarse search (round 6; see the phase in k_pitch): what it keeps in LDS.
constexpr int PK_DCK = 41;                       // a stream's row of check points of the coarse lags' energy scan (one per group of four lags; an odd pitch)
constexpr int PK_RMAX = 24;                      // lags of one stream that may survive the approximate search (more: the block takes the full search)
constexpr int PK_CAP = 384;                      // ... and of the block's 16 streams together
constexpr int PK_PLW = XLP / 2 + 8;              // halfwords between two streams' bf16 planes (the 432 even rows of pitch_buf): 220 words, so that the
                                                 // eight streams of a region sit on eight different banks for the FIR's word stores
constexpr float PK_EPS = 0.0083f;                // |approximate - reference| <= PK_EPS sqrt(|x|^2 |y window|^2): two bf16 roundings 2^-7, the matrix
                                                 // cores' f32 accumulation and the reference's own sequential f32 sum well inside the rest
struct PkCert {
    float denck[PK_SPB][PK_DCK];                 // the running energy before coarse lag 4 k (find_best_pitch, ref: src/pitch.rs:380-402): check points, as for the fine lags
    float bsum[PK_SPB][27];                      // |.|^2 of the 27 blocks of 16 even rows (any nonzero value: at least 2^-120)
    unsigned mask[PK_SPB][5];                    // bit L: coarse lag L of the stream survived
    unsigned count, full, pad_[2];               // survivors of the block; != 0: the block takes the full search
    unsigned short list[PK_CAP];                 // survivors (stream << 8 | lag) ...
    float slotv[PK_SPB][PK_RMAX], slotd[PK_SPB][PK_RMAX];   // ... their exact sums and the energy each of them saw, by rank within the stream
    int slotl[PK_SPB][PK_RMAX];                  // ... and their lags
    alignas(16) unsigned short plane[(PK_SPB / 2) * PK_PLW + 16];   // bf16 even rows of streams 8 .. 15 (streams 0 .. 7: over ckf / cky, idle until the scans);
                                                                    // + the 32 bytes the last fragment read of the last stream runs over (masked).  Nothing
                                                                    // shares the planes' bytes: the search reads them while its first waves list survivors
};
struct alignas(16) PkLds {
    float pb[PK_ODD + PK_HALF];                  // the decimated window, then (in place) pitch_buf
    float ckf[PK_NCKF][PK_SPB];                  // running energy of the fine lags before lag 8 m (find_best_pitch, ref: src/pitch.rs:380-402)
    float cky[PK_NCKY][PK_SPB];                  // running energy yy of remove_doubling after step 5 m (ref: src/pitch.rs:133-142); [0] = xx
    union {
        struct { float xc[NLAG1][PK_SPB], ysq[NLAG1][PK_SPB]; } c;   // full coarse search: squared positive cross-correlation (else NaN), running energy per lag
        PkCert a;                                                      // certified coarse search
        struct {                                                       // from the fine search on
            float part[PK_NC][4][PK_SPB];        // inner-product partials [slot][q][stream] (ref: src/pitch.rs:225-244)
            float yy[32][PK_SPB];                // yy_lookup at the candidate periods
            float ye[10][PK_SPB];                // the running energy the fine lags of the two windows saw
            int cand[32][PK_SPB];                // candidate periods of remove_doubling
            int lo[2][PK_SPB];                   // first fine lag of the two windows
            int tsel[PK_SPB];                    // the period the decision loop chose
            float xx[PK_SPB], lgain[PK_SPB];     // per stream: |x|^2, the previous frame's gain ...
            int t0[PK_SPB], pprev[PK_SPB];       // ... the period before remove_doubling, the previous frame's period / 2
            float kxy[16][PK_SPB], kyy[16][PK_SPB], kg[16][PK_SPB];   // per candidate divisor k: its xy, yy, gain ...
            int kpass[16][PK_SPB];               // ... and whether it replaces the best so far
            int any_refine;                      // some stream of the block left t0: the +-1 refinement needs inner products
        } f;
    } u;
};
static_assert(sizeof(PkLds) <= 80 * 1024, "two blocks per CU");
static_assert(offsetof(PkLds, ckf) % 16 == 0 && offsetof(PkLds, u) % 16 == 0 && offsetof(PkCert, plane) % 16 == 0 && (PK_PLW * 2) % 16 == 0, "16-byte fragment reads");
static_assert(sizeof(float) * (PK_NCKF + PK_NCKY) * PK_SPB >= sizeof(unsigned short) * (PK_SPB / 2) * PK_PLW + 32, "the first eight planes fit over the check points");

// Window and FIR mapping: thread = (stream col, chunk ch of 32 rows), 27 chunks (the block's last 80 threads idle here); a
// chunk starts on an even row, so every LDS address of its 16 row pairs is the thread's base plus a constant.
constexpr int PK_CH = 32, PK_NCH = XLP / PK_CH;
static_assert(PK_NCH * PK_CH == XLP && PK_NCH * PK_SPB <= PK_T, "");

// a frame's window from the decimated-history ring: 16 lanes share a 64-byte segment of a tile row; with it the frame's five FIR
// taps (k_lpc's output, kept by ring slot)
__device__ __forceinline__ void pk_window_load(const Buffers &b, const StepParams *sp, int tile, int q0, int tid, float (&v)[PK_CH],
                                               float (&fir)[5])
{
    const int col = tid & 15, ch = tid >> 4;
    if (ch >= PK_NCH) {   // (defined on every path: otherwise the previous window stays live through the whole frame for the register allocator)
#pragma unroll
        for (int i = 0; i < PK_CH; i++) v[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; i++) fir[i] = 0.0f;
        return;
    }
    const int slot = sp->slot;
    {
        const float *lp = NNN_TI(b.lpc, b.nslot * 10, tile, q0 + col) + (size_t)(slot * 10) * TILE;
#pragma unroll
        for (int i = 0; i < 5; i++) fir[i] = lp[(size_t)(5 + i) * TILE];
    }
    const float *base = b.dec + ((size_t)tile * dec_len(b.nslot) + (size_t)dec_base(slot, b.nslot)) * TILE + q0;   // uniform
    const unsigned off = (unsigned)(ch * PK_CH) * TILE + (unsigned)col;
#pragma unroll
    for (int i = 0; i < PK_CH; i++) v[i] = base[off + (unsigned)i * TILE];
    if (ch == 0) v[0] = NNN_TI(b.xlp0, b.nslot, tile, q0 + col)[(size_t)slot * TILE];   // x_lp[0] is special (ref: src/pitch.rs:458)
}

// 4-way interleaved inner-product partials (ref: src/pitch.rs:225-244) of NCAND candidates against the fixed operand
// p[384 ..]: lane (s, q) accumulates x[4m+q] * y_c[4m+q] over m in order, one read of x serving every candidate;
// the caller combines ((s0+s1)+s2)+s3.  y_c starts at row yr[c].  Rows 4m + const of one stream are 32 floats apart: one
// ds_read2_b32 brings taps m, m + 1 as a register pair, one v_pk_mul_f32 forms both products, two adds in order.
template <int NCAND>
__device__ __forceinline__ void pk_inner(const float *pb, int s, int q, const int (&yr)[NCAND], float (&acc)[NCAND])
{
    constexpr int U = NCAND >= 4 ? 2 : 3;   // tap pairs per register set
    static_assert(120 % (4 * U) == 0, "");
    const float *xp = pb + pk_at(PITCH_MAX / 2 + q, s);
    const float *yp[NCAND];
#pragma unroll
    for (int c = 0; c < NCAND; c++) { acc[c] = 0.0f; yp[c] = pb + pk_at(yr[c] + q, s); }
    // two register sets in turn: the rows of the next taps travel while these are summed (round 6: with one set the compiler's loop was
    // "load, wait, use" -- every trip paid the LDS latency in full)
    v2f xa[U], ya[NCAND][U], xb[U], yb[NCAND][U];
#define NNN_LD(X, Y, M) do { _Pragma("unroll") for (int u_ = 0; u_ < U; u_++) { const int o_ = 32 * ((M) + 2 * u_); X[u_] = mk2(xp[o_], xp[o_ + 32]); \
        _Pragma("unroll") for (int c_ = 0; c_ < NCAND; c_++) Y[c_][u_] = mk2(yp[c_][o_], yp[c_][o_ + 32]); } } while (0)
#define NNN_ACC(X, Y) do { _Pragma("unroll") for (int u_ = 0; u_ < U; u_++) _Pragma("unroll") for (int c_ = 0; c_ < NCAND; c_++) { \
        const v2f pr_ = pk_mul(X[u_], Y[c_][u_]); acc[c_] = sadd(acc[c_], pr_.x); acc[c_] = sadd(acc[c_], pr_.y); } } while (0)
    NNN_LD(xa, ya, 0);
#pragma nounroll
    for (int m0 = 0; m0 < 120 - 4 * U; m0 += 4 * U) {
        NNN_LD(xb, yb, m0 + 2 * U);
        NNN_ACC(xa, ya);
        NNN_LD(xa, ya, m0 + 4 * U);
        NNN_ACC(xb, yb);
    }
    NNN_LD(xb, yb, 120 - 2 * U);
    NNN_ACC(xa, ya);
    NNN_ACC(xb, yb);
#undef NNN_LD
#undef NNN_ACC
}

// Lag K of the autocorrelation of a stream's 864-value window in LDS (`pbs` = L.pb + stream: row r at pk_at(r, 0)): the reference's
// sequential sum over i = 0 .. 859 and its tail (ref: src/pitch.rs:433-446), one lag per wave so that the lag is a compile-time
// offset into a sliding run of rows held in registers: a row is read once per lag, a step is one multiply and one dependent add.
// `blk0`, `c0`: the sum's first 16 blk0 steps were taken elsewhere (k_hp2's head waves: they need none of the new frame) and gave c0.
template <int K>
__device__ __forceinline__ float pk_autocorr(const float *pbs, int blk0 = 0, float c0 = 0.0f)
{
    const float *E = pbs + 8 * blk0 * PK_SPB, *O = E + PK_ODD;
    auto row = [&](const float *e, const float *o, int j) { return (j & 1) ? o[(j >> 1) * PK_SPB] : e[(j >> 1) * PK_SPB]; };
    float run[20];   // run[j] = x[16 blk + j]
#pragma unroll
    for (int j = 0; j < 20; j++) run[j] = row(E, O, j);
    E = pbs; O = pbs + PK_ODD;
    float c = c0;
    constexpr int NBLK = 52;   // 52 blocks of 16 steps, then 28 steps on rows 832 .. 863
#pragma nounroll   // (two blocks per trip -- the run's hand-over a renaming instead of twenty moves -- measured: no change)
    for (int blk = blk0; blk < NBLK; blk++) {
        float nxt[16];   // rows 16 (blk + 1) + 4 .. + 19 travel while this block's steps are summed
        const float *En = E + (8 * (blk + 1) + 2) * PK_SPB, *On = O + (8 * (blk + 1) + 2) * PK_SPB;
#pragma unroll
        for (int j = 0; j < 16; j++) nxt[j] = row(En, On, j);
#pragma unroll
        for (int j = 0; j < 16; j++) c += run[j] * run[j + K];
#pragma unroll
        for (int j = 0; j < 4; j++) run[j] = run[16 + j];
#pragma unroll
        for (int j = 0; j < 16; j++) run[4 + j] = nxt[j];
    }
    float last[12];   // rows 852 .. 863
#pragma unroll
    for (int j = 0; j < 12; j++) last[j] = row(E + (8 * NBLK + 10) * PK_SPB, O + (8 * NBLK + 10) * PK_SPB, j);
    auto x = [&](int i) { return i < 16 * NBLK + 20 ? run[i - 16 * NBLK] : last[i - 16 * NBLK - 20]; };   // rows 832 .. 863 (static index)
#pragma unroll
    for (int i = 16 * NBLK; i < XLP - 4; i++) c += x(i) * x(i + K);
    float d = 0.0f;   // tail d_K = sum_{i = K + 860}^{863} x[i] x[i - K], added after the main sum
#pragma unroll
    for (int i = K + XLP - 4; i < XLP; i++) d += x(i) * x(i - K);
    return c + d;
}

// ---- the serial energy scans of k_pitch on quad lanes (round 6) -------------------------------------------------------------------
// find_best_pitch and remove_doubling carry three running energies through the frame -- the energy every coarse lag sees, the one every
// fine lag sees, yy_lookup (ref: src/pitch.rs:380-402, :133-142) -- each a chain of several hundred f32 additions whose order is the
// reference's.  One wave issues one instruction every ~4.5 cycles whatever its lane count, and with lane = stream a step was six or seven
// instructions (two loads, two squares, a difference, the add, the clamp) on 16 of 64 lanes: the chains, not the arithmetic of the search,
// were the frame's critical path (round 6 stamps: 10 + 8 us of a block's 42).  Here a wave takes ONE chain for the block's 16 streams,
// lane = (stream, q): the four lanes of a quad fetch and square the rows of four consecutive steps at once, and every lane then adds the
// four terms in order, each add taking its operand from a quad lane through DPP -- the same additions in the same order, 2.25 to 4.25
// instructions per step.  Rows are requested a group ahead of the adds that use them.
constexpr int PK_FINE_K = (NLAG2 + 3) / 4;       // 74 groups of four fine lags
constexpr int PK_YY_B = 19;                      // 19 runs of twenty steps of yy_lookup (380 steps: the last four are only ever replayed)
// (the four terms are fetched by four independent DPP moves, then added by plain instructions: an add that takes its operand through DPP
// waits two more states on the sum it has just written and runs at a third of the rate -- measured, 21 against 9 cycles a step)
__device__ __forceinline__ Quad4 pk_quad4(float t)
{
    Quad4 r = quad_all(t);   // (nnn_mfma.h: four DPP moves; the tests' interpreter: one rendezvous of the wave's lanes)
    keep_rw(r.t0); keep_rw(r.t1); keep_rw(r.t2); keep_rw(r.t3);
    return r;
}
__device__ __forceinline__ float pk_add4(float y, float t)
{
    const Quad4 r = pk_quad4(t);
    y = y + r.t0;
    y = y + r.t1;
    y = y + r.t2;
    return y + r.t3;
}
// the energy every coarse lag sees (even rows only), ref: src/pitch.rs:83 -> :380-402: its start, 1 + |y4[0 .. 239]|^2 ...
__device__ __forceinline__ float pk_chain_coarse_start(const float *pbs, int cq)
{
    const float *pq = pbs + cq * PK_SPB;   // y4[4 k + cq] at pq[64 k]
    float ysq = 1.0f;
    float nx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) nx[i] = pq[(4 * i) * PK_SPB];
#pragma nounroll
    for (int k0 = 0; k0 < 60; k0 += 4) {
        float cur[4];
#pragma unroll
        for (int i = 0; i < 4; i++) cur[i] = nx[i];
        const int kn = k0 + 4 < 60 ? k0 + 4 : k0;
#pragma unroll
        for (int i = 0; i < 4; i++) nx[i] = pq[(4 * (kn + i)) * PK_SPB];
#pragma unroll
        for (int i = 0; i < 4; i++) ysq = pk_add4(ysq, cur[i] * cur[i]);
    }
    return ysq;
}
// ... and groups k0 .. k1 - 1 of four lags (lag L drops y4[L] and takes y4[L + 240]); check point dn[k] = the energy before lag 4 k
constexpr int PK_COARSE_K = 38;                  // (147 lags: the 148th .. 152nd are computed and dropped)
__device__ __forceinline__ float pk_chain_coarse(const float *pbs, int cq, float ysq, int k0, int k1, float *dn)
{
    float na[2], nd[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int L = 4 * (k0 + i) + cq < NLAG1 ? 4 * (k0 + i) + cq : NLAG1 - 1;
        na[i] = pbs[(L + 240) * PK_SPB];
        nd[i] = pbs[L * PK_SPB];
    }
#pragma nounroll
    for (int k = k0; k < k1; k += 2) {
        float ca[2], cd[2];
#pragma unroll
        for (int i = 0; i < 2; i++) { ca[i] = na[i]; cd[i] = nd[i]; }
        const int kn = k + 2 < k1 ? k + 2 : k;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int L = 4 * (kn + i) + cq < NLAG1 ? 4 * (kn + i) + cq : NLAG1 - 1;
            na[i] = pbs[(L + 240) * PK_SPB];
            nd[i] = pbs[L * PK_SPB];
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float t = ca[i] * ca[i] - cd[i] * cd[i];
            const Quad4 r = pk_quad4(t);
            if (cq == 0) dn[k + i] = ysq;
            ysq = fmaxf(ysq + r.t0, 1.0f);
            ysq = fmaxf(ysq + r.t1, 1.0f);
            ysq = fmaxf(ysq + r.t2, 1.0f);
            ysq = fmaxf(ysq + r.t3, 1.0f);
        }
    }
    return ysq;
}
// the energy every fine lag sees (ref: src/pitch.rs:97 -> :380-402): its start, 1 + |rows 0 .. 479|^2 in row order ...
__device__ __forceinline__ float pk_chain_fine_start(const float *pbs, int cq, float ysq, int k0, int k1)   // rows 4 k0 .. 4 k1 - 1 (k0, k1 multiples of 4)
{
    const float *pq = pbs + ((cq & 1) ? PK_ODD : 0) + (cq >> 1) * PK_SPB;   // row 4 k + cq at pq[32 k]
    float nx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) nx[i] = pq[(2 * (k0 + i)) * PK_SPB];
#pragma nounroll
    for (int k = k0; k < k1; k += 4) {
        float cur[4];
#pragma unroll
        for (int i = 0; i < 4; i++) cur[i] = nx[i];
        const int kn = k + 4 < k1 ? k + 4 : k;
#pragma unroll
        for (int i = 0; i < 4; i++) nx[i] = pq[(2 * (kn + i)) * PK_SPB];
#pragma unroll
        for (int i = 0; i < 4; i++) ysq = pk_add4(ysq, cur[i] * cur[i]);
    }
    return ysq;
}
// ... and groups k0 .. k1 - 1 of four lags: lag 4 k + cq drops row 4 k + cq and takes row 4 k + cq + 480; check point ckf[m] = the energy
// before lag 8 m (k even)
__device__ __forceinline__ float pk_chain_fine(const float *pbs, int cq, int cs, float ysq, int k0, int k1, float (*ckf)[PK_SPB])
{
    const float *pq = pbs + ((cq & 1) ? PK_ODD : 0) + (cq >> 1) * PK_SPB;
    float na[2], nd[2];
#pragma unroll
    for (int i = 0; i < 2; i++) { na[i] = pq[(2 * (k0 + i) + 240) * PK_SPB]; nd[i] = pq[(2 * (k0 + i)) * PK_SPB]; }
#pragma nounroll
    for (int k = k0; k < k1; k += 2) {
        float ca[2], cd[2];
#pragma unroll
        for (int i = 0; i < 2; i++) { ca[i] = na[i]; cd[i] = nd[i]; }
        const int kn = k + 2 < k1 ? k + 2 : k;
#pragma unroll
        for (int i = 0; i < 2; i++) { na[i] = pq[(2 * (kn + i) + 240) * PK_SPB]; nd[i] = pq[(2 * (kn + i)) * PK_SPB]; }
        if (cq == 0) ckf[k >> 1][cs] = ysq;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float t = ca[i] * ca[i] - cd[i] * cd[i];
            const Quad4 r = pk_quad4(t);
            ysq = fmaxf(ysq + r.t0, 1.0f);
            ysq = fmaxf(ysq + r.t1, 1.0f);
            ysq = fmaxf(ysq + r.t2, 1.0f);
            ysq = fmaxf(ysq + r.t3, 1.0f);
        }
    }
    return ysq;
}
// xx = yy_lookup[0] = |rows 384 .. 863|^2 as inner_prod sums it: four interleaved partial sums combined ((s0 + s1) + s2) + s3
// (ref: src/pitch.rs:133-136, :225-244) -- one partial per quad lane
__device__ __forceinline__ float pk_chain_yy_start(const float *pbs, int cq)
{
    const float *pq = pbs + ((cq & 1) ? PK_ODD : 0) + (192 + (cq >> 1)) * PK_SPB;   // row 384 + 4 t + cq at pq[32 t]
    float sq = 0.0f;
    float nx[8];
#pragma unroll
    for (int i = 0; i < 8; i++) nx[i] = pq[(2 * i) * PK_SPB];
#pragma nounroll
    for (int t0 = 0; t0 < 120; t0 += 8) {
        float cur[8];
#pragma unroll
        for (int i = 0; i < 8; i++) cur[i] = nx[i];
        const int tn = t0 + 8 < 120 ? t0 + 8 : t0;
#pragma unroll
        for (int i = 0; i < 8; i++) nx[i] = pq[(2 * (tn + i)) * PK_SPB];
#pragma unroll
        for (int i = 0; i < 8; i++) sq += cur[i] * cur[i];
    }
    return ((quad_lane<0>(sq) + quad_lane<1>(sq)) + quad_lane<2>(sq)) + quad_lane<3>(sq);
}
// runs b0 .. b1 - 1 of twenty steps of yy_lookup (ref: src/pitch.rs:137-142): step j takes row 384 - j and drops row 864 - j; check point
// cky[m] = yy after step 5 m.  Lane cq of a quad prepares steps 4 k + 1 + cq.
__device__ __forceinline__ float pk_chain_yy(const float *pbs, int cq, int cs, float yy, int b0, int b1, float (*cky)[PK_SPB])
{
    const float *pq = pbs + ((cq & 1) ? 0 : PK_ODD) + (191 - (cq >> 1)) * PK_SPB;   // row 383 - 4 k - cq at pq[-32 k]
    float na[5], nc[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { na[i] = pq[-(2 * (5 * b0 + i)) * PK_SPB]; nc[i] = pq[(240 - 2 * (5 * b0 + i)) * PK_SPB]; }
#pragma nounroll
    for (int bk = b0; bk < b1; bk++) {
        float ca[5], cc[5];
#pragma unroll
        for (int i = 0; i < 5; i++) { ca[i] = na[i]; cc[i] = nc[i]; }
        const int bn = bk + 1 < b1 ? bk + 1 : bk;
#pragma unroll
        for (int i = 0; i < 5; i++) { na[i] = pq[-(2 * (5 * bn + i)) * PK_SPB]; nc[i] = pq[(240 - 2 * (5 * bn + i)) * PK_SPB]; }
#pragma unroll
        for (int i = 0; i < 5; i++) {   // steps 20 bk + 4 i + 1 .. + 4: the run's check points fall behind step 5, 10, 15, 20
            const float t = ca[i] * ca[i] - cc[i] * cc[i];
            const Quad4 r = pk_quad4(t);
            yy = yy + r.t0;
            if (i == 1 && cq == 0) cky[4 * bk + 1][cs] = yy;
            yy = yy + r.t1;
            if (i == 2 && cq == 0) cky[4 * bk + 2][cs] = yy;
            yy = yy + r.t2;
            if (i == 3 && cq == 0) cky[4 * bk + 3][cs] = yy;
            yy = yy + r.t3;
            if (i == 4 && cq == 0) cky[4 * bk + 4][cs] = yy;
        }
    }
    return yy;
}

// sum_{j < 240} x[j] y[j] in order (ref: src/pitch.rs:296-363, one lag), rows PK_SPB floats apart: two register sets in turn, so that the
// rows of the next eight taps travel while these eight are summed (the compiler rotates a one-set prefetch back into "load, wait, use")
__device__ __forceinline__ float pk_dot240(const float *xp, const float *yp)
{
    float c = 0.0f;
    v2f xa[4], ya[4], xb[4], yb[4];
#define NNN_LD(X, Y, J) do { _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) { \
        X[i_] = mk2(xp[((J) + 2 * i_) * PK_SPB], xp[((J) + 2 * i_ + 1) * PK_SPB]); Y[i_] = mk2(yp[((J) + 2 * i_) * PK_SPB], yp[((J) + 2 * i_ + 1) * PK_SPB]); } } while (0)
#define NNN_ACC(X, Y) do { _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) { const v2f pr_ = pk_mul(X[i_], Y[i_]); c = sadd(c, pr_.x); c = sadd(c, pr_.y); } } while (0)
    NNN_LD(xa, ya, 0);
#pragma nounroll
    for (int j = 0; j < 224; j += 16) {
        NNN_LD(xb, yb, j + 8);
        NNN_ACC(xa, ya);
        NNN_LD(xa, ya, j + 16);
        NNN_ACC(xb, yb);
    }
    NNN_LD(xb, yb, 232);
    NNN_ACC(xa, ya);
    NNN_ACC(xb, yb);
#undef NNN_LD
#undef NNN_ACC
    return c;
}

constexpr int PK_SEG_FS = 120;
#ifndef NNN_PK_SEG_F1
#define NNN_PK_SEG_F1 44
#endif
constexpr int PK_SEG_F1 = NNN_PK_SEG_F1;   // groups of fine lags wave 5 has scanned when the survivors' exact sums are done; the rest beside find_best
static_assert(PK_SEG_F1 % 2 == 0 && PK_SEG_F1 <= PK_FINE_K, "");


// (defined behind the transforms, further down: the X transform of a one-frame call in rider blocks of k_pitch's launch)
__device__ __forceinline__ void xt_rider(const Buffers &b, const StepParams *sp, int rb, void *lds);

#ifndef NNN_BISECT_A
#define NNN_BISECT_A 0
#endif
#ifndef NNN_BISECT_B
#define NNN_BISECT_B 0
#endif
#ifndef NNN_BISECT_C
#define NNN_BISECT_C 0
#endif
#ifndef NNN_BISECT_D
#define NNN_BISECT_D 0
#endif
#ifndef NNN_PK_PRIO
#define NNN_PK_PRIO 0
#endif
#ifndef NNN_PK_LATE_WINDOW
#define NNN_PK_LATE_WINDOW 1
#endif
#ifndef NNN_PK_MINWAVES
#define NNN_PK_MINWAVES 4   // waves per SIMD: two blocks of 8 waves per CU, <= 128 registers
#endif
// `chain` != 0: one workgroup per (frame, quarter tile), work item = frame * blocks_per_frame + quarter tile.  Everything but the
// decision loop of remove_doubling is independent from frame to frame, so the frames of a group run side by side and a workgroup
// waits -- just before that loop -- for the flag its predecessor (same streams, previous frame: a lower work item) sets once its
// pitch and gain are in memory.  A workgroup takes its work item from a device-wide ticket counter when it STARTS (`tbase` = the
// counter's value before this launch), not from its block index: the holder of item i then knows that every item below i has been
// taken by a workgroup that is already running, so the wait always ends -- whatever order the hardware dispatches workgroups in
// (HIP promises none; in the observed in-order dispatch ticket and block index coincide).  `seq0` numbers the group's first frame;
// flag values are frame numbers, so a flag left by an earlier use of the scratch set never matches.  `chain` == 0: one workgroup
// per quarter tile loops over the frames.
// `lpc_here` != 0 (one-frame launches of a few thousand streams, the real-time tick): the LPC analysis runs here, on five of the block's
// waves ahead of the FIR, instead of as a launch of its own (k_lpc_wide) ahead of this one -- round 2's arrangement, which costs the
// block 9 us with six waves waiting; for a group of frames that was the kernel's worst phase, for a lone frame it is cheaper than the
// 14.5 us launch plus its gap on the call's critical path.  Same sums in the same order: bit-identical to k_lpc / k_lpc_wide.
// LPC: the instantiation that can run the LPC analysis (`lpc_here`): its autocorrelation holds 36 registers beside the prefetched window, which
// costs the frame loop of the groups' instantiation spills at the kernel's 128-register limit (round 6).
template <bool LPC>
__global__ void __launch_bounds__(PK_T, NNN_PK_MINWAVES) k_pitch(Buffers b, const StepParams *sp0, int g, int chain, int seq0, unsigned tbase, int lpc_here_,
                                                                 int riders)
{
    __shared__ PkLds L;
    const int lpc_here = LPC ? lpc_here_ : 0;
    if (riders > 0 && (int)blockIdx.x >= riders) {   // (one-frame launches only: chain == 0, the pitch blocks are blocks 0 .. riders - 1)
        xt_rider(b, sp0, (int)blockIdx.x - riders, &L);
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane0 = threadIdx.x & 63;
    int lane = lane0, s = lane & 15, q = lane >> 4;  // lane = (stream, chain)
    int item = (int)blockIdx.x;
    if (chain) {
        if (threadIdx.x == 0) L.u.f.any_refine = (int)(ticket_take(b.ticket) - tbase);
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(L.u.f.any_refine);
        __syncthreads();   // (the field is written again further down)
    }
    // Workgroup b runs on XCD b mod 8 (observed dispatch order; a speed matter only).  The four quarter-tile blocks of tile t are
    // sent to XCD t mod 8 -- the one whose L2 holds the tile's decimated history, written there by k_hp's block t: consecutive
    // block indices would spread them over four XCDs, each fetching the same lines.
    const int per = b.S_pad / PK_SPB;   // blocks per frame
    const int f_begin = chain ? item / per : 0, f_end = chain ? f_begin + 1 : g;
    int tile, sub_;
    xcd_tile_block(item - f_begin * per, b.NT, TILE / PK_SPB, tile, sub_);
    const int q0 = sub_ * PK_SPB;   // first stream of this block within its tile
    if (tile * TILE + q0 >= b.S) return;   // (a block whose streams are all padding -- the last tile of a batch that is not a multiple of 64 -- has nothing to do)
    const int min_period = PITCH_MIN / 2, max_period = PITCH_MAX / 2;
    const bool dec_lane = wave == 0 && lane0 < PK_SPB;                // lane = stream decisions
    int last_period = 0;
    float last_gain = 0.0f;
    if (dec_lane && f_begin == 0) {
        last_period = NNN_TI(b.last_period, 1, tile, q0 + s)[0];
        last_gain = NNN_TI(b.last_gain, 1, tile, q0 + s)[0];
    }
    // (the head waves' sums travel with the window: asked for where they are used, their trip to memory stood on the critical path)
    const float head_c0 = (lpc_here == 2 && wave < 5 && lane0 < PK_SPB) ? b.lpc_head[((size_t)tile * 5 + wave) * TILE + q0 + lane0] : 0.0f;
    float win[PK_CH], fir[5];
    pk_window_load(b, sp0 + f_begin, tile, q0, (int)threadIdx.x, win, fir);
    for (int f = f_begin; f < f_end; f++) {
        lane = launder_v(lane0);   // keep the frame loop's addresses inside the loop (see launder_v)
        s = lane & 15;
        q = lane >> 4;
        const int sl = q0 + s;
        NNN_STAMP(b, 0);
        const int tid = 64 * wave + lane, col = tid & 15, ch = tid >> 4;
        const int qi = ((q & 1) << 1) | (q >> 1);   // inner-product partial of this lane: q = 0, 2 | 1, 3 over the wave's quarters
        const int chc = ch < PK_NCH ? ch : PK_NCH - 1;   // (idle threads shadow the last chunk's reads and store nothing)
        float *chE = L.pb + (chc * (PK_CH / 2)) * PK_SPB + col, *chO = chE + PK_ODD;   // this thread's chunk: rows 2m / 2m + 1 at ch?[16 m]
        // ---- the window -> LDS
        if (ch < PK_NCH) {
#pragma unroll
            for (int m = 0; m < PK_CH / 2; m++) { chE[m * PK_SPB] = win[2 * m]; chO[m * PK_SPB] = win[2 * m + 1]; }
        }
        __syncthreads();
        NNN_STAMP(b, 1);
        // (the autocorrelation and the Levinson recursion that stood here -- two waves busy for a fifth of the block's time, six waiting
        // -- are k_lpc's now: lane = stream, ahead of this launch; the FIR taps arrive with the window)
        if (lpc_here) {
            // ... except for a lone frame: one lag per wave on waves 0 .. 4, lane = stream; the window is in LDS (row r of stream s at
            // pk_at(r, s), row 0 already the frame's special first element)
            float *acs = &L.u.c.xc[0][0], *firs = &L.u.c.xc[8][0];   // [5][16] each, in space the coarse search takes later
            if (wave < 5 && lane < PK_SPB) {   // wave w: lag w of the block's 16 streams (lane = stream)
                const float *pbs = L.pb + lane;
                // (`lpc_here` == 2: k_hp2's head waves took the first LPC_HEAD_BLK blocks of every sum while the frame was being filtered)
                const int blk0 = lpc_here == 2 ? LPC_HEAD_BLK : 0;
                const float c0 = head_c0;
                float a;
                if (wave == 0) a = pk_autocorr<0>(pbs, blk0, c0);
                else if (wave == 1) a = pk_autocorr<1>(pbs, blk0, c0);
                else if (wave == 2) a = pk_autocorr<2>(pbs, blk0, c0);
                else if (wave == 3) a = pk_autocorr<3>(pbs, blk0, c0);
                else a = pk_autocorr<4>(pbs, blk0, c0);
                acs[wave * PK_SPB + lane] = a;
            }
            __syncthreads();
            if (dec_lane) {
                float ac[5], taps[5];
#pragma unroll
                for (int i = 0; i < 5; i++) ac[i] = acs[i * PK_SPB + s];
                lpc_finish(b, tile, sl, sp0[f].slot, ac, taps);
#pragma unroll
                for (int i = 0; i < 5; i++) firs[i * PK_SPB + s] = taps[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 5; i++) fir[i] = firs[i * PK_SPB + col];
        }
        // ---- FIR5 with zero initial memory, in place (ref: src/pitch.rs:407-429): the chunk's inputs are still in the
        //      thread's registers, the five rows before it come from LDS before anyone overwrites them
        {
            float v[PK_CH + 5];
            {
                const int hb = chc > 0 ? 0 : 3 * PK_SPB;   // chunk 0 has no rows before it: read in range, use zeros
                const float h0 = chO[hb - 3 * PK_SPB], h1 = chE[hb - 2 * PK_SPB], h2 = chO[hb - 2 * PK_SPB], h3 = chE[hb - 1 * PK_SPB],
                            h4 = chO[hb - 1 * PK_SPB];
                v[0] = chc > 0 ? h0 : 0.0f; v[1] = chc > 0 ? h1 : 0.0f; v[2] = chc > 0 ? h2 : 0.0f; v[3] = chc > 0 ? h3 : 0.0f;
                v[4] = chc > 0 ? h4 : 0.0f;
#pragma unroll
                for (int u = 0; u < PK_CH; u++) v[5 + u] = win[u];
            }
            __syncthreads();   // every chunk has its inputs
            if (ch < PK_NCH) {
                // Two consecutive outputs are independent sums with the same coefficients: the halves of packed instructions (each half
                // rounds like the single instruction; products and sums in the reference's order, left to right).  Output pair m needs
                // the inputs as pairs in both alignments, VE[i] = (v[2i], v[2i+1]) and VO[i] = (v[2i+1], v[2i+2]): 10 packed
                // instructions per two outputs where scalar code issued 20 (round 5; the second alignment costs a register move per pair).
                const v2f N01 = mk2(fir[0], fir[1]), N23 = mk2(fir[2], fir[3]), N4 = mk2(fir[4], fir[4]);
                float *tap = b.taps ? NNN_TIF(b, xlp_ti, XLP, f, tile, q0 + col) + (size_t)(ch * PK_CH) * TILE : nullptr;
#pragma unroll
                for (int m = 0; m < PK_CH / 2; m++) {
                    const v2f VE0 = mk2(v[2 * m], v[2 * m + 1]), VO0 = mk2(v[2 * m + 1], v[2 * m + 2]);
                    const v2f VE1 = mk2(v[2 * m + 2], v[2 * m + 3]), VO1 = mk2(v[2 * m + 3], v[2 * m + 4]);
                    const v2f VE2 = mk2(v[2 * m + 4], v[2 * m + 5]), VO2 = mk2(v[2 * m + 5], v[2 * m + 6]);
                    // out = x + n0 m0 + n1 m1 + n2 m2 + n3 m3 + n4 m4, left to right (m0 = previous input, ...)
                    v2f o = pk_add(VO2, pk_mul_bx(N01, VE2));
                    o = pk_add(o, pk_mul_by(N01, VO1));
                    o = pk_add(o, pk_mul_bx(N23, VE1));
                    o = pk_add(o, pk_mul_by(N23, VO0));
                    o = pk_add(o, pk_mul_bx(N4, VE0));
                    if (tap) { tap[(size_t)(2 * m) * TILE] = o.x; tap[(size_t)(2 * m + 1) * TILE] = o.y; }
                    chE[m * PK_SPB] = o.x;
                    chO[m * PK_SPB] = o.y;
                }
            }
            if (ch < PK_NCH) {
                // for the certified coarse search: the chunk's 16 even rows -- the 4x-decimated signal -- as bf16 (to nearest) and their energy.
                // Read back from LDS: held in registers through the FIR they cost the kernel spills (it sits at its 128-register limit there).
                const float *ce = L.pb + launder_v((chc * (PK_CH / 2)) * PK_SPB + col);
                unsigned *pl = (unsigned *)((col < PK_SPB / 2 ? (unsigned short *)&L.ckf[0][0] : L.u.a.plane) + (col & 7) * PK_PLW + (PK_CH / 2) * ch);
                unsigned nz = 0;
                float bs = 0.0f;
#pragma unroll
                for (int m = 0; m < PK_CH / 2; m += 2) {
                    const float e0 = ce[m * PK_SPB], e1 = ce[(m + 1) * PK_SPB];
                    pl[m >> 1] = pk_bf16_rn(e0, e1);
                    bs += e0 * e0;
                    bs += e1 * e1;
                    nz |= (__float_as_uint(e0) | __float_as_uint(e1)) << 1;
                }
                // (a block with any nonzero value has a nonzero sum -- squares underflow; a NaN or an infinity stays what it is)
                L.u.a.bsum[col][ch] = (nz != 0 && bs < 0x1p-120f) ? 0x1p-120f : bs;
            }
            if (tid < PK_SPB * 5 + 2) (&L.u.a.mask[0][0])[tid] = 0u;   // the survivors' masks, count and the full-search flag
        }
#if !NNN_PK_LATE_WINDOW
        if (f + 1 < f_end) pk_window_load(b, sp0 + f + 1, tile, q0, tid, win, fir);   // the next frame's window travels behind this frame's work
#endif
        __syncthreads();
        NNN_STAMP(b, 4);
        // ---- coarse search (ref: src/pitch.rs:83-84 -> :296-363, :372-405).  find_best_pitch returns the two lags with the largest
        //      corr^2 / energy and nothing else of the 147 cross-correlations is ever used, so most of them need not be exact
        //      (round 6).  Certified search:
        //      (1) every correlation APPROXIMATELY, on the matrix cores: D[i][j] = sum_k A[i][k] B[k][j] with A[i][k] = x4[k - i]
        //          (zero outside 0 .. 239), B[k][j] = y4[k + 16 j] is lag i + 16 j -- eight v_mfma_f32_16x16x32_bf16 per stream, the
        //          operands the signal rounded to bf16 (the FIR left that plane in LDS).  Rigorous error, from the energies of the 16-value
        //          blocks the window covers: |approx - reference| <= e = PK_EPS sqrt(|x4|^2 W_j), W_j >= |y4[16 j .. 16 j + 255]|^2.
        //      (2) with the exact running energy den_L of every lag (the reference's own serial scan): lag L certainly scores at least
        //          lo_L^2 = (c_L - e)^2 / den_L (if c_L - e > 0) and at most hi_L^2 = (c_L + e)^2 / den_L.  Let T be the SECOND largest lo.
        //          A lag with hi_L < T is beaten by two lags whatever its exact value: it cannot be in the final pair, and -- being below
        //          both of them by more than the comparisons' own rounding -- it cannot change which of the others end up there
        //          (DESIGN.md section 4.1 has the argument).  Everything else SURVIVES: typically two to five lags per stream.
        //      (3) the survivors' sums exactly, in the reference's order, lane = (stream, lag); find_best_pitch over them in lag order.
        //      Blocks where that does not apply -- a stream with non-finite or extreme values, fewer than two certain lags and many
        //      candidates, parity taps that want all 147 values -- take the full search below: round 5's code, every lag exact.
        //      Roles: waves 0 .. 4 the search (streams wave, wave + 5, wave + 10 side by side, so that one's latencies are another's issue
        //      slots), waves 5, 6, 7 the three serial energy scans (see pk_chain_*), which run beside it in pieces cut at the search's
        //      barriers; wave 6, whose scan starts with the shortest sum, takes the sixteenth stream behind it.
        //      (2) does NOT wait for the coarse lags' energy scan: it bounds den_L from both sides with the block energies and the bf16
        //      plane (|den_L - (1 + |y4[L .. L + 239]|^2)| is the scan's own rounding, <= 2^-15 (1 + |y4|^2)); the scan's exact values are
        //      first needed by find_best_pitch over the survivors.
        const float *pE = L.pb + s, *pO = pE + PK_ODD;   // rows 2m / 2m + 1 of this lane's stream at p?[16 m]
        const int li = lane & 15, kg = lane >> 4;        // matrix fragments: row / column, k group
        const int cs = lane >> 2, cq = lane & 3;         // scan waves: stream, quad lane
        const float *pcs = L.pb + cs;
        const int wv = launder_s(wave);                  // (keeps this phase's wave-uniform addresses inside the frame loop, see launder_v)
        float chain_y = 0.0f;                            // scan waves: the running energy
        // the search for NS streams s0, s0 + 5, ... on the calling wave (NS a compile-time constant: waves 0 .. 4 take three streams each, wave 6 --
        // whose scan starts with the shortest sum -- the sixteenth behind it)
        auto search = [&](auto ns_, const int s0) {
            constexpr int NS = decltype(ns_)::value;
            f32x4 cacc[NS];                              // approximate correlations of the wave's streams
            // x4[u] sits at halfword 192 + u of the plane; a row of A reaches 15 halfwords before x4[0] (first k-step) and 31 behind x4[239]
            // (last k-step): masked.  Halfword e of lane (li, kg) is x4[32 t + 8 kg + e - li].
            unsigned m0[4], m7[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int n0 = li - 8 * kg, n1 = 16 + li - 8 * kg;
                m0[w] = (2 * w >= n0 ? 0xffffu : 0u) | (2 * w + 1 >= n0 ? 0xffff0000u : 0u);
                m7[w] = (2 * w < n1 ? 0xffffu : 0u) | (2 * w + 1 < n1 ? 0xffff0000u : 0u);
            }
            const unsigned sh = (unsigned)(li & 1) * 16u;
            const char *pl[NS], *pa[NS];
#pragma unroll
            for (int u = 0; u < NS; u++) {
                const int sq = s0 + 5 * u;
                pl[u] = (const char *)((sq < PK_SPB / 2 ? (const unsigned short *)&L.ckf[0][0] : L.u.a.plane) + (sq & 7) * PK_PLW);
                pa[u] = pl[u] + ((384 + 16 * kg - 2 * li) & ~3);   // A: the word that holds halfword 192 + 8 kg - li
                pl[u] += 16 * kg + 32 * li;                        // B: halfword 8 kg + 16 li
                cacc[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int t = 0; t < 8; t++) {
#pragma unroll
                for (int u = 0; u < NS; u++) {
                    const uint4 bq = *(const uint4 *)(pl[u] + 64 * t);
                    const unsigned *d = (const unsigned *)(pa[u] + 64 * t);
                    const unsigned d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
                    uint4 aq = make_uint4(align_bits(d1, d0, sh), align_bits(d2, d1, sh), align_bits(d3, d2, sh), align_bits(d4, d3, sh));
                    if (t == 0) { aq.x &= m0[0]; aq.y &= m0[1]; aq.z &= m0[2]; aq.w &= m0[3]; }
                    if (t == 7) { aq.x &= m7[0]; aq.y &= m7[1]; aq.z &= m7[2]; aq.w &= m7[3]; }
                    cacc[u] = mfma_16x16x32_bf16(aq, bq, cacc[u]);
                }
            }
            NNN_STAMPW(b, 2, wave == 0);
            if (b.taps == 2) {   // (test mode: the cross-correlation tap keeps NaN where a lag was ruled out)
                for (int u = 0; u < NS; u++)
                    for (int i = lane; i < NLAG1; i += 64) NNN_TIF(b, xc1, NLAG1, f, tile, q0 + s0 + 5 * u)[(size_t)i * TILE] = __builtin_nanf("");
            }
            // ---- (2) who survives.  Element r of lane (li, kg) of D is row 4 kg + r, column li: lag 16 li + 4 kg + r.
            const int lj = li < 10 ? li : 9;   // (columns 10 .. 15 hold no lag: they follow column 9 and are masked)
            float chp[NS][4], clo[NS][4], m1[NS], m2[NS];
            unsigned cfl = 0;                  // bit u: stream u has nothing but zeros in x4; bit 4 + u: stream u is not ordinary
#pragma unroll
            for (int u = 0; u < NS; u++) {
                const int sq = s0 + 5 * u;
                const float *bsr = L.u.a.bsum[sq];
                // W_j = blocks j .. j + 15 (>= the energy of the 240-value window of every lag of column j); |x4|^2 = blocks 12 .. 26 = W_12
                float wub = 0.0f;
#pragma unroll
                for (int n = 0; n < 4; n++) { const int ix = li + 4 * kg + n; wub += ix < 27 ? bsr[ix] : 0.0f; }
                wub += wave_xor16(wub, lane);
                wub += wave_xor32(wub, lane);
                const float xxu = lane_value(wub, 12), wtot = lane_value(wub, 0) + lane_value(wub, 11);
                // the search is certified for ordinary values only: no NaN or infinity, no product of the comparisons in find_best_pitch
                // near the ends of the f32 range (|corr| <= sqrt(|x4|^2 W) <= 2^41, energies <= 2^41 + 1: corr^2 * energy < 2^124)
                const bool odd = !(wtot <= 0x1p41f) || !(xxu >= 0x1p-60f);
                const bool xzero = xxu == 0.0f;   // every x4 is zero: every correlation is zero (or NaN), none is > 0 -- no survivor
                cfl |= (xzero ? 1u : 0u) << u | ((odd && !xzero) ? 16u : 0u) << u;
                const float e = PK_EPS * fast_sqrt(xxu) * fast_sqrt(wub);   // (two roots: the product of two small energies would underflow)
                // den_L from both sides.  The window of lag 16 j + i is the tail of block j from value i on, blocks j + 1 .. j + 14 whole (their
                // f32 energies) and the first i values of block j + 15: the two partial blocks from the bf16 plane, this lane's quarter of each
                // as suffix / prefix sums, the other quarters' totals from the lanes that hold them.
                const unsigned short *pv = (sq < PK_SPB / 2 ? (const unsigned short *)&L.ckf[0][0] : L.u.a.plane) + (sq & 7) * PK_PLW + 16 * lj + 4 * kg;
                const uint2 av = *(const uint2 *)pv, cv = *(const uint2 *)(pv + 240);
                const float a0 = __uint_as_float(av.x << 16), a1 = __uint_as_float(av.x & 0xffff0000u), a2 = __uint_as_float(av.y << 16), a3 = __uint_as_float(av.y & 0xffff0000u);
                const float c0 = __uint_as_float(cv.x << 16), c1 = __uint_as_float(cv.x & 0xffff0000u), c2 = __uint_as_float(cv.y << 16), c3 = __uint_as_float(cv.y & 0xffff0000u);
                float sfx[4], pfx[4];
                sfx[3] = a3 * a3; sfx[2] = a2 * a2 + sfx[3]; sfx[1] = a1 * a1 + sfx[2]; sfx[0] = a0 * a0 + sfx[1];
                pfx[0] = 0.0f; pfx[1] = c0 * c0; pfx[2] = pfx[1] + c1 * c1; pfx[3] = pfx[2] + c2 * c2;
                const float qa = sfx[0], qc = pfx[3] + c3 * c3;
                const float qa1 = wave_xor16(qa, lane), qa2 = wave_xor32(qa, lane), qa3 = wave_xor32(qa1, lane);   // the quarter sums of lanes kg ^ 1, kg ^ 2, kg ^ 3
                const float qc1 = wave_xor16(qc, lane), qc2 = wave_xor32(qc, lane), qc3 = wave_xor32(qc1, lane);
                const float sa = ((kg ^ 1) > kg ? qa1 : 0.0f) + ((kg ^ 2) > kg ? qa2 : 0.0f) + ((kg ^ 3) > kg ? qa3 : 0.0f);   // the quarters behind this one
                const float sc = ((kg ^ 1) < kg ? qc1 : 0.0f) + ((kg ^ 2) < kg ? qc2 : 0.0f) + ((kg ^ 3) < kg ? qc3 : 0.0f);   // the quarters ahead of this one
                const float s14 = wub - bsr[lj] - bsr[lj + 15];
                const float slack = 0x1p-22f * wub + 0x1p-15f * (1.0f + wtot);   // the subtraction above; the serial scan's own rounding
                m1[u] = -INFINITY;
                m2[u] = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int Lr = 4 * kg + r + 16 * li;
                    const bool valid = li < 10 && Lr < NLAG1;
                    const float pr = (sfx[r] + sa) + (pfx[r] + sc);   // the partial blocks' share, within 2^-6.98 (bf16 values squared)
                    const float dlo = fmaxf(1.0f + (s14 + pr * (1.0f - 0x1p-6f) - slack), 1.0f), dhi = 1.0f + (s14 + pr * (1.0f + 0x1p-6f) + slack);
                    const float ch = cacc[u][r] + e;
                    chp[u][r] = (valid && ch > 0.0f) ? ch * fast_rsq(dlo) : -1.0f;   // hi_L where the lag may be positive at all
                    clo[u][r] = valid ? (cacc[u][r] - e) * fast_rsq(dhi) : -INFINITY;
                    const float lo_ = fminf(m1[u], clo[u][r]);
                    m1[u] = fmaxf(m1[u], clo[u][r]);
                    m2[u] = fmaxf(m2[u], lo_);
                }
            }
            // the second largest lo of each stream: rows of 16 lanes by DPP, the four rows through scalars
#define NNN_TOP2(u, n1, n2) do { const float a1_ = (n1), a2_ = (n2), lo_ = fminf(m1[u], a1_); m1[u] = fmaxf(m1[u], a1_); m2[u] = fmaxf(lo_, fmaxf(m2[u], a2_)); } while (0)
#pragma unroll
            for (int u = 0; u < NS; u++) NNN_TOP2(u, row_partner<0>(m1[u]), row_partner<0>(m2[u]));
#pragma unroll
            for (int u = 0; u < NS; u++) NNN_TOP2(u, row_partner<1>(m1[u]), row_partner<1>(m2[u]));
#pragma unroll
            for (int u = 0; u < NS; u++) NNN_TOP2(u, row_partner<2>(m1[u]), row_partner<2>(m2[u]));
#pragma unroll
            for (int u = 0; u < NS; u++) NNN_TOP2(u, row_partner<3>(m1[u]), row_partner<3>(m2[u]));
#pragma unroll
            for (int u = 0; u < NS; u++) {
                const float b1 = lane_value(m1[u], 16), b2 = lane_value(m2[u], 16), c1 = lane_value(m1[u], 32), c2 = lane_value(m2[u], 32),
                            d1 = lane_value(m1[u], 48), d2 = lane_value(m2[u], 48);
                m1[u] = lane_value(m1[u], 0);
                m2[u] = lane_value(m2[u], 0);
                NNN_TOP2(u, b1, b2);
                NNN_TOP2(u, c1, c2);
                NNN_TOP2(u, d1, d2);
            }
#undef NNN_TOP2
            bool want_full = false;
#pragma unroll
            for (int u = 0; u < NS; u++) {
                const int sq = s0 + 5 * u;
                // two lags are certainly positive and certainly in the ordinary range: T = m2, less the rounding of this arithmetic, v_rsq_f32's
                // ulp and the margin of (2); else every lag that may be positive survives
                const bool two = m2[u] > 0x1p-40f;
                const float thr = two ? m2[u] * (1.0f - 0x1p-10f) : 0.0f;
                const bool xzero = (cfl >> u) & 1u;
                unsigned long long bal[4];
                unsigned tot = 0;
                bool keep[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    keep[r] = !xzero && chp[u][r] > 0.0f && chp[u][r] >= thr;
                    bal[r] = wave_ballot(keep[r]);
                    tot += (unsigned)__builtin_popcountll(bal[r]);
                }
                if (tot != 0) {
                    unsigned base = 0;
                    if (lane == 0) base = lds_add_u32(&L.u.a.count, tot);
                    base = __float_as_uint(lane_value(__uint_as_float(base), 0));
                    want_full |= base + tot > (unsigned)PK_CAP;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int Lr = 4 * kg + r + 16 * li;
                        const unsigned ix = base + lane_rank(bal[r], lane);
                        if (keep[r] && ix < (unsigned)PK_CAP) {
                            L.u.a.list[ix] = (unsigned short)((sq << 8) | Lr);
                            lds_or_u32(&L.u.a.mask[sq][Lr >> 5], 1u << (Lr & 31));
                        }
                        base += (unsigned)__builtin_popcountll(bal[r]);
                    }
                }
                want_full |= ((cfl >> (4 + u)) & 1u) != 0 || tot > (unsigned)PK_RMAX;
            }
            if (lane == 0 && (want_full || b.taps == 1)) L.u.a.full = 1u;
        };
        if (wv < 5) search(std::integral_constant<int, 3>(), wv);
        else if (wv == 7) {
            chain_y = pk_chain_coarse_start(pcs, cq);
            chain_y = pk_chain_coarse(pcs, cq, chain_y, 0, PK_COARSE_K, L.u.a.denck[cs]);
            NNN_STAMPW(b, 3, true);
        } else if (wv == 5) {
            chain_y = pk_chain_fine_start(pcs, cq, 1.0f, 0, PK_SEG_FS);
            NNN_STAMPW(b, 28, true);
        } else if (wv == 6) {
            chain_y = pk_chain_yy_start(pcs, cq);
            NNN_STAMPW(b, 29, true);
            search(std::integral_constant<int, 1>(), 15);
        }
        __syncthreads();   // (the planes are read: the scans' check points may take their place; the survivors are listed)
        NNN_STAMP(b, 5);
        const bool full = L.u.a.full != 0;           // block-uniform
        const unsigned nsurv = L.u.a.count;
        if (full) {
            __syncthreads();   // (everyone has read the flag: the full search's arrays take the space)
            // ---- full search: the cross-correlation of every lag on waves 0..2, the running energy of the coarse lags on wave 3
            const int grp = 4 * wave + q;
            if (grp < PK_NG) {
                // xcorr[L] = sum_j x4[j] y4[L + j], x4[j] = p[384 + 2j], y4[m] = p[2m] (the even rows, compact): a sequential sum
                // per lag (ref: src/pitch.rs:296-363).  Lags L0 .. L0 + 11 in six packed accumulators, L0 + 12 single.  The y window
                // w[i] = y4[L0 + j + i] is kept as pairs in both alignments, PE[t] = (w[2t], w[2t+1]) and PO[t] = (w[2t+1], w[2t+2]):
                // tap k multiplies pairs w[k+2n], w[k+2n+1], whichever alignment that is, with x4[j+k] from either half of its pair.
                const int L0 = PK_LC * grp;
                const float *yb = L.pb + L0 * PK_SPB + s, *xb = L.pb + 192 * PK_SPB + s;
                v2f acc2[PK_NP], PE[PK_NT], PO[PK_NT];
                float acc1 = 0.0f;
#pragma unroll
                for (int n = 0; n < PK_NP; n++) acc2[n] = mk2(0.0f, 0.0f);
#pragma unroll
                for (int t = 0; t < PK_NT - PK_JB / 2; t++) {
                    PE[t] = mk2(yb[(2 * t) * PK_SPB], yb[(2 * t + 1) * PK_SPB]);
                    PO[t] = mk2(yb[(2 * t + 1) * PK_SPB], yb[(2 * t + 2) * PK_SPB]);
                }
#pragma unroll 5   // (the window's register pairs come round after five steps: no moves at the back edge)
                for (int j = 0; j < 240; j += PK_JB) {
                    const float *yj = yb + j * PK_SPB, *xj = xb + j * PK_SPB;
                    v2f X[PK_JB / 2];
#pragma unroll
                    for (int t = 0; t < PK_JB / 2; t++) X[t] = mk2(xj[(2 * t) * PK_SPB], xj[(2 * t + 1) * PK_SPB]);
#pragma unroll
                    for (int t = PK_NT - PK_JB / 2; t < PK_NT; t++) {
                        PE[t] = mk2(yj[(2 * t) * PK_SPB], yj[(2 * t + 1) * PK_SPB]);
                        PO[t] = mk2(yj[(2 * t + 1) * PK_SPB], yj[(2 * t + 2) * PK_SPB]);
                    }
#pragma unroll
                    for (int k = 0; k < PK_JB; k++) {
                        const v2f xp2 = X[k >> 1];
#pragma unroll
                        for (int n = 0; n < PK_NP; n++) {
                            const v2f wp = (k & 1) ? PO[(k >> 1) + n] : PE[(k >> 1) + n];
                            acc2[n] = pk_add(acc2[n], (k & 1) ? pk_mul_by(xp2, wp) : pk_mul_bx(xp2, wp));
                        }
                        const float w1 = (k & 1) ? PO[(k >> 1) + PK_NP].x : PE[(k >> 1) + PK_NP].x;
                        acc1 = sadd(acc1, ((k & 1) ? xp2.y : xp2.x) * w1);
                    }
#pragma unroll
                    for (int t = 0; t < PK_NT - PK_JB / 2; t++) { PE[t] = PE[t + PK_JB / 2]; PO[t] = PO[t + PK_JB / 2]; }
                }
                float acc[PK_LC];
#pragma unroll
                for (int n = 0; n < PK_NP; n++) { acc[2 * n] = acc2[n].x; acc[2 * n + 1] = acc2[n].y; }
                acc[PK_LC - 1] = acc1;
#pragma unroll
                for (int i = 0; i < PK_LC; i++)   // (what find_best_pitch needs of a correlation: its square if it is positive)
                    if (L0 + i < NLAG1) L.u.c.xc[L0 + i][s] = acc[i] > 0.0f ? acc[i] * acc[i] : __builtin_nanf("");
                if (b.taps) {
                    float *o = NNN_TIF(b, xc1, NLAG1, f, tile, sl);
#pragma unroll
                    for (int i = 0; i < PK_LC; i++)
                        if (L0 + i < NLAG1) o[(size_t)(L0 + i) * TILE] = acc[i];
                }
            }
            if (wave == PK_XW && lane < PK_SPB) {
                // the running energy every coarse lag sees in find_best_pitch (ref: src/pitch.rs:83 -> :380-402): even rows only
                float ysq = 1.0f;
#pragma nounroll
                for (int j0 = 0; j0 < 240; j0 += 8) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = pE[(j0 + i) * PK_SPB];
#pragma unroll
                    for (int i = 0; i < 8; i++) ysq += v[i] * v[i];
                }
#pragma nounroll
                for (int i0 = 0; i0 < NLAG1; i0 += 7) {
                    float a[7], d[7];
#pragma unroll
                    for (int i = 0; i < 7; i++) { a[i] = pE[(i0 + i + 240) * PK_SPB]; d[i] = pE[(i0 + i) * PK_SPB]; }
#pragma unroll
                    for (int i = 0; i < 7; i++) {
                        L.u.c.ysq[i0 + i][s] = ysq;
                        ysq += a[i] * a[i] - d[i] * d[i];
                        ysq = fmaxf(ysq, 1.0f);
                    }
                }
            }
            __syncthreads();
        }
        NNN_STAMP(b, 63);
        // ---- find_best_pitch over the coarse lags (ref: src/pitch.rs:372-405, call site :83-84): over the survivors' exact sums (certified
        //      search: waves 0 .. 4 and 7 make them, wave 0 scans) or over all 147 (full search: wave 0, a serial scan); beside it, on
        //      waves 5 and 6, the rest of the two energy scans whose results are looked up later in the frame
        int lo1 = 0, lo2 = 0;
        if (wv == 5) chain_y = pk_chain_fine(pcs, cq, cs, chain_y, 0, PK_SEG_F1, L.ckf);   // (the rest beside find_best, below)
        else if (wv == 6) {
            if (cq == 0) L.cky[0][cs] = chain_y;   // xx = yy_lookup[0]
            chain_y = pk_chain_yy(pcs, cq, cs, chain_y, 0, PK_YY_B, L.cky);
        } else if (wv != 7 && !full) {
            // (3) lane = (stream, lag) of the survivor list: xcorr[L] = sum_j x4[j] y4[L + j], x4[j] = p[384 + 2j], y4[m] = p[2m], a sequential
            //     sum (ref: src/pitch.rs:296-363), two taps per packed multiply, the adds in order
#pragma nounroll
            for (unsigned e0 = 64u * (unsigned)wv; e0 < nsurv; e0 += 64u * 5u) {
                const unsigned en = e0 + (unsigned)lane;
                if (en < nsurv) {
                    const unsigned ent = L.u.a.list[en];
                    const int se = (int)(ent >> 8), Le = (int)(ent & 255u);
                    const float *xp = L.pb + 192 * PK_SPB + se, *yp = L.pb + Le * PK_SPB + se;
                    const float c = pk_dot240(xp, yp);
                    // its place among the stream's survivors, in lag order
                    const unsigned *mk = L.u.a.mask[se];
                    int rk = 0;
#pragma unroll
                    for (int w = 0; w < 5; w++) {
                        const unsigned mw = mk[w], below = w < (Le >> 5) ? mw : (w == (Le >> 5) ? mw & ((1u << (Le & 31)) - 1u) : 0u);
                        rk += __builtin_popcount(below);
                    }
                    // the energy the lag saw in find_best_pitch: the scan's own steps from the check point below it (<= 3)
                    float dy = L.u.a.denck[se][Le >> 2];
                    {
                        const float *yq = L.pb + (Le & ~3) * PK_SPB + se;
                        float ra[3], rd[3];
#pragma unroll
                        for (int i = 0; i < 3; i++) { ra[i] = yq[(i + 240) * PK_SPB]; rd[i] = yq[i * PK_SPB]; }
#pragma unroll
                        for (int i = 0; i < 3; i++) {
                            const float yn = fmaxf(dy + (ra[i] * ra[i] - rd[i] * rd[i]), 1.0f);
                            dy = i < (Le & 3) ? yn : dy;
                        }
                    }
                    L.u.a.slotv[se][rk] = c;
                    L.u.a.slotl[se][rk] = Le;
                    L.u.a.slotd[se][rk] = dy;
                    if (b.taps) NNN_TIF(b, xc1, NLAG1, f, tile, q0 + se)[(size_t)Le * TILE] = c;
                }
            }
        }
        if (NNN_PK_PRIO && wv >= 5) wave_prio<0>();
        if (full) {
            if (wv == 5) chain_y = pk_chain_fine(pcs, cq, cs, chain_y, PK_SEG_F1, PK_FINE_K, L.ckf);
            if (dec_lane) {
                BestPitch bp;
                bp.init();
                float c[7], e[7];
#pragma unroll
                for (int i = 0; i < 7; i++) { c[i] = L.u.c.xc[i][s]; e[i] = L.u.c.ysq[i][s]; }
#pragma nounroll
                for (int i0 = 0; i0 < NLAG1; i0 += 7) {
                    float cn[7], en[7];   // the next seven lags travel while these are judged
                    const int i1 = i0 + 7 < NLAG1 ? i0 + 7 : i0;
#pragma unroll
                    for (int i = 0; i < 7; i++) { cn[i] = L.u.c.xc[i1 + i][s]; en[i] = L.u.c.ysq[i1 + i][s]; }
#pragma unroll
                    for (int i = 0; i < 7; i++) bp.update_sq(i0 + i, c[i], e[i]);
#pragma unroll
                    for (int i = 0; i < 7; i++) { c[i] = cn[i]; e[i] = en[i]; }
                }
                if (b.taps) {
                    int *o = (int *)NNN_TIF(b, best1, 2, f, tile, sl);
                    o[0] = bp.best;
                    o[TILE] = bp.second;
                }
                lo1 = 2 * bp.best - 2;
                lo2 = 2 * bp.second - 2;
                NNN_STAMP(b, 61);
            }
        } else {
            __syncthreads();   // (the survivors' sums of waves 0 .. 4, the coarse lags' energies of wave 7)
            NNN_STAMP(b, 27);
            if (wv == 5) chain_y = pk_chain_fine(pcs, cq, cs, chain_y, PK_SEG_F1, PK_FINE_K, L.ckf);
            if (wave == 0) {
                // find_best_pitch over the stream's survivors in lag order, with the energy each of them saw
                BestPitch bp;
                bp.init();
                int ns = 0;
#pragma unroll
                for (int w = 0; w < 5; w++) ns += lane < PK_SPB ? __builtin_popcount(L.u.a.mask[s][w]) : 0;
                float nc = 0.0f, nd = 1.0f;
                int nl = 0;
                if (ns > 0) { nl = L.u.a.slotl[s][0]; nc = L.u.a.slotv[s][0]; nd = L.u.a.slotd[s][0]; }
                for (int k = 0; wave_ballot(k < ns) != 0ull; k++) {
                    const int cl = nl;
                    const float cc = nc, cd = nd;
                    if (k + 1 < ns) { nl = L.u.a.slotl[s][k + 1]; nc = L.u.a.slotv[s][k + 1]; nd = L.u.a.slotd[s][k + 1]; }   // (the next one travels)
                    if (k < ns) bp.update(cl, cc, cd);
                }
                if (dec_lane) {
                    if (b.taps) {
                        int *o = (int *)NNN_TIF(b, best1, 2, f, tile, sl);
                        o[0] = bp.best;
                        o[TILE] = bp.second;
                    }
                    lo1 = 2 * bp.best - 2;
                    lo2 = 2 * bp.second - 2;
                }
                NNN_STAMP(b, 61);
            }
        }
        // (wave 0 alone read the search's arrays in this phase; the scans' waves write their check points only: its lanes may put the fine
        // search's windows into the space -- in the partial sums' layout -- before the barrier)
        if (wave == 0) wave_lds_sync();
        if (dec_lane) {
            L.u.f.lo[0][s] = lo1;
            L.u.f.lo[1][s] = lo2;
            if (s == 0) L.u.f.any_refine = 0;
        }
        __syncthreads();   // the coarse arrays are dead: their space takes the partial sums from here on
        NNN_STAMP(b, 6);
        // ---- fine cross-correlation at the <= 10 lags within +-2 of 2*best / 2*second (ref: src/pitch.rs:88-96): wave w
        //      takes lags lo1 + w and lo2 + w
        if (wave < 5) {
            const int la = L.u.f.lo[0][s] + wave, lb = L.u.f.lo[1][s] + wave;
            const bool va = la >= 0 && la < NLAG2, vb = lb >= 0 && lb < NLAG2;
            const int yr[2] = {va ? la : 0, vb ? lb : 0};
            float acc[2];
            pk_inner<2>(L.pb, s, qi, yr, acc);
            L.u.f.part[wave][qi][s] = acc[0];
            L.u.f.part[5 + wave][qi][s] = acc[1];
        } else if (wave == 5) {   // (beside it, on a wave the cross-correlation leaves idle)
            if (q < 2) {
                // the energy lags lo .. lo + 4 of window q saw: from the check point below the first of them, the scan's own
                // steps (at most 7 + 5; every row is requested before the first step is taken)
                const int lo = L.u.f.lo[q][s], first = lo > 0 ? lo : 0;
                const int i0 = (first < NLAG2 ? first : NLAG2 - 1) & ~(PK_CKF - 1);
                float y = L.ckf[i0 / PK_CKF][s], ev[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                float ra[PK_CKF + 4], rd[PK_CKF + 4];
#pragma unroll
                for (int st = 0; st < PK_CKF + 4; st++) {
                    const int i = i0 + st < NLAG2 ? i0 + st : NLAG2 - 1;
                    ra[st] = L.pb[pk_at(i + 480, s)];
                    rd[st] = L.pb[pk_at(i, s)];
                }
#pragma unroll
                for (int st = 0; st < PK_CKF + 4; st++) {
#pragma unroll
                    for (int c = 0; c < 5; c++) ev[c] = (i0 + st == lo + c) ? y : ev[c];
                    y += ra[st] * ra[st] - rd[st] * rd[st];
                    y = fmaxf(y, 1.0f);
                }
#pragma unroll
                for (int c = 0; c < 5; c++) L.u.f.ye[5 * q + c][s] = ev[c];
            }
        }
        __syncthreads();
        NNN_STAMP(b, 7);
        // ---- find_best_pitch over the fine lags: xcorr is zero outside the two 5-lag windows, so only they can update the
        //      best pitch; replayed in increasing lag order with the energy each of them saw.  Then the candidate periods.
        Xc2 xc;
        int t0 = 0;
        float xx = 0.0f;
        if (dec_lane) {
            xx = L.cky[0][s];
            xc.lo1 = lo1;
            xc.lo2 = lo2;
            float ye[10];
#pragma unroll
            for (int c = 0; c < 10; c++) {
                const int lag = c < 5 ? lo1 + c : lo2 + (c - 5);
                const bool valid = lag >= 0 && lag < NLAG2;
                float v = L.u.f.part[c][0][s] + L.u.f.part[c][1][s] + L.u.f.part[c][2][s] + L.u.f.part[c][3][s];
                v = fmaxf(v, -1.0f);
                xc.v[c] = valid ? v : 0.0f;
                ye[c] = valid ? L.u.f.ye[c][s] : 0.0f;
            }
            if (b.taps) {
                float *o = NNN_TIF(b, xc2, 10, f, tile, sl);
#pragma unroll
                for (int c = 0; c < 10; c++) o[(size_t)c * TILE] = xc.v[c];
            }
            NNN_STAMP(b, 59);
            BestPitch bp;
            bp.init();
            const int loA = min(lo1, lo2), loB = max(lo1, lo2);
            const bool a_first = lo1 <= lo2;
#pragma unroll
            for (int u = 0; u < 10; u++) {
                const int i = u < 5 ? loA + u : loB + (u - 5);
                const bool on = i >= 0 && i < NLAG2 && (u < 5 || i > loA + 4);
                // the energy of the lower window's lags sits at ye[0..4] when lo1 <= lo2, at ye[5..9] otherwise
                const float e = (u < 5) == a_first ? ye[u < 5 ? u : u - 5] : ye[5 + (u < 5 ? u : u - 5)];
                // (lag i is the u-th of its window: no search needed; where the windows overlap the values are the same)
                const float cv = (u < 5) == a_first ? xc.v[u < 5 ? u : u - 5] : xc.v[5 + (u < 5 ? u : u - 5)];
                if (on) bp.update(i, cv, e);
            }
            int offset = 0;
            if (bp.best > 0 && bp.best < NLAG2 - 1) {
                float a = xc.at(bp.best - 1), bb = xc.at(bp.best), c = xc.at(bp.best + 1);
                if (c - a > 0.7f * (bb - a)) offset = 1;
                else if (a - c > 0.7f * (bb - c)) offset = -1;
            }
            const int psr = 2 * bp.best - offset;
            if (b.taps) NNN_TIF(b, psearch, 1, f, tile, sl)[0] = psr;
            // ---- remove_doubling: the candidates of the decision loop + the two neighbours of t0 (slots 23, 24): if the loop keeps
            //      t0 the final +-1 refinement needs no inner products of its own.  Divisors k >= 13 never get past the loop's
            //      `t1 < min_period` break (t0 <= 383: (2 t0 + 13) / 26 <= 29), so slots exist for k = 2 .. PK_KMAX = 12 only.
            NNN_STAMP(b, 60);
            t0 = (PITCH_MAX - psr) / 2;
            if (t0 > max_period - 1) t0 = max_period - 1;
        #pragma unroll
            for (int e = 0; e < PK_NSLOT; e++) {   // (unrolled: k is a constant in every copy)
                int t;
                if (e == 0) t = t0;
                else if (e >= PK_NE) t = e == PK_NE ? t0 - 1 : t0 + 1;
                else {
                    const int k = 2 + (e - 1) / 2;
                    const int t1 = (2 * t0 + k) / (2 * k);
                    if ((e - 1) & 1) {
                        const int sc = kSecondCheck[k];
                        t = (k == 2) ? ((t1 + t0 > max_period) ? t0 : t0 + t1) : (2 * sc * t0 + k) / (2 * k);
                    } else t = t1;
                }
                L.u.f.cand[e][s] = t;
            }

            L.u.f.xx[s] = xx;
            L.u.f.t0[s] = t0;
        }
        __syncthreads();
        NNN_STAMP(b, 53);
        // ---- yy_lookup at the candidate periods (ref: src/pitch.rs:138-142): one candidate per lane (s, q) of waves 0..5; from the
        //      check point below T, at most four of the scan's steps.  (Read by the decision loop, behind the next barrier.)
        {
            const int e = 4 * wave + q;
            if (e < PK_NE) {
                const int T = L.u.f.cand[e][s], m = T / PK_CKY;
                float y = L.cky[m][s];
                float ra[PK_CKY - 1], rc[PK_CKY - 1];
#pragma unroll
                for (int st = 1; st < PK_CKY; st++) {
                    const int j = PK_CKY * m + st, jc = j <= 384 ? j : 384;   // step j: row 384 - j enters, row 864 - j leaves
                    ra[st - 1] = L.pb[pk_at(384 - jc, s)];
                    rc[st - 1] = L.pb[pk_at(864 - jc, s)];
                }
#pragma unroll
                for (int st = 1; st < PK_CKY; st++) {
                    const float yn = y + (ra[st - 1] * ra[st - 1] - rc[st - 1] * rc[st - 1]);
                    y = PK_CKY * m + st <= T ? yn : y;
                }
                L.u.f.yy[e][s] = fmaxf(y, 0.0f);
            }
        }
        // ---- the candidates' inner products against p[384 ..]: wave w takes slots w, w + 8, w + 16 (wave 0 also slot 24)
        static_assert(PK_NSLOT == 25, "three slots per wave and one more");
        if (wave == 0) {
            int yr[4];
#pragma unroll
            for (int c = 0; c < 4; c++) yr[c] = max_period - L.u.f.cand[8 * c][s];
            float acc[4];
            pk_inner<4>(L.pb, s, qi, yr, acc);
#pragma unroll
            for (int c = 0; c < 4; c++) L.u.f.part[8 * c][qi][s] = acc[c];
        } else {
            int yr[3];
#pragma unroll
            for (int c = 0; c < 3; c++) yr[c] = max_period - L.u.f.cand[wave + 8 * c][s];
            float acc[3];
            pk_inner<3>(L.pb, s, qi, yr, acc);
#pragma unroll
            for (int c = 0; c < 3; c++) L.u.f.part[wave + 8 * c][qi][s] = acc[c];
        }
#if NNN_PK_LATE_WINDOW   // the next frame's window is requested here, behind the candidates' inner products (round 6; round 5: ahead of them; until then behind the FIR): its 32 registers are free
                        // through the cross-correlation and the searches, and the ~8 us left of the frame still cover the trip (k_pitch -2.6 %; 0 = as before)
        pk_window_load(b, sp0 + f + 1, tile, q0, f + 1 < f_end ? tid : PK_T, win, fir);   // (the group's last frame: nothing to load, and no old value kept)
#endif
        if (dec_lane) {
            if (chain && f > 0) {
                // the previous frame of these streams is another workgroup's: wait for its flag, then take its pitch and gain
                const int *flag = (const int *)NNN_TIF(b, pflag, 1, f - 1, tile, q0);
                // (the ticket order guarantees the wait ends; the limit is wall time on the constant clock, not a spin count, so that a
                // predecessor slowed by a shared GPU or a stalled queue is waited for: giving up invalidates the streams' state for good)
                const long long t_wait = realtime_ticks();
                unsigned spins = 0;
                bool lost = false;
                while (flag_read(flag) != seq0 + f - 1 && !lost) {
                    chain_pause();
                    if ((++spins & 255u) == 0 && realtime_ticks() - t_wait > b.handoff_ticks) lost = true;
                }
                if (lost) *b.fault = 1;   // never seen (cannot happen, see the ticket order above): reported to the host, not hung on
                last_period = NNN_TIF(b, pitch, 1, f - 1, tile, sl)[0];
                last_gain = NNN_TIF(b, pgain, 1, f - 1, tile, sl)[0];
            }
            L.u.f.pprev[s] = last_period / 2;
            L.u.f.lgain[s] = last_gain;
        }
        __syncthreads();
        NNN_STAMP(b, 54);
        // ---- decision loop (ref: src/pitch.rs:150-206).  Whether divisor k replaces the best candidate depends on t0, the previous
        //      frame and k's own inner products, not on the other divisors: lane (stream, k) judges k = 2 .. 12 on three waves, the
        //      stream's lane then takes the last k that passed (the loop's break at the first t1 < min_period cuts a suffix: t1
        //      falls with k).
        if (wave < 3) {
            const int k = 2 + 4 * wave + q;
            if (k <= PK_KMAX) {
                auto ipv = [&](int e) { return L.u.f.part[e][0][s] + L.u.f.part[e][1][s] + L.u.f.part[e][2][s] + L.u.f.part[e][3][s]; };
                const int e1 = 1 + 2 * (k - 2), e2 = e1 + 1;
                const int t1 = L.u.f.cand[e1][s], t0s = L.u.f.t0[s];
                const float xxs = L.u.f.xx[s], lg = L.u.f.lgain[s];
                const float g0 = pitch_gain(ipv(0), xxs, L.u.f.yy[0][s]);
                const float xy = (ipv(e1) + ipv(e2)) / 2.0f;
                const float yy = (L.u.f.yy[e1][s] + L.u.f.yy[e2][s]) / 2.0f;
                const float g1 = pitch_gain(xy, xxs, yy);
                int d = t1 - L.u.f.pprev[s];
                if (d < 0) d = -d;
                float cont;
                if (d <= 1) cont = lg;
                else if (d <= 2 && 5 * k * k < t0s) cont = lg / 2.0f;
                else cont = 0.0f;
                float thresh;
                if (t1 < 3 * min_period) thresh = fmaxf(0.85f * g0 - cont, 0.4f);
                else if (t1 < 2 * min_period) thresh = fmaxf(0.9f * g0 - cont, 0.5f);
                else thresh = fmaxf(0.7f * g0 - cont, 0.3f);
                L.u.f.kpass[k][s] = (t1 >= min_period && g1 > thresh) ? 1 : 0;
                L.u.f.kxy[k][s] = xy;
                L.u.f.kyy[k][s] = yy;
                L.u.f.kg[k][s] = g1;
            }
        }
        __syncthreads();
        NNN_STAMP(b, 58);
        int t = 0;
        float pg = 0.0f, gg = 0.0f;
        if (dec_lane) {
            auto ipv = [&](int e) { return L.u.f.part[e][0][s] + L.u.f.part[e][1][s] + L.u.f.part[e][2][s] + L.u.f.part[e][3][s]; };
            t = t0;
            float best_xy = ipv(0), best_yy = L.u.f.yy[0][s];
            gg = pitch_gain(best_xy, xx, best_yy);
            int kw = 0;
#pragma unroll
            for (int k = 2; k <= PK_KMAX; k++) kw = L.u.f.kpass[k][s] ? k : kw;
            if (kw) {
                best_xy = L.u.f.kxy[kw][s];
                best_yy = L.u.f.kyy[kw][s];
                gg = L.u.f.kg[kw][s];
                t = L.u.f.cand[1 + 2 * (kw - 2)][s];
            }
            best_xy = fmaxf(best_xy, 0.0f);
            pg = (best_yy <= best_xy) ? 1.0f : best_xy / (best_yy + 1.0f);
            L.u.f.tsel[s] = t;
            if (t != t0) L.u.f.any_refine = 1;
        }
        __syncthreads();
        NNN_STAMP(b, 55);
        // ---- final +-1 refinement: the inner products at t - 1, t, t + 1 (the same sums whichever way they are obtained)
        const bool refine = L.u.f.any_refine != 0;   // block-uniform
        if (refine) {
            if (wave < 3) {
                const int yr[1] = {max_period - (L.u.f.tsel[s] + wave - 1)};
                float acc[1];
                pk_inner<1>(L.pb, s, qi, yr, acc);
                L.u.f.part[32 + wave][qi][s] = acc[0];
            }
            __syncthreads();
        }
        NNN_STAMP(b, 56);
        if (dec_lane) {
            auto ipv = [&](int e) { return L.u.f.part[e][0][s] + L.u.f.part[e][1][s] + L.u.f.part[e][2][s] + L.u.f.part[e][3][s]; };
            float x3[3];
            if (t == t0) { x3[0] = ipv(PK_NE); x3[1] = ipv(0); x3[2] = ipv(PK_NE + 1); }
            else { x3[0] = ipv(32); x3[1] = ipv(33); x3[2] = ipv(34); }
            int offset = 0;
            if (x3[2] - x3[0] > 0.7f * (x3[1] - x3[0])) offset = 1;
            else if (x3[0] - x3[2] > 0.7f * (x3[1] - x3[2])) offset = -1;
            pg = fminf(pg, gg);
            int res = 2 * t + offset;
            if (res < PITCH_MIN) res = PITCH_MIN;
            NNN_TIF(b, pitch, 1, f, tile, sl)[0] = res;
            NNN_TIF(b, pgain, 1, f, tile, sl)[0] = pg;
            last_period = res;
            last_gain = pg;
            if (chain) {
                __threadfence();   // every stream's pitch and gain before the flag
                if (lane0 == 0 && seq0 + f != b.dbg_withhold) flag_publish((int *)NNN_TIF(b, pflag, 1, f, tile, q0), seq0 + f);
            }
        }
        NNN_STAMP(b, 57);
        // (the next frame's first writes to anything this frame still reads sit behind barriers wave 0 takes part in)
    }
    if (dec_lane && f_end == g) {
        NNN_TI(b.last_period, 1, tile, q0 + s)[0] = last_period;
        NNN_TI(b.last_gain, 1, tile, q0 + s)[0] = last_gain;
    }
}

// ---------------------------------------------------------------------------------------------
// Real 960-point transforms as 480-point complex FFTs in LDS (Stockham autosort, radices 8 x 6 x 10,
// one wave per transform) plus the split/merge step.  The reference's FFT is third-party
// (easyfft 0.4.2 -> realfft 3.5.0 -> rustfft 6.4.1; call sites src/features.rs:264,290),
// un-normalised in both directions.
// ---------------------------------------------------------------------------------------------
// Everything from here to the end of k_fft_x, and k_synth further down, is downstream of an FFT: the reference itself is only
// defined to f32 rounding there (its FFT picks AVX / SSE / scalar code at run time) and parity is a tolerance.  A multiply fuses
// with an add exactly where the source says fmaf (complex products, band sums); nowhere else.  Round 3 let the compiler fuse at
// will in these regions (#pragma clang fp contract(fast): -5 % / -1 % static vector instructions in k_fft_xp / k_synth, no measured
// time, profiles/r3_experiments_ab.txt block A).  Round 4 took that back: the same source then rounds the same way in every kernel
// it is inlined into, and the fused back end (k_back) -- the same transforms and synthesis inside another kernel, where the
// compiler's choices came out differently in a third of the spectrum's bins -- gives the bits of k_fft_xp / k_synth, so a stream
// may change back end from call to call.  NNN_FFT_CONTRACT=1 builds the freely fusing variant for A/B runs.
#ifndef NNN_FFT_CONTRACT
#define NNN_FFT_CONTRACT 0
#endif
#if NNN_FFT_CONTRACT
#pragma clang fp contract(fast)
#endif
constexpr int NFFT = 480;

__device__ __forceinline__ float2 cmulf(float2 a, float2 w)
{
    return make_float2(fmaf(a.x, w.x, -a.y * w.y), fmaf(a.x, w.y, a.y * w.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 c) { return make_float2(a.x + c.x, a.y + c.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 c) { return make_float2(a.x - c.x, a.y - c.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)

__device__ __forceinline__ void bfly2(float2 &a, float2 &c) { float2 t = csub(a, c); a = cadd(a, c); c = t; }

__device__ __forceinline__ void dft8(float2 *v)
{
    const float h = 0.70710678118654752440f;
    float2 a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7];
    bfly2(a0, a4); bfly2(a1, a5); bfly2(a2, a6); bfly2(a3, a7);
    a5 = make_float2((a5.x + a5.y) * h, (a5.y - a5.x) * h);   // * exp(-i pi/4)
    a6 = mul_mi(a6);                                          // * exp(-i pi/2)
    a7 = make_float2((a7.y - a7.x) * h, (-a7.x - a7.y) * h);  // * exp(-3i pi/4)
    bfly2(a0, a2); bfly2(a1, a3); bfly2(a4, a6); bfly2(a5, a7);
    a3 = mul_mi(a3); a7 = mul_mi(a7);
    bfly2(a0, a1); bfly2(a2, a3); bfly2(a4, a5); bfly2(a6, a7);
    v[0] = a0; v[4] = a1; v[2] = a2; v[6] = a3; v[1] = a4; v[5] = a5; v[3] = a6; v[7] = a7;
}

__device__ __forceinline__ void dft3(float2 &v0, float2 &v1, float2 &v2)
{
    const float s = 0.86602540378443864676f;  // sin(2 pi / 3)
    float2 t1 = cadd(v1, v2);
    float2 t2 = csub(v1, v2);
    float2 m = make_float2(fmaf(-0.5f, t1.x, v0.x), fmaf(-0.5f, t1.y, v0.y));
    v0 = cadd(v0, t1);
    // v1 = m + (-i s t2), v2 = m - (-i s t2): the products fused into the sums (written out: the compiler's own contraction is off)
    v1 = make_float2(fmaf(s, t2.y, m.x), fmaf(-s, t2.x, m.y));
    v2 = make_float2(fmaf(-s, t2.y, m.x), fmaf(s, t2.x, m.y));
}

__device__ __forceinline__ void dft5(float2 &v0, float2 &v1, float2 &v2, float2 &v3, float2 &v4)
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;  // cos(2pi/5), cos(4pi/5)
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;   // sin(2pi/5), sin(4pi/5)
    float2 a1 = cadd(v1, v4), b1 = csub(v1, v4);
    float2 a2 = cadd(v2, v3), b2 = csub(v2, v3);
    float2 x0 = v0;
    float2 m1 = make_float2(fmaf(c2, a2.x, fmaf(c1, a1.x, x0.x)), fmaf(c2, a2.y, fmaf(c1, a1.y, x0.y)));
    float2 m2 = make_float2(fmaf(c1, a2.x, fmaf(c2, a1.x, x0.x)), fmaf(c1, a2.y, fmaf(c2, a1.y, x0.y)));
    float2 n1 = make_float2(fmaf(s2, b2.y, s1 * b1.y), -fmaf(s2, b2.x, s1 * b1.x));   // -i (s1 b1 + s2 b2)
    float2 n2 = make_float2(fmaf(-s1, b2.y, s2 * b1.y), -fmaf(-s1, b2.x, s2 * b1.x));  // -i (s2 b1 - s1 b2)
    v0 = make_float2(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
    v1 = cadd(m1, n1);
    v4 = csub(m1, n1);
    v2 = cadd(m2, n2);
    v3 = csub(m2, n2);
}

// 6 = 2 x 3 and 10 = 2 x 5 by the prime-factor map (no inner twiddles):
// input n = (N2 n1 + 2 n2) mod N, output k = (N2 k1 + c k2) mod N with c = 4 (N = 6) or 6 (N = 10).
__device__ __forceinline__ void dft6(float2 *v)
{
    float2 a0 = v[0], a1 = v[2], a2 = v[4], b0 = v[3], b1 = v[5], b2 = v[1];
    dft3(a0, a1, a2);
    dft3(b0, b1, b2);
    v[0] = cadd(a0, b0); v[3] = csub(a0, b0);
    v[4] = cadd(a1, b1); v[1] = csub(a1, b1);
    v[2] = cadd(a2, b2); v[5] = csub(a2, b2);
}

__device__ __forceinline__ void dft10(float2 *v)
{
    float2 a0 = v[0], a1 = v[2], a2 = v[4], a3 = v[6], a4 = v[8];
    float2 b0 = v[5], b1 = v[7], b2 = v[9], b3 = v[1], b4 = v[3];
    dft5(a0, a1, a2, a3, a4);
    dft5(b0, b1, b2, b3, b4);
    v[0] = cadd(a0, b0); v[5] = csub(a0, b0);
    v[6] = cadd(a1, b1); v[1] = csub(a1, b1);
    v[2] = cadd(a2, b2); v[7] = csub(a2, b2);
    v[8] = cadd(a3, b3); v[3] = csub(a3, b3);
    v[4] = cadd(a4, b4); v[9] = csub(a4, b4);
}

template <int R> __device__ __forceinline__ void dftR(float2 *v)
{
    if (R == 8) dft8(v);
    else if (R == 6) dft6(v);
    else dft10(v);
}

// The wave = stream transform kernels run FFT_SPB streams per block (one wave each) so that the block's waves share
// one copy of the read-only tables in LDS: every table read sits on a wave's dependent chain, and from LDS it costs
// ~100 cycles instead of a trip to L2.  After the tables are in place (one __syncthreads) the waves never meet again:
// each synchronises only with itself (wave_lds_sync) on its own LDS region.
constexpr int FFT_SPB = 4;
// Per-bin arrays that the band sums read (lane = a segment of <= 8 consecutive bins: neighbouring lanes 8 floats apart, every
// read 8-way bank-conflicted) are kept skewed, bin k at k + k / 8: neighbouring
// lanes then sit 9 floats apart.  Lanes that walk the bins in order (k = lane + 64 u) pay nothing: the skew of their index is a
// per-lane constant.  k_fft_xp 19.6 -> 18.9 us per frame at 4096 streams, 325 -> 321 at 65536 (same box).
__device__ __forceinline__ int bsk(int k) { return k + (k >> 3); }
constexpr int BSK_LEN = 400 + 400 / 8;
// NNN_FFT_LANE_TW=1 (on since round 5; 0 builds the variant without): the second and third pass's twiddles as the lanes use them -- a
// lane's twiddles are constants of the lane, one LDS read each instead of index, wrap and sign (five vector instructions a piece) -- in
// k_synth, whose blocks copy the tables once per group of frames (-6.6 % vector instructions, -1.5 % time: profiles/r4_experiments_ab.txt M,
// profiles/r5_experiments_ab.txt C); the kernels whose blocks copy the tables per stream-frame keep the half circle and copy the part of
// the image before these tables only.  Same products of the same factors: bit-identical to the variant without.
#ifndef NNN_FFT_LANE_TW
#define NNN_FFT_LANE_TW 1
#endif
constexpr int FFT_TW2 = 2 * 5 * 64, FFT_TW3 = 9 * 64;
struct alignas(16) FftLds {
    float2 tw[NFFT];           // exp(-2 pi i k / 960), k < 480; the other half of the circle is the negation
    float frac[BSK_LEN];       // triangular band weights (ref: src/lib.rs:65-82), skewed (bsk)
    unsigned char band[400];   // band of each bin
    short seg[256];            // band-sum segmentation (see band_sums_par): k0[64], count[64], first segment[32] and segments[32] per
                               // interval, [192 + s]: segments behind segment s in its interval
    float dct[NB * NB];        // DCT table (ref: src/lib.rs:118-127): the feature head's two transforms read 44 of its rows per stream-frame
                               // (from global memory they were half of k_fft_xp's vector-memory instructions; same time either way)
    float pad_[2];
#if NNN_FFT_LANE_TW
    float2 tw2[FFT_TW2];       // fft_pass<6, 8>: [it][r - 1][lane]; copied only by the kernels that use them (fft_tables_load)
    float2 tw3[FFT_TW3];       // fft_pass<10, 48>: [r - 1][lane]
#endif
};
static_assert(sizeof(FftLds) % 16 == 0, "copied as 16-byte pieces");
#if NNN_FFT_LANE_TW
constexpr int FFT_TABLES_SHORT = (int)offsetof(FftLds, tw2);
#else
constexpr int FFT_TABLES_SHORT = (int)sizeof(FftLds);
#endif
static_assert(FFT_TABLES_SHORT % 16 == 0, "");
// Fills the block's tables from the image the host built in exactly this layout (Buffers::fft_img): a straight copy of 16-byte
// pieces.  (Building them in the kernel from the plain tables -- skewed index, byte and short conversions, scattered narrow LDS
// stores -- cost k_fft_xp 190 of its 1530 vector instructions per stream-frame.)  Every thread of the block calls it, the caller
// synchronises.
__device__ __forceinline__ void fft_tables_load(FftLds &t, const Buffers &b, bool lane_tw = false)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const uint4 *src = (const uint4 *)b.fft_img;
    uint4 *dst = (uint4 *)&t;
    const int n = (lane_tw ? (int)sizeof(FftLds) : FFT_TABLES_SHORT) / 16;
    // every piece a thread copies is requested before the first one is stored: one trip to the L2 per block instead of one per piece
    // (until round 5 the loop waited for each load before it asked for the next: three to five trips on the block's critical path)
    constexpr int MAXIT = 5;   // blocks of >= 256 threads
    static_assert(sizeof(FftLds) / 16 <= (size_t)MAXIT * 256, "");
    uint4 r[MAXIT];
#pragma unroll
    for (int k = 0; k < MAXIT; k++) {
        const int i = tid + k * nt;
        if (i < n) r[k] = ld_global_u4(src + i);
    }
#pragma unroll
    for (int k = 0; k < MAXIT; k++) {
        const int i = tid + k * nt;
        if (i < n) dst[i] = r[k];
    }
    for (int i = tid + MAXIT * nt; i < n; i += nt) dst[i] = ld_global_u4(src + i);   // blocks of fewer than 256 threads: the rest, piece by piece
}
// the host's side of it
__host__ inline void fft_tables_image(FftLds &t, const float2 *tw960, const float *bin_frac, const int *bin_band, const int *seg, const float *dct)
{
    memset(&t, 0, sizeof(t));
    for (int i = 0; i < NB * NB; i++) t.dct[i] = dct[i];
    for (int i = 0; i < NFFT; i++) t.tw[i] = tw960[i];
#if NNN_FFT_LANE_TW
    auto at = [&](int k) { const float2 w = tw960[k >= NFFT ? k - NFFT : k]; return k >= NFFT ? make_float2(-w.x, -w.y) : w; };   // tw960_at
    for (int it = 0; it < 2; it++)       // fft_pass<6, 8>: butterfly j = lane + 64 it < 80, k = j % 8, twiddle (r k 20) % 960
        for (int r = 1; r < 6; r++)
            for (int l = 0; l < 64; l++) t.tw2[(it * 5 + r - 1) * 64 + l] = at((r * ((l + 64 * it) % 8) * 20) % 960);
    for (int r = 1; r < 10; r++)         // fft_pass<10, 48>: butterfly j = lane < 48, k = j, twiddle (r k 2) % 960
        for (int l = 0; l < 64; l++) t.tw3[(r - 1) * 64 + l] = at((r * (l % 48) * 2) % 960);
#endif
    for (int i = 0; i < 400; i++) {
        t.frac[i + (i >> 3)] = bin_frac[i];
        t.band[i] = (unsigned char)bin_band[i];
    }
    for (int i = 0; i < 192; i++) t.seg[i] = (short)seg[i];
    for (int iv = 0; iv < NB - 1; iv++)
        for (int i = 0; i < seg[160 + iv]; i++) t.seg[192 + seg[128 + iv] + i] = (short)(seg[160 + iv] - 1 - i);
}
__device__ __forceinline__ float2 tw960_at(const float2 *tw, int k)   // k in [0, 960)
{
    const float2 w = tw[k >= NFFT ? k - NFFT : k];
    return k >= NFFT ? make_float2(-w.x, -w.y) : w;
}

// one Stockham pass of the 480-point transform, in place: every lane pulls its butterflies into registers,
// the wave synchronises, then scatters the results (autosort order).  Radix R, NS = product of the radices
// already applied; tw = the LDS half-table of exp(-2 pi i k / 960).  One buffer per transform keeps LDS small
// enough for a full complement of waves per CU.
// The first pass scatters with a stride of 8 elements (64 bytes): 32 lanes on 4 bank pairs, every store 8-way conflicted
// -- as many LDS cycles as all other accesses of the transform together.  Its output (and the second pass's input) is
// therefore skewed, element i at i + i / 8 (stride 9: conflict-free); the buffer holds NFFT_BUF elements for that.
// The second pass's output (the third's input) is padded the same way for the same reason (round 5): its butterflies j and j + 8 of one
// 16-lane store group wrote elements 48 apart -- the same banks -- so every block of 48 elements now starts 8 further on (element i at
// i + 8 (i / 48), 552 elements): the two halves of a store group sit 56 apart, 8 modulo 16, and the third pass reads with stride 56.
constexpr int NFFT_BUF = 560;
constexpr int FFT_P2PAD = 8;
// LT: `tw` is the pass's own per-lane twiddle table ([it][r - 1][lane], FftLds::tw2 / tw3) instead of the half circle
template <int R, int NS, bool SKEW_IN, bool SKEW_OUT, bool LT = false>
__device__ __forceinline__ void fft_pass(float2 *buf, const float2 *tw, int lane)
{
    constexpr int NBF = NFFT / R, IT = (NBF + 63) / 64;
    constexpr bool PAD_OUT = NS == 8 && R == 6, PAD_IN = NS == 48 && R == 10;   // the second pass's padded output = the third's input
    static_assert(!PAD_IN || NBF == 48, "");
    static_assert(!SKEW_IN || NBF % 8 == 0, "skewed reads assume r * NBF is a multiple of 8");
    static_assert(!SKEW_OUT || (NS == 1 && R == 8), "skewed writes are the first pass's");
    float2 v[IT][R];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int j = lane + 64 * it;
        if (j < NBF) {
            const int k = j % NS;
            const int jj = SKEW_IN ? j + (j >> 3) : j;
#pragma unroll
            for (int r = 0; r < R; r++) v[it][r] = buf[jj + (SKEW_IN ? r * NBF + r * NBF / 8 : (PAD_IN ? r * (NBF + FFT_P2PAD) : r * NBF))];
            if (NS > 1) {
                constexpr int step = 960 / (NS * R);
#pragma unroll
                for (int r = 1; r < R; r++) v[it][r] = cmulf(v[it][r], LT ? tw[(it * (R - 1) + r - 1) * 64 + lane] : tw960_at(tw, (r * k * step) % 960));
            }
            dftR<R>(v[it]);
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int j = lane + 64 * it;
        if (j < NBF) {
            const int base = SKEW_OUT ? 9 * j : (j / NS) * (NS * R + (PAD_OUT ? FFT_P2PAD : 0)) + (j % NS);   // skewed: element 8 j + r at 9 j + r
#pragma unroll
            for (int r = 0; r < R; r++) buf[base + r * NS] = v[it][r];
        }
    }
    wave_lds_sync();
}

// forward 480-point FFT in place (natural order in, natural order out); buf has room for NFFT_BUF elements
__device__ __forceinline__ void fft480(float2 *buf, const float2 *tw, int lane)
{
    fft_pass<8, 1, false, true>(buf, tw, lane);
    fft_pass<6, 8, true, false>(buf, tw, lane);
    fft_pass<10, 48, false, false>(buf, tw, lane);
}
// The same with the input handed over in registers in the first pass's own order -- lane j < 60 holds elements j + 60 r, r < 8
// (lanes 60..63 hold anything) -- instead of staged in buf: both callers can produce their input in that order, which saves the
// staging store, the first pass's reads and a synchronisation per transform.  Every earlier reader of buf must be done.
// RL (the fused back end, a wave of 128 registers that also holds two spectra): the lane index is laundered between the passes, so that
// each pass forms its LDS addresses and twiddle indices where it starts instead of all of them up front (see launder_v)
template <bool RL = false, bool LT = false>
__device__ __forceinline__ void fft480_regs(float2 (&v)[8], float2 *buf, const float2 *tw, int lane)
{
    dft8(v);
    if (lane < NFFT / 8) {
#pragma unroll
        for (int r = 0; r < 8; r++) buf[9 * lane + r] = v[r];   // skewed, as fft_pass<8, 1, false, true> leaves it
    }
    wave_lds_sync();
    if (RL) lane = launder_v(lane);
#if NNN_FFT_LANE_TW
    if (LT) {   // (tw = FftLds::tw of a block that copied the whole image)
        const float2 *tw2 = (const float2 *)((const char *)tw + (offsetof(FftLds, tw2) - offsetof(FftLds, tw)));
        fft_pass<6, 8, true, false, true>(buf, tw2, lane);
        if (RL) lane = launder_v(lane);
        fft_pass<10, 48, false, false, true>(buf, tw2 + FFT_TW2, lane);
        return;
    }
#endif
    fft_pass<6, 8, true, false>(buf, tw, lane);
    if (RL) lane = launder_v(lane);
    fft_pass<10, 48, false, false>(buf, tw, lane);
}
constexpr int FFT_P1 = NFFT / 8;   // butterflies (= lanes at work) of the first pass

// band sums in the reference's accumulation order (ref: src/lib.rs:65-82): out[b] first receives
// the frac-weighted terms of interval b-1, then the (1-frac)-weighted terms of interval b.
__device__ __forceinline__ float band_sum(const float *v, int bnd, const float *bin_frac)
{
    const int e_lo = bnd >= 1 ? kEband[bnd - 1] : 0, e_mid = kEband[bnd], e_hi = bnd < NB - 1 ? kEband[bnd + 1] : 0;
    float acc = 0.0f;
    if (bnd >= 1)
        for (int k = 4 * e_lo; k < 4 * e_mid; k++) acc += bin_frac[k] * v[k];
    if (bnd < NB - 1)
        for (int k = 4 * e_mid; k < 4 * e_hi; k++) acc += (1.0f - bin_frac[k]) * v[k];
    if (bnd == 0 || bnd == NB - 1) acc *= 2.0f;
    return acc;
}

// Parallel band sums for tolerance-only quantities (every band energy is downstream of an FFT): the 21 band intervals are cut
// into 54 segments of 4 or 8 bins (table t.seg; an 8-bin segment starts on a multiple of 8, so a segment's bins are contiguous
// in the skewed arrays too).  Lane = segment forms the two triangularly weighted partial sums of its bins for up to NQ
// quantities -- eight unrolled steps, the shorter segments masked; the segments of an interval sit on consecutive lanes and are
// summed across lanes by a segmented suffix sum (four shuffle rounds: intervals have at most 11 segments); lane = band then takes
// the frac-weighted total of the interval below it and the (1 - frac)-weighted total of its own (ref: src/lib.rs:65-82).
// (Until round 3 the partial sums went through LDS and lane = band looped over up to 11 of them, twice: with the per-bin loop
// that was a third of k_fft_xp's vector instructions, issued for 22 or 54 of 64 lanes.)  Every lane of the wave must call it.
template <int NQ>
__device__ __forceinline__ void band_sums_par(const FftLds &t, const float *const (&v)[NQ], float (&out)[NQ], int lane)
{
    const short *seg = t.seg;
    // lane = slot: k0, bin count (0: an idle slot) and the number of the interval's segments behind this one.  The slots of an interval
    // never straddle a row of 16 lanes (the host leaves slots idle for that, 59 of 64 in use), so the suffix sum's four rounds are DPP
    // row moves (round 5: they were wave shuffles -- an index computation and an LDS-crossbar trip each).
    const int k0 = seg[lane], cnt = seg[64 + lane], rem = seg[192 + lane];
    const int ks0 = bsk(k0);
    float pa[NQ], pb[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { pa[q] = 0.0f; pb[q] = 0.0f; }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const float fr = t.frac[ks0 + u], om = 1.0f - fr;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            float x = v[q][ks0 + u];
            x = u < cnt ? x : 0.0f;
            pa[q] = fmaf(om, x, pa[q]);
            pb[q] = fmaf(fr, x, pb[q]);
        }
    }
    // segmented suffix sum: afterwards the first segment of every interval holds the interval's totals
#define NNN_SUFFIX_ROUND(D)                                                          \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) {                                 \
        const float ta = dpp_row_down<D>(pa[q]), tb = dpp_row_down<D>(pb[q]);        \
        pa[q] += rem >= D ? ta : 0.0f;                                               \
        pb[q] += rem >= D ? tb : 0.0f;                                               \
    }
    NNN_SUFFIX_ROUND(1) NNN_SUFFIX_ROUND(2) NNN_SUFFIX_ROUND(4) NNN_SUFFIX_ROUND(8)
#undef NNN_SUFFIX_ROUND
    // lane = band: interval `lane - 1` from below, interval `lane` above
    const int bnd = lane < NB ? lane : 0;
    const int lo = seg[128 + (bnd >= 1 ? bnd - 1 : 0)], hi = seg[128 + (bnd < NB - 1 ? bnd : 0)];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const float fb = wave_read(pb[q], lo), fa = wave_read(pa[q], hi);
        float o = (bnd >= 1 ? fb : 0.0f) + (bnd < NB - 1 ? fa : 0.0f);
        if (bnd == 0 || bnd == NB - 1) o *= 2.0f;
        out[q] = lane < NB ? o : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// K8  fft_xp: transform_input (window, real FFT, normalise, band energy) at lag 0 and at lag = pitch, one after the
//     other in the same wave: X never leaves the registers between its own transform and the band correlation with P
//     (ref: src/features.rs:119-135, 281-298, src/lib.rs:65-82, 150-155).  One wave per stream, one launch per frame
//     group.  Ends with the head of the feature stage (ref: src/features.rs:135-170).  WITH_P = false: the lag-0
//     transform and its band energies only (the clean / noise states of the training rows).
// ---------------------------------------------------------------------------------------------
// one DCT output (ref: src/lib.rs:139-148): sequential sum over the 22 inputs, scaled in double
__device__ __forceinline__ float dct_out(const float *x, const float *dct, int i)
{
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < NB; j++) sum += x[j] * dct[j * NB + i];
    return (float)((double)sum * 0.30151134457776363 /* sqrt(2/22) */);
}

struct __attribute__((packed, aligned(4))) SamplePair { float x, y; };   // two consecutive samples: one 8-byte load at 4-byte alignment
// windowed 960 samples ending `lag` samples before the newest one -> Z (packed as 480 complex), transform in place,
// spectrum bins into Y (lane owns bins lane + 64 u), scaled by wnorm
// the 960 samples ending `lag` samples before the newest one, as the sample pairs n = j + 60 r of the transform's first pass
__device__ __forceinline__ void window_load(const float *h, int ring, int rb, int lag, int lane, SamplePair (&sm)[8])
{
    int start = rb + (HIST - WINDOW) - lag;   // in (0, 2 ring)
    if (start >= ring) start -= ring;
    const int j = lane < FFT_P1 ? lane : FFT_P1 - 1;   // (lanes 60..63 shadow lane 59 and store nothing)
    // Pair r sits 8 FFT_P1 r bytes behind pair 0, less the ring's length when that is past the ring's end: of x and x - 4 ring taken as
    // unsigned numbers the smaller is the one in range.  Three 32-bit instructions per pair and an offset the load adds to the
    // stream's (wave-uniform) base itself; as signed indices with a compare and a 64-bit address each, it was seven.
    const unsigned x0 = 4u * (unsigned)(start + 2 * j), ring4 = 4u * (unsigned)ring;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const unsigned x = x0 + (unsigned)(8 * FFT_P1 * r), y = x - ring4;
        sm[r] = *(const SamplePair *)((const char *)h + (x < y ? x : y));   // (the pair that starts on the ring's last sample reads the copy of sample 0 kept behind it)
    }
}
template <bool RL = false>
__device__ __forceinline__ void window_rfft(const Buffers &b, const SamplePair (&sm)[8], const float2 (&w)[8], const FftLds &t,
                                            float2 *Z, float2 (&Y)[8], int lane, bool first)
{
    // sample pairs n = j + 60 r straight into the first pass's registers (w holds the window in the same order)
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = make_float2(sm[r].x * w[r].x, sm[r].y * w[r].y);
    if (first) __syncthreads();   // tables in place; from here on every wave is on its own
    fft480_regs<RL>(v, Z, t.tw, lane);
    if (RL) lane = launder_v(lane);
    // Bins k and 480 - k come from the same two transform outputs (E[480 - k] = conj E[k], O[480 - k] = conj O[k], the twiddle
    // of 480 - k is -conj of k's): a lane takes them as a pair -- one read of each output, one twiddle, one complex product for
    // both -- and owns bins rfft_slot_bin(lane, u): k = lane + 64 u in slots u < 4 (k <= 240), 480 - k in slot 4 + u (k < 240).
    // (the split step's factor 1/2 rides on the analysis window, Buffers::window_a -- an exact scaling.  The normalisation 1 / 480
    // stays here: folded into the window as well it rounds every coefficient a second time, a window that is no longer the
    // reference's bit for bit, and on a signal with a huge slowly decaying component -- the high-passed DC step of the edge-case
    // set -- the leakage of that component differs enough to move the gains by three times the reference's own f32 / f64 spread.)
    const float wn = b.wnorm;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int k = lane + 64 * u;
        if (k <= NFFT / 2) {
            const float2 zk = Z[k], zn = Z[k ? NFFT - k : 0];
            const float2 e = make_float2(zk.x + zn.x, zk.y - zn.y);
            const float2 o = make_float2(zk.y + zn.y, -(zk.x - zn.x));   // (zk - conj zn) / i
            const float2 wo = cmulf(o, t.tw[k]);
            Y[u] = make_float2((e.x + wo.x) * wn, (e.y + wo.y) * wn);
            Y[4 + u] = make_float2((e.x - wo.x) * wn, -(e.y - wo.y) * wn);   // conj(E - W O)
        }
    }
    wave_lds_sync();   // the transform has been read: its buffer now takes the per-bin products for the band sums
}
// bin of slot u of a lane's spectrum registers (see window_rfft); -1: an empty slot
__device__ __forceinline__ int rfft_slot_bin(int lane, int u)
{
    const int k = lane + 64 * (u & 3);
    if (u < 4) return k <= NFFT / 2 ? k : -1;
    return k < NFFT / 2 ? NFFT - k : -1;
}

// What the fused back end (k_back, nnn_back.hip) keeps of a frame's transforms instead of sending it through HBM: both spectra in the
// registers of the stream's wave (slot order of window_rfft), the three per-band quantities the pitch filter needs on lanes 0..21, the
// silence flag; the feature head's 28 outputs go to `cnw` (LDS) for the feature stage that follows on the same wave.
struct XpKeep {
    float2 X[8], P[8];
    float ex, ep, xn;     // lane < NB: band energies of X and P, normalised correlation
    int silent;           // wave-uniform
    int sl;               // in: the stream's row in its tile
    float *cnw;           // in: LDS staging of the frame's cepstrum (22) + pitch-correlation DCT (6)
    int *flag;            // in: one LDS word of the wave (the silence flag travels through it)
};
// a spectrum in the wave's registers (slot order, see window_rfft) <-> its row in memory: pair (slot u, slot 4 + u) = (bin k, bin 480 - k)
// of lane j's k = j + 64 u as one 16-byte access at float4 index 64 u + j (FSTR in nnn_layout.h)
__device__ __forceinline__ void spectrum_store(float2 *row, const float2 (&S)[8], int lane)
{
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (lane + 64 * u <= NFFT / 2) ((float4 *)row)[64 * u + lane] = make_float4(S[u].x, S[u].y, S[4 + u].x, S[4 + u].y);
}
__device__ __forceinline__ void spectrum_load(const float2 *row, float2 (&S)[8], int lane)
{
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const float4 v = lane + 64 * u <= NFFT / 2 ? ((const float4 *)row)[64 * u + lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        S[u] = make_float2(v.x, v.y);
        S[4 + u] = make_float2(v.z, v.w);
    }
}

// The pitch-lagged spectrum P the same way, except that the partners of slot 0 -- bins 417 .. 480, which the pitch filter never reads (its
// gain is zero from bin 400 up, ref: src/lib.rs:84-97) -- stay out of the row: 64 lone bins (8 bytes each) first, then the pairs of slots
// 1 .. 3; with the parity taps on the 64 partners follow behind (P_TAIL).  13 % fewer bytes for P than whole pairs.
constexpr int P_PAIRS0 = 64, P_TAIL = 64 + 2 * 177;   // float2 index of the first pair (slot 1) and of the taps-only partners of slot 0
__device__ __forceinline__ void spectrum_store_p(float2 *row, const float2 (&S)[8], int lane, bool taps)
{
    row[lane] = S[0];
    if (taps) row[P_TAIL + lane] = S[4];
#pragma unroll
    for (int u = 1; u < 4; u++)
        if (lane + 64 * u <= NFFT / 2) ((float4 *)(row + P_PAIRS0))[64 * (u - 1) + lane] = make_float4(S[u].x, S[u].y, S[4 + u].x, S[4 + u].y);
}
__device__ __forceinline__ void spectrum_load_p(const float2 *row, float2 (&S)[8], int lane)
{
    S[0] = row[lane];
    S[4] = make_float2(0.0f, 0.0f);   // (bins 417 .. 480: never read)
#pragma unroll
    for (int u = 1; u < 4; u++) {
        const float4 v = lane + 64 * u <= NFFT / 2 ? ((const float4 *)(row + P_PAIRS0))[64 * (u - 1) + lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        S[u] = make_float2(v.x, v.y);
        S[4 + u] = make_float2(v.z, v.w);
    }
}

#ifndef NNN_FH_STRIDE
#define NNN_FH_STRIDE 24
#endif
constexpr int FH_STRIDE = NNN_FH_STRIDE;   // floats between the feature head's three staged band arrays (22 used of each)
static_assert(FH_STRIDE >= NB, "");
// XR (fused, one-frame calls): X and its band energies are in memory already -- computed by rider blocks of k_pitch's launch, which needs
// nothing of what the pitch analysis finds (xt_rider) -- and are fetched instead of computed.
template <bool WITH_P, bool FUSED = false, bool XR = false>
__device__ __forceinline__ void transform_inputs(const Buffers &b, const StepParams *sp, int tile_in, int sub, FftLds &t, float2 *Z, float *part,
                                                 XpKeep *keep = nullptr)
{
    int tile = tile_in;
    // (fused: the lane index is laundered again between the stages -- see launder_v -- so that the addresses of a later stage are formed
    // where it starts instead of at the top of the function, where the first build kept dozens of them alive in spilled registers)
    int lane = threadIdx.x & 63;
    int sl = FUSED ? keep->sl : sub * FFT_SPB + (int)(threadIdx.x >> 6), s = tile * TILE + sl;
#define NNN_FUSED_RELAUNDER() do { if (FUSED) { lane = launder_v(lane); tile = launder_s(tile); sl = launder_s(sl); s = tile * TILE + sl; } } while (0)
    const int ring = ring_len(b.nslot), rb = ring_base(sp->slot, b.nslot);
    float2 w[8];   // the window at sample pairs j + 60 r: the order of the transforms' first pass (window_rfft)
#pragma unroll
    for (int r = 0; r < 8; r++) w[r] = ((const float2 *)b.window_a)[(lane < FFT_P1 ? lane : FFT_P1 - 1) + FFT_P1 * r];
    const int lag = WITH_P ? NNN_TI(b.pitch, 1, tile, sl)[0] : 0;
    if (!FUSED) fft_tables_load(t, b);   // (the fused kernel loads them once per launch)
    const float *h = b.hist + (size_t)__builtin_amdgcn_readfirstlane(s) * hist_stride(b.nslot);   // (wave = stream: a scalar base)
    // both windows' samples are requested now: the second transform's used to be requested when it started, a trip to memory on
    // the wave's critical path per stream-frame (these kernels move enough bytes for that to show)
    SamplePair sx[8], spw[8];
    if (!XR) window_load(h, ring, rb, 0, lane, sx);
#ifndef NNN_FFT_LATE_P   // (A/B knob: the second window requested where its transform starts, as before)
    if (WITH_P && !FUSED) window_load(h, ring, rb, lag, lane, spw);   // (fused: sixteen registers it has not got; requested after the first transform)
#endif
    float2 X[8];
    float2 *dx = b.X + (size_t)s * FSTR;
    if (XR) spectrum_load(dx, X, lane);
    else window_rfft<FUSED>(b, sx, w, t, Z, X, lane, !FUSED);
    // NNN_PROBE_XP (developer probe, wrong audio, timing only): the spectra are not stored here and k_synth reads them from a
    // region small enough to stay in the XCD's L2 -- an upper bound on what keeping X and P on chip between the transforms
    // and the synthesis (a fused back end) could gain from the removed HBM round trip.
#ifndef NNN_PROBE_XP
    if (!XR && (!FUSED || b.taps)) spectrum_store(dx, X, lane);   // (fused: the spectra stay in registers; memory sees them for the parity taps only)
#endif
    NNN_FUSED_RELAUNDER();
    float *vv = (float *)Z, *vc = vv + BSK_LEN;   // per-bin quantities of the band sums, skewed (bsk)
    float exv;
    if (XR) {
        exv = lane < NB ? NNN_TI(b.ex, NB, tile, sl)[(size_t)lane * TILE] : 0.0f;
    } else {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = rfft_slot_bin(lane, u);
            if (k >= 0 && k < 400) vv[bsk(k)] = fmaf(X[u].y, X[u].y, X[u].x * X[u].x);
        }
        wave_lds_sync();
        const float *const v[1] = {vv};
        float o[1];
        band_sums_par<1>(t, v, o, lane);
        exv = o[0];
        if (lane < NB) NNN_TI(b.ex, NB, tile, sl)[(size_t)lane * TILE] = exv;
    }
    if (!WITH_P) return;
    wave_lds_sync();
    NNN_FUSED_RELAUNDER();
    float2 Y[8];
#ifdef NNN_FFT_LATE_P
    window_load(h, ring, rb, lag, lane, spw);
#else
    if (FUSED) window_load(b.hist + (size_t)__builtin_amdgcn_readfirstlane(s) * hist_stride(b.nslot), ring, rb, lag, lane, spw);
#endif
    if (FUSED) {   // (the window again, from the L2: sixteen registers less across the first transform and the band sums of a wave that has 128)
#pragma unroll
        for (int r = 0; r < 8; r++) w[r] = ((const float2 *)b.window_a)[(lane < FFT_P1 ? lane : FFT_P1 - 1) + FFT_P1 * r];
    }
    window_rfft<FUSED>(b, spw, w, t, Z, Y, lane, false);
    float2 *dp = b.P + (size_t)s * FSTR;
#ifndef NNN_PROBE_XP
    if (!FUSED || b.taps) spectrum_store_p(dp, Y, lane, b.taps != 0);
#endif
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int k = rfft_slot_bin(lane, u);
        if (k >= 0 && k < 400) {
            vv[bsk(k)] = fmaf(Y[u].y, Y[u].y, Y[u].x * Y[u].x);
            vc[bsk(k)] = fmaf(X[u].y, Y[u].y, X[u].x * Y[u].x);
        }
    }
    wave_lds_sync();
    NNN_FUSED_RELAUNDER();
    const float *const v[2] = {vv, vc};
    float o[2];
    band_sums_par<2>(t, v, o, lane);
    // Head of the feature stage (ref: src/features.rs:135-170), here because everything it needs is at hand and this
    // launch covers a whole frame group: the correlation normalised by the band energies, the floored log energies,
    // the silence test, and the two DCTs -- lane = band.  Same operations in the same order as when one lane did it all.
    wave_lds_sync();
    NNN_FUSED_RELAUNDER();
    float *xc = part, *ly = part + FH_STRIDE, *exl = part + 2 * FH_STRIDE;
    float lyv = -2.0f, xnv = 0.0f;
    if (lane < NB) {
        const float xn = o[1] / sqrtf(0.001f + exv * o[0]);
        xnv = xn;
        NNN_TI(b.ep, NB, tile, sl)[(size_t)lane * TILE] = o[0];
        NNN_TI(b.exp_, NB, tile, sl)[(size_t)lane * TILE] = xn;
        xc[lane] = xn;
        exl[lane] = exv;
        lyv = log10f(1e-2f + exv);
    }
    // The floors of the log energies (ref: src/features.rs:150-158) are a 22-step recurrence -- l_i = max(ly_i, max_{j<i} l_j - 7,
    // follow_i - 1.5) with follow decaying by 1.5 per band -- that one lane used to walk while the wave waited.  Unrolled it is
    // l_i = max over j <= i of ly_j - c(i - j) with c(0) = 0, c(d) = min(7, 1.5 d) (c is subadditive, so floors of floors add
    // nothing), plus the two start values; and since c(d) = 7 from d = 5 on: four neighbours and a prefix maximum five bands
    // back, lane = band, through shuffles.  Same values up to the rounding of the decay (one multiply instead of repeated
    // subtraction); a NaN energy is ignored by the max exactly as in the recurrence.  (Every lane takes part in the shuffles.)
    float lfl;
    {
        float pm = lyv;   // inclusive prefix maximum over the bands below (lanes past the bands hold -2: never above a log energy)
#pragma unroll
        for (int d = 1; d < 32; d *= 2) {
            const float tt = wave_read(pm, lane - d);   // (lanes below d read some other lane and keep their own value)
            pm = lane >= d ? fmaxf(pm, tt) : pm;
        }
        float m = fmaxf(fmaxf(lyv, -2.0f - 7.0f), -2.0f - 1.5f * (float)(lane + 1));
#pragma unroll
        for (int d = 1; d <= 4; d++) {
            const float tt = wave_read(lyv, lane - d) - 1.5f * (float)d;
            m = lane >= d ? fmaxf(m, tt) : m;
        }
        const float t5 = wave_read(pm, lane - 5) - 7.0f;
        lfl = lane >= 5 ? fmaxf(m, t5) : m;
    }
    if (lane < NB) ly[lane] = lfl;
    wave_lds_sync();
    if (lane == 0) {   // the silence test on the band energies summed in band order (ref: src/features.rs:160)
        float e = 0.0f;
#pragma unroll
        for (int i = 0; i < NB; i++) e += exl[i];
        NNN_TI(b.silence, 1, tile, sl)[0] = e < 0.04f ? 1 : 0;
        if (FUSED) keep->flag[0] = e < 0.04f ? 1 : 0;
    }
    // the two DCTs side by side: lanes 0..21 the cepstrum of the floored log energies, lanes 32..37 the first six coefficients
    // of the pitch correlation (ref: src/features.rs:141-147, 167-169; src/lib.rs:139-148)
    {
        const bool second = lane >= 32;
        const int i = second ? lane - 32 : lane;
        if (i < (second ? 6 : NB)) {
            float *cn = NNN_TI(b.cn, 28, tile, sl);
            float c = dct_out(second ? xc : ly, t.dct, i);
            if (second) c -= i == 0 ? 1.3f : (i == 1 ? 0.9f : 0.0f);
            else c -= i == 0 ? 12.0f : (i == 1 ? 4.0f : 0.0f);
            if (FUSED) keep->cnw[(second ? NB : 0) + i] = c;   // (the feature stage follows on this wave)
            else cn[(size_t)((second ? NB : 0) + i) * TILE] = c;
        }
    }
    if (FUSED) {
#pragma unroll
        for (int u = 0; u < 8; u++) { keep->X[u] = X[u]; keep->P[u] = Y[u]; }
        keep->ex = exv;
        keep->ep = o[0];
        keep->xn = xnv;
        wave_lds_sync();
        keep->silent = keep->flag[0];
    }
#undef NNN_FUSED_RELAUNDER
}

// five waves per SIMD (96 registers, no spill) and, with LDS kept to 27.6 KB per block, five blocks per CU: k_fft_xp -3.8 % against four
// (profiles/r5_experiments_ab.txt Q)
#ifndef NNN_FFT_MINWAVES
#define NNN_FFT_MINWAVES 5
#endif
// Block index -> (frame, tile, four streams of the tile) for the g frames of a group.  Batches of a multiple of 8 tiles: the blocks an
// XCD receives (i mod 8, in index order) are the g frames of one quartet of streams, then the next quartet's, tile t on XCD t mod 8:
// consecutive frames of a stream share three quarters of the history samples their two windows read, and blocks that run
// side by side on one XCD fetch them into its L2 once (frame-major order -- all streams of frame 0, then frame 1 ... -- puts
// 250 MB of other streams' samples between two uses of a line).
__device__ __forceinline__ void fft_block(const Buffers &b, int g, int &frame, int &tile, int &sub)
{
    xcd_tile_block_units((int)blockIdx.x, b.NT, TILE / FFT_SPB, g, frame, tile, sub);
}
__global__ void __launch_bounds__(64 * FFT_SPB, NNN_FFT_MINWAVES) k_fft_xp(Buffers b, const StepParams *sp, int g)
{
    // (the part of the tables this kernel copies and reads: with the staging below kept to what it holds, 27.6 KB -- a fifth of the CU's LDS)
    __shared__ __attribute__((aligned(16))) char tbuf[FFT_TABLES_SHORT];
    FftLds &t = *(FftLds *)tbuf;
    __shared__ float2 Z[FFT_SPB][NFFT_BUF];
    __shared__ float part[FFT_SPB][3 * FH_STRIDE];   // the feature head's staging: correlation, log energies, band energies
    int frame, tile, sub;
    fft_block(b, g, frame, tile, sub);
    if (tile * TILE + sub * FFT_SPB >= b.S) return;   // (a block whose streams are all padding -- the last tile of a batch that is not a multiple of 64 -- has nothing to do)
    b = frame_view(b, frame);
    const int wave = threadIdx.x >> 6;
    transform_inputs<true>(b, sp + frame, tile, sub, t, Z[wave], part[wave]);
}

// Rider blocks of a one-frame k_pitch launch (blocks `riders` ..): the frame's lag-0 transform X and its band energies, eight streams
// per block (wave = stream) -- the part of the back end that needs nothing from the pitch analysis, done while the pitch blocks, one per
// compute unit and bound by their own latency chains, leave most issue slots free.  The fused back end then fetches X instead of
// computing it (k_back<.., XR>).  The code is k_fft_x's; `lds` is the pitch kernel's own LDS block, which a rider block has to itself.
struct XtLds { FftLds t; float2 Z[8][NFFT_BUF]; float part[8][4]; };
__device__ __forceinline__ void xt_rider(const Buffers &b, const StepParams *sp, int rb, void *lds)
{
    static_assert(sizeof(XtLds) <= sizeof(PkLds) && PK_T == 512 && FFT_SPB == 4, "");
    XtLds &x = *(XtLds *)lds;
    const int wave = threadIdx.x >> 6;
    if ((rb >> 3) * TILE + 8 * (rb & 7) >= b.S) return;
    transform_inputs<false>(b, sp, rb >> 3, 2 * (rb & 7), x.t, x.Z[wave], x.part[wave]);   // rows 8 (rb % 8) + wave of tile rb / 8
}

// lag-0 transform and band energies only (training rows: clean and noise states)
__global__ void __launch_bounds__(64 * FFT_SPB) k_fft_x(Buffers b, const StepParams *sp, int g)
{
    __shared__ FftLds t;
    __shared__ float2 Z[FFT_SPB][NFFT_BUF];
    __shared__ float part[FFT_SPB][4];   // (the feature head's staging: unused without the second transform)
    int frame, tile, sub;
    fft_block(b, g, frame, tile, sub);
    if (tile * TILE + sub * FFT_SPB >= b.S) return;
    b = frame_view(b, frame);
    const int wave = threadIdx.x >> 6;
    transform_inputs<false>(b, sp + frame, tile, sub, t, Z[wave], part[wave]);
}

#pragma clang fp contract(off)
// ---------------------------------------------------------------------------------------------
// K9  features: the 42 RNN inputs from band energies, pitch and the cepstral history.
//     ref: src/features.rs:135-219, src/lib.rs:139-148.  lane = stream; runs on wave 0 of the RNN kernel.
// ---------------------------------------------------------------------------------------------

// The feature stage runs inside the RNN kernel, lane = stream, in three steps so that only the truly serial part
// sits on one wave: (1) wave 0: band energies -> correlation DCT, log-energy DCT (the new cepstrum), silence
// flag, while waves 1..7 stage the 8 x 22 cepstral ring in LDS; (2) wave 0: ring update and delta features;
// (3) all waves: the 28 pairwise cepstral distances of the spectral-variability feature; wave 0 finishes.
struct FeatHead {
    float fpitch;
    bool silent;
};

// The head of the feature stage (correlation normalisation, log energies, silence test, both DCTs) is done by
// k_fft_p; this picks its results up: the new cepstrum (rows 0..21) and the pitch-correlation DCT (rows 22..27) go to
// the block's LDS staging (`cn`).  `lane` = the stream's row in its 64-stream tile (global layouts), `ll` = its column
// in the staging, `ls` = the staging's row stride (columns per block).
__device__ __forceinline__ void features_load(const Buffers &b, int tile, int lane, int ll, int ls, FeatHead &h, float *cn)
{
    const float *cg = NNN_TI(b.cn, 28, tile, lane);
    float v[28];
#pragma unroll
    for (int i = 0; i < 28; i++) v[i] = cg[(size_t)i * TILE];
    const int pitch = NNN_TI(b.pitch, 1, tile, lane)[0];
    h.silent = NNN_TI(b.silence, 1, tile, lane)[0] != 0;
    h.fpitch = 0.01f * ((float)pitch - 300.0f);
#pragma unroll
    for (int i = 0; i < 28; i++) cn[i * ls + ll] = v[i];
}

// ring update + delta features (wave 0, after the ring has been staged in crs)
__device__ __forceinline__ void features_deltas(const Buffers &b, int tile, int lane, int ll, int ls, const FeatHead &h, float *crs,
                                                const float *cn, float (&fr)[NFEAT])
{
    if (h.silent) {   // "if there's no audio, avoid messing up the state" (ref: src/features.rs:160-166)
#pragma unroll
        for (int i = 0; i < NFEAT; i++) fr[i] = 0.0f;
        return;
    }
    int *midp = NNN_TI(b.mem_id, 1, tile, lane);
    float *cm = NNN_TI(b.ceps_mem, CEPS_MEM * NB, tile, lane);
    int mem_id = midp[0];
    const int c0 = mem_id, c1 = mem_id < 1 ? CEPS_MEM + mem_id - 1 : mem_id - 1;
    const int c2 = mem_id < 2 ? CEPS_MEM + mem_id - 2 : mem_id - 2;
    float c[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) {
        c[k] = cn[k * ls + ll];
        cm[(size_t)(c0 * NB + k) * TILE] = c[k];
        crs[(c0 * NB + k) * ls + ll] = c[k];
    }
    mem_id += 1;
    if (mem_id == CEPS_MEM) mem_id = 0;
    midp[0] = mem_id;
#pragma unroll
    for (int i = 0; i < NB; i++) fr[i] = c[i];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float v1 = crs[(c1 * NB + i) * ls + ll], v2 = crs[(c2 * NB + i) * ls + ll];
        const float v0 = c[i];
        fr[i] = v0 + v1 + v2;
        fr[NB + i] = v0 - v2;
        fr[NB + 6 + i] = v0 - 2.0f * v1 + v2;
        fr[NB + 12 + i] = cn[(NB + i) * ls + ll];
    }
    fr[40] = h.fpitch;
    fr[41] = 0.0f;
}

// pair p of the 28 unordered pairs (i < j) of ring rows
__device__ __forceinline__ void pair_of(int p, int &i, int &j)
{
    i = 0;
    int rem = p;
#pragma unroll
    for (int u = 0; u < 7; u++) {
        const int cnt = 7 - u;
        if (i == u && rem >= cnt) { rem -= cnt; i = u + 1; }
    }
    j = i + 1 + rem;
}

// squared cepstral distance of one pair, summed over the 22 bands in order (ref: src/features.rs:203-208)
__device__ __forceinline__ float pair_dist(const float *crs, int p, int lane, int ls)
{
    int i, j;
    pair_of(p, i, j);
    float dist = 0.0f;
#pragma unroll
    for (int k = 0; k < NB; k++) {
        float d = crs[(i * NB + k) * ls + lane] - crs[(j * NB + k) * ls + lane];
        dist += d * d;
    }
    return dist;
}

// spectral variability = mean_i min_{j != i} dist(i, j) - 2.1 from the 28 staged pair distances
__device__ __forceinline__ float spectral_variability(const float *dists, int lane, int ls)
{
    float mind[CEPS_MEM];
#pragma unroll
    for (int i = 0; i < CEPS_MEM; i++) mind[i] = 1e15f;
    int p = 0;
#pragma unroll
    for (int i = 0; i < CEPS_MEM; i++)
#pragma unroll
        for (int j = i + 1; j < CEPS_MEM; j++) {
            const float d = dists[p * ls + lane];
            mind[i] = fminf(mind[i], d);
            mind[j] = fminf(mind[j], d);
            p++;
        }
    float sv = 0.0f;
#pragma unroll
    for (int i = 0; i < CEPS_MEM; i++) sv += mind[i];
    return sv / (float)CEPS_MEM - 2.1f;
}

// ---------------------------------------------------------------------------------------------
// K9b features, stand-alone: the same feature stage as the RNN kernel's prologue for callers that stop at the 42
//     features (training-data generation, ref: src/training.rs:113-160).  One 64-stream tile per block, 8 waves.
// ---------------------------------------------------------------------------------------------
constexpr int FEAT_WAVES = 8;
__global__ void __launch_bounds__(64 * FEAT_WAVES) k_features(Buffers b0, int g)
{
    __shared__ float crs[CEPS_MEM * NB * TILE];   // staged cepstral ring
    __shared__ float dists[28 * TILE];            // new cepstrum / correlation DCT, then the pair distances
    const int wave = (int)(threadIdx.x >> 6), lane = threadIdx.x & 63, tile = blockIdx.x;
    // the frames of a group one after the other (the cepstral ring is the tile's own state, carried through memory)
#pragma unroll 1
    for (int fr_i = 0; fr_i < g; fr_i++) {
    const Buffers b = frame_view(b0, fr_i);
    if (fr_i) __syncthreads();
    FeatHead fh;
    float fr[NFEAT];
    if (wave == 0) {
        features_load(b, tile, lane, lane, TILE, fh, dists);
    } else {
        const float *cm = NNN_TI(b.ceps_mem, CEPS_MEM * NB, tile, lane);
        for (int r = wave - 1; r < CEPS_MEM * NB; r += FEAT_WAVES - 1) crs[r * TILE + lane] = cm[(size_t)r * TILE];
    }
    __syncthreads();
    if (wave == 0) features_deltas(b, tile, lane, lane, TILE, fh, crs, dists, fr);
    __syncthreads();
    for (int p = wave; p < 28; p += FEAT_WAVES) dists[p * TILE + lane] = pair_dist(crs, p, lane, TILE);
    __syncthreads();
    if (wave == 0) {
        if (!fh.silent) fr[41] = spectral_variability(dists, lane, TILE);
        float *f = NNN_TI(b.feat, NFEAT, tile, lane);
#pragma unroll
        for (int k = 0; k < NFEAT; k++) f[(size_t)k * TILE] = fr[k];
    }
    }
}

// One training row per stream (ref: src/training.rs:136-158): the combined signal's 42 features, 22 ideal band gains
// sqrt((Ex_clean + 1e-3) / (Ex_combined + 1e-3)) capped at 1 (-1 where both energies are below 5e-2, and from the
// band cutoff up; cutoff 0 on silent frames), 22 noise levels log10(Ex_noise + 1e-2), and the caller's VAD label.
constexpr int TRAIN_COLS = NFEAT + 2 * NB + 1;
__global__ void __launch_bounds__(64) k_train_rows(Buffers comb, Buffers clean, Buffers noise, const int *cutoff, const float *vad,
                                                   float *rows)
{
    __shared__ float row[TILE][TRAIN_COLS + 1];
    // block index = frame * tiles + tile: the frames of a group in one launch, frame f's labels and rows f * S further on
    const int NTl = comb.S_pad / TILE, frame = (int)blockIdx.x / NTl;
    const int lane = threadIdx.x, tile = (int)blockIdx.x - frame * NTl, s = tile * TILE + lane;
    comb = frame_view(comb, frame);
    clean = frame_view(clean, frame);
    noise = frame_view(noise, frame);
    cutoff += (size_t)frame * comb.S;
    vad += (size_t)frame * comb.S;
    rows += (size_t)frame * comb.S * TRAIN_COLS;
    if (s < comb.S) {
        const bool silent = NNN_TI(comb.silence, 1, tile, lane)[0] != 0;
        const int cut = silent ? 0 : cutoff[s];
        const float *f = NNN_TI(comb.feat, NFEAT, tile, lane);
        for (int k = 0; k < NFEAT; k++) row[lane][k] = f[(size_t)k * TILE];
        const float *ec = NNN_TI(clean.ex, NB, tile, lane), *ex = NNN_TI(comb.ex, NB, tile, lane), *en = NNN_TI(noise.ex, NB, tile, lane);
        for (int i = 0; i < NB; i++) {
            const float c = ec[(size_t)i * TILE], x = ex[(size_t)i * TILE];
            float g = -1.0f;
            if (i < cut && !(c < 5e-2f && x < 5e-2f)) g = fminf(sqrtf((c + 1e-3f) / (x + 1e-3f)), 1.0f);
            row[lane][NFEAT + i] = g;
            row[lane][NFEAT + NB + i] = log10f(en[(size_t)i * TILE] + 1e-2f);
        }
        row[lane][NFEAT + 2 * NB] = vad[s];
    }
    __syncthreads();
    // rows of the tile are contiguous in the output: write them coalesced
    const int n = (comb.S - tile * TILE < TILE ? comb.S - tile * TILE : TILE) * TRAIN_COLS;
    float *o = rows + (size_t)tile * TILE * TRAIN_COLS;
    for (int i = lane; i < n; i += 64) o[i] = row[i / TRAIN_COLS][i % TRAIN_COLS];
}

// ---------------------------------------------------------------------------------------------
// K10 rnn: dense + 3 GRUs + 2 dense, i8-origin weights, activations via the 201-entry tanh table.
//     ref: src/rnn.rs:251-272, 292-327, 343-379, 402-410; src/util.rs:29-53.
//     Batched GEMMs on the matrix cores: a tile's activations live in LDS as [stream][column] matrices in three bf16 planes
//     (x = hi + mid + lo exactly), the weights are small integers (exact in bf16) packed on the host in MFMA B-fragment order,
//     so v_mfma_f32_16x16x32_bf16 accumulates exact products.  Two kernels share these pieces: k_rnn (layers one after the
//     other, any model the format allows) and k_rnn_wf (layers of different frames side by side, the built-in shape class).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float tansig_approx(float x, const float *tab)
{
    // ref: src/util.rs:29-45, written without branches so that a lane's many evaluations overlap (the table
    // read is a dependent LDS access).  Same arithmetic for |x| < 8; the saturation tests (which also catch
    // NaN exactly like the reference's reversed comparisons) select the result at the end.
    const float ax = fabsf(x);
    float fi = floorf(0.5f + 25.0f * ax);
    fi = fminf(fi, 200.0f);   // only reached when the result is discarded (|x| >= 8 or NaN)
    const float xr = ax - 0.04f * fi;
    const float y0 = tab[(int)fi];
    const float dy = 1.0f - y0 * y0;
    const float y = y0 + xr * dy * (1.0f - y0 * xr);
    const float r = x < 0.0f ? -y : y;
    return !(x < 8.0f) ? 1.0f : (!(x > -8.0f) ? -1.0f : r);
}
__device__ __forceinline__ float sigmoid_approx(float x, const float *tab) { return 0.5f + 0.5f * tansig_approx(0.5f * x, tab); }
__device__ __forceinline__ float activate(int act, float x, const float *tab)
{
    if (act == 0) return tansig_approx(x, tab);
    if (act == 1) return sigmoid_approx(x, tab);
    return fmaxf(x, 0.0f);
}

constexpr int RNN_WAVES = 8;

__device__ __forceinline__ unsigned short bf16_rn(float x)   // round to nearest even
{
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// x = hi + mid + lo exactly (8 + 8 + 8 significand bits), one bf16 plane each.  Truncation, not rounding: hi is the top
// half of x's word, the remainder x - hi is exact and has <= 16 significant bits, its top half is mid, and what is left
// has <= 8 bits and is a bf16 as it stands.
__device__ __forceinline__ void store_split(unsigned short *P, int plane_stride, int idx, float x)
{
    const unsigned uh = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(uh);
    const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(um);
    P[idx] = (unsigned short)(uh >> 16);
    P[idx + plane_stride] = (unsigned short)(um >> 16);
    P[idx + 2 * plane_stride] = (unsigned short)(__float_as_uint(r2) >> 16);
}
__device__ __forceinline__ float load_split(const unsigned short *P, int plane_stride, int idx)
{
    return (bf16_f32(P[idx]) + bf16_f32(P[idx + plane_stride])) + bf16_f32(P[idx + 2 * plane_stride]);
}

// NNN_RNN_GEMM_PLANES = 2 (developer probe): the products use the hi and mid planes only -- activations truncated to 16
// significand bits -- while the planes are stored and the states kept as before: what dropping a third of the MFMAs and operand reads is
// worth, and what 16-bit activations do to gains and VAD, before anything is rebuilt around two planes.
#ifndef NNN_RNN_GEMM_PLANES
#define NNN_RNN_GEMM_PLANES 3
#endif
constexpr int GPL = NNN_RNN_GEMM_PLANES;
// Weight fragments of one GEMM group: all k-steps (up to KSMAX) are requested together so that a layer pays
// one trip to the Infinity Cache / HBM instead of one per k-step.
constexpr int KSMAX = 4;
template <int NG, int KS = KSMAX> struct Frags { uint4 f[KS][NG]; };

template <int NG, int G0, int KS = KSMAX>
__device__ __forceinline__ void load_frags(Frags<NG, KS> &fr, const GemmDesc &g, const uint4 *__restrict__ Bnb, int lane)
{
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int gi = 0; gi < NG; gi++)
            fr.f[ks][gi] = ks < g.ksteps ? Bnb[((G0 + gi) * g.ksteps + ks) * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
}

// acc[G0 + g][mb] += A[16 (mb0 + mb) .. +15][kbase ..] * B(gate G0 + g), g < NG, mb < MB, over all k-steps and
// the three activation planes.  Bnb points at this neuron block's fragments ([gate][k-step][lane]).
template <int NG, int MB, int G0, int KS = KSMAX>
__device__ __forceinline__ void gemm_acc(f32x4 (&acc)[3][2], const unsigned short *A, int plane_stride, int row_w, int mb0,
                                         const GemmDesc &g, const uint4 *__restrict__ Bnb, int lane, const Frags<NG, KS> &fr)
{
    const unsigned short *a0 = A + (size_t)(mb0 * 16 + (lane & 15)) * row_w + g.kbase + 8 * (lane >> 4);
    const size_t mb_stride = (size_t)16 * row_w;
    // Software pipeline over (k-step, stream block): the three plane fragments of the next step are read from
    // LDS before the current step's 3*NG MFMAs issue, so LDS latency hides behind matrix work.
    const int steps = g.ksteps * MB;
    uint4 cur[3], nxt[3];
#pragma unroll
    for (int pl = 0; pl < GPL; pl++) cur[pl] = *(const uint4 *)(a0 + (size_t)pl * plane_stride);
    for (int ks = 0; ks < g.ksteps; ks++) {
        uint4 bfr[NG];
        if (ks < KS) {
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
                bfr[gi] = fr.f[0][gi];
#pragma unroll
                for (int u = 1; u < KS; u++) bfr[gi] = (ks == u) ? fr.f[u][gi] : bfr[gi];
            }
        } else {   // more k-steps than the fragment set holds: fetch as we go
#pragma unroll
            for (int gi = 0; gi < NG; gi++) bfr[gi] = Bnb[((G0 + gi) * g.ksteps + ks) * 64 + lane];
        }
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
            const int step = ks * MB + mb;
            if (step + 1 < steps) {
                const int ks1 = (mb + 1 < MB) ? ks : ks + 1, mb1 = (mb + 1 < MB) ? mb + 1 : 0;
                const unsigned short *ap = a0 + mb1 * mb_stride + ks1 * 32;
#pragma unroll
                for (int pl = 0; pl < GPL; pl++) nxt[pl] = *(const uint4 *)(ap + (size_t)pl * plane_stride);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < GPL; pl++)
#pragma unroll
                for (int gi = 0; gi < NG; gi++) acc[G0 + gi][mb] = mfma_16x16x32_bf16(cur[pl], bfr[gi], acc[G0 + gi][mb]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < GPL; pl++) cur[pl] = nxt[pl];
        }
    }
}

struct RnnLds {
    const float *tab;
    int *live;            // rows whose frame is not silent (this frame)
    unsigned short *IN;   // input operand matrix [rm][in_w], 3 planes
    unsigned short *RS;   // r * state of the layer in progress [rm][rec_w], 3 planes
    int in_ps, rs_ps;     // plane strides (elements)
    int rm;               // stream rows of this block: 32 or 16
};

// One GRU layer (ref: src/rnn.rs:292-327) as three GEMM groups on the matrix cores.  A wave owns one
// (neuron block, MB stream blocks) unit: z, r and the input part of the candidate accumulate together,
// r * state goes through LDS (every candidate needs all of it), then the recurrent part of the candidate and the state
// update.  The state itself lives in the layer's own LDS planes `SP` (row stride `sw`; three bf16 planes hold an f32
// exactly) for all frames of the launch: recurrent operand of the GEMMs, read back by the owning wave for the update, and
// rewritten by it -- a layer costs two barriers, and no state travels to HBM and back between frames.
// `idle` runs on waves without a unit while the others are in the first GEMM phase (the next frame's features).
template <int MB, class Idle>
__device__ __forceinline__ void gru_layer(const Buffers &b, const LayerDesc &L, const RnnPlan &pl, const RnnLds &lds, unsigned short *SP,
                                          int sw, const uint4 *__restrict__ Wq, const float *__restrict__ fpar,
                                          int wave, int lane, Idle &&idle)
{
    const float scale = 1.0f / 256.0f;
    const int groups = (lds.rm >> 4) / MB, units = L.nb * groups;
    const bool mine = wave < units;
    const int nbi = mine ? wave / groups : 0, mb0 = (wave % groups) * MB;
    const int neuron = nbi * 16 + (lane & 15);
    const bool nvalid = mine && neuron < L.n;
    const int sp_ps = lds.rm * sw;
    const uint4 *Bin = Wq + L.in.wofs + (size_t)nbi * 3 * L.in.ksteps * 64;
    const uint4 *Brec = Wq + L.rec.wofs + (size_t)nbi * 3 * L.rec.ksteps * 64;
    // all weight fragments of this layer start travelling now
    Frags<3> f_in;
    Frags<2> f_zr;
    Frags<1> f_h;
    load_frags<3, 0>(f_in, L.in, Bin, lane);
    load_frags<2, 0>(f_zr, L.rec, Brec, lane);
    load_frags<1, 2>(f_h, L.rec, Brec, lane);
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; g++) bias[g] = (neuron < L.n) ? fpar[L.bias + g * L.n + neuron] : 0.0f;
    NNN_STAMP(b, 16);
    f32x4 acc[3][2];
    float zz[2][4], sold[2][4];
    if (mine) {
#pragma unroll
        for (int g = 0; g < 3; g++) {
#pragma unroll
            for (int mb = 0; mb < MB; mb++) acc[g][mb] = f32x4{bias[g], bias[g], bias[g], bias[g]};
        }
        gemm_acc<2, MB, 0>(acc, SP, sp_ps, sw, mb0, L.rec, Brec, lane, f_zr);   // (recurrent part first: the order k_rnn_wf uses)
        gemm_acc<3, MB, 0>(acc, lds.IN, lds.in_ps, pl.in_w, mb0, L.in, Bin, lane, f_in);
        // r * state: the columns of this neuron block, plus (last block) the padding up to the GEMM's k range, as zeros
        const int kcols = 32 * L.rec.ksteps;
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = (mb0 + mb) * 16 + 4 * (lane >> 4) + q;
                const float so = nvalid ? load_split(SP, sp_ps, row * sw + neuron) : 0.0f;   // the three planes hold the state exactly
                sold[mb][q] = so;
                zz[mb][q] = sigmoid_approx(scale * acc[0][mb][q], lds.tab);
                const float rs = so * sigmoid_approx(scale * acc[1][mb][q], lds.tab);
                if (neuron < kcols) store_split(lds.RS, lds.rs_ps, row * pl.rec_w + neuron, rs);
                if (nbi == L.nb - 1 && neuron + 16 < kcols) store_split(lds.RS, lds.rs_ps, row * pl.rec_w + neuron + 16, 0.0f);
            }
        NNN_STAMP(b, 22);
    } else {
        idle();
    }
    lds_barrier();   // r * state complete; every wave is done reading the old state planes
    NNN_STAMP(b, 18);
    if (mine) {
        gemm_acc<1, MB, 2>(acc, lds.RS, lds.rs_ps, pl.rec_w, mb0, L.rec, Brec, lane, f_h);
        NNN_STAMP(b, 23);
        if (nvalid) {
#pragma unroll
            for (int mb = 0; mb < MB; mb++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = (mb0 + mb) * 16 + 4 * (lane >> 4) + q;
                    const float hh = activate(L.act, scale * acc[2][mb][q], lds.tab);
                    const float z = zz[mb][q], so = sold[mb][q];
                    float snew = z * so + (1.0f - z) * hh;
                    snew = lds.live[row] ? snew : so;   // silent frames leave the state alone (ref: src/denoise.rs:100)
                    store_split(lds.IN, lds.in_ps, row * pl.in_w + L.out_col + neuron, snew);
                    store_split(SP, sp_ps, row * sw + neuron, snew);
                }
        }
        NNN_STAMP(b, 26);
    }
    lds_barrier();
}

// a layer's state: its stream-major array in HBM (the rows of this block) <-> its LDS planes, all threads of the block
__device__ __forceinline__ void gru_state_io(const LayerDesc &L, int rm, float *state, unsigned short *SP, int sw, bool load)
{
    const int n = rm * L.n;
    for (int e = (int)threadIdx.x; e < n; e += 64 * RNN_WAVES) {
        const int row = e / L.n, col = e - row * L.n;
        if (load) store_split(SP, rm * sw, row * sw + col, state[e]);
        else state[e] = load_split(SP, rm * sw, row * sw + col);
    }
}

// dense layer on the matrix cores: act(W x + b) for every (neuron block, 16-stream block) unit, units dealt round-robin to
// the block's waves (a layer wider than the waves are many takes several rounds); `sink(row, neuron, value)` per output
template <class Sink>
__device__ __forceinline__ void dense_layer(const LayerDesc &L, const RnnPlan &pl, const RnnLds &lds, const uint4 *__restrict__ Wq,
                                            const float *__restrict__ fpar, int wave, int lane, Sink &&sink)
{
    const int mbt = lds.rm >> 4, units = L.nb * mbt;
    for (int unit = wave; unit < units; unit += RNN_WAVES) {
        const int nbi = unit / mbt, mb0 = unit % mbt;
        const int neuron = nbi * 16 + (lane & 15);
        const uint4 *Bnb = Wq + L.in.wofs + (size_t)nbi * L.in.ksteps * 64;
        Frags<1> fr;
        load_frags<1, 0>(fr, L.in, Bnb, lane);
        const float bv = neuron < L.n ? fpar[L.bias + neuron] : 0.0f;
        f32x4 acc[3][2];
        acc[0][0] = f32x4{bv, bv, bv, bv};
        gemm_acc<1, 1, 0>(acc, lds.IN, lds.in_ps, pl.in_w, mb0, L.in, Bnb, lane, fr);
        if (neuron < L.n) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                sink(mb0 * 16 + 4 * (lane >> 4) + q, neuron, activate(L.act, acc[0][0][q] * (1.0f / 256.0f), lds.tab));
        }
    }
}

// pair index of rows i < j among the 8 ring rows, in spectral_variability's order
__device__ __forceinline__ int pair_index(int i, int j) { return i * (15 - i) / 2 + (j - i - 1); }

// The feature stage of one frame for one row (lane = stream), start to finish without leaving the lane: cepstral-ring
// update, delta features, spectral variability (ref: src/features.rs:170-219).  The ring (`crs`) and the 28 pairwise
// cepstral distances (`dc`) stay in LDS for all frames of the launch: a new cepstrum changes only the 7 distances it
// takes part in, the other 21 are the same sums over the same rows as the reference recomputes.  Writes the 42 features as
// three bf16 planes into the staging `FS` (row stride FS_W) and the row's live flag for that frame.
constexpr int FS_W = 56;   // 48 feature columns + 8: 16-byte rows, odd multiple of 16 bytes
__device__ __forceinline__ void features_row(const Buffers &b, int f, int tile, int trow, int ll, int rm, float *crs, float *dc,
                                             unsigned short *FS, int fs_w, int *live_next, int &mem_id)
{
    const float *cg = NNN_TIF(b, cn, 28, f, tile, trow);
    float cn[28];
#pragma unroll
    for (int i = 0; i < 28; i++) cn[i] = cg[(size_t)i * TILE];
    const int pitch = NNN_TIF(b, pitch, 1, f, tile, trow)[0];
    const bool silent = NNN_TIF(b, silence, 1, f, tile, trow)[0] != 0;
    float fr[NFEAT];
    if (silent) {   // "if there's no audio, avoid messing up the state" (ref: src/features.rs:160-166)
#pragma unroll
        for (int i = 0; i < NFEAT; i++) fr[i] = 0.0f;
    } else {
        float *cm = NNN_TI(b.ceps_mem, CEPS_MEM * NB, tile, trow);
        const int c0 = mem_id, c1 = mem_id < 1 ? CEPS_MEM + mem_id - 1 : mem_id - 1;
        const int c2 = mem_id < 2 ? CEPS_MEM + mem_id - 2 : mem_id - 2;
#pragma unroll
        for (int k = 0; k < NB; k++) {
            cm[(size_t)(c0 * NB + k) * TILE] = cn[k];
            crs[(c0 * NB + k) * rm + ll] = cn[k];
        }
        mem_id = mem_id + 1 == CEPS_MEM ? 0 : mem_id + 1;
#pragma unroll
        for (int i = 0; i < NB; i++) fr[i] = cn[i];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const float v1 = crs[(c1 * NB + i) * rm + ll], v2 = crs[(c2 * NB + i) * rm + ll];
            const float v0 = cn[i];
            fr[i] = v0 + v1 + v2;
            fr[NB + i] = v0 - v2;
            fr[NB + 6 + i] = v0 - 2.0f * v1 + v2;
            fr[NB + 12 + i] = cn[NB + i];
        }
        fr[40] = 0.01f * ((float)pitch - 300.0f);
        // the 7 distances the new row takes part in, each summed over the 22 bands in order (ref: src/features.rs:203-208)
        for (int j = 0; j < CEPS_MEM; j++) {
            if (j == c0) continue;
            float dist = 0.0f;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const float d = cn[k] - crs[(j * NB + k) * rm + ll];
                dist += d * d;
            }
            dc[pair_index(j < c0 ? j : c0, j < c0 ? c0 : j) * rm + ll] = dist;
        }
        fr[41] = spectral_variability(dc, ll, rm);
    }
    live_next[ll] = silent ? 0 : 1;
    if (b.taps) {
        float *fo = NNN_TIF(b, feat, NFEAT, f, tile, trow);
#pragma unroll
        for (int k = 0; k < NFEAT; k++) fo[(size_t)k * TILE] = fr[k];
    }
#pragma unroll
    for (int k = 0; k < NFEAT; k++) store_split(FS, rm * fs_w, ll * fs_w + k, fr[k]);
}

// ---------------------------------------------------------------------------------------------
// K10 rnn: the feature stage's recurrent part and the network (ref: src/rnn.rs:343-379) for the `g` frames of a group in
//     one launch.  `rm` stream rows (32 or 16 of a 64-stream tile) per block, 8 waves; GEMMs on the matrix cores with
//     exact products (bf16 weights, activations as three bf16 planes).  Across the frames of the launch the GRU states stay in
//     registers and LDS, the cepstral ring and its pair distances in LDS; the last wave prepares frame f + 1's features
//     while the others are inside frame f's GRU GEMMs (when the layer shapes leave it without a unit).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int rnn_state_w(const LayerDesc &L) { return 32 * L.rec.ksteps + 8; }

__global__ void __launch_bounds__(64 * RNN_WAVES) k_rnn(Buffers b, RnnPlan pl, const uint4 *__restrict__ Wq,
                                                          const float *__restrict__ fpar, int tile0, int rm, int g)
{
    HIP_DYNAMIC_SHARED(float, lds_raw)
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane0 = threadIdx.x & 63;
    int wave = wave0, lane = lane0, tid = threadIdx.x;
    const int per = TILE / rm, mbt = rm >> 4;
    int tile, sub;
    xcd_tile_block((int)blockIdx.x, (tile0 & 7) ? 1 : (int)gridDim.x / per, per, tile, sub);
    tile += tile0;                                         // tile0: first tile of this model's run
    const int r0 = sub * rm;                               // first row of the tile handled here
    if (tile * TILE + r0 >= b.S) return;   // (a block whose streams are all padding -- the last tile of a batch that is not a multiple of 64 -- has nothing to do)
    const bool rowl = lane < rm;                           // lane = stream phases: this lane has a row
    const int trow = r0 + (rowl ? lane : 0);               // its row in the tile
    // ---- LDS carve-up (rnn_lds_bytes on the host mirrors it)
    float *tab = lds_raw;
    int *live = (int *)(lds_raw + 256), *live_next = live + 64;
    unsigned short *IN = (unsigned short *)(lds_raw + 256 + 128);
    const int in_ps = rm * pl.in_w, rs_ps = rm * pl.rec_w;
    unsigned short *RS = IN + 3 * in_ps;
    const int sw_v = rnn_state_w(pl.vad), sw_n = rnn_state_w(pl.noise), sw_dn = rnn_state_w(pl.dn);
    unsigned short *SPv = RS + 3 * rs_ps, *SPn = SPv + 3 * rm * sw_v, *SPdn = SPn + 3 * rm * sw_n;
    unsigned short *FS = SPdn + 3 * rm * sw_dn;
    float *crs = (float *)(FS + 3 * rm * FS_W);        // cepstral ring [8 * 22][rm]
    float *dc = crs + CEPS_MEM * NB * rm;              // pair distances [28][rm]
    RnnLds lds{tab, live, IN, RS, in_ps, rs_ps, rm};
    float *sv = b.gru_v + ((size_t)tile * TILE * b.gru_v_w + (size_t)r0 * pl.vad.n),
          *sn = b.gru_n + ((size_t)tile * TILE * b.gru_n_w + (size_t)r0 * pl.noise.n),
          *sdn = b.gru_dn + ((size_t)tile * TILE * b.gru_dn_w + (size_t)r0 * pl.dn.n);
    NNN_STAMP(b, 8);
    // stream blocks per wave unit of a GRU: one, or both of a 32-row block's when the layer has more than four neuron blocks
    // (units = neuron blocks x stream-block groups must stay within the 8 waves)
    const int mb_v = pl.vad.nb * mbt <= RNN_WAVES ? 1 : 2;
    const int mb_n = pl.noise.nb * mbt <= RNN_WAVES ? 1 : 2;
    const int mb_dn = pl.dn.nb * mbt <= RNN_WAVES ? 1 : 2;
#define NNN_MB(mb, CALL)            \
    {                               \
        if ((mb) == 2) { CALL(2) }  \
        else { CALL(1) }            \
    }
    // ---- once per launch: zero every operand plane (padding columns must read as 0), activation table, ring, states
    {
        uint4 *z = (uint4 *)IN;
        const int n16 = (int)(((char *)crs - (char *)IN) / 16);
        for (int i = tid; i < n16; i += 64 * RNN_WAVES) z[i] = make_uint4(0u, 0u, 0u, 0u);
        for (int i = tid; i < 201; i += 64 * RNN_WAVES) tab[i] = b.tansig[i];
        const float *cm = NNN_TI(b.ceps_mem, CEPS_MEM * NB, tile, trow);
        constexpr int PER = (CEPS_MEM * NB + RNN_WAVES - 1) / RNN_WAVES;   // 22 rows per wave, all in flight
        float stg[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int r = wave + i * RNN_WAVES;
            stg[i] = (rowl && r < CEPS_MEM * NB) ? cm[(size_t)r * TILE] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int r = wave + i * RNN_WAVES;
            if (rowl && r < CEPS_MEM * NB) crs[r * rm + lane] = stg[i];
        }
    }
    lds_barrier();
    gru_state_io(pl.vad, rm, sv, SPv, sw_v, true);
    gru_state_io(pl.noise, rm, sn, SPn, sw_n, true);
    gru_state_io(pl.dn, rm, sdn, SPdn, sw_dn, true);
    if (rowl)
        for (int p = wave; p < 28; p += RNN_WAVES) dc[p * rm + lane] = pair_dist(crs, p, lane, rm);
    int mem_id = (wave == RNN_WAVES - 1 && rowl) ? NNN_TI(b.mem_id, 1, tile, trow)[0] : 0;
    lds_barrier();
    // frame 0's features (the last wave; the others have nothing to do yet)
    if (wave == RNN_WAVES - 1 && rowl) features_row(b, 0, tile, trow, lane, rm, crs, dc, FS, FS_W, live_next, mem_id);
    NNN_STAMP(b, 9);
    for (int f = 0; f < g; f++) {
        // keep the frame loop's addresses inside the loop (see launder_v)
        lane = launder_v(lane0);
        wave = launder_s(wave0);
        tid = 64 * wave + lane;
        lds_barrier();   // features of frame f staged; the previous frame is done with the input matrix
        {   // staged features -> their columns of the input matrix; live flags
            const int n8 = rm * 6;   // 48 columns = 6 x 16 bytes per row and plane
            for (int i = tid; i < 3 * n8; i += 64 * RNN_WAVES) {
                const int plx = i / n8, rem = i - plx * n8, row = rem / 6, c8 = rem - row * 6;
                *(uint4 *)(IN + (size_t)plx * in_ps + row * pl.in_w + pl.cF + 8 * c8) =
                    *(const uint4 *)(FS + (size_t)plx * rm * FS_W + row * FS_W + 8 * c8);
            }
            if (tid < 64) live[tid] = live_next[tid];
        }
        lds_barrier();
        NNN_STAMP(b, 10);
        bool feat_done = (f + 1 >= g);   // wave-uniform: the next frame's features are staged (or there is no next frame)
        auto feat_next = [&]() {
            if (wave == RNN_WAVES - 1 && !feat_done) {
                if (rowl) features_row(b, f + 1, tile, trow, lane, rm, crs, dc, FS, FS_W, live_next, mem_id);
                feat_done = true;
            }
        };
        auto no_idle = []() {};
        // input dense (ref: src/rnn.rs:353-355)
        dense_layer(pl.dense, pl, lds, Wq, fpar, wave, lane, [&](int row, int neuron, float v) {
            store_split(IN, in_ps, row * pl.in_w + pl.dense.out_col + neuron, v);
        });
        lds_barrier();
        NNN_STAMP(b, 11);
#define NNN_GRU_V(M) gru_layer<M>(b, pl.vad, pl, lds, SPv, sw_v, Wq, fpar, wave, lane, no_idle);
#define NNN_GRU_N(M) gru_layer<M>(b, pl.noise, pl, lds, SPn, sw_n, Wq, fpar, wave, lane, feat_next);
#define NNN_GRU_DN(M) gru_layer<M>(b, pl.dn, pl, lds, SPdn, sw_dn, Wq, fpar, wave, lane, feat_next);
        NNN_MB(mb_v, NNN_GRU_V)                                             // ref: src/rnn.rs:356-358
        NNN_STAMP(b, 12);
        if (wave == RNN_WAVES - 1 && rowl) {   // vad output, 1 x nv, lane = stream (ref: src/rnn.rs:359)
            float acc = fpar[pl.vo_b];
            for (int k = 0; k < pl.vad.n; k++) acc = fmaf(fpar[pl.vo_w + k], load_split(IN, in_ps, lane * pl.in_w + pl.cV + k), acc);
            NNN_TIF(b, vad, 1, f, tile, trow)[0] = live[lane] ? activate(pl.act_vo, acc * (1.0f / 256.0f), tab) : 0.0f;
        }
        NNN_MB(mb_n, NNN_GRU_N)                                             // ref: src/rnn.rs:361-366
        NNN_STAMP(b, 13);
        NNN_MB(mb_dn, NNN_GRU_DN)                                           // ref: src/rnn.rs:368-377
        NNN_STAMP(b, 14);
        // gains (ref: src/rnn.rs:378) and smoothing g = max(g, 0.6 lastg) (ref: src/denoise.rs:106-109)
        dense_layer(pl.out, pl, lds, Wq, fpar, wave, lane, [&](int lrow, int band, float v) {
            const int row = r0 + lrow;
            const bool lv = live[lrow] != 0;
            const float gr = lv ? v : 0.0f;
            NNN_TIF(b, g_raw, NB, f, tile, row)[(size_t)band * TILE] = gr;
            float gs = 0.0f;
            if (lv) {
                float *lg = NNN_TI(b.lastg, NB, tile, row) + (size_t)band * TILE;
                gs = fmaxf(gr, 0.6f * *lg);
                *lg = gs;
            }
            NNN_TIF(b, g, NB, f, tile, row)[(size_t)band * TILE] = gs;
        });
        feat_next();   // layer shapes that keep every wave busy: the next frame's features go last
        NNN_STAMP(b, 15);
    }
    // ---- states back to HBM (the last layer's update is behind its closing barrier)
    gru_state_io(pl.vad, rm, sv, SPv, sw_v, false);
    gru_state_io(pl.noise, rm, sn, SPn, sw_n, false);
    gru_state_io(pl.dn, rm, sdn, SPdn, sw_dn, false);
    if (wave == RNN_WAVES - 1 && rowl) NNN_TI(b.mem_id, 1, tile, trow)[0] = mem_id;
#undef NNN_MB
}

// ---------------------------------------------------------------------------------------------
// K10w rnn, layer-pipelined ("wavefront"): the same computation as k_rnn for models of the built-in shape class (at most 2 / 2 /
//     3 / 6 neuron blocks of 16 in the input dense, vad, noise and denoise layers), 16 stream rows per block, 12 waves.
//     The chain of a frame -- dense, three GRUs of two phases each, output dense -- is eleven dependent phases of ~1-2 us of
//     mostly latency; here the layers of DIFFERENT frames run side by side, each on its own waves: in tick t the vad GRU works
//     on frame t, the noise GRU on frame t - 1, the denoise GRU on frame t - 2, the output layer on frame t - 3, the input
//     dense and the feature stage on frame t + 1, and a tick is two phases (everybody's first GEMM phase; everybody's second).
//     A group of g frames takes g + 4 ticks instead of 11 g phases.
//     Every layer reads its inputs from its own LDS matrix, written by its producers one to three ticks earlier into the
//     slot of that frame (noise: 2 slots, denoise: 3), so nothing is overwritten before its last reader has passed; all
//     matrices use the column coordinates of the packed weights (k_rnn's single input matrix), each holding the window its
//     layer reads.  Roles (every phase of every layer is a latency chain, so no wave carries two of the long ones): waves 0..5
//     denoise neuron block w, the first two also an input dense unit each in the second phase (their shortest); waves 6..8
//     noise block w - 6 and, between them, the feature fan-out; waves 9, 10 vad block w - 9 plus an output dense unit each;
//     wave 11 the feature stage (lane = (row, part)) and the vad output.  Measured per tick at 65 536 streams
//     (scripts/gpu_stamps_rnn.sh): first phase 3.7-3.9 us on every role, second 1.5-2.6 us (round 2a: 5.8 + 3.4, both set by
//     the features wave and the vad waves, which then also carried the input dense layer and the fan-out).
// ---------------------------------------------------------------------------------------------
constexpr int WF_ROWS = 16, WF_WAVES = 12, WF_FS_W = 72;   // feature staging: the input dense layer's two k-steps wide + 8

struct WfGru {   // what a GRU unit keeps from its first phase to its second
    f32x4 acc[3][2];
    float zz[4], sold[4];
};
// The recurrent weight fragments of a wave's GRU unit (and its biases) stay in its registers for the whole launch; the input
// fragments -- too many to keep beside them at three waves per SIMD -- are requested at the head of the first phase and
// travel behind the recurrent GEMM.  Recurrent GEMMs of the shape class have <= 3 k-steps.
constexpr int WF_KS_REC = 3;
struct WfWeights {
    Frags<2, WF_KS_REC> zr;
    Frags<1, WF_KS_REC> h;
    float bias[3];
};
__device__ __forceinline__ void wf_load_weights(WfWeights &w, const LayerDesc &L, const uint4 *__restrict__ Wq, const float *__restrict__ fpar,
                                                int nbi, int lane)
{
    const uint4 *Brec = Wq + L.rec.wofs + (size_t)nbi * 3 * L.rec.ksteps * 64;
    load_frags<2, 0, WF_KS_REC>(w.zr, L.rec, Brec, lane);
    load_frags<1, 2, WF_KS_REC>(w.h, L.rec, Brec, lane);
    const int neuron = nbi * 16 + (lane & 15);
#pragma unroll
    for (int g = 0; g < 3; g++) w.bias[g] = neuron < L.n ? fpar[L.bias + g * L.n + neuron] : 0.0f;
}

// first phase of GRU layer L for neuron block nbi: z, r, input part of the candidate; r * state -> RS (ref: src/rnn.rs:292-318)
__device__ __forceinline__ void wf_gru_a(const LayerDesc &L, const unsigned short *Ain, int in_w, const unsigned short *SP, unsigned short *RS,
                                         int sw, const uint4 *__restrict__ Wq, const WfWeights &w, const float *tab, int nbi, int lane, WfGru &u)
{
    const float scale = 1.0f / 256.0f;
    const int neuron = nbi * 16 + (lane & 15);
    const bool nvalid = neuron < L.n;
    const uint4 *Bin = Wq + L.in.wofs + (size_t)nbi * 3 * L.in.ksteps * 64;
    const uint4 *Brec = Wq + L.rec.wofs + (size_t)nbi * 3 * L.rec.ksteps * 64;
    Frags<3> f_in;
    load_frags<3, 0>(f_in, L.in, Bin, lane);
#pragma unroll
    for (int g = 0; g < 3; g++) u.acc[g][0] = f32x4{w.bias[g], w.bias[g], w.bias[g], w.bias[g]};
    const int ps = WF_ROWS * sw;
    gemm_acc<2, 1, 0, WF_KS_REC>(u.acc, SP, ps, sw, 0, L.rec, Brec, lane, w.zr);
    gemm_acc<3, 1, 0>(u.acc, Ain, WF_ROWS * in_w, in_w, 0, L.in, Bin, lane, f_in);
    const int kcols = 32 * L.rec.ksteps;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int row = 4 * (lane >> 4) + q;
        const float so = nvalid ? load_split(SP, ps, row * sw + neuron) : 0.0f;   // the three planes hold the state exactly
        u.sold[q] = so;
        u.zz[q] = sigmoid_approx(scale * u.acc[0][0][q], tab);
        const float rs = so * sigmoid_approx(scale * u.acc[1][0][q], tab);
        if (neuron < kcols) store_split(RS, ps, row * sw + neuron, rs);
        if (nbi == L.nb - 1 && neuron + 16 < kcols) store_split(RS, ps, row * sw + neuron + 16, 0.0f);
    }
}

// second phase: recurrent part of the candidate on r * state, state update (ref: src/rnn.rs:319-326); sink(row, neuron, new state)
template <class Sink>
__device__ __forceinline__ void wf_gru_b(const LayerDesc &L, const unsigned short *RS, int sw, const uint4 *__restrict__ Wq, const WfWeights &w,
                                         const float *tab, const int *live, int nbi, int lane, WfGru &u, Sink &&sink)
{
    const float scale = 1.0f / 256.0f;
    const int neuron = nbi * 16 + (lane & 15);
    const uint4 *Brec = Wq + L.rec.wofs + (size_t)nbi * 3 * L.rec.ksteps * 64;
    gemm_acc<1, 1, 2, WF_KS_REC>(u.acc, RS, WF_ROWS * sw, sw, 0, L.rec, Brec, lane, w.h);
    if (neuron < L.n) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = 4 * (lane >> 4) + q;
            const float hh = activate(L.act, scale * u.acc[2][0][q], tab);
            const float z = u.zz[q], so = u.sold[q];
            float snew = z * so + (1.0f - z) * hh;
            snew = live[row] ? snew : so;   // silent frames leave the state alone (ref: src/denoise.rs:100)
            sink(row, neuron, snew);
        }
    }
}

// The feature stage of one frame on all 64 lanes of the features wave: lane = (row, part), row = lane & 15, part = lane >> 4.
// Same arithmetic as features_row (which one lane per row runs start to finish); the four parts of a row share the new
// cepstrum through LDS (`cnb`), split the 7 pair distances the new ring row takes part in and the 42 outputs.  The frame's
// inputs (written by k_fft_xp / k_pitch2, in HBM) are requested a tick ahead: WfFeatIn.
// (they live in the registers of the features wave's otherwise unused weight set: one register allocation serves all roles)
struct WfFeatIn {
    float mine[7];
    int pitch, silent;
};
__device__ __forceinline__ void wf_features_load(WfWeights &w, const Buffers &b, int f, int tile, int r0, int lane)
{
    const int row = lane & 15, part = lane >> 4, trow = r0 + row;
    const float *cg = NNN_TIF(b, cn, 28, f, tile, trow);
    unsigned m[7];
#pragma unroll
    for (int i = 0; i < 7; i++) m[i] = __float_as_uint(cg[(size_t)(7 * part + i) * TILE]);
    w.zr.f[0][0] = make_uint4(m[0], m[1], m[2], m[3]);
    w.zr.f[0][1] = make_uint4(m[4], m[5], m[6], (unsigned)NNN_TIF(b, pitch, 1, f, tile, trow)[0]);
    w.h.f[0][0].x = (unsigned)NNN_TIF(b, silence, 1, f, tile, trow)[0];
}
__device__ __forceinline__ WfFeatIn wf_features_in(const WfWeights &w)
{
    WfFeatIn in;
    const uint4 a = w.zr.f[0][0], c = w.zr.f[0][1];
    in.mine[0] = __uint_as_float(a.x); in.mine[1] = __uint_as_float(a.y); in.mine[2] = __uint_as_float(a.z); in.mine[3] = __uint_as_float(a.w);
    in.mine[4] = __uint_as_float(c.x); in.mine[5] = __uint_as_float(c.y); in.mine[6] = __uint_as_float(c.z);
    in.pitch = (int)c.w;
    in.silent = (int)w.h.f[0][0].x;
    return in;
}
// outputs kb + part of the feature stage, all of one kind (compile-time, so every address below is a base + a constant):
// KIND 0: ceps + ring[-1] + ring[-2]; 1: the value itself; 2: ceps - ring[-2]; 3: ceps - 2 ring[-1] + ring[-2] -- each as
// (v0 + a) + c with features_row's a and c, zeros included.  src_b / rb: first input of the kind / first ring band, for part 0.
template <int KIND, int KB, int SRC_B, int RB, bool HALF>
__device__ __forceinline__ void wf_feature_outputs(const Buffers &b, int f, int tile, int trow, int part, bool silent, const float *cnb_l,
                                                   const float *r1_l, const float *r2_l, unsigned short *fs_l)
{
    constexpr int rm = WF_ROWS;
    if (HALF && part >= 2) return;   // (the kind's last two outputs)
    const float v0 = cnb_l[SRC_B * rm];
    const float v1 = (KIND == 0 || KIND == 3) ? r1_l[RB * rm] : 0.0f;
    const float v2 = (KIND == 0 || KIND == 2 || KIND == 3) ? r2_l[RB * rm] : 0.0f;
    const float a = KIND == 0 ? v1 : (KIND == 3 ? -(2.0f * v1) : 0.0f);
    const float c = (KIND == 0 || KIND == 3) ? v2 : (KIND == 2 ? -v2 : 0.0f);
    float v = (v0 + a) + c;
    v = silent ? 0.0f : v;
    if (b.taps) NNN_TIF(b, feat, NFEAT, f, tile, trow)[(size_t)(KB + part) * TILE] = v;
    store_split(fs_l, rm * WF_FS_W, KB, v);
}

__device__ __forceinline__ void wf_features(const Buffers &b, const WfFeatIn &in, int f, int tile, int r0, int lane, float *crs, float *dc,
                                            float *cnb, unsigned short *FS, int *live_f, int &mem_id)
{
    constexpr int rm = WF_ROWS;
    const int row = lane & 15, part = lane >> 4, trow = r0 + row;
    const int pitch = in.pitch;
    const bool silent = in.silent != 0;
    NNN_STAMPW(b, 8, f == 3);
#pragma unroll
    for (int i = 0; i < 7; i++) cnb[(7 * part + i) * rm + row] = in.mine[i];
    wave_lds_sync();
    if (part == 0) live_f[row] = silent ? 0 : 1;
    const int c0 = mem_id, c1 = mem_id < 1 ? CEPS_MEM + mem_id - 1 : mem_id - 1;
    const int c2 = mem_id < 2 ? CEPS_MEM + mem_id - 2 : mem_id - 2;
    NNN_STAMPW(b, 9, f == 3);
    if (!silent) {   // "if there's no audio, avoid messing up the state" (ref: src/features.rs:160-166)
        float *cm = NNN_TI(b.ceps_mem, CEPS_MEM * NB, tile, trow);
        for (int k = part; k < NB; k += 4) {
            const float v = cnb[k * rm + row];
            cm[(size_t)(c0 * NB + k) * TILE] = v;
            crs[(c0 * NB + k) * rm + row] = v;
        }
        mem_id = mem_id + 1 == CEPS_MEM ? 0 : mem_id + 1;
        NNN_STAMPW(b, 10, f == 3);
        // the 7 distances the new row takes part in, each summed over the 22 bands in order (ref: src/features.rs:203-208);
        // they read the new row from cnb and the others from the ring (rows j != c0 are not being written).  A lane takes
        // partners part and part + 4 together (the fourth part's second one is a shadow of its first, not stored): one read of
        // the new row serves both sums and the two chains hide each other's latency.
        const int ia = part, ib = part + 4 < CEPS_MEM - 1 ? part + 4 : part;
        const int ja = ia < c0 ? ia : ia + 1, jb = ib < c0 ? ib : ib + 1;
        const float *ra = crs + (ja * NB) * rm + row, *rb = crs + (jb * NB) * rm + row, *nw = cnb + row;
        float da = 0.0f, db = 0.0f;
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const float x = nw[k * rm];
            const float ea = x - ra[k * rm], eb = x - rb[k * rm];
            da += ea * ea;
            db += eb * eb;
        }
        dc[pair_index(ja < c0 ? ja : c0, ja < c0 ? c0 : ja) * rm + row] = da;
        if (part + 4 < CEPS_MEM - 1) dc[pair_index(jb < c0 ? jb : c0, jb < c0 ? c0 : jb) * rm + row] = db;
    }
    NNN_STAMPW(b, 11, f == 3);
    wave_lds_sync();
    // outputs 0..39, four at a time (one per part) and one kind at a time: the operations of features_row in its order on the
    // new cepstrum and the two ring rows before it
    {
        const float *cnb_l = cnb + part * rm + row;
        const float *r1_l = crs + (c1 * NB + part) * rm + row, *r2_l = crs + (c2 * NB + part) * rm + row;
        unsigned short *fs_l = FS + row * WF_FS_W + part;
        wf_feature_outputs<0, 0, 0, 0, false>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<0, 4, 4, 4, true>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<1, 6, 6, 0, false>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<1, 10, 10, 0, false>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<1, 14, 14, 0, false>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<1, 18, 18, 0, false>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<2, NB, 0, 0, false>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<2, NB + 4, 4, 4, true>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<3, NB + 6, 0, 0, false>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<3, NB + 10, 4, 4, true>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
        wf_feature_outputs<1, NB + 12, NB, 0, false>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);       // cn[22 + ..]: the correlation DCT
        wf_feature_outputs<1, NB + 16, NB + 4, 0, true>(b, f, tile, trow, part, silent, cnb_l, r1_l, r2_l, fs_l);
    }
    NNN_STAMPW(b, 12, f == 3);
    if (part < 2) {   // k = 40 (pitch) on part 0, k = 41 (spectral variability) on part 1
        float v = part == 0 ? 0.01f * ((float)pitch - 300.0f) : spectral_variability(dc, row, rm);
        v = silent ? 0.0f : v;
        if (b.taps) NNN_TIF(b, feat, NFEAT, f, tile, trow)[(size_t)(40 + part) * TILE] = v;
        store_split(FS, rm * WF_FS_W, row * WF_FS_W + 40 + part, v);
    }
    NNN_STAMPW(b, 13, f == 3);
}

// dense unit nbi of layer L on the 16 rows of the block; sink(row, neuron, value).  Its weights are requested by wf_dense_load,
// placed ahead of other work of the phase so that they travel behind it.
template <int KS> struct WfDenseW {
    Frags<1, KS> fr;
    float bias;
};
template <int KS>
__device__ __forceinline__ void wf_dense_load(WfDenseW<KS> &w, const LayerDesc &L, const uint4 *__restrict__ Wq, const float *__restrict__ fpar,
                                              int nbi, int lane)
{
    load_frags<1, 0, KS>(w.fr, L.in, Wq + L.in.wofs + (size_t)nbi * L.in.ksteps * 64, lane);
    const int neuron = nbi * 16 + (lane & 15);
    w.bias = neuron < L.n ? fpar[L.bias + neuron] : 0.0f;
}
template <int KS, class Sink>
__device__ __forceinline__ void wf_dense(const LayerDesc &L, const unsigned short *Ain, int in_w, const uint4 *__restrict__ Wq, const WfDenseW<KS> &w,
                                         const float *tab, int nbi, int lane, Sink &&sink)
{
    const int neuron = nbi * 16 + (lane & 15);
    const uint4 *Bnb = Wq + L.in.wofs + (size_t)nbi * L.in.ksteps * 64;
    f32x4 acc[3][2];
    acc[0][0] = f32x4{w.bias, w.bias, w.bias, w.bias};
    gemm_acc<1, 1, 0, KS>(acc, Ain, WF_ROWS * in_w, in_w, 0, L.in, Bnb, lane, w.fr);
    if (neuron < L.n) {
#pragma unroll
        for (int q = 0; q < 4; q++) sink(4 * (lane >> 4) + q, neuron, activate(L.act, acc[0][0][q] * (1.0f / 256.0f), tab));
    }
}

#ifndef NNN_WF_MINWAVES
#define NNN_WF_MINWAVES 3   // waves per SIMD the register allocation must allow (12 waves = 3 per SIMD at <= 168 registers)
#endif
struct WfPlan {   // LDS strides (bf16 elements) of the per-layer matrices, set by the host from the model
    int w_v, w_n, w_dn;         // input windows: vad [cD ..], noise [cV ..], denoise [0 ..]
    int sw_v, sw_n, sw_dn;      // state / r * state matrices
};

__host__ __device__ constexpr WfPlan wf_plan_of(const RnnPlan &pl)
{
    return WfPlan{32 * pl.vad.in.ksteps + 8, 32 * pl.noise.in.ksteps + 8, 32 * pl.dn.in.ksteps + 8,
                  32 * pl.vad.rec.ksteps + 8, 32 * pl.noise.rec.ksteps + 8, 32 * pl.dn.rec.ksteps + 8};
}
// SH: a shape class with a compile-time packing plan (SH::plan(), e.g. BkShapeBuiltin of nnn_back.hip: every model of the built-in
// layer sizes) or WfShapeAny (the plan comes with the launch).  With a compile-time plan only the six activation kinds are taken from
// the launch's plan: every stride, column and fragment offset is a constant -- the run-time form keeps some fifty of them in scalar
// registers, more than the wave has, and pays for it in v_readlane / v_writelane spill traffic inside the tick loop (round 5: a sixth
// of the loop's vector instructions).  Same arithmetic either way.
struct WfShapeAny { static constexpr bool fixed = false; };
template <class SH> struct WfFixed { static constexpr bool value = true; };
template <> struct WfFixed<WfShapeAny> { static constexpr bool value = false; };
template <class SH>
__global__ void __launch_bounds__(64 * WF_WAVES, NNN_WF_MINWAVES) k_rnn_wf(Buffers b, RnnPlan pl_rt, WfPlan wp_rt, const uint4 *__restrict__ Wq,
                                                            const float *__restrict__ fpar, int tile0, int g)
{
    RnnPlan pl = pl_rt;
    WfPlan wp = wp_rt;
    if constexpr (WfFixed<SH>::value) {
        constexpr RnnPlan p0 = SH::plan();
        constexpr WfPlan w0 = wf_plan_of(p0);
        pl = p0;
        pl.dense.act = pl_rt.dense.act; pl.vad.act = pl_rt.vad.act; pl.noise.act = pl_rt.noise.act; pl.dn.act = pl_rt.dn.act;
        pl.out.act = pl_rt.out.act; pl.act_vo = pl_rt.act_vo;
        wp = w0;
    }
    HIP_DYNAMIC_SHARED(float, lds_raw)
    constexpr int rm = WF_ROWS;
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane0 = threadIdx.x & 63;
    int wave = wave0, lane = lane0;
    int tile, sub;
    xcd_tile_block((int)blockIdx.x, (tile0 & 7) ? 1 : (int)gridDim.x / (TILE / rm), TILE / rm, tile, sub);
    tile += tile0;
    const int r0 = sub * rm;                                 // first row of the tile handled here
    if (tile * TILE + r0 >= b.S) return;   // (a block whose streams are all padding -- the last tile of a batch that is not a multiple of 64 -- has nothing to do)
    const bool rowl = lane0 < rm;
    const int trow = r0 + (rowl ? lane0 : 0);
    NNN_STAMP(b, 50);
    // ---- LDS carve-up (rnn_wf_lds_bytes on the host mirrors it)
    float *tab = lds_raw;
    int *live = (int *)(lds_raw + 256);                  // [8][16]: live flags of frame f at slot f mod 8 (written a tick before the
                                                         // first reader, read until three ticks after)
    unsigned short *Xv = (unsigned short *)(lds_raw + 256 + 128);
    unsigned short *Xn = Xv + 3 * rm * wp.w_v;           // 2 slots
    unsigned short *Xdn = Xn + 2 * 3 * rm * wp.w_n;      // 3 slots
    unsigned short *SPv = Xdn + 3 * 3 * rm * wp.w_dn, *SPn = SPv + 3 * rm * wp.sw_v, *SPdn = SPn + 3 * rm * wp.sw_n;
    unsigned short *RSv = SPdn + 3 * rm * wp.sw_dn, *RSn = RSv + 3 * rm * wp.sw_v, *RSdn = RSn + 3 * rm * wp.sw_n;
    unsigned short *FS = RSdn + 3 * rm * wp.sw_dn;
    float *crs = (float *)(FS + 3 * rm * WF_FS_W);       // cepstral ring [8 * 22][rm]
    float *dc = crs + CEPS_MEM * NB * rm;                // pair distances [28][rm]
    float *cnb = dc + 28 * rm;                           // the frame's own cepstrum + pitch-correlation DCT [28][rm] (features wave)
    const int cD = pl.dense.out_col, cV = pl.cV, cF = pl.cF;
    float *sv = b.gru_v + ((size_t)tile * TILE * b.gru_v_w + (size_t)r0 * pl.vad.n),
          *sn = b.gru_n + ((size_t)tile * TILE * b.gru_n_w + (size_t)r0 * pl.noise.n),
          *sdn = b.gru_dn + ((size_t)tile * TILE * b.gru_dn_w + (size_t)r0 * pl.dn.n);
    // ---- once per launch: zero every operand plane, activation table, cepstral ring, states, pair distances
    {
        uint4 *z = (uint4 *)Xv;
        const int n16 = (int)(((char *)crs - (char *)Xv) / 16);
        for (int i = (int)threadIdx.x; i < n16; i += 64 * WF_WAVES) z[i] = make_uint4(0u, 0u, 0u, 0u);
        for (int i = (int)threadIdx.x; i < 201; i += 64 * WF_WAVES) tab[i] = b.tansig[i];
        const float *cm = NNN_TI(b.ceps_mem, CEPS_MEM * NB, tile, trow);
        constexpr int PER = (CEPS_MEM * NB + WF_WAVES - 1) / WF_WAVES;
        float stg[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int r = wave + i * WF_WAVES;
            stg[i] = (rowl && r < CEPS_MEM * NB) ? cm[(size_t)r * TILE] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int r = wave + i * WF_WAVES;
            if (rowl && r < CEPS_MEM * NB) crs[r * rm + lane] = stg[i];
        }
    }
    lds_barrier();
    {
        auto load_state = [&](const LayerDesc &L, const float *state, unsigned short *SP, int sw) {
            for (int e = (int)threadIdx.x; e < rm * L.n; e += 64 * WF_WAVES) {
                const int row = e / L.n, col = e - row * L.n;
                store_split(SP, rm * sw, row * sw + col, state[e]);
            }
        };
        load_state(pl.vad, sv, SPv, wp.sw_v);
        load_state(pl.noise, sn, SPn, wp.sw_n);
        load_state(pl.dn, sdn, SPdn, wp.sw_dn);
        if (rowl)
            for (int p = wave; p < 28; p += WF_WAVES) dc[p * rm + lane] = pair_dist(crs, p, lane, rm);
    }
    int mem_id = wave == WF_WAVES - 1 ? NNN_TI(b.mem_id, 1, tile, r0 + (lane & 15))[0] : 0;   // every part of a row keeps a copy
    // this wave's GRU unit: its weights stay in registers for all ticks
    const int W_N = 6, W_V = 9, W_F = 11;   // first wave of the noise / vad roles; the features wave
    WfWeights wts;
    if (wave < W_N) wf_load_weights(wts, pl.dn, Wq, fpar, wave < pl.dn.nb ? wave : 0, lane);
    else if (wave < W_V) wf_load_weights(wts, pl.noise, Wq, fpar, wave - W_N < pl.noise.nb ? wave - W_N : 0, lane);
    else if (wave < W_F) wf_load_weights(wts, pl.vad, Wq, fpar, wave - W_V < pl.vad.nb ? wave - W_V : 0, lane);
    else wf_load_weights(wts, pl.vad, Wq, fpar, 0, lane);   // (the features wave keeps its prefetched inputs there)
    lds_barrier();
    NNN_STAMP(b, 51);
    WfGru ua;   // the GRU unit of this wave (denoise / noise / vad by role), first phase -> second phase
    if (wave == WF_WAVES - 1 && g > 0) wf_features_load(wts, b, 0, tile, r0, lane);
    // the features wave is all vector ALU on one wave while the GRU waves wait on matrix results: let it issue first
    if (wave0 == WF_WAVES - 1) wf_setprio_high();
    for (int t = -1; t < g + 3; t++) {
        lane = launder_v(lane0);   // keep the tick loop's addresses inside the loop (see launder_v)
        wave = launder_s(wave0);
        const int fv = t, fn = t - 1, fd = t - 2, fo = t - 3, ff = t + 1;   // the frame each layer works on in this tick
        const bool on_v = fv >= 0 && fv < g, on_n = fn >= 0 && fn < g, on_d = fd >= 0 && fd < g, on_o = fo >= 0 && fo < g,
                   on_f = ff >= 0 && ff < g;
        unsigned short *Xn_n = Xn + (fn & 1) * 3 * rm * wp.w_n;               // noise input of frame fn
        unsigned short *Xdn_d = Xdn + ((fd + 3) % 3) * 3 * rm * wp.w_dn;      // denoise input of frame fd
        // role stamps of a mid-group tick: slots 30 + 5 role + {0: tick start, 1: first phase done, 2: past barrier, 3: second phase done, 4: past barrier}
        const int srole = wave0 == 0 ? 0 : (wave0 == 6 ? 1 : (wave0 == 9 ? 2 : (wave0 == 11 ? 3 : -1)));
        NNN_STAMPW(b, 30 + 5 * srole, t == 2 && srole >= 0);
        // ---------------- first phase
        if (wave < W_N) {
            if (on_d && wave < pl.dn.nb) wf_gru_a(pl.dn, Xdn_d, wp.w_dn, SPdn, RSdn, wp.sw_dn, Wq, wts, tab, wave, lane, ua);
        } else if (wave < W_V) {
            if (on_n && wave - W_N < pl.noise.nb) wf_gru_a(pl.noise, Xn_n - cV, wp.w_n, SPn, RSn, wp.sw_n, Wq, wts, tab, wave - W_N, lane, ua);
        } else if (wave < W_F) {
            WfDenseW<3> dw;   // the output layer reads the denoise state: <= 96 columns in this shape class
            const bool mine_o = on_o && wave - W_V < pl.out.nb;   // (the output layer has 22 neurons: two units, one per vad wave)
            if (on_v && wave - W_V < pl.vad.nb) wf_gru_a(pl.vad, Xv - cD, wp.w_v, SPv, RSv, wp.sw_v, Wq, wts, tab, wave - W_V, lane, ua);
            if (mine_o) {   // gains of frame fo (ref: src/rnn.rs:378) and smoothing g = max(g, 0.6 lastg) (ref: src/denoise.rs:106-109)
                const int *lv = live + 16 * (fo & 7);
                {
                    const int nbi = wave - W_V;
                    wf_dense_load(dw, pl.out, Wq, fpar, wave - W_V, lane);
                    wf_dense(pl.out, SPdn, wp.sw_dn, Wq, dw, tab, nbi, lane, [&](int lrow, int band, float v) {
                        const int row = r0 + lrow;
                        const bool on = lv[lrow] != 0;
                        const float gr = on ? v : 0.0f;
                        NNN_TIF(b, g_raw, NB, fo, tile, row)[(size_t)band * TILE] = gr;
                        float gs = 0.0f;
                        if (on) {
                            float *lg = NNN_TI(b.lastg, NB, tile, row) + (size_t)band * TILE;
                            gs = fmaxf(gr, 0.6f * *lg);
                            *lg = gs;
                        }
                        NNN_TIF(b, g, NB, fo, tile, row)[(size_t)band * TILE] = gs;
                    });
                }
            }
        } else {
            if (on_f) wf_features(b, wf_features_in(wts), ff, tile, r0, lane, crs, dc, cnb, FS, live + 16 * (ff & 7), mem_id);
            if (ff + 1 >= 0 && ff + 1 < g) wf_features_load(wts, b, ff + 1, tile, r0, lane);   // the next tick's inputs start travelling
        }
        NNN_STAMPW(b, 31 + 5 * srole, t == 2 && srole >= 0);
#if defined(NNN_STAMPS) && NNN_WFSTAMP_PHASE == 1
        NNN_STAMPW(b, 14 + wave0, t == 2);   // every wave's end of the first phase (scripts/gpu_stamps_rnn.sh)
#endif
        lds_barrier();
        NNN_STAMPW(b, 32 + 5 * srole, t == 2 && srole >= 0);
        // ---------------- second phase
        if (wave < W_N) {
            // the denoise units' second phase is the shortest of all: the first waves also take the input dense layer of
            // frame ff (ref: src/rnn.rs:353-355) on the features staged in the first phase; two k-steps (42 features)
            WfDenseW<2> dw;
            const bool mine_d = on_f && wave < pl.dense.nb;
            if (on_d && wave < pl.dn.nb)
                wf_gru_b(pl.dn, RSdn, wp.sw_dn, Wq, wts, tab, live + 16 * (fd & 7), wave, lane, ua,
                         [&](int row, int n, float v) { store_split(SPdn, rm * wp.sw_dn, row * wp.sw_dn + n, v); });
            if (mine_d) {
                unsigned short *Xnn = Xn + (ff & 1) * 3 * rm * wp.w_n;
                wf_dense_load(dw, pl.dense, Wq, fpar, wave, lane);
                wf_dense(pl.dense, FS - cF, WF_FS_W, Wq, dw, tab, wave, lane, [&](int row, int n, float v) {
                    store_split(Xv, rm * wp.w_v, row * wp.w_v + n, v);                      // vad window starts at cD
                    store_split(Xnn, rm * wp.w_n, row * wp.w_n + (cD - cV) + n, v);
                });
            }
        } else if (wave < W_V) {
            if (on_n && wave - W_N < pl.noise.nb) {
                unsigned short *Xd = Xdn + (fn % 3) * 3 * rm * wp.w_dn;   // the denoise layer's input of the same frame
                wf_gru_b(pl.noise, RSn, wp.sw_n, Wq, wts, tab, live + 16 * (fn & 7), wave - W_N, lane, ua, [&](int row, int n, float v) {
                    store_split(SPn, rm * wp.sw_n, row * wp.sw_n + n, v);
                    store_split(Xd, rm * wp.w_dn, row * wp.w_dn + n, v);
                });
            }
            if (on_f) {
                // feature fan-out, dealt over the noise waves: the staged features of frame ff -> the noise and denoise inputs
                // of that frame (48 columns)
                unsigned short *Xnn = Xn + (ff & 1) * 3 * rm * wp.w_n, *Xd = Xdn + (ff % 3) * 3 * rm * wp.w_dn;
                for (int i = 64 * (wave - W_N) + lane; i < 3 * rm * 6; i += 64 * (W_V - W_N)) {
                    const int plx = i / (rm * 6), rem = i - plx * rm * 6, row = rem / 6, c8 = rem - row * 6;
                    const uint4 v = *(const uint4 *)(FS + (size_t)plx * rm * WF_FS_W + row * WF_FS_W + 8 * c8);
                    *(uint4 *)(Xnn + (size_t)plx * rm * wp.w_n + row * wp.w_n + (cF - cV) + 8 * c8) = v;
                    *(uint4 *)(Xd + (size_t)plx * rm * wp.w_dn + row * wp.w_dn + cF + 8 * c8) = v;
                }
            }
        } else if (wave < W_F) {
            if (on_v && wave - W_V < pl.vad.nb) {
                unsigned short *Xnn = Xn + (fv & 1) * 3 * rm * wp.w_n, *Xd = Xdn + (fv % 3) * 3 * rm * wp.w_dn;
                wf_gru_b(pl.vad, RSv, wp.sw_v, Wq, wts, tab, live + 16 * (fv & 7), wave - W_V, lane, ua, [&](int row, int n, float v) {
                    store_split(SPv, rm * wp.sw_v, row * wp.sw_v + n, v);
                    store_split(Xnn, rm * wp.w_n, row * wp.w_n + n, v);               // noise window starts at cV
                    store_split(Xd, rm * wp.w_dn, row * wp.w_dn + cV + n, v);
                });
            }
        } else {
            if (on_n && rowl) {
                // vad output of frame fn, 1 x nv, lane = stream (ref: src/rnn.rs:359), from the copy of that frame's vad state in
                // the noise layer's input (columns 0.. of its window; not rewritten before tick fn + 2)
                const unsigned short *Xv1 = Xn + (fn & 1) * 3 * rm * wp.w_n;
                float acc = fpar[pl.vo_b];
                for (int k = 0; k < pl.vad.n; k++) acc = fmaf(fpar[pl.vo_w + k], load_split(Xv1, rm * wp.w_n, lane * wp.w_n + k), acc);
                NNN_TIF(b, vad, 1, fn, tile, trow)[0] = live[16 * (fn & 7) + lane] ? activate(pl.act_vo, acc * (1.0f / 256.0f), tab) : 0.0f;
            }
        }
        NNN_STAMPW(b, 33 + 5 * srole, t == 2 && srole >= 0);
#if defined(NNN_STAMPS) && NNN_WFSTAMP_PHASE == 2
        NNN_STAMPW(b, 14 + wave0, t == 2);   // ... or of the second
#endif
        lds_barrier();
        NNN_STAMPW(b, 34 + 5 * srole, t == 2 && srole >= 0);
    }
    // ---- states back to HBM
    {
        auto save_state = [&](const LayerDesc &L, float *state, const unsigned short *SP, int sw) {
            for (int e = (int)threadIdx.x; e < rm * L.n; e += 64 * WF_WAVES) {
                const int row = e / L.n, col = e - row * L.n;
                state[e] = load_split(SP, rm * sw, row * sw + col);
            }
        };
        save_state(pl.vad, sv, SPv, wp.sw_v);
        save_state(pl.noise, sn, SPn, wp.sw_n);
        save_state(pl.dn, sdn, SPdn, wp.sw_dn);
    }
    if (wave0 == WF_WAVES - 1 && rowl) NNN_TI(b.mem_id, 1, tile, trow)[0] = mem_id;   // (part 0 of every row)
    NNN_STAMP(b, 52);
}

#if NNN_FFT_CONTRACT
#pragma clang fp contract(fast)   // (k_synth: downstream of the transforms, see the note above them)
#endif
// interpolated band gain at bin k (ref: src/lib.rs:84-97): zero for k >= 400
__device__ __forceinline__ float interp_gain(const float *g, int k, const float *bin_frac, const unsigned char *bin_band)
{
    if (k >= 400) return 0.0f;
    int i = bin_band[k];
    float frac = bin_frac[bsk(k)];   // (the LDS table is skewed)
    return fmaf(frac, g[i + 1], (1.0f - frac) * g[i]);
}
// the same for two gain vectors at once (one look-up of the bin's band and weight serves both)
__device__ __forceinline__ void interp_gain2(const float *ga, const float *gb, int k, const float *bin_frac, const unsigned char *bin_band,
                                             float &ra, float &rb)
{
    ra = 0.0f;
    rb = 0.0f;
    if (k >= 400) return;
    const int i = bin_band[k];
    const float frac = bin_frac[bsk(k)], om = 1.0f - frac;
    ra = fmaf(frac, ga[i + 1], om * ga[i]);
    rb = fmaf(frac, gb[i + 1], om * gb[i]);
}

// ---------------------------------------------------------------------------------------------
// K11 synth: pitch filter, band renormalisation, gains, inverse FFT, window, overlap-add.
//     ref: src/features.rs:223-275, src/denoise.rs:103-114.  One wave per stream; the launch loops over the `g` frames of
//     its group with the overlap memory in registers (read and written once per group, not per frame).
// ---------------------------------------------------------------------------------------------
// ---- pitch filter, band renormalisation, gains, inverse transform, overlap-add for the stream of this wave (ref: src/features.rs:223-275,
//      src/denoise.rs:103-114) on spectra in the wave's registers in the transforms' own bin order (rfft_slot_bin): the frame body of
//      k_synth (spectra from memory) and of the fused back end (spectra straight from its transforms).  b_* are the lane's band
//      (lane < NB) quantities.  The overlap memory is `smv` (sample quads of lane j: samples 4 j + 256 u .. + 3), loaded and stored
//      around the frame when SMV_IO (the fused kernel: eight registers it has not got across a frame) or carried by the caller.
// The band and the interpolation weight of a lane's eight bins are constants of the lane: k_synth, which loops over the frames of a group,
// reads them from the tables once per launch and keeps them in registers (BinConst; round 5: two LDS reads and their index arithmetic less
// per bin and use, three uses per frame: k_synth -2.7 %); the fused back end, which has no registers to spare, looks them up where it
// needs them (null).  Same products of the same factors either way.  (At 128 registers the kernel now spills one 64-bit value, the address
// of the stream's overlap memory: stored before the frame loop, reloaded once behind it -- two scratch accesses per launch, none per frame.)
struct BinConst { int band[8]; float frac[8]; };   // band: -1 = no gain there (bins from 400 up, empty slots)
// PLAIN (round 5): the call's boundary format is process_frame's own -- f32 in the range of an i16, one channel -- known when the kernel is launched:
// the conversions, the channel arithmetic and their branches are compiled out of the instantiation the bench and most device-buffer callers run.
template <bool SMV_IO, bool PLAIN = false>
__device__ __forceinline__ void synth_frame(const Buffers &b, const StepParams *sp, int f, int tile, int sl, int s, int lane, const FftLds &t, float2 *A,
                                         float *r, float2 (&Xr)[8], const float2 (&Pk)[8], float b_ex, float b_ep, float b_xp, float b_graw,
                                         float b_g, float vadv, bool live, float *sm, float4 (&smq)[2], const BinConst *bc = nullptr)
{
    if (SMV_IO) {
#pragma unroll
        for (int u = 0; u < 2; u++) smq[u] = lane + 64 * u < FRAME / 4 ? ((const float4 *)sm)[lane + 64 * u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    float *ebuf = (float *)A, *r2 = r + NB, *gg = r + 2 * NB;
    float *vad_out = sp->vad;
    const int fmt = PLAIN ? (int)PCM_F32 : sp->fmt;
    const int ch = PLAIN ? 1 : sp->channels, grp = s / ch, elem = pcm_elem_bytes(fmt), sstride = ch * elem;
    char *o = sp->out + (long long)grp * sp->group_stride + (long long)(s - grp * ch) * elem;
    const bool store = s < b.S && !sp->discard;
    int bmask = 1 << NB;             // lane 0: this frame's branch mask (bit 22: silent)
    if (live) {
        const bool up = b_xp > b_graw;   // the branch the parity tests compare (ref: src/features.rs:227)
        const int mask = (int)(wave_ballot(up && lane < NB) & ((1ull << NB) - 1));   // bit i: band i took `exp > g`
        if (lane < NB) {
            float v;
            if (up) v = 1.0f;
            else {
                float exp_sq = b_xp * b_xp, g_sq = b_graw * b_graw;
                v = exp_sq * (1.0f - g_sq) / (0.001f + g_sq * (1.0f - exp_sq));
            }
            v = sqrtf(fminf(fmaxf(v, 0.0f), 1.0f));
            v *= sqrtf(b_ex / (1e-8f + b_ep));
            r[lane] = v;
            gg[lane] = b_g;
        }
        wave_lds_sync();
        if (lane == 0) {
            NNN_TIF(b, branch, 1, f, tile, sl)[0] = mask;
            bmask = mask;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = rfft_slot_bin(lane, u);
            if (k >= 0) {
                float2 X = Xr[u];
                const float2 P = k < 400 ? Pk[u] : make_float2(0.0f, 0.0f);   // from bin 400 up the filter gain is zero
                float rf;
                if (!SMV_IO && bc) {
                    const int i = bc->band[u] < 0 ? 0 : bc->band[u];
                    const float fr = bc->frac[u];
                    rf = bc->band[u] < 0 ? 0.0f : fmaf(fr, r[i + 1], (1.0f - fr) * r[i]);
                } else rf = interp_gain(r, k, t.frac, t.band);
                X.x = fmaf(P.x, rf, X.x);
                X.y = fmaf(P.y, rf, X.y);
                Xr[u] = X;
                if (k < 400) ebuf[bsk(k)] = fmaf(X.y, X.y, X.x * X.x);
            }
        }
        wave_lds_sync();
        {
            const float *const v[1] = {ebuf};
            float ne[1];
            band_sums_par<1>(t, v, ne, lane);
            if (lane < NB) r2[lane] = sqrtf(b_ex / (1e-8f + ne[0]));
        }
        wave_lds_sync();
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = rfft_slot_bin(lane, u);
            if (k >= 0) {
                float rf, gf;
                if (!SMV_IO && bc) {
                    const int i = bc->band[u] < 0 ? 0 : bc->band[u];
                    const float fr = bc->frac[u], om = 1.0f - fr;
                    rf = bc->band[u] < 0 ? 0.0f : fmaf(fr, r2[i + 1], om * r2[i]);
                    gf = bc->band[u] < 0 ? 0.0f : fmaf(fr, gg[i + 1], om * gg[i]);
                } else interp_gain2(r2, gg, k, t.frac, t.band, rf, gf);
                Xr[u].x *= rf; Xr[u].y *= rf;
                Xr[u].x *= gf; Xr[u].y *= gf;
            }
        }
    } else if (lane == 0) {
        NNN_TIF(b, branch, 1, f, tile, sl)[0] = 1 << NB;
    }
    if (sp->log && s < b.S) {   // parity-test record of this frame: pitch index, branch mask, smoothed gains
        unsigned *lg = sp->log + (size_t)s * FRAME_LOG_WORDS;
        if (lane < NB) lg[2 + lane] = __float_as_uint(live ? b_g : 0.0f);
        if (lane == 0) {
            lg[0] = (unsigned)NNN_TIF(b, pitch, 1, f, tile, sl)[0];
            lg[1] = (unsigned)bmask;
        }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int k = rfft_slot_bin(lane, u);
        if (k >= 0) A[k] = Xr[u];
    }
    wave_lds_sync();
    // complex-to-real 960-point inverse as a 480-point complex inverse (see k_synth)
    float2 zin[8];
    {
        const int j = lane < FFT_P1 ? lane : FFT_P1 - 1;
#pragma unroll
        for (int rr = 0; rr < 8; rr++) {
            const int k = j + FFT_P1 * rr;
            float2 a = A[k], c = A[NFFT - k];
            float2 e2 = make_float2(a.x + c.x, a.y - c.y);
            float2 d = make_float2(a.x - c.x, a.y + c.y);
            float2 w = t.tw[k];
            w.y = -w.y;
            float2 o2 = cmulf(d, w);
            zin[rr] = make_float2(e2.y + o2.x, e2.x - o2.y);
        }
    }
    wave_lds_sync();   // the spectrum has been read: the transform takes its buffer
    // (from here on a lane owns sample quads: samples 4 j + 256 u .. + 3 of both halves, j = lane, u < 2 -- 16-byte reads of the
    // transform, the window and the overlap memory, 16-byte stores of the audio; same arithmetic per sample as with pairs)
    float4 wlo[2], whi[2];   // the two window halves
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int n = lane + 64 * u;
        const bool on = n < FRAME / 4;
        wlo[u] = on ? ((const float4 *)b.window_s)[n] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // (window / 2: the inverse transform's halving rides on it)
        whi[u] = on ? ((const float4 *)b.window_s)[FRAME / 4 + n] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    fft480_regs<true, NNN_FFT_LANE_TW && !SMV_IO>(zin, A, t.tw, lane);   // time samples: x[2n] = A[n].y, x[2n+1] = A[n].x
    if (lane == 0 && vad_out && s < b.S) vad_out[s] = vadv;
    const bool quad_ok = ch == 1 && (((size_t)o) & (size_t)(4 * elem - 1)) == 0;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int n = lane + 64 * u;
        if (n < FRAME / 4) {
            const float4 lo = ((const float4 *)A)[n], hi = ((const float4 *)A)[n + FRAME / 4];   // (A[2n], A[2n + 1]) each
            const float u0 = hi.y * whi[u].x, u1 = hi.x * whi[u].y, u2 = hi.w * whi[u].z, u3 = hi.z * whi[u].w;   // (x / 2) * w and x * (w / 2) are the same float
            if (store) {
                const float y0 = fmaf(lo.y, wlo[u].x, smq[u].x), y1 = fmaf(lo.x, wlo[u].y, smq[u].y), y2 = fmaf(lo.w, wlo[u].z, smq[u].z),
                            y3 = fmaf(lo.z, wlo[u].w, smq[u].w);
                if (quad_ok && fmt == PCM_F32) ((float4 *)o)[n] = make_float4(y0, y1, y2, y3);
                else if (quad_ok && fmt == PCM_I16)
                    ((uint2 *)o)[n] = make_uint2((unsigned)(unsigned short)pcm_to_i16(y0) | ((unsigned)(unsigned short)pcm_to_i16(y1) << 16),
                                                 (unsigned)(unsigned short)pcm_to_i16(y2) | ((unsigned)(unsigned short)pcm_to_i16(y3) << 16));
                else if (quad_ok) ((float4 *)o)[n] = make_float4(pcm_to_unit(y0), pcm_to_unit(y1), pcm_to_unit(y2), pcm_to_unit(y3));
                else {
                    pcm_store(o + (long long)(4 * n) * sstride, fmt, y0);
                    pcm_store(o + (long long)(4 * n + 1) * sstride, fmt, y1);
                    pcm_store(o + (long long)(4 * n + 2) * sstride, fmt, y2);
                    pcm_store(o + (long long)(4 * n + 3) * sstride, fmt, y3);
                }
            }
            smq[u] = make_float4(u0, u1, u2, u3);
            if (SMV_IO) ((float4 *)sm)[n] = smq[u];
        }
    }
    wave_lds_sync();   // A is refilled by the next frame
}

#ifndef NNN_SYN_MINWAVES
#define NNN_SYN_MINWAVES 4
#endif
template <bool PLAIN>
__global__ void __launch_bounds__(64 * FFT_SPB, NNN_SYN_MINWAVES) k_synth(Buffers b, const StepParams *sp0, int g)
{
    __shared__ FftLds t;
    __shared__ float2 A_[FFT_SPB][NFFT_BUF];   // also the per-bin energies of the band renormalisation (before A is filled)
    __shared__ float r_[FFT_SPB][3 * NB];
    const int wave = threadIdx.x >> 6;
    float2 *A = A_[wave];
    float *r = r_[wave];
    int tile, sub;
    xcd_tile_block((int)blockIdx.x, b.NT, TILE / FFT_SPB, tile, sub);
    if (tile * TILE + sub * FFT_SPB >= b.S) return;   // (a block whose streams are all padding -- the last tile of a batch that is not a multiple of 64 -- has nothing to do)
    const int lane0 = threadIdx.x & 63, sl = sub * FFT_SPB + wave, s = tile * TILE + sl;
    int lane = lane0;
    fft_tables_load(t, b, NNN_FFT_LANE_TW != 0);
    float *sm = b.synth_mem + (size_t)s * FRAME;
    float4 smq[2];   // overlap memory as sample quads, carried from frame to frame in registers
#pragma unroll
    for (int u = 0; u < 2; u++) smq[u] = lane + 64 * u < FRAME / 4 ? ((const float4 *)sm)[lane + 64 * u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();   // tables in place; from here on every wave is on its own (a silent stream skips the filter)
    BinConst bc;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int k = rfft_slot_bin(lane0, u);
        const bool on = k >= 0 && k < 400;
        bc.band[u] = on ? (int)t.band[on ? k : 0] : -1;
        bc.frac[u] = on ? t.frac[bsk(on ? k : 0)] : 0.0f;
    }
    for (int f = 0; f < g; f++) {
        lane = launder_v(lane0);   // keep the frame loop's addresses inside the loop (see launder_v)
        const size_t fo = (size_t)b.S_pad * (size_t)f;   // this frame's scratch set
#ifdef NNN_PROBE_XP
        const float2 *Xg = b.X + (size_t)(s & 255) * FSTR, *Pg = b.P + (size_t)(s & 255) * FSTR;
#else
        const float2 *Xg = b.X + (fo + s) * FSTR, *Pg = b.P + (fo + s) * FSTR;
#endif
        // every global load of this frame is independent of its own results: issue them all now.  The spectra arrive as the
        // transforms held them, (bin k, bin 480 - k) pairs in 16-byte loads (spectrum_load)
        const bool live = NNN_TIF(b, silence, 1, f, tile, sl)[0] == 0;
        float2 Xr[8], Pr[8];
        spectrum_load(Xg, Xr, lane);
        spectrum_load_p(Pg, Pr, lane);
        float b_ex = 0.0f, b_ep = 0.0f, b_xp = 0.0f, b_graw = 0.0f, b_g = 0.0f;
        if (lane < NB) {
            b_ex = NNN_TIF(b, ex, NB, f, tile, sl)[(size_t)lane * TILE];
            b_ep = NNN_TIF(b, ep, NB, f, tile, sl)[(size_t)lane * TILE];
            b_xp = NNN_TIF(b, exp_, NB, f, tile, sl)[(size_t)lane * TILE];
            b_graw = NNN_TIF(b, g_raw, NB, f, tile, sl)[(size_t)lane * TILE];
            b_g = NNN_TIF(b, g, NB, f, tile, sl)[(size_t)lane * TILE];
        }
        const float vadv = NNN_TIF(b, vad, 1, f, tile, sl)[0];
        synth_frame<false, PLAIN>(b, sp0 + f, f, tile, sl, s, lane, t, A, r, Xr, Pr, b_ex, b_ep, b_xp, b_graw, b_g, vadv, live, sm, smq, &bc);
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
        if (lane0 + 64 * u < FRAME / 4) ((float4 *)sm)[lane0 + 64 * u] = smq[u];
}

#pragma clang fp contract(off)
// the activation functions on their own, for the direct known-answer sweep of the parity tests (ref: src/util.rs:29-53)
__global__ void k_activation_kat(const float *tansig, const float *x, float *y, int act, int n)
{
    __shared__ float tab[201];
    for (int i = threadIdx.x; i < 201; i += blockDim.x) tab[i] = tansig[i];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = activate(act, x[i], tab);
}

// Per-frame launch parameters of a call (one thread per frame): entry t describes frame t of the call (v = frame 0).
__global__ void k_fill_params(StepParams *tab, StepParams v, int n, int nslot)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    tab[t] = step_params_at(v, t, nslot);
}

#endif   // NNN_ONLY_HP

}  // namespace nnn
