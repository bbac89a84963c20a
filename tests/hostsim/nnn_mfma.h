// tests/hostsim/nnn_mfma.h -- TEST-ONLY shadow of nnnoiseless_amd/csrc/nnn_mfma.h for the SIMT
// interpreter: the 64 fibers of a wave rendezvous and the tile product is computed in plain C++
// with the fragment layout documented for v_mfma_f32_16x16x32_bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <time.h>

namespace nnn {

struct f32x4 {
    float x, y, z, w;
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

namespace detail {
struct MfmaIn { uint4 a, b; f32x4 c; };
static inline float bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline void mfma_tile(const void *const *ins, void *const *outs, int nl)
{
    static float A[16][32], B[32][16], C[16][16];
    for (int l = 0; l < nl; l++) {
        if (!ins[l]) continue;
        const MfmaIn *in = (const MfmaIn *)ins[l];
        unsigned short ha[8], hb[8];
        memcpy(ha, &in->a, 16);
        memcpy(hb, &in->b, 16);
        for (int e = 0; e < 8; e++) {
            A[l & 15][8 * (l >> 4) + e] = bf16_to_f32(ha[e]);
            B[8 * (l >> 4) + e][l & 15] = bf16_to_f32(hb[e]);
        }
        for (int q = 0; q < 4; q++) C[4 * (l >> 4) + q][l & 15] = in->c[q];
    }
    for (int l = 0; l < nl; l++) {
        if (!outs[l]) continue;
        f32x4 *d = (f32x4 *)outs[l];
        for (int q = 0; q < 4; q++) {
            int i = 4 * (l >> 4) + q, j = l & 15;
            float acc = C[i][j];
            for (int k = 0; k < 32; k++) acc = fmaf(A[i][k], B[k][j], acc);
            (*d)[q] = acc;
        }
    }
}
}  // namespace detail

static inline f32x4 mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c)
{
    detail::MfmaIn in = {a, b, c};
    f32x4 d;
    hostsim::wave_collective(&in, &d, detail::mfma_tile);
    return d;
}

// ---- the matrix-core DFT's primitives ----
namespace detail {
static inline float f16_to_f32(unsigned short h)
{
    const unsigned sgn = (unsigned)(h & 0x8000) << 16, e = (h >> 10) & 31, m = h & 1023;
    if (e == 0) { float f = ldexpf((float)m, -24); return sgn ? -f : f; }
    if (e == 31) { unsigned u = sgn | 0x7f800000u | (m << 13); float f; memcpy(&f, &u, 4); return f; }
    unsigned u = sgn | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}
static inline unsigned short f32_to_f16_rn(float f)   // round to nearest even, overflow to infinity, subnormals kept
{
    unsigned u; memcpy(&u, &f, 4);
    const unsigned sgn = (u >> 16) & 0x8000;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (unsigned short)(sgn | 0x7c00 | (u > 0x7f800000u ? 0x200 : 0));
    if (u >= 0x477ff000u) return (unsigned short)(sgn | 0x7c00);   // rounds to >= 65520: infinity
    if (u < 0x38800000u) {                                           // below the smallest normal half: a multiple of 2^-24
        float a; memcpy(&a, &u, 4);
        const float q = a * 16777216.0f;                            // exact
        const float r = nearbyintf(q);                              // (default rounding mode: to nearest even)
        return (unsigned short)(sgn | (unsigned)r);
    }
    const unsigned mant = u & 0x7fffffu, e = (u >> 23) - 112;
    unsigned h = (e << 10) | (mant >> 13);
    const unsigned rem = mant & 0x1fff;
    if (rem > 0x1000 || (rem == 0x1000 && (h & 1))) h++;
    return (unsigned short)(sgn | h);
}
static inline void mfma_tile_f16(const void *const *ins, void *const *outs, int nl)
{
    static float A[16][32], B[32][16], C[16][16];
    for (int l = 0; l < nl; l++) {
        if (!ins[l]) continue;
        const MfmaIn *in = (const MfmaIn *)ins[l];
        unsigned short ha[8], hb[8];
        memcpy(ha, &in->a, 16);
        memcpy(hb, &in->b, 16);
        for (int e = 0; e < 8; e++) {
            A[l & 15][8 * (l >> 4) + e] = f16_to_f32(ha[e]);
            B[8 * (l >> 4) + e][l & 15] = f16_to_f32(hb[e]);
        }
        for (int q = 0; q < 4; q++) C[4 * (l >> 4) + q][l & 15] = in->c[q];
    }
    for (int l = 0; l < nl; l++) {
        if (!outs[l]) continue;
        f32x4 *d = (f32x4 *)outs[l];
        for (int q = 0; q < 4; q++) {
            int i = 4 * (l >> 4) + q, j = l & 15;
            float acc = C[i][j];
            for (int k = 0; k < 32; k++) acc = fmaf(A[i][k], B[k][j], acc);
            (*d)[q] = acc;
        }
    }
}
static inline void umax_fn(const void *const *ins, void *const *outs, int nl)
{
    unsigned m = 0;
    for (int l = 0; l < nl; l++) if (ins[l]) { unsigned v = *(const unsigned *)ins[l]; if (v > m) m = v; }
    for (int l = 0; l < nl; l++) if (outs[l]) *(unsigned *)outs[l] = m;
}
}  // namespace detail
static inline f32x4 mfma_16x16x32_f16(uint4 a, uint4 b, f32x4 c)
{
    detail::MfmaIn in = {a, b, c};
    f32x4 d;
    hostsim::wave_collective(&in, &d, detail::mfma_tile_f16);
    return d;
}
static inline unsigned pk_f16_rn(float a, float b) { return (unsigned)detail::f32_to_f16_rn(a) | ((unsigned)detail::f32_to_f16_rn(b) << 16); }
static inline float f16_resid_lo(float a, unsigned h) { return a - detail::f16_to_f32((unsigned short)(h & 0xffff)); }
static inline float f16_resid_hi(float a, unsigned h) { return a - detail::f16_to_f32((unsigned short)(h >> 16)); }
static inline unsigned pk_bf16_trunc(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
static inline float amax3(float a, float b, float m)
{
    const float x = fabsf(a), y = fabsf(b);
    float r = m;
    if (x > r) r = x;   // (a NaN operand compares false: ignored)
    if (y > r) r = y;
    return r;
}
template <int D> static inline float dpp_row_down(float x) { return __shfl_down(x, D, 16); }   // (own value where the source is past the row)
static inline float wave_read(float x, int src) { return __shfl(x, src & 63); }
static inline unsigned wave_max_u32(unsigned x)
{
    unsigned m = 0;
    hostsim::wave_collective(&x, &m, detail::umax_fn);
    return m;
}

static inline void lds_barrier() { __syncthreads(); }
// the interpreter's lanes are fibers, not lock-step: a wave-level rendezvous (the shuffle machinery) stands in
static inline void wave_lds_sync() { (void)__shfl(0, 0); }
template <class T> static inline T ld_global(const void *p) { T v; memcpy(&v, p, sizeof(T)); return v; }
static inline uint4 ld_global_u4(const void *p) { return ld_global<uint4>(p); }
template <class T> static inline void st_global(void *p, T v) { memcpy(p, &v, sizeof(T)); }

static inline void wf_setprio_high() {}
template <int P> static inline void wave_prio() {}
static inline int launder_v(int x) { return x; }
static inline int launder_s(int x) { return x; }
static inline void keep_v(float) {}
static inline void keep_rw(float &) {}

struct v2f { float x, y; };
static inline v2f operator*(v2f a, v2f b) { v2f r = {a.x * b.x, a.y * b.y}; return r; }
static inline v2f operator+(v2f a, v2f b) { v2f r = {a.x + b.x, a.y + b.y}; return r; }
static inline v2f mk2(float x, float y) { v2f r = {x, y}; return r; }
static inline v2f pk_mul(v2f a, v2f b) { return mk2(a.x * b.x, a.y * b.y); }
static inline v2f pk_mul_bx(v2f a, v2f b) { return mk2(a.x * b.x, a.x * b.y); }
static inline v2f pk_mul_by(v2f a, v2f b) { return mk2(a.y * b.x, a.y * b.y); }
static inline v2f pk_add(v2f a, v2f b) { return mk2(a.x + b.x, a.y + b.y); }
static inline v2f pk_add_bx(v2f a, v2f b) { return mk2(a.x + b.x, a.y + b.x); }
static inline v2f pk_add_by(v2f a, v2f b) { return mk2(a.x + b.y, a.y + b.y); }
static inline float sadd(float a, float b) { return a + b; }
static inline float smul(float a, float b) { return a * b; }

// ---- the certified coarse pitch search's primitives ----
static inline unsigned short bf16_rn_bits(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN (quiet)
    u += 0x7fffu + ((u >> 16) & 1u);                                                   // to nearest even; overflow rounds to infinity
    return (unsigned short)(u >> 16);
}
static inline unsigned pk_bf16_rn(float a, float b) { return (unsigned)bf16_rn_bits(a) | ((unsigned)bf16_rn_bits(b) << 16); }
static inline unsigned align_bits(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((unsigned long long)hi << 32) | lo) >> (sh & 31)); }
static inline unsigned lds_add_u32(unsigned *p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }   // (fibers run one at a time)
static inline void lds_or_u32(unsigned *p, unsigned v) { *p |= v; }
static inline float wave_xor16(float x, int) { return __shfl_xor(x, 16); }
static inline float wave_xor32(float x, int) { return __shfl_xor(x, 32); }
template <int STEP> static inline float row_partner(float x)
{
    const int lane = (int)(threadIdx.x & 63);
    const int src = STEP == 0 ? lane ^ 1 : STEP == 1 ? lane ^ 2 : STEP == 2 ? ((lane & ~7) | (7 - (lane & 7))) : ((lane & ~15) | (15 - (lane & 15)));
    return __shfl(x, src);
}
template <int K> static inline float quad_lane(float y) { return __shfl(y, (int)((threadIdx.x & 60) | K)); }
// the four values of the caller's quad in ONE rendezvous of the wave's fibers (k_pitch's serial scans call this once per four steps: as four
// shuffles it was most of the interpreter's time)
struct Quad4 { float t0, t1, t2, t3; };
namespace detail {
static inline void quad_all_fn(const void *const *ins, void *const *outs, int nl)
{
    for (int l = 0; l < nl; l++) {
        if (!outs[l]) continue;
        float q[4];
        for (int k = 0; k < 4; k++) { const void *p = ins[(l & ~3) | k]; q[k] = p ? *(const float *)p : 0.0f; }
        Quad4 r = {q[0], q[1], q[2], q[3]};
        *(Quad4 *)outs[l] = r;
    }
}
}  // namespace detail
static inline Quad4 quad_all(float y)
{
    Quad4 r;
    hostsim::wave_collective(&y, &r, detail::quad_all_fn);
    return r;
}
static inline unsigned lane_rank(unsigned long long m, int lane) { return (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull)); }
static inline float lane_value(float x, int l) { return __shfl(x, l); }
static inline float fast_sqrt(float x) { return sqrtf(x); }
static inline float fast_rsq(float x) { return 1.0f / sqrtf(x); }

// (the interpreter runs the workgroups of a launch one after the other in index order: a flag is always set when it is read)
static inline void flag_publish(int *flag, int value) { *flag = value; }
static inline int flag_read(const int *flag) { return *flag; }
static inline unsigned ticket_take(unsigned *ctr) { return (*ctr)++; }
namespace detail {
static inline void ballot_fn(const void *const *ins, void *const *outs, int nl)
{
    unsigned long long m = 0;
    for (int l = 0; l < nl; l++) if (ins[l] && *(const bool *)ins[l]) m |= 1ull << l;
    for (int l = 0; l < nl; l++) if (outs[l]) *(unsigned long long *)outs[l] = m;
}
}  // namespace detail
static inline unsigned long long wave_ballot(bool p)
{
    unsigned long long m = 0;
    hostsim::wave_collective(&p, &m, detail::ballot_fn);
    return m;
}
static inline bool wave_any(bool) { return true; }   // (a skipped no-op update and an executed one leave the same state)
static inline void chain_pause() {}
static inline long long realtime_ticks()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 100000000ll + ts.tv_nsec / 10;
}
static inline void __threadfence() {}

}  // namespace nnn
