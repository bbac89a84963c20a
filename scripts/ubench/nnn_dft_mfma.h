// nnn_dft_mfma.h -- the 480-point complex DFT behind the 960-point real transforms (ref: the real_fft_using / real_ifft_using calls of
// src/features.rs:264,290; the arithmetic lives in the un-vendored rustfft) as two matrix products on the matrix cores.
//
//   n = 30 n1 + n2, k = k1 + 16 k2 (n1, k1 < 16; n2, k2 < 30):
//   Z[k1 + 16 k2] = sum_n2 w30^(n2 k2) * ( w480^(n2 k1) * sum_n1 z[30 n1 + n2] w16^(n1 k1) )
//
//   stage 1   C1[n2][k1]   = A[n2][(n1, re|im)] * B1[(n1, re|im)][(k1, re|im)]     data is the A operand: 2 row tiles x 2 column tiles, K = 32
//   twiddle   Y[n2][k1]    = C1[n2][k1] * w480^(n2 k1)                             on the vector ALU, in the accumulator layout
//   stage 2   D[k2][k1]    = A2[(k2, re|im)][(n2, re|im)] * Y[(n2, re|im)][k1]     data is the B operand: 4 row tiles, 2 k-steps
//
// The accumulator layout of stage 1 (lane = column k1, four consecutive rows n2) IS the B-operand layout of stage 2 up to the order of the
// k index, which only permutes the constant matrix's columns: nothing is transposed between the stages, and the transform touches LDS
// once, to hand its result over in natural order.  One wave = one transform (the callers' wave = stream mapping); the constant
// fragments live in the wave's registers (DftRegs: 48 or 72 of them + 16 for the twiddles) for as many transforms as the wave runs.
//
// Operands are f32 values split into planes the matrix cores take, products of planes accumulated in f32:
//   DftF16   x = hi + lo in IEEE half precision (two round-to-nearest steps, 22 significant bits), products hi*hi, hi*lo, lo*hi; the data
//            is scaled by a power of two per transform (largest magnitude into [0.5, 1)) so that the planes stay inside half precision's
//            exponent range, and scaled back on the way out.  36 matrix instructions per transform.  Measured error (profiles/
//            r5_dft_mfma_probe.txt): that of an f32 FFT.
//   DftBf16  x = hi + mid + lo in bf16 by truncation (exact, 24 bits; no scaling: bf16 has f32's exponent range), the six products of
//            weight >= 2^-16: 72 matrix instructions.  Error below an f32 FFT's.
// Everything downstream of these transforms is tolerance-only.
#pragma once
#include "nnn_layout.h"
#include <nnn_mfma.h>
#include <math.h>
#include <string.h>

namespace nnn {

constexpr int DFT_K1 = 16, DFT_K2 = 30;   // 480 = 16 x 30

// ---- which element of the 480 a lane's input register r (0 .. 7) holds: lane = (m = lane % 16, g = lane / 16), r = 4 t + j:
//      n = 30 (4 g + j) + 16 t + m, or -1 (t = 1, m >= 14: rows 30, 31 of the padded 32; such a slot must hold a finite value no larger
//      in magnitude than the transform's real inputs -- a duplicate of one of them -- and is otherwise ignored)
__host__ __device__ inline int dft_in_n(int lane, int r)
{
    const int m = lane & 15, g = lane >> 4, t = r >> 2, j = r & 3, n2 = 16 * t + m;
    return n2 < 30 ? 30 * (4 * g + j) + n2 : -1;
}
// the same with the padding rows redirected to a valid element (what a loader may fetch for them)
__host__ __device__ inline int dft_in_n_clamped(int lane, int r)
{
    const int n = dft_in_n(lane, r);
    return n >= 0 ? n : dft_in_n(lane & ~15, r);   // (row m = 0 of the same tile)
}

struct DftF16 {
    static constexpr int NPL = 2;
    static constexpr bool SCALED = true;
    static constexpr int NPROD = 3;
    // (data plane, constant plane), smallest weight first
    __host__ __device__ static constexpr int pd(int p) { return p == 0 ? 1 : 0; }
    __host__ __device__ static constexpr int pc(int p) { return p == 1 ? 1 : 0; }
    __device__ static __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) { return mfma_16x16x32_f16(a, b, c); }
    __device__ static __forceinline__ void split_pair(float a, float b, unsigned (&pl)[NPL])
    {
        const unsigned h = pk_f16_rn(a, b);
        pl[0] = h;
        pl[1] = pk_f16_rn(f16_resid_lo(a, h), f16_resid_hi(b, h));
    }
    static constexpr unsigned SIGNS = 0x80008000u;
};
struct DftBf16 {
    static constexpr int NPL = 3;
    static constexpr bool SCALED = false;
    static constexpr int NPROD = 6;
    __host__ __device__ static constexpr int pd(int p) { return p == 0 ? 2 : (p == 1 ? 0 : (p == 2 ? 1 : (p == 3 ? 1 : 0))); }
    __host__ __device__ static constexpr int pc(int p) { return p == 0 ? 0 : (p == 1 ? 2 : (p == 2 ? 1 : (p == 3 ? 0 : (p == 4 ? 1 : 0)))); }
    __device__ static __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) { return mfma_16x16x32_bf16(a, b, c); }
    __device__ static __forceinline__ void split_pair(float a, float b, unsigned (&pl)[NPL])
    {
        pl[0] = pk_bf16_trunc(a, b);
        const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
        pl[1] = pk_bf16_trunc(ra, rb);
        const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
        pl[2] = pk_bf16_trunc(sa, sb);
    }
    static constexpr unsigned SIGNS = 0x80008000u;
};

// the constant fragments and twiddles as a wave holds them
template <class SC> struct DftRegs {
    uint4 b1[2][SC::NPL];       // stage 1, column tile c (0: real parts of the 16 outputs k1, 1: imaginary parts)
    uint4 a2[2][2][SC::NPL];    // stage 2, row block b (k2 = 16 b + m), kind (0: cos, 1: sin)
    float2 tw[8];               // w480^(n2 k1) at the lane's accumulator rows: [4 t + q], n2 = 16 t + 4 g + q, k1 = lane % 16
};
// their image in memory: fragment f of lane l at uint4 index f * 64 + l, the twiddles behind them ([i][lane] float2)
template <class SC> struct DftImage {
    static constexpr int NFRAG = 6 * SC::NPL;
    static constexpr size_t BYTES = (size_t)NFRAG * 64 * 16 + 8 * 64 * 8;
};

template <class SC>
__device__ __forceinline__ void dft_regs_load(DftRegs<SC> &c, const void *img, int lane)
{
    const uint4 *f = (const uint4 *)img + lane;
    int i = 0;
#pragma unroll
    for (int cc = 0; cc < 2; cc++)
#pragma unroll
        for (int p = 0; p < SC::NPL; p++) c.b1[cc][p] = ld_global_u4(f + 64 * (i++));
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int kd = 0; kd < 2; kd++)
#pragma unroll
            for (int p = 0; p < SC::NPL; p++) c.a2[b][kd][p] = ld_global_u4(f + 64 * (i++));
    const float2 *t = (const float2 *)((const uint4 *)img + 64 * DftImage<SC>::NFRAG) + lane;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const v2f w = ld_global<v2f>(t + 64 * r);
        c.tw[r] = make_float2(w.x, w.y);
    }
}

// acc += sum over the scheme's plane products of data x constant; the data is the A operand (stage 1) or the B operand (stage 2)
template <class SC, bool DATA_IS_A>
__device__ __forceinline__ f32x4 dft_mm(const uint4 (&data)[SC::NPL], const uint4 (&cst)[SC::NPL], f32x4 acc)
{
#pragma unroll
    for (int p = 0; p < SC::NPROD; p++)
        acc = DATA_IS_A ? SC::mfma(data[SC::pd(p)], cst[SC::pc(p)], acc) : SC::mfma(cst[SC::pc(p)], data[SC::pd(p)], acc);
    return acc;
}

// eight floats (a lane's k = 8 g .. 8 g + 7 of one operand) -> the scheme's planes, element pairs (0,1), (2,3), (4,5), (6,7)
template <class SC>
__device__ __forceinline__ void dft_split8(const float (&x)[8], uint4 (&pl)[SC::NPL])
{
    unsigned w[4][SC::NPL];
#pragma unroll
    for (int i = 0; i < 4; i++) SC::split_pair(x[2 * i], x[2 * i + 1], w[i]);
#pragma unroll
    for (int p = 0; p < SC::NPL; p++) pl[p] = make_uint4(w[0][p], w[1][p], w[2][p], w[3][p]);
}

// The transform.  v: the 480 inputs in the order of dft_in_n; Z: the wave's LDS buffer, 480 outputs in natural order (the caller
// synchronises the wave before reading them, and every earlier reader of Z must be done).  Returns the factor the outputs still have
// to be multiplied by (1 unless UNSCALE is false and the scheme scales its data).
template <class SC, bool UNSCALE = true>
__device__ __forceinline__ float dft480_mfma(float2 (&v)[8], float2 *Z, const DftRegs<SC> &c, int lane)
{
    float inv = 1.0f;
    if (SC::SCALED) {
        float m = 0.0f;
#pragma unroll
        for (int r = 0; r < 8; r++) m = amax3(v[r].x, v[r].y, m);
        const unsigned e = wave_max_u32(__float_as_uint(m)) >> 23;             // biased exponent of the largest magnitude (255: inf / NaN)
        int sf = 253 - (int)e;                                                  // scale 2^(126 - e): the maximum lands in [0.5, 1)
        sf = sf < 1 ? 1 : sf;
        const float s = __uint_as_float((unsigned)sf << 23);
        inv = __uint_as_float((unsigned)(254 - sf) << 23);                      // exactly 1 / s
#pragma unroll
        for (int r = 0; r < 8; r++) { v[r].x *= s; v[r].y *= s; }
    }
    // ---- stage 1: rows n2 (two tiles), K = (n1, re|im) = 32
    f32x4 c1[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const float x[8] = {v[4 * t].x, v[4 * t].y, v[4 * t + 1].x, v[4 * t + 1].y, v[4 * t + 2].x, v[4 * t + 2].y, v[4 * t + 3].x, v[4 * t + 3].y};
        uint4 a[SC::NPL];
        dft_split8<SC>(x, a);
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            c1[t][cc] = dft_mm<SC, true>(a, c.b1[cc], z);
        }
    }
    // ---- twiddle, then the planes of stage 2's B operand: k-step "re" = the 8 real parts (t, q), k-step "im" the imaginary parts
    float yr[8], yi[8];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float ar = c1[t][0][q], ai = c1[t][1][q];
            const float2 w = c.tw[4 * t + q];
            yr[4 * t + q] = fmaf(ar, w.x, -ai * w.y);
            yi[4 * t + q] = fmaf(ar, w.y, ai * w.x);
        }
    uint4 bre[SC::NPL], bim[SC::NPL], bnre[SC::NPL];
    dft_split8<SC>(yr, bre);
    dft_split8<SC>(yi, bim);
#pragma unroll
    for (int p = 0; p < SC::NPL; p++)   // the planes of -x are the planes of x with the signs flipped
        bnre[p] = make_uint4(bre[p].x ^ SC::SIGNS, bre[p].y ^ SC::SIGNS, bre[p].z ^ SC::SIGNS, bre[p].w ^ SC::SIGNS);
    // ---- stage 2: D_re = cos * Yre + sin * Yim, D_im = cos * Yim + sin * (-Yre)
    const int k1 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 dre = dft_mm<SC, false>(bim, c.a2[b][1], z);
        f32x4 dim = dft_mm<SC, false>(bnre, c.a2[b][1], z);
        dre = dft_mm<SC, false>(bre, c.a2[b][0], dre);
        dim = dft_mm<SC, false>(bim, c.a2[b][0], dim);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k2 = 16 * b + 4 * g + q;
            if (k2 < DFT_K2) {
                float2 o = make_float2(dre[q], dim[q]);
                if (SC::SCALED && UNSCALE) { o.x *= inv; o.y *= inv; }
                Z[k1 + 16 * k2] = o;
            }
        }
    }
    return UNSCALE ? 1.0f : inv;
}

// ---------------------------------------------------------------------------------------------
// host side: the image of the constants
// ---------------------------------------------------------------------------------------------
namespace dft_host {
inline float f16_value(unsigned short h)
{
    const int e = (h >> 10) & 31, m = h & 1023;
    const float f = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(m + 1024), e - 25);
    return (h & 0x8000) ? -f : f;
}
inline unsigned short f16_rn(float f)   // |f| <= 1 here: round to nearest even, subnormals kept
{
    unsigned u;
    memcpy(&u, &f, 4);
    const unsigned sgn = (u >> 16) & 0x8000;
    u &= 0x7fffffffu;
    if (u < 0x38800000u) {
        float a;
        memcpy(&a, &u, 4);
        return (unsigned short)(sgn | (unsigned)nearbyintf(a * 16777216.0f));
    }
    const unsigned mant = u & 0x7fffffu, e = (u >> 23) - 112;
    unsigned h = (e << 10) | (mant >> 13);
    const unsigned rem = mant & 0x1fff;
    if (rem > 0x1000 || (rem == 0x1000 && (h & 1))) h++;
    return (unsigned short)(sgn | h);
}
inline unsigned short bf16_rn(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
inline float bf16_value(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// a constant (|w| <= 1, given in double) as the scheme's planes, each plane rounded to nearest
template <class SC> inline void planes(double w, unsigned short (&pl)[SC::NPL]);
template <> inline void planes<DftF16>(double w, unsigned short (&pl)[2])
{
    pl[0] = f16_rn((float)w);
    pl[1] = f16_rn((float)(w - (double)f16_value(pl[0])));
}
template <> inline void planes<DftBf16>(double w, unsigned short (&pl)[3])
{
    pl[0] = bf16_rn((float)w);
    double r = w - (double)bf16_value(pl[0]);
    pl[1] = bf16_rn((float)r);
    r -= (double)bf16_value(pl[1]);
    pl[2] = bf16_rn((float)r);
}
}  // namespace dft_host

// fills `img` (DftImage<SC>::BYTES bytes) -- see the layout comments at the top of the file
template <class SC> inline void dft_mfma_image(void *img)
{
    const double PI2 = 6.283185307179586476925286766559;
    unsigned short *h = (unsigned short *)img;   // fragment f, lane l, element e: h[(f * 64 + l) * 8 + e]
    memset(img, 0, DftImage<SC>::BYTES);
    auto put = [&](int frag0, int l, int e, double w) {   // the planes of one constant into fragments frag0 .. frag0 + NPL - 1
        unsigned short pl[SC::NPL];
        dft_host::planes<SC>(w, pl);
        for (int p = 0; p < SC::NPL; p++) h[((size_t)(frag0 + p) * 64 + l) * 8 + e] = pl[p];
    };
    for (int l = 0; l < 64; l++) {
        const int lo = l & 15, g = l >> 4;
        // stage 1, B operand: column k1 = lo, rows k = 8 g + e <-> n1 = 4 g + e / 2, part = e % 2
        for (int e = 0; e < 8; e++) {
            const int n1 = 4 * g + e / 2, part = e & 1;
            const double th = PI2 * (double)((n1 * lo) % 16) / 16.0;
            put(0 * SC::NPL, l, e, part == 0 ? cos(th) : sin(th));     // real outputs:  Xre cos + Xim sin
            put(1 * SC::NPL, l, e, part == 0 ? -sin(th) : cos(th));    // imaginary:    -Xre sin + Xim cos
        }
        // stage 2, A operand: row m = lo <-> k2 = 16 b + lo, columns k = 8 g + e <-> n2 = 16 (e / 4) + 4 g + e % 4
        for (int b = 0; b < 2; b++)
            for (int e = 0; e < 8; e++) {
                const int k2 = 16 * b + lo, n2 = 16 * (e >> 2) + 4 * g + (e & 3);
                const bool on = k2 < DFT_K2 && n2 < DFT_K2;
                const double ph = PI2 * (double)((n2 * k2) % 30) / 30.0;
                put((2 + 2 * b + 0) * SC::NPL, l, e, on ? cos(ph) : 0.0);
                put((2 + 2 * b + 1) * SC::NPL, l, e, on ? sin(ph) : 0.0);
            }
    }
    float2 *tw = (float2 *)((uint4 *)img + 64 * DftImage<SC>::NFRAG);
    for (int r = 0; r < 8; r++)
        for (int l = 0; l < 64; l++) {
            const int n2 = 16 * (r >> 2) + 4 * (l >> 4) + (r & 3), k1 = l & 15;
            const double a = PI2 * (double)((n2 * k1) % 480) / 480.0;
            tw[r * 64 + l] = n2 < DFT_K2 ? make_float2((float)cos(a), (float)-sin(a)) : make_float2(0.0f, 0.0f);
        }
}

}  // namespace nnn
