// nnn_mfma.h -- the CDNA4-specific primitives of the kernels: one matrix instruction, an LDS-only barrier and
// explicitly-global memory accesses.
#pragma once
#include <hip/hip_runtime.h>

namespace nnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// v_mfma_f32_16x16x32_bf16: D(16x16 f32) = A(16x32 bf16) * B(32x16 bf16) + C, one wave.
// Fragments: lane l holds A[l & 15][8 (l >> 4) .. +7], B[8 (l >> 4) .. +7][l & 15] as 8 packed bf16,
// and C/D[4 (l >> 4) + q][l & 15] in element q.
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// ---- the primitives of the matrix-core DFT probe (scripts/ubench/nnn_dft_mfma.h: measured in round 5, not adopted) ----
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// v_mfma_f32_16x16x32_f16: the same tile shapes and fragment layouts as the bf16 form, operands in IEEE half precision
__device__ __forceinline__ f32x4 mfma_16x16x32_f16(uint4 a, uint4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// (a, b) rounded to nearest-even half precision, a in the low half
__device__ __forceinline__ unsigned pk_f16_rn(float a, float b)
{
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a - (float)(low / high half of h), exact when h holds a's own rounding: one v_fma_mix_f32 each
__device__ __forceinline__ float f16_resid_lo(float a, unsigned h)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
    return r;
}
__device__ __forceinline__ float f16_resid_hi(float a, unsigned h)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
    return r;
}
// the upper halves of (a, b) as one word, a's in the low half: two floats truncated to bf16
__device__ __forceinline__ unsigned pk_bf16_trunc(float a, float b)
{
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// max(|a|, |b|, m) (a NaN operand is ignored)
__device__ __forceinline__ float amax3(float a, float b, float m)
{
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}
// lane i of every row of 16 lanes receives lane i + D's value of the same row (a lane whose source would be past the row's end keeps its
// own): one DPP move (row_shl) where a shuffle is an index computation and a trip through the LDS crossbar
template <int D> __device__ __forceinline__ float dpp_row_down(float x)
{
    static_assert(D >= 1 && D <= 15, "");
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0x100 + D, 0xf, 0xf, false));
}
// x of lane `src` (any lane of the wave, taken modulo 64): the bare ds_bpermute_b32 -- one address instruction at most where __shfl /
// __shfl_up re-derive the lane number and clamp against a width every time (eleven vector instructions a call in k_fft_xp's feature head)
__device__ __forceinline__ float wave_read(float x, int src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, x)));
}
// maximum over the wave's 64 lanes, wave-uniform (four DPP steps inside the rows of 16, then the four rows through scalars)
__device__ __forceinline__ unsigned wave_max_u32(unsigned x)
{
    x = max(x, (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    x = max(x, (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    x = max(x, (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xf, 0xf, true));   // row_half_mirror
    x = max(x, (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xf, 0xf, true));   // row_mirror
    const unsigned a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16), c = __builtin_amdgcn_readlane(x, 32),
                   d = __builtin_amdgcn_readlane(x, 48);
    return max(max(a, b), max(c, d));
}

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier, no vmcnt(0).  Unlike
// __syncthreads() it does not drain outstanding global loads/stores, so requests issued early (weights, states)
// keep travelling across phase boundaries.  Only for phases that exchange data through LDS.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Orders one wave's LDS traffic: lanes of a wave may read what other lanes of the SAME wave wrote before this point.
// s_waitcnt lgkmcnt(0) (every LDS access this wave has issued is complete) and a compiler barrier; no s_barrier, and
// no vmcnt drain: the other waves of the block are not involved (each works on its own LDS region).  Relying on the
// LDS unit's in-order execution alone (no wait at all) produced rare single-stream glitches on some runs.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Loads/stores through pointers that reach a kernel inside a struct read from memory (the per-frame StepParams):
// the compiler cannot prove them global and would emit flat_* instructions, which also occupy the LDS counter.
template <class T> __device__ __forceinline__ T ld_global(const void *p)
{
    return *(const __attribute__((address_space(1))) T *)p;
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_global_u4(const void *p)   // 16 bytes, 16-byte aligned
{
    const u32x4_t v = *(const __attribute__((address_space(1))) u32x4_t *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <class T> __device__ __forceinline__ void st_global(void *p, T v)
{
    *(__attribute__((address_space(1))) T *)p = v;
}

// Opaque identity on a per-lane / wave-uniform value.  Inside a loop over the frames of a group it makes everything derived
// from the value loop-variant for the compiler: otherwise every address of every phase is hoisted out of the frame loop as
// loop invariant, hundreds of registers wide, and spilled (measured: 300 spills in k_rnn without it, none with it).
__device__ __forceinline__ void wf_setprio_high() { __builtin_amdgcn_s_setprio(3); }
template <int P> __device__ __forceinline__ void wave_prio() { __builtin_amdgcn_s_setprio(P); }   // issue priority of the calling wave (0 .. 3)
__device__ __forceinline__ int launder_v(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}
// marks a value as used (an 8-byte LDS read whose second half is not needed stays an 8-byte read)
__device__ __forceinline__ void keep_v(float x) { asm volatile("" ::"v"(x)); }
// makes a value opaque where it stands (the compiler neither moves its producer below this point nor folds it into its consumer)
__device__ __forceinline__ void keep_rw(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ int launder_s(int x)
{
    asm volatile("" : "+s"(x));
    return x;
}

// Packed FP32 (two lanes of a 64-bit register pair per instruction), spelled out where the pairing matters: the
// compiler's own SLP pairing of scalar code put the two products of one v_pk_mul_f32 in registers that the next
// instruction wanted somewhere else (one v_mov per product in the inner-product loops).  Products and sums round exactly
// like v_mul_f32 / v_add_f32.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f mk2(float x, float y) { v2f r = {x, y}; return r; }
__device__ __forceinline__ float smul(float a, float b)
{
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a single add that stays a single add (sequential sums of packed products: the compiler's pairing across accumulators
// would cost register moves)
__device__ __forceinline__ float sadd(float a, float b)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#ifndef NNN_NO_PK
#define NNN_NO_PK 0
#endif
#if NNN_NO_PK   // experiment (round 6): every packed helper as two plain instructions -- same roundings, half-rate issue slots each
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b) { return mk2(smul(a.x, b.x), smul(a.y, b.y)); }
__device__ __forceinline__ v2f pk_mul_bx(v2f a, v2f b) { return mk2(smul(a.x, b.x), smul(a.x, b.y)); }
__device__ __forceinline__ v2f pk_mul_by(v2f a, v2f b) { return mk2(smul(a.y, b.x), smul(a.y, b.y)); }
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) { return mk2(sadd(a.x, b.x), sadd(a.y, b.y)); }
__device__ __forceinline__ v2f pk_add_bx(v2f a, v2f b) { return mk2(sadd(a.x, b.x), sadd(a.y, b.x)); }
__device__ __forceinline__ v2f pk_add_by(v2f a, v2f b) { return mk2(sadd(a.x, b.y), sadd(a.y, b.y)); }
#else
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b)       // (a.x b.x, a.y b.y)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_mul_bx(v2f a, v2f b)    // (a.x b.x, a.x b.y): a's low half against both of b
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_mul_by(v2f a, v2f b)    // (a.y b.x, a.y b.y)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_add(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_add_bx(v2f a, v2f b)    // (a.x + b.x, a.y + b.x): b's low half onto both of a
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_add_by(v2f a, v2f b)    // (a.x + b.y, a.y + b.y)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#endif

// ---- the certified coarse pitch search's primitives (k_pitch, round 6) ----
// (a, b) rounded to nearest-even bf16, a in the low half: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16_rn(float a, float b)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// bits [sh + 31 : sh] of the 64-bit value hi:lo (sh in 0 .. 31): v_alignbit_b32
__device__ __forceinline__ unsigned align_bits(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
// LDS atomics of one workgroup (relaxed: a barrier orders them against the readers)
__device__ __forceinline__ unsigned lds_add_u32(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_or_u32(unsigned *p, unsigned v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// x of lane (lane ^ 16) / (lane ^ 32)
__device__ __forceinline__ float wave_xor16(float x, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, __builtin_bit_cast(int, x))); }
__device__ __forceinline__ float wave_xor32(float x, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, x))); }
// the four exchange steps of a reduction over a row of 16 lanes, as DPP moves: partner lane ^ 1, lane ^ 2, 7 - lane within its eight, 15 - lane
// within its row (a reduction needs disjoint partners, not a butterfly)
template <int STEP> __device__ __forceinline__ float row_partner(float x)
{
    const int b = __builtin_bit_cast(int, x);
    constexpr int ctl = STEP == 0 ? 0xB1 : STEP == 1 ? 0x4E : STEP == 2 ? 0x141 : 0x140;
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(b, ctl, 0xf, 0xf, true));
}
// y of lane K of the caller's quad (lanes 4 i .. 4 i + 3): a DPP operand of the instruction that uses it (the serial energy scans of k_pitch:
// the four lanes of a quad prepare four consecutive steps, every lane adds them in order)
template <int K> __device__ __forceinline__ float quad_lane(float y)
{
    const int b = __builtin_bit_cast(int, y);
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(b, K * 85, 0xf, 0xf, true));
}
// y of all four lanes of the caller's quad
struct Quad4 { float t0, t1, t2, t3; };
__device__ __forceinline__ Quad4 quad_all(float y) { Quad4 r = {quad_lane<0>(y), quad_lane<1>(y), quad_lane<2>(y), quad_lane<3>(y)}; return r; }
// how many lanes below the caller have their bit set in a wave-wide mask (v_mbcnt_lo / _hi)
__device__ __forceinline__ unsigned lane_rank(unsigned long long m, int) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }
// the value lane `l` holds, wave-uniform (l a compile-time constant)
__device__ __forceinline__ float lane_value(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32, 1 ulp
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32, 1 ulp

// Frame-to-frame hand-off between workgroups of one launch (k_pitch): the producer makes its results visible device-wide
// and then stores the flag; the consumer polls the flag and only then reads the results.
__device__ __forceinline__ void flag_publish(int *flag, int value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int flag_read(const int *flag)
{
    return __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
// next value of a device-wide counter (one lane calls it)
__device__ __forceinline__ unsigned ticket_take(unsigned *ctr)
{
    return __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true when the predicate holds on any active lane of the wave (wave-uniform)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
// bit l: the predicate on lane l (wave-uniform)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ void chain_pause() { __builtin_amdgcn_s_sleep(4); }
// the constant 100 MHz clock (s_memrealtime): wall time, whatever the shader clock does
__device__ __forceinline__ long long realtime_ticks() { return (long long)__builtin_amdgcn_s_memrealtime(); }

}  // namespace nnn
