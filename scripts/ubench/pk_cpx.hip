// Micro-benchmark and known-answer check (not part of the product; round 6): the transforms' complex arithmetic as hand-packed FP32 --
// a complex value lives in one 64-bit register pair, an add / subtract / multiply by -i is ONE v_pk_add_f32 (op_sel swaps the halves, neg_lo /
// neg_hi negate them), a complex product one v_pk_mul_f32 + one v_pk_fma_f32 -- against the scalar code the product compiles today
// (-fno-slp-vectorize).  Same roundings per component: the two must agree bit for bit.  Work unit: seven twiddle products + an 8-point DFT.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o pk_cpx pk_cpx.hip && ./pk_cpx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float v2f __attribute__((ext_vector_type(2)));
#define DI __device__ __forceinline__
// ---- scalar (the product's code)
DI float2 cmulf(float2 a, float2 w) { return make_float2(fmaf(a.x, w.x, -a.y * w.y), fmaf(a.x, w.y, a.y * w.x)); }
DI float2 cadd(float2 a, float2 c) { return make_float2(a.x + c.x, a.y + c.y); }
DI float2 csub(float2 a, float2 c) { return make_float2(a.x - c.x, a.y - c.y); }
DI float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
DI void bfly2(float2 &a, float2 &c) { float2 t = csub(a, c); a = cadd(a, c); c = t; }
DI void dft8(float2 *v)
{
    const float h = 0.70710678118654752440f;
    float2 a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7];
    bfly2(a0, a4); bfly2(a1, a5); bfly2(a2, a6); bfly2(a3, a7);
    a5 = make_float2((a5.x + a5.y) * h, (a5.y - a5.x) * h);
    a6 = mul_mi(a6);
    a7 = make_float2((a7.y - a7.x) * h, (-a7.x - a7.y) * h);
    bfly2(a0, a2); bfly2(a1, a3); bfly2(a4, a6); bfly2(a5, a7);
    a3 = mul_mi(a3); a7 = mul_mi(a7);
    bfly2(a0, a1); bfly2(a2, a3); bfly2(a4, a5); bfly2(a6, a7);
    v[0] = a0; v[4] = a1; v[2] = a2; v[6] = a3; v[1] = a4; v[5] = a5; v[3] = a6; v[7] = a7;
}
// ---- packed
DI v2f padd(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DI v2f psub(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
DI v2f padd_mi(v2f x, v2f a) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(a)); return r; }   // x + (-i) a
DI v2f psub_mi(v2f x, v2f a) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(a)); return r; }   // x - (-i) a
DI v2f pnegadd_mi(v2f x, v2f a) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(r) : "v"(x), "v"(a)); return r; }   // -x + (-i) a
DI v2f pscale(v2f a, v2f c) { v2f r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "s"(c)); return r; }   // a * c.x (c wave-uniform)
DI v2f pcmul(v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));       // (-a.y w.y, a.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));                 // (a.x w.x + t.x, a.x w.y + t.y)
    return r;
}
DI void pdft8(v2f *v)
{
    const v2f h = {0.70710678118654752440f, 0.70710678118654752440f};
    v2f a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7];
#define BF(A, C) do { const v2f t_ = psub(A, C); A = padd(A, C); C = t_; } while (0)
#define BF_MI(A, C) do { const v2f t_ = psub_mi(A, C); A = padd_mi(A, C); C = t_; } while (0)   // the butterfly of A and (-i) C
    BF(a0, a4); BF(a1, a5); BF(a2, a6); BF(a3, a7);
    a5 = pscale(padd_mi(a5, a5), h);
    a7 = pscale(pnegadd_mi(a7, a7), h);
    BF(a0, a2); BF(a1, a3); BF_MI(a4, a6); BF(a5, a7);
    BF(a0, a1); BF_MI(a2, a3); BF(a4, a5); BF_MI(a6, a7);
    v[0] = a0; v[4] = a1; v[2] = a2; v[6] = a3; v[1] = a4; v[5] = a5; v[3] = a6; v[7] = a7;
}
template <int MODE> __global__ void __launch_bounds__(1024) k(float *io, long long *cyc, int n)
{
    const int tid = threadIdx.x;
    float2 w[7];
    for (int i = 0; i < 7; i++) w[i] = make_float2(io[16 + 2 * i], io[17 + 2 * i]);
    long long t0, t1;
    if (MODE == 0) {
        float2 v[8];
        for (int i = 0; i < 8; i++) v[i] = make_float2(io[2 * i] + tid * 1e-3f, io[2 * i + 1]);
        t0 = clock64();
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int r = 1; r < 8; r++) v[r] = cmulf(v[r], w[r - 1]);
            dft8(v);
#pragma unroll
            for (int r = 0; r < 8; r++) { asm volatile("" : "+v"(v[r].x), "+v"(v[r].y)); }
        }
        t1 = clock64();
        for (int i = 0; i < 8; i++) { io[64 + 16 * tid + 2 * i] = v[i].x; io[64 + 16 * tid + 2 * i + 1] = v[i].y; }
    } else {
        v2f v[8], pw[7];
        for (int i = 0; i < 7; i++) pw[i] = v2f{w[i].x, w[i].y};
        for (int i = 0; i < 8; i++) v[i] = v2f{io[2 * i] + tid * 1e-3f, io[2 * i + 1]};
        t0 = clock64();
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int r = 1; r < 8; r++) v[r] = pcmul(v[r], pw[r - 1]);
            pdft8(v);
#pragma unroll
            for (int r = 0; r < 8; r++) { asm volatile("" : "+v"(v[r])); }
        }
        t1 = clock64();
        for (int i = 0; i < 8; i++) { io[64 + 16 * tid + 2 * i] = v[i].x; io[64 + 16 * tid + 2 * i + 1] = v[i].y; }
    }
    if ((tid & 63) == 0) { cyc[2 * (tid >> 6)] = t0; cyc[2 * (tid >> 6) + 1] = t1; }
}
int main()
{
    float *io; long long *cyc; const int NIO = 64 + 16 * 1024;
    hipMalloc(&io, NIO * 4); hipMalloc(&cyc, 32 * 8);
    float h[64];
    for (int i = 0; i < 16; i++) h[i] = 0.3f + 0.17f * i - 0.011f * i * i;
    for (int i = 0; i < 7; i++) { h[16 + 2 * i] = cosf(0.37f * (i + 1)); h[17 + 2 * i] = -sinf(0.37f * (i + 1)); }   // unit twiddles: the values stay bounded
    static float out[2][64 + 16 * 1024];
    printf("clock64 ticks of a wave's own time per unit of work (7 complex products + an 8-point DFT: 92 scalar / 41 packed arithmetic instructions)\n");
    printf("%-24s %8s %8s %8s\n", "waves per SIMD", "1", "2", "4");
    for (int mode = 0; mode < 2; mode++) {
        printf("%-24s", mode ? "hand-packed" : "scalar");
        for (int W = 1; W <= 4; W *= 2) {
            const int n = 512;
            hipMemcpy(io, h, sizeof h, hipMemcpyHostToDevice);
            if (mode) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256 * W), 0, 0, io, cyc, n); else hipLaunchKernelGGL(k<0>, dim3(1), dim3(256 * W), 0, 0, io, cyc, n);
            hipDeviceSynchronize();
            long long cc[32]; hipMemcpy(cc, cyc, sizeof cc, hipMemcpyDeviceToHost);
            double s = 0;
            for (int w = 0; w < 4 * W; w++) s += (double)(cc[2 * w + 1] - cc[2 * w]);
            printf(" %8.1f", s / (4 * W) / n);
            if (W == 1) hipMemcpy(out[mode], io, NIO * 4, hipMemcpyDeviceToHost);
        }
        printf("\n");
    }
    int bad = 0;
    for (int i = 64; i < 64 + 16 * 256; i++) bad += memcmp(&out[0][i], &out[1][i], 4) != 0;
    printf("values that differ between the two (of %d, after 512 rounds): %d   e.g. %.9g %.9g\n", 16 * 256, bad, out[0][64 + 5], out[1][64 + 5]);
    return bad != 0;
}
