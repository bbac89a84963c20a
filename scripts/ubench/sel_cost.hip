// Micro-benchmark (not part of the product; round 6): what a select costs a SIMD against the ways round it -- v_cndmask_b32 (VCC or a scalar
// pair as the mask), v_and_b32 with a lane mask of all ones / zeros, a multiply by 0 / 1, and the moves / integer instructions the kernels' index
// arithmetic is made of.  SIMD ticks per wave instruction, 8 independent chains per wave, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o sel_cost sel_cost.hip && ./sel_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int n)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = out[(threadIdx.x + i) & 63];
    const float one = out[64 + (threadIdx.x & 63)];
    const unsigned long long m = __ballot(one > 1.0f);   // a scalar pair the compiler cannot fold
    unsigned ai[8]; for (int i = 0; i < 8; i++) ai[i] = __float_as_uint(a[i]);
    const unsigned onei = __float_as_uint(one);
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[0]), "v"(one) : "vcc");
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 64; r++) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(one));
                if (MODE == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(one) : );
                if (MODE == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(one), "s"(m));
                if (MODE == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ai[c]) : "v"(onei));
                if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(one));
                if (MODE == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(a[c]) : "v"(one));
                if (MODE == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ai[c]) : "v"(onei));
                if (MODE == 7) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(ai[c]));
                if (MODE == 8) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(ai[c]) : "v"(onei));
                if (MODE == 9) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(ai[c]) : "v"(onei));
                if (MODE == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[c]), "v"(one) : "vcc");
                if (MODE == 11) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(*(unsigned long long *)&ai[0]) : "v"(a[c]), "v"(one));
                if (MODE == 12) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(one));
                if (MODE == 13) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[c]) : "v"(one));
                if (MODE == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(one) : "vcc");                 // compare + select through VCC (per instruction)
                if (MODE == 17) { unsigned long long mm; asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a[c]), "=&s"(mm) : "v"(one)); }   // ... through a scalar pair
                if (MODE == 18) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(one) : "vcc");
                if (MODE == 19) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_add_f32 %2, %2, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]), "+v"(a[(c + 1) & 7]) : "v"(one) : "vcc");
                if (MODE == 14) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[c]) : "v"(one));
                if (MODE == 15) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[c]) : "v"(one));
            }
        }
    }
    long long t1 = clock64();
    float s = 0.0f;
    for (int i = 0; i < 8; i++) s += a[i] + __uint_as_float(ai[i]);
    out[128 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int MODE> void run(const char *name, float *out, long long *cyc)
{
    printf("%-44s", name);
    for (int W = 1; W <= 4; W *= 2) {
        const int n = 64;
        hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(256 * W), 0, 0, out, cyc, n); hipDeviceSynchronize();
        hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(256 * W), 0, 0, out, cyc, n); hipDeviceSynchronize();
        long long cc[32]; hipMemcpy(cc, cyc, sizeof cc, hipMemcpyDeviceToHost);
        double s = 0;
        for (int w = 0; w < 4 * W; w++) s += (double)(cc[2 * w + 1] - cc[2 * w]);
        printf(" %6.2f", s / (4 * W) / ((double)n * 64 * 8) / W / (MODE >= 16 && MODE <= 19 ? 2 : 1));
    }
    printf("\n");
}
int main()
{
    float *out; long long *cyc; hipMalloc(&out, (128 + 1024) * 4); hipMalloc(&cyc, 32 * 8);
    float h[128]; for (int i = 0; i < 128; i++) h[i] = i < 64 ? 1.0f + i * 1e-3f : 1.0000001f; hipMemcpy(out, h, sizeof h, hipMemcpyHostToDevice);
    printf("SIMD ticks (clock64) per wave instruction, 8 independent chains per wave; W = 1, 2, 4 waves per SIMD\n");
    run<0>("v_add_f32", out, cyc); run<15>("v_sub_f32", out, cyc); run<4>("v_mul_f32", out, cyc); run<14>("v_max_f32", out, cyc); run<12>("v_fma_f32", out, cyc);
    run<1>("v_cndmask_b32 (vcc)", out, cyc); run<2>("v_cndmask_b32_e64 (scalar pair)", out, cyc);
    run<3>("v_and_b32", out, cyc); run<5>("v_mov_b32", out, cyc); run<13>("v_mov_b32 DPP", out, cyc);
    run<6>("v_add_u32", out, cyc); run<7>("v_lshlrev_b32", out, cyc); run<8>("v_lshl_add_u32", out, cyc); run<9>("v_mul_u32_u24", out, cyc);
    run<16>("v_cmp vcc + v_cndmask vcc (per pair / 2)", out, cyc); run<17>("v_cmp_e64 + v_cndmask_e64 (per pair / 2)", out, cyc);
    run<18>("v_cmp vcc, s_nop, v_cndmask vcc (per pair / 2)", out, cyc); run<19>("v_cmp vcc, v_add, v_cndmask vcc (per 3 / 2)", out, cyc);
    run<10>("v_cmp_lt_f32 (vcc)", out, cyc); run<11>("v_cmp_lt_f32_e64 (scalar pair)", out, cyc);
    return 0;
}
