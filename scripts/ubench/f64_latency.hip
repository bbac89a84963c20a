// Micro-benchmark (not part of the product): issue rate and dependent latency of the f64 instructions of the high-pass recurrence,
// one wave alone on a SIMD.  hipcc --offload-arch=gfx950 -O3 -o f64_latency f64_latency.hip && ./f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int MODE> __global__ void k(double *out, long long *cyc, int n)
{
    double a = out[threadIdx.x], b = a + 1.0, c = a + 2.0, d = a + 3.0;
    float f = (float)a, g = f + 1.0f, h = f + 2.0f, e = f + 3.0f;
    const double one = out[64 + threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(one));
            if (MODE == 1) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(f)); }
            if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(one));
            if (MODE == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(one));
            if (MODE == 4) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(one)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(b) : "v"(one));
                             asm volatile("v_add_f64 %0, %0, %1" : "+v"(c) : "v"(one)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(one)); }
            if (MODE == 5) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(g) : "v"(b));
                             asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(h) : "v"(c)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(e) : "v"(d)); }
            if (MODE == 6) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(f)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(b) : "v"(g));
                             asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c) : "v"(h)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(e)); }
            if (MODE == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(g));
            if (MODE == 8) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(one)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(b) : "v"(one));
                             asm volatile("v_mul_f64 %0, %0, %1" : "+v"(c) : "v"(one)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(one)); }
            if (MODE == 9) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(one)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(b) : "v"(one));
                             asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(c) : "v"(one)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(one)); }
            if (MODE == 10) {   // the recurrence's chain as the compiler issues it: cvt, add, mul, fma, add, cvt
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(f)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(one));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(one)); asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a) : "v"(one));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(one)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a)); }
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = a + b + c + d + (double)(f + g + h + e);
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name, int per_rep, double *out, long long *cyc, int lanes = 64)
{
    const int n = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(lanes), 0, 0, out, cyc, n); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(lanes), 0, 0, out, cyc, n * 16); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double ops = (double)n * 16 * REP * per_rep;
    printf("%-44s %7.2f ns per instruction   (%6.2f clock64 ticks)\n", name, ms * 1e6 / ops, (double)c / ops);
}
int main()
{
    double *out; long long *cyc; hipMalloc(&out, 128 * 8); hipMalloc(&cyc, 8);
    double h[128]; for (int i = 0; i < 128; i++) h[i] = i < 64 ? 1.0 + i * 1e-3 : 1.0000001; hipMemcpy(out, h, sizeof h, hipMemcpyHostToDevice);
    run<0>("v_add_f64, dependent", 1, out, cyc);
    run<3>("v_mul_f64, dependent", 1, out, cyc);
    run<2>("v_fma_f64, dependent", 1, out, cyc);
    run<1>("v_cvt_f32_f64 + v_cvt_f64_f32, dependent", 2, out, cyc);
    run<7>("v_add_f32, dependent", 1, out, cyc);
    run<4>("v_add_f64, 4 independent chains", 4, out, cyc);
    run<8>("v_mul_f64, 4 independent chains", 4, out, cyc);
    run<9>("v_fma_f64, 4 independent chains", 4, out, cyc);
    run<5>("v_cvt_f32_f64, independent", 4, out, cyc);
    run<6>("v_cvt_f64_f32, independent", 4, out, cyc);
    run<10>("recurrence chain (6 instructions)", 6, out, cyc);
    // does a wave with fewer live lanes issue faster?  (the high-pass recurrence is one chain per stream: 32-lane waves would be twice as many)
    run<4>("v_add_f64, 4 independent chains, 32 lanes", 4, out, cyc, 32);
    run<4>("v_add_f64, 4 independent chains, 16 lanes", 4, out, cyc, 16);
    run<10>("recurrence chain, 32 lanes", 6, out, cyc, 32);
    run<10>("recurrence chain, 16 lanes", 6, out, cyc, 16);
    run<5>("v_cvt_f32_f64, independent, 32 lanes", 4, out, cyc, 32);
    return 0;
}
