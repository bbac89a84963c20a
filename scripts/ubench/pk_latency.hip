// Micro-benchmark (not part of the product; round 6): issue interval of v_pk_*_f32 against v_*_f32 by the number of independent chains a wave
// interleaves (1, 2, 4, 8) and by waves per SIMD -- does a packed instruction cost a SIMD one issue slot or two once waves compete, and do the
// op_sel / neg modifiers change that?
//   hipcc --offload-arch=gfx950 -O3 -o pk_latency pk_latency.hip && ./pk_latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE, int CH> __global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int n)
{
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; i++) { a[i] = out[(threadIdx.x + i) & 63]; p[i] = v2f{a[i], a[i] + 1.0f}; }
    const float one = out[64 + (threadIdx.x & 63)];
    const v2f one2 = v2f{one, one};
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 64; r++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(one));
                if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(one2));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(p[c]) : "v"(one2));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[0,1,1]" : "+v"(p[c]) : "v"(one2));
                if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(one));
                if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(p[c]) : "v"(one2));
            }
        }
    }
    long long t1 = clock64();
    float s = 0.0f;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[128 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int MODE, int CH> void run1(float *out, long long *cyc)
{
    for (int W = 1; W <= 4; W *= 2) {
        const int n = 64;
        hipLaunchKernelGGL((k<MODE, CH>), dim3(1), dim3(256 * W), 0, 0, out, cyc, n); hipDeviceSynchronize();
        hipLaunchKernelGGL((k<MODE, CH>), dim3(1), dim3(256 * W), 0, 0, out, cyc, n); hipDeviceSynchronize();
        long long cc[32]; hipMemcpy(cc, cyc, sizeof cc, hipMemcpyDeviceToHost);
        double s = 0;
        for (int w = 0; w < 4 * W; w++) s += (double)(cc[2 * w + 1] - cc[2 * w]);
        printf(" %6.2f", s / (4 * W) / ((double)n * 64 * CH) / W);   // SIMD ticks per wave instruction
    }
    printf(" |");
}
template <int MODE> void run(const char *name, float *out, long long *cyc)
{
    printf("%-52s", name);
    run1<MODE, 1>(out, cyc); run1<MODE, 2>(out, cyc); run1<MODE, 4>(out, cyc); run1<MODE, 8>(out, cyc);
    printf("\n");
}
int main()
{
    float *out; long long *cyc; hipMalloc(&out, (128 + 1024) * 4); hipMalloc(&cyc, 32 * 8);
    float h[128]; for (int i = 0; i < 128; i++) h[i] = i < 64 ? 1.0f + i * 1e-3f : 1.0000001f; hipMemcpy(out, h, sizeof h, hipMemcpyHostToDevice);
    printf("SIMD ticks (clock64) per wave instruction; columns: independent chains per wave 1 | 2 | 4 | 8, each at W = 1, 2, 4 waves per SIMD\n");
    run<0>("v_add_f32", out, cyc);
    run<4>("v_fma_f32", out, cyc);
    run<1>("v_pk_add_f32", out, cyc);
    run<2>("v_pk_add_f32 op_sel / neg_hi (x + (-i) a)", out, cyc);
    run<5>("v_pk_mul_f32 op_sel / neg_lo", out, cyc);
    run<3>("v_pk_fma_f32 op_sel_hi", out, cyc);
    return 0;
}
