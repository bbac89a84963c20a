// Micro-benchmark (not part of the product): vector-ALU issue cost per wave instruction by instruction class and by waves per SIMD.
// One workgroup of 256 * W threads on one CU = W waves on every SIMD, each wave running 8 independent chains of one instruction class.
// Question it answers (round 6): does a SIMD take 4 cycles for every wave instruction (round 5's ceiling model), or 2 for plain
// f32 and 4 for packed f32 once more than one wave feeds it?
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int n, int live)
{
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; i++) { a[i] = out[(threadIdx.x + i) & 63]; p[i] = v2f{a[i], a[i] + 1.0f}; }
    const float one = out[64 + (threadIdx.x & 63)];
    const v2f one2 = v2f{one, one};
    __shared__ float lds[4096];
    lds[threadIdx.x] = one; lds[threadIdx.x + 1024] = one; lds[threadIdx.x + 2048] = one; lds[threadIdx.x + 3072] = one;
    __syncthreads();
    const float *lp = lds + (threadIdx.x & 1023);
    long long t0 = clock64();
    if ((int)(threadIdx.x & 63) < live) {
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int r = 0; r < REP; r++) {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(one));
                    if (MODE == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(one));
                    if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(one));
                    if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(one2));
                    if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(one2));
                    if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[c]) : "v"(one2));
                    if (MODE == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[c]) : "v"(one));
                    if (MODE == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(one) : );
                    if (MODE == 8) {   // the inner-product step of k_pitch: one packed multiply, two ordered adds
                        v2f pr;
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(p[c]), "v"(one2));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(pr.x));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(pr.y));
                    }
                    if (MODE == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(a[c]) : "v"(p[c].x));
                    if (MODE == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[c]) : "v"(one));
                    if (MODE == 11) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1" : "+v"(a[c]), "+v"(p[c].x) : "v"(one));
                    if (MODE == 12) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[c]) : "v"(one));
                    if (MODE == 13) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(one));
                }
            }
        }
    }
    long long t1 = clock64();
    float s = 0.0f;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[128 + threadIdx.x] = s + lp[0];
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int MODE> void run(const char *name, int per, float *out, long long *cyc)
{
    const int n = 64;
    printf("%-40s", name);
    for (int live = 64; live >= 16; live -= 48)
        for (int W = 1; W <= 4; W *= 2) {
            hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256 * W), 0, 0, out, cyc, 4, live); hipDeviceSynchronize();
            hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256 * W), 0, 0, out, cyc, n, live); hipDeviceSynchronize();
            long long cc[32]; hipMemcpy(cc, cyc, sizeof cc, hipMemcpyDeviceToHost);
            long long lo = cc[0], hi = cc[1];
            for (int w = 0; w < 4 * W; w++) { lo = cc[2 * w] < lo ? cc[2 * w] : lo; hi = cc[2 * w + 1] > hi ? cc[2 * w + 1] : hi; }
            const long long c = hi - lo;
            // cycles the SIMD spends per wave instruction: (last wave's end - first wave's start) / (instructions per wave * waves on its SIMD)
            printf("  %5.2f", (double)c / ((double)n * REP * 8 * per * W));
        }
    printf("\n");
}
int main()
{
    float *out; long long *cyc; hipMalloc(&out, (128 + 1024) * 4); hipMalloc(&cyc, 32 * 8);
    float h[128]; for (int i = 0; i < 128; i++) h[i] = i < 64 ? 1.0f + i * 1e-3f : 1.0000001f; hipMemcpy(out, h, sizeof h, hipMemcpyHostToDevice);
    printf("clock64 ticks of SIMD time per wave instruction (s_memtime = shader clock)\n");
    printf("%-40s  %s\n", "class", "64 lanes: W=1   W=2   W=4 | 16 lanes: W=1   W=2   W=4   (W = waves per SIMD)");
    run<0>("v_add_f32", 1, out, cyc);
    run<1>("v_mul_f32", 1, out, cyc);
    run<2>("v_fma_f32", 1, out, cyc);
    run<6>("v_max_f32", 1, out, cyc);
    run<7>("v_cndmask_b32", 1, out, cyc);
    run<9>("v_mov_b32", 1, out, cyc);
    run<10>("v_and_b32", 1, out, cyc);
    run<12>("v_alignbit_b32", 1, out, cyc);
    run<13>("v_perm_b32", 1, out, cyc);
    run<3>("v_pk_add_f32", 1, out, cyc);
    run<4>("v_pk_mul_f32", 1, out, cyc);
    run<5>("v_pk_fma_f32", 1, out, cyc);
    run<8>("v_pk_mul_f32 + 2 v_add_f32 (per instr)", 3, out, cyc);
    run<11>("v_mul_f32 + v_add_f32 (per instr)", 2, out, cyc);
    return 0;
}
