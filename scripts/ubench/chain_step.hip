// Micro-benchmark (not part of the product): cost per step of k_pitch's serial energy scans, by how the four terms of a quad reach the adding lane.
// A wave runs ONE dependent chain y = y + t0; y = y + t1; ... (optionally clamped: y = max(y + t, 1)); the terms of four consecutive steps are made by
// the four lanes of a quad.  Variants: (0) four v_mov_b32 DPP moves, then plain adds (the product until round 6); (1) the same with v_max_f32 DPP
// (x, x) in the move's place; (2) v_or_b32 DPP; (3) through LDS: one ds_write_b32 per lane, one ds_read_b128 per lane, a group ahead of the adds;
// (4) the adds alone (terms in registers: the floor).
//   hipcc --offload-arch=gfx950 -O3 -o chain_step chain_step.hip && ./chain_step
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, bool CLAMP> __global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int n)
{
    __shared__ float xch[16][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float y = out[lane & 3] , t = out[4 + lane], one = out[70];
    float *xw = &xch[wave][0][lane];
    const float4 *xr = (const float4 *)&xch[wave][0][lane & ~3];
    xw[0] = t; xw[64] = t;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float a, b, c, d;
            t = t * one;   // (the step's own term: one multiply)
            if (MODE == 0) {
                asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(t));
            } else if (MODE == 1) {
                asm volatile("v_max_f32_dpp %0, %4, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %4, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                             "v_max_f32_dpp %2, %4, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %3, %4, %4 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(t));
            } else if (MODE == 2) {
                asm volatile("v_or_b32_dpp %0, %4, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_or_b32_dpp %1, %4, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                             "v_or_b32_dpp %2, %4, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\tv_or_b32_dpp %3, %4, %4 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(t));
            } else if (MODE == 3) {
                // this group's terms were written a trip ago; write the next group's, then read this one's (LDS operations of a wave run in order)
                const float4 q = xr[(r & 1) * 16];
                xw[((r + 1) & 1) * 64] = t;
                a = q.x; b = q.y; c = q.z; d = q.w;
            } else { a = t; b = one; c = t; d = one; }
            if (CLAMP) { y = fmaxf(y + a, 1.0f); y = fmaxf(y + b, 1.0f); y = fmaxf(y + c, 1.0f); y = fmaxf(y + d, 1.0f); }
            else { y = y + a; y = y + b; y = y + c; y = y + d; }
            asm volatile("" : "+v"(y));
        }
    }
    long long t1 = clock64();
    out[128 + threadIdx.x] = y + t;
    if (lane == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}
template <int MODE, bool CLAMP> void run(const char *name, float *out, long long *cyc)
{
    const int n = 256;
    printf("%-58s", name);
    for (int W = 1; W <= 4; W *= 2) {
        hipLaunchKernelGGL((k<MODE, CLAMP>), dim3(1), dim3(256 * W), 0, 0, out, cyc, 4); hipDeviceSynchronize();
        hipLaunchKernelGGL((k<MODE, CLAMP>), dim3(1), dim3(256 * W), 0, 0, out, cyc, n); hipDeviceSynchronize();
        long long cc[32]; hipMemcpy(cc, cyc, sizeof cc, hipMemcpyDeviceToHost);
        double s = 0;
        for (int w = 0; w < 4 * W; w++) s += (double)(cc[2 * w + 1] - cc[2 * w]);
        printf("  %6.2f", s / (4 * W) / ((double)n * 16 * 4));   // a wave's own ticks per chain step
    }
    printf("\n");
}
int main()
{
    float *out; long long *cyc; hipMalloc(&out, (128 + 1024) * 4); hipMalloc(&cyc, 32 * 8);
    float h[128]; for (int i = 0; i < 128; i++) h[i] = 1.0f + i * 1e-3f; h[70] = 1.0000001f; hipMemcpy(out, h, sizeof h, hipMemcpyHostToDevice);
    printf("clock64 ticks of a wave's own time per chain step (four steps per group; W = waves per SIMD, every wave running its own chain)\n");
    printf("%-58s  %s\n", "how the quad's four terms reach the adding lane", "   W=1     W=2     W=4");
    run<0, false>("v_mov_b32 DPP x 4, adds", out, cyc);
    run<1, false>("v_max_f32 DPP (x, x) x 4, adds", out, cyc);
    run<2, false>("v_or_b32 DPP (x, x) x 4, adds", out, cyc);
    run<3, false>("ds_write_b32 + ds_read_b128 a group ahead, adds", out, cyc);
    run<4, false>("terms in registers (floor), adds", out, cyc);
    run<0, true>("v_mov_b32 DPP x 4, add + clamp", out, cyc);
    run<1, true>("v_max_f32 DPP (x, x) x 4, add + clamp", out, cyc);
    run<2, true>("v_or_b32 DPP (x, x) x 4, add + clamp", out, cyc);
    run<3, true>("ds_write_b32 + ds_read_b128 a group ahead, add + clamp", out, cyc);
    run<4, true>("terms in registers (floor), add + clamp", out, cyc);
    return 0;
}
