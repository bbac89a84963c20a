// scripts/repro_graph_replay.cpp -- stand-alone repro attempt for the round-1 observation "replaying a captured MULTI-STREAM
// DAG back to back crashes the ROCm 7.2 runtime" (DESIGN.md round 1, section 4).  The product no longer uses hipGraphs at all
// (a frame group is seven eager launches), so this is documentation of the runtime behaviour, not a product path.
//
// Shape of the round-1 graph: a capture on stream A that forks to streams B and C through events, runs chains of small kernels
// on all three with cross-stream event waits in the middle (the frame-to-frame recurrences), joins back into A; instantiated
// once and launched many times back to back without a synchronise in between.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/repro_graph_replay.cpp -o /tmp/repro && /tmp/repro [replays] [chain]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("FAIL %s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)

__global__ void k_step(float *p, int n, float a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * a + 1.0f;
}

int main(int argc, char **argv)
{
    const int replays = argc > 1 ? atoi(argv[1]) : 2000, chain = argc > 2 ? atoi(argv[2]) : 8;
    const int n = 1 << 16;
    float *buf[3];
    hipStream_t st[3];
    hipEvent_t fork, join[2], mid[3][16];
    for (int i = 0; i < 3; i++) {
        CHK(hipMalloc((void **)&buf[i], n * sizeof(float)));
        CHK(hipMemset(buf[i], 0, n * sizeof(float)));
        CHK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        for (int k = 0; k < 16; k++) CHK(hipEventCreateWithFlags(&mid[i][k], hipEventDisableTiming));
    }
    CHK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) CHK(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));
    hipGraph_t g;
    hipGraphExec_t ex;
    CHK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
    CHK(hipEventRecord(fork, st[0]));
    CHK(hipStreamWaitEvent(st[1], fork, 0));
    CHK(hipStreamWaitEvent(st[2], fork, 0));
    for (int k = 0; k < chain && k < 16; k++)
        for (int i = 0; i < 3; i++) {
            if (k > 0) CHK(hipStreamWaitEvent(st[i], mid[(i + 2) % 3][k - 1], 0));   // lane i waits for lane i - 1's previous step
            hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, st[i], buf[i], n, 0.5f);
            CHK(hipEventRecord(mid[i][k], st[i]));
        }
    CHK(hipEventRecord(join[0], st[1]));
    CHK(hipEventRecord(join[1], st[2]));
    CHK(hipStreamWaitEvent(st[0], join[0], 0));
    CHK(hipStreamWaitEvent(st[0], join[1], 0));
    CHK(hipStreamEndCapture(st[0], &g));
    CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int r = 0; r < replays; r++) CHK(hipGraphLaunch(ex, st[0]));   // back to back, no synchronise
    CHK(hipStreamSynchronize(st[0]));
    float h = 0.0f;
    CHK(hipMemcpy(&h, buf[0], sizeof(float), hipMemcpyDeviceToHost));
    printf("OK: %d replays of a 3-stream, %d-step captured DAG completed; buf[0][0] = %f (expected 2.0)\n", replays, chain, h);
    return 0;
}
