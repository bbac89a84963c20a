// tests/hostsim/hip/hip_runtime.h -- TEST-ONLY stand-in for <hip/hip_runtime.h>.
//
// Lets the unmodified product sources (nnnoiseless_amd/csrc/*.hip) be compiled with g++ and
// executed on the CPU by a tiny SIMT interpreter: every thread of a workgroup is a ucontext
// fiber, __syncthreads() and the wave shuffles are rendezvous points, __shared__ is a
// function-local static.  It exists so kernel LOGIC (indexing, summation order, barriers)
// can be checked against the oracle in the GPU-less build container.  It is not a product
// path: nothing under nnnoiseless_amd/ knows it exists, and the package loader only ever
// opens the hipcc-built library.
#pragma once

#include <ucontext.h>

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __constant__ static
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)hostsim::dyn_lds;

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = {x, y}; return r; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
typedef void *hipStream_t;
typedef struct hostsim_event { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace hostsim {
struct Sim {
    uint3_ threadIdx, blockIdx;
    dim3 blockDim, gridDim;
};
extern Sim g;
extern char dyn_lds[160 * 1024];
void yield_barrier();
unsigned long long yield_shfl_u64(unsigned long long v, int src_lane_or_mask, int mode, int width);
void run_grid(dim3 grid, dim3 block, const std::function<void()> &body);
// every live lane of the calling wave deposits (in, out); fn runs once with all of them
typedef void (*WaveFn)(const void *const *ins, void *const *outs, int nlanes);
void wave_collective(const void *in, void *out, WaveFn fn);
}  // namespace hostsim

#define threadIdx (hostsim::g.threadIdx)
#define blockIdx (hostsim::g.blockIdx)
#define blockDim (hostsim::g.blockDim)
#define gridDim (hostsim::g.gridDim)

static inline void __syncthreads() { hostsim::yield_barrier(); }

// wave shuffles (wave = 64 consecutive threads); mode 0 = idx, 1 = xor, 2 = down, 3 = up
template <class T> static inline T hostsim_shfl(T v, int a, int mode, int width) {
    static_assert(sizeof(T) <= 8, "shfl payload");
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    bits = hostsim::yield_shfl_u64(bits, a, mode, width);
    T r;
    memcpy(&r, &bits, sizeof(T));
    return r;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) { return hostsim_shfl(v, src, 0, width); }
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { return hostsim_shfl(v, mask, 1, width); }
template <class T> static inline T __shfl_down(T v, int d, int width = 64) { return hostsim_shfl(v, d, 2, width); }
template <class T> static inline T __shfl_up(T v, int d, int width = 64) { return hostsim_shfl(v, d, 3, width); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // only used on uniform values

template <class T> static inline T min(T a, T b) { return b < a ? b : a; }
template <class T> static inline T max(T a, T b) { return a < b ? b : a; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline int hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return 0; }

static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline unsigned long long __builtin_readcyclecounter() { return 0; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }

// ---- runtime API subset -------------------------------------------------------------------
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2 };
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < h; r++) memmove((char *)d + r * dp, (const char *)s + r * sp, w);
    return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 2; return hipSuccess; }   // (two ordinals, one interpreter: what the node tests need)
static inline hipError_t hipDeviceGetPCIBusId(char *, int, int) { return 1; }   // (the interpreter's "device" has no PCI function)
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
hipError_t hipEventCreate(hipEvent_t *e);
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }

// graphs: the simulator has none; report "unsupported" so the host code takes its eager path
typedef void *hipGraph_t;
typedef void *hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return 1; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *) { return 1; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, void *, void *, size_t) { return 1; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

template <class... KArgs, class... Args>
static inline void hipLaunchKernelGGL(void (*k)(KArgs...), dim3 grid, dim3 block, size_t, hipStream_t, Args... args) {
    hostsim::run_grid(grid, block, [&]() { k(static_cast<KArgs>(args)...); });
}
