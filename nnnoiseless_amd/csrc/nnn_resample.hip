// nnn_resample.hip -- batched 16-tap windowed-sinc resampler (include/nnn_resample.h).
// ref: src/nnnoiseless.rs:15-32, 40-46, 106-131 (Resample<RS>); dasp_interpolate 0.11.0 sinc::Sinc (not in the reference tree,
// restated from its published source: unpinned).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/nnn_resample.h"

extern "C" const char *nnn_last_error(void);
int nnn_set_error(const char *msg);   // nnn_batch.hip

namespace {

constexpr int RS_TAPS = 16, RS_DEPTH = RS_TAPS / 2;

// One output sample: the fold of Sinc::interpolate.  `e` points at the ring as it stands (frames[i] = e[i], i < 16, oldest first);
// w[2 n] / w[2 n + 1] = the left / right tap weight of step n (f64), nd = max_depth, idx = the interpolator's idx.
__device__ __forceinline__ float sinc_fold(const float *e, const double *w, int nd, int idx)
{
    float v = 0.0f;
    for (int n = 0; n < nd; n++) {
        v += (float)(w[2 * n] * (double)e[(idx - n) & (RS_TAPS - 1)]);          // frames[nl - n]
        v += (float)(w[2 * n + 1] * (double)e[(idx + 1 + n) & (RS_TAPS - 1)]);  // frames[nr + n] (the ring index wraps)
    }
    return v;
}

struct RsOut {   // per output sample of a call, shared by all streams
    int consumed;   // source samples of this call pushed before it
    int idx, nd;    // interpolator idx and max_depth at that point
    int pad;
};

__global__ void k_resample(const float *in, size_t in_stride, const float *hist, float *out, size_t out_stride, const RsOut *sched,
                           const double *w, int n_out, int n_streams)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (m >= n_out || s >= n_streams) return;
    const RsOut o = sched[m];
    float e[RS_TAPS];
#pragma unroll
    for (int i = 0; i < RS_TAPS; i++) {
        const int p = o.consumed - RS_TAPS + i;   // position in this call's input; negative: the 16 samples before it
        e[i] = p >= 0 ? in[(size_t)s * in_stride + p] : hist[(size_t)s * RS_TAPS + (RS_TAPS + p)];
    }
    out[(size_t)s * out_stride + m] = sinc_fold(e, w + (size_t)m * RS_TAPS, o.nd, o.idx);
}

// the last 16 source samples (of history ++ this call's input) become the next call's history
__global__ void k_resample_hist(const float *in, size_t in_stride, float *hist, long n_in, int n_streams)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, s = t / RS_TAPS, i = t % RS_TAPS;
    if (s >= n_streams) return;
    const long p = n_in - RS_TAPS + i;
    const float v = p >= 0 ? in[(size_t)s * in_stride + p] : hist[(size_t)s * RS_TAPS + (RS_TAPS + p)];
    __syncthreads();   // (a block covers whole streams: 256 threads = 16 streams; every read of hist precedes the writes)
    hist[(size_t)s * RS_TAPS + i] = v;
}

}  // namespace

struct nnn_resampler {
    int n_streams = 0, device = 0;
    double ratio = 1.0, pos = 0.0;
    int idx = 0;
    long credit = 0;                       // source samples already pushed on behalf of the next output (they arrived early)
    float *hist = nullptr;                 // device [n_streams][16]
    RsOut *sched = nullptr;
    double *w = nullptr;
    long cap = 0;                          // outputs the schedule buffers hold
    hipStream_t stream = nullptr;
    std::vector<float> stage_in, stage_out;
    float *d_in = nullptr, *d_out = nullptr;
    size_t d_in_cap = 0, d_out_cap = 0;
};

static int rfail(const char *msg) { return nnn_set_error(msg); }

extern "C" nnn_resampler *nnn_resampler_create(int n_streams, double ratio, int device)
{
    if (n_streams <= 0 || !(ratio > 0.0)) { rfail("resampler: n_streams and ratio must be positive"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { rfail("resampler: no such HIP device"); return nullptr; }
    nnn_resampler *r = new nnn_resampler();
    r->n_streams = n_streams;
    r->ratio = ratio;
    r->device = device;
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&r->hist, (size_t)n_streams * RS_TAPS * sizeof(float)) != hipSuccess ||
        hipMemset(r->hist, 0, (size_t)n_streams * RS_TAPS * sizeof(float)) != hipSuccess) {
        rfail("resampler: allocation failed");
        nnn_resampler_destroy(r);
        return nullptr;
    }
    return r;
}

extern "C" void nnn_resampler_destroy(nnn_resampler *r)
{
    if (!r) return;
    hipSetDevice(r->device);
    if (r->stream) hipStreamSynchronize(r->stream);
    hipFree(r->hist); hipFree(r->sched); hipFree(r->w); hipFree(r->d_in); hipFree(r->d_out);
    if (r->stream) hipStreamDestroy(r->stream);
    delete r;
}

extern "C" int nnn_resampler_reset(nnn_resampler *r)
{
    if (!r) return rfail("null resampler");
    hipSetDevice(r->device);
    hipStreamSynchronize(r->stream);
    if (hipMemset(r->hist, 0, (size_t)r->n_streams * RS_TAPS * sizeof(float)) != hipSuccess) return rfail("resampler: reset failed");
    r->pos = 0.0;
    r->idx = 0;
    r->credit = 0;
    return 0;
}

extern "C" long nnn_resampler_max_output(const nnn_resampler *r, long n_in)
{
    return r ? (long)ceil(((double)n_in + 2.0) / r->ratio) + 2 : 0;
}

extern "C" int nnn_resampler_process_device(nnn_resampler *r, const float *d_in, long n_in, size_t in_stride, float *d_out,
                                            long cap_out, size_t out_stride, long *n_out, void *hip_stream)
{
    if (!r || !n_out) return rfail("null argument");
    *n_out = 0;
    if (n_in < 0 || cap_out < 0 || (n_in > 0 && !d_in) || (cap_out > 0 && !d_out)) return rfail("resampler: bad buffer");
    // every source sample of a call is consumed by it (the ring moves past all of them), so every output they complete must
    // have room: a smaller cap_out would silently drop outputs and leave the bookkeeping behind the ring
    if (cap_out < nnn_resampler_max_output(r, n_in)) return rfail("resampler: cap_out smaller than nnn_resampler_max_output(r, n_in)");
    if (hipSetDevice(r->device) != hipSuccess) return rfail("resampler: no such HIP device");
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : r->stream;
    // ---- the position sequence of Resample::next_sample (src/nnnoiseless.rs:107-120) and the weights of Sinc::interpolate
    std::vector<RsOut> sched;
    std::vector<double> w;
    const double pi = 3.14159265358979323846;
    double pos = r->pos;
    int idx = r->idx;
    long consumed = 0, credit = r->credit;
    while ((long)sched.size() < cap_out) {
        double p = pos + r->ratio;
        long need = 0;
        while (p >= 1.0) { p -= 1.0; need++; }
        const long need_now = need - credit;    // `credit` of them went into the ring with an earlier call
        if (consumed + need_now > n_in) break;  // this output waits for the next call's samples
        pos = p;
        consumed += need_now;
        credit = 0;
        for (long k = 0; k < need_now && idx < RS_DEPTH; k++) idx++;   // Sinc::next_source_frame
        const int nl = idx, nr = idx + 1, rightmost = nl + RS_DEPTH, leftmost = nr - RS_DEPTH;
        const int nd = rightmost >= RS_TAPS ? RS_TAPS - RS_DEPTH : (leftmost < 0 ? RS_DEPTH + leftmost : RS_DEPTH);
        RsOut o{(int)consumed, idx, nd, 0};
        sched.push_back(o);
        const double phil = pos, phir = 1.0 - pos;
        for (int n = 0; n < RS_DEPTH; n++) {
            double a = pi * (phil + (double)n);
            double first = a == 0.0 ? 1.0 : sin(a) / a, second = 0.5 + 0.5 * cos(a / (double)RS_DEPTH);
            w.push_back(first * second);
            a = pi * (phir + (double)n);
            first = a == 0.0 ? 1.0 : sin(a) / a;
            second = 0.5 + 0.5 * cos(a / (double)RS_DEPTH);
            w.push_back(first * second);
        }
    }
    const long n = (long)sched.size();
    if (n > r->cap) {
        hipStreamSynchronize(st);
        hipFree(r->sched); hipFree(r->w);
        r->sched = nullptr; r->w = nullptr; r->cap = 0;
        const long cap = n + n / 2 + 1024;
        if (hipMalloc((void **)&r->sched, (size_t)cap * sizeof(RsOut)) != hipSuccess ||
            hipMalloc((void **)&r->w, (size_t)cap * RS_TAPS * sizeof(double)) != hipSuccess)
            return rfail("resampler: allocation failed");
        r->cap = cap;
    }
    if (n > 0) {
        // (pageable host vectors: hipMemcpyAsync from them is synchronous with respect to the host copy, so they may go out of scope)
        if (hipMemcpyAsync(r->sched, sched.data(), (size_t)n * sizeof(RsOut), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(r->w, w.data(), (size_t)n * RS_TAPS * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
            return rfail("resampler: schedule upload failed");
        hipLaunchKernelGGL(k_resample, dim3((unsigned)((n + 255) / 256), (unsigned)r->n_streams), dim3(256), 0, st, d_in, in_stride,
                           (const float *)r->hist, d_out, out_stride, (const RsOut *)r->sched, (const double *)r->w, (int)n, r->n_streams);
    }
    // Source samples of this call beyond those its completed outputs pulled belong to the next output, whose other samples have
    // not arrived.  The reference pulls a sample only inside the loop of the output that needs it; pushing these now is the same
    // thing (a push only moves the ring, and the ring is looked at when that output is interpolated): they are counted as
    // `credit` against that output's need, and the position keeps its value.
    if (n_in > 0)
        hipLaunchKernelGGL(k_resample_hist, dim3((unsigned)((r->n_streams * RS_TAPS + 255) / 256)), dim3(256), 0, st, d_in, in_stride, r->hist,
                           n_in, r->n_streams);
    if (hipGetLastError() != hipSuccess) return rfail("resampler: launch failed");
    {
        const long early = n_in - consumed;
        for (long k = 0; k < early && idx < RS_DEPTH; k++) idx++;
        r->credit = credit + early;
        r->pos = pos;
        r->idx = idx;
    }
    *n_out = n;
    return 0;
}

extern "C" int nnn_resampler_process_host(nnn_resampler *r, const float *in, long n_in, float *out, long cap_out, long *n_out)
{
    if (!r || !n_out) return rfail("null argument");
    if (hipSetDevice(r->device) != hipSuccess) return rfail("resampler: no such HIP device");
    const size_t S = (size_t)r->n_streams, nin = S * (size_t)(n_in > 0 ? n_in : 0), nout = S * (size_t)(cap_out > 0 ? cap_out : 0);
    if (nin > r->d_in_cap) {
        hipStreamSynchronize(r->stream);
        hipFree(r->d_in);
        r->d_in = nullptr; r->d_in_cap = 0;
        if (hipMalloc((void **)&r->d_in, (nin + nin / 2) * sizeof(float)) != hipSuccess) return rfail("resampler: allocation failed");
        r->d_in_cap = nin + nin / 2;
    }
    if (nout > r->d_out_cap) {
        hipStreamSynchronize(r->stream);
        hipFree(r->d_out);
        r->d_out = nullptr; r->d_out_cap = 0;
        if (hipMalloc((void **)&r->d_out, (nout + nout / 2) * sizeof(float)) != hipSuccess) return rfail("resampler: allocation failed");
        r->d_out_cap = nout + nout / 2;
    }
    if (nin && hipMemcpyAsync(r->d_in, in, nin * sizeof(float), hipMemcpyHostToDevice, r->stream) != hipSuccess) return rfail("resampler: host staging failed");
    if (int rc = nnn_resampler_process_device(r, r->d_in, n_in, (size_t)n_in, r->d_out, cap_out, (size_t)cap_out, n_out, r->stream)) return rc;
    if (*n_out > 0) {
        // rows of cap_out on the device, rows of cap_out on the host: one 2-D copy of the produced part
        if (hipMemcpy2DAsync(out, (size_t)cap_out * sizeof(float), r->d_out, (size_t)cap_out * sizeof(float), (size_t)*n_out * sizeof(float), S,
                             hipMemcpyDeviceToHost, r->stream) != hipSuccess)
            return rfail("resampler: copy back failed");
    }
    if (hipStreamSynchronize(r->stream) != hipSuccess) return rfail("resampler: synchronize failed");
    return 0;
}
