// tests/hostsim/hostsim.cpp -- TEST-ONLY SIMT interpreter behind tests/hostsim/hip/hip_runtime.h.
// One workgroup at a time; each thread is a ucontext fiber; barriers and wave shuffles are
// rendezvous points resolved by a round-robin scheduler.
#include <hip/hip_runtime.h>

#include <chrono>

// the shim maps the CUDA-style builtins onto hostsim::g; this file IS hostsim, so drop them
#undef threadIdx
#undef blockIdx
#undef blockDim
#undef gridDim

namespace hostsim {

Sim g;
char dyn_lds[160 * 1024];

namespace {
enum State { RUNNABLE, AT_BARRIER, AT_SHFL, AT_COLL, DONE };
struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    State st = RUNNABLE;
    unsigned long long xchg = 0;  // shuffle payload
    int shfl_arg = 0, shfl_mode = 0, shfl_width = 64;
    unsigned long long shfl_result = 0;
    const void *coll_in = nullptr;
    void *coll_out = nullptr;
    WaveFn coll_fn = nullptr;
};
const size_t kStack = 512 * 1024;
std::vector<Fiber> fibers;
ucontext_t sched_ctx;
int cur = -1;
const std::function<void()> *cur_body = nullptr;
dim3 cur_block;

void set_tid(int t) {
    g.threadIdx.x = t % cur_block.x;
    g.threadIdx.y = (t / cur_block.x) % cur_block.y;
    g.threadIdx.z = t / (cur_block.x * cur_block.y);
}

void trampoline() {
    (*cur_body)();
    fibers[cur].st = DONE;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}
}  // namespace

void yield_barrier() {
    int me = cur;
    fibers[me].st = AT_BARRIER;
    swapcontext(&fibers[me].ctx, &sched_ctx);
    set_tid(me);
}

unsigned long long yield_shfl_u64(unsigned long long v, int a, int mode, int width) {
    int me = cur;
    Fiber &f = fibers[me];
    f.xchg = v;
    f.shfl_arg = a;
    f.shfl_mode = mode;
    f.shfl_width = width;
    f.st = AT_SHFL;
    swapcontext(&f.ctx, &sched_ctx);
    set_tid(me);
    return fibers[me].shfl_result;
}

void wave_collective(const void *in, void *out, WaveFn fn) {
    int me = cur;
    Fiber &f = fibers[me];
    f.coll_in = in;
    f.coll_out = out;
    f.coll_fn = fn;
    f.st = AT_COLL;
    swapcontext(&f.ctx, &sched_ctx);
    set_tid(me);
}

static void resolve_coll(int w0, int w1) {
    const void *ins[64];
    void *outs[64];
    WaveFn fn = nullptr;
    for (int t = w0; t < w0 + 64; t++) {
        bool on = t < w1 && fibers[t].st == AT_COLL;
        ins[t - w0] = on ? fibers[t].coll_in : nullptr;
        outs[t - w0] = on ? fibers[t].coll_out : nullptr;
        if (on) fn = fibers[t].coll_fn;
    }
    if (fn) fn(ins, outs, 64);
    for (int t = w0; t < w1; t++)
        if (fibers[t].st == AT_COLL) fibers[t].st = RUNNABLE;
}

static void resolve_shfl(int wave_begin, int wave_end) {
    for (int t = wave_begin; t < wave_end; t++) {
        Fiber &f = fibers[t];
        if (f.st != AT_SHFL) continue;
        int lane = t - wave_begin, w = f.shfl_width, base = lane / w * w, src;
        switch (f.shfl_mode) {
        case 0: src = base + (f.shfl_arg % w + w) % w; break;
        case 1: src = lane ^ f.shfl_arg; if (src / w != lane / w) src = lane; break;
        case 2: src = lane + f.shfl_arg; if (src >= base + w) src = lane; break;
        default: src = lane - f.shfl_arg; if (src < base) src = lane; break;
        }
        int st = wave_begin + src;
        f.shfl_result = (st < wave_end && fibers[st].st == AT_SHFL) ? fibers[st].xchg : f.xchg;
    }
    for (int t = wave_begin; t < wave_end; t++)
        if (fibers[t].st == AT_SHFL) fibers[t].st = RUNNABLE;
}

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if ((int)fibers.size() < nthreads) {
        size_t old = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = old; i < fibers.size(); i++) fibers[i].stack = (char *)malloc(kStack);
    }
    g.blockDim = block;
    g.gridDim = grid;
    cur_block = block;
    cur_body = &body;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                g.blockIdx.x = bx; g.blockIdx.y = by; g.blockIdx.z = bz;
                for (int t = 0; t < nthreads; t++) {
                    Fiber &f = fibers[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    f.st = RUNNABLE;
                    makecontext(&f.ctx, trampoline, 0);
                }
                for (;;) {
                    bool ran = false;
                    for (int t = 0; t < nthreads; t++) {
                        if (fibers[t].st != RUNNABLE) continue;
                        cur = t;
                        set_tid(t);
                        swapcontext(&sched_ctx, &fibers[t].ctx);
                        ran = true;
                    }
                    // release shuffles wave by wave when every live lane of the wave waits on one
                    for (int w0 = 0; w0 < nthreads; w0 += 64) {
                        int w1 = w0 + 64 < nthreads ? w0 + 64 : nthreads;
                        bool any = false, all = true;
                        for (int t = w0; t < w1; t++) {
                            if (fibers[t].st == AT_SHFL) any = true;
                            else if (fibers[t].st != DONE) all = false;
                        }
                        if (any && all) { resolve_shfl(w0, w1); ran = true; }
                        bool anyc = false, allc = true;
                        for (int t = w0; t < w1; t++) {
                            if (fibers[t].st == AT_COLL) anyc = true;
                            else if (fibers[t].st != DONE) allc = false;
                        }
                        if (anyc && allc) { resolve_coll(w0, w1); ran = true; }
                    }
                    // release the barrier when every live thread waits on it
                    bool anyb = false, allb = true, alldone = true;
                    for (int t = 0; t < nthreads; t++) {
                        if (fibers[t].st == AT_BARRIER) anyb = true;
                        else if (fibers[t].st != DONE) allb = false;
                        if (fibers[t].st != DONE) alldone = false;
                    }
                    if (alldone) break;
                    if (anyb && allb) {
                        for (int t = 0; t < nthreads; t++)
                            if (fibers[t].st == AT_BARRIER) fibers[t].st = RUNNABLE;
                        ran = true;
                    }
                    if (!ran) {
                        fprintf(stderr, "hostsim: deadlock (divergent barrier/shuffle) in block %u,%u\n", bx, by);
                        abort();
                    }
                }
            }
    cur = -1;
}

}  // namespace hostsim

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new hostsim_event{0.0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = now_ms(); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
