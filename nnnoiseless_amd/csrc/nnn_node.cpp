// nnn_node.cpp -- all the GPUs of a node behind one object (include/nnn_node.h): contiguous stream shards, one nnn_batch and one host
// thread per device, fan-out and join inside every call.  Host-only code above the batch ABI: everything that touches a device goes
// through nnn_batch_*.  (ref: the reference's hosts walk a vector of independent states, src/nnnoiseless.rs:305-320.)
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nnn_node.h"

int nnn_set_error(const char *msg);   // nnn_batch.hip

// Pins the calling thread to the CPUs local to a device's PCI function ("0-31,128-159" in the kernel's cpulist syntax, from
// nnn_device_local_cpulist): a worker that stages pageable buffers and enqueues for GPU i should run on the socket GPU i hangs off.
// Best effort: an empty or unparsable list leaves the thread where it is.
static bool pin_to_cpulist(const char *list)
{
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    for (const char *p = list; *p;) {
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        p = e;
        if (*p == '-') {
            b = strtol(p + 1, &e, 10);
            if (e == p + 1) break;
            p = e;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0) { CPU_SET((int)c, &set); n++; }
        if (*p == ',') p++;
        else break;
    }
    return n > 0 && pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0;
}

namespace {
// a host thread that runs the jobs its shard is handed, one at a time
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = 0;
    std::string err;
    void loop()
    {
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit) return;
                j = job;
                has_job = false;
            }
            const int r = j();
            std::string e = r ? nnn_last_error() : "";   // (the error text is thread-local: carried back to the caller's thread)
            {
                std::lock_guard<std::mutex> lk(mu);
                rc = r;
                err = e;
                done = true;
            }
            cv.notify_all();
        }
    }
    void post(std::function<int()> j)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
};
}  // namespace

struct nnn_node {
    struct Shard {
        nnn_batch *b = nullptr;
        int device = 0, lo = 0, hi = 0;
        Worker *w = nullptr;
        std::vector<float> vad;   // host calls: this shard's VAD block [n_frames][hi - lo] before it is spread over the node's rows
        std::string cpus;         // the CPUs its worker is pinned to ("" = not pinned)
    };
    std::vector<Shard> shards;
    int n_streams = 0;
    bool failed = false;          // a call failed on some shard: the shards may sit at different frame counts.  Sticky until nnn_node_reset
                                  // (every process call refuses), like a batch's own fault.
    std::string fail_text;
    bool threads = true;          // NNN_NODE_THREADS=0: the shards one after the other on the caller's thread (debugging; the
                                  // test-only interpreter build runs kernels on the calling thread and is not re-entrant)
};

// Every shard runs fn(shard index) -- all of them, also when one fails, on the shards' own threads or (NNN_NODE_THREADS=0, one shard)
// one after the other on the caller's: either way the other shards are joined and have advanced.  Returns the first failure in
// shard order with its text in nnn_last_error().
static int fan_out(nnn_node *n, const std::function<int(int)> &fn)
{
    int rc = 0;
    std::string err;
    if (!n->threads || n->shards.size() == 1) {
        for (size_t i = 0; i < n->shards.size(); i++) {
            const int r = fn((int)i);
            if (r && !rc) {
                rc = r;
                err = nnn_last_error();
            }
        }
    } else {
        for (size_t i = 0; i < n->shards.size(); i++) n->shards[i].w->post([&fn, i] { return fn((int)i); });
        for (size_t i = 0; i < n->shards.size(); i++) {
            const int r = n->shards[i].w->wait();
            if (r && !rc) {
                rc = r;
                err = n->shards[i].w->err;
            }
        }
    }
    if (rc) nnn_set_error(err.c_str());
    return rc;
}
// a processing call: refused while the node is failed; a failure makes it so
static int process_call(nnn_node *n, const std::function<int(int)> &fn)
{
    if (n->failed) {
        std::string t = "node failed earlier (reset it): " + n->fail_text;
        return nnn_set_error(t.c_str());
    }
    const int rc = fan_out(n, fn);
    if (rc) {
        n->failed = true;
        n->fail_text = nnn_last_error();
        nnn_set_error(n->fail_text.c_str());
    }
    return rc;
}

extern "C" void nnn_node_destroy(nnn_node *n)
{
    if (!n) return;
    for (auto &s : n->shards) {
        if (s.w) {
            {
                std::lock_guard<std::mutex> lk(s.w->mu);
                s.w->quit = true;
            }
            s.w->cv.notify_all();
            if (s.w->th.joinable()) s.w->th.join();
            delete s.w;
        }
        if (s.b) nnn_batch_destroy(s.b);
    }
    delete n;
}

extern "C" nnn_node *nnn_node_create(const RNNModel *model, int n_streams, const int *devices, int n_devices, const nnn_batch_opts *opts)
{
    if (!devices || n_devices <= 0) {
        nnn_set_error("need at least one device");
        return nullptr;
    }
    if (n_streams < n_devices) {
        nnn_set_error("fewer streams than devices");
        return nullptr;
    }
    nnn_node *n = new nnn_node();
    n->n_streams = n_streams;
    if (const char *e = getenv("NNN_NODE_THREADS")) n->threads = atoi(e) != 0;
    n->shards.resize((size_t)n_devices);
    const int base = n_streams / n_devices, rem = n_streams % n_devices;
    for (int i = 0; i < n_devices; i++) {   // contiguous, balanced: nnnoiseless_amd/shard.py's shard_range
        nnn_node::Shard &s = n->shards[(size_t)i];
        s.device = devices[i];
        s.lo = i * base + (i < rem ? i : rem);
        s.hi = s.lo + base + (i < rem ? 1 : 0);
    }
    // One worker per shard, pinned to the CPUs local to its device's PCI function (the socket its GPU hangs off: a worker stages pageable
    // buffers and enqueues for that GPU), then every worker makes its own batch: the thread is pinned BEFORE it allocates and uploads, so the page-locked
    // staging buffers and the tables' host copies come from its GPU's socket.  The creations themselves still run one after another:
    // nnn_batch_create_opts holds the library's runtime lock for its whole length (allocation and legacy-stream copies must not overlap
    // another thread's stream capture, nnn_batch.hip), so eight devices take eight creations' time.
    if (n->threads && n_devices > 1)
        for (auto &s : n->shards) {
            s.w = new Worker();
            s.w->th = std::thread([w = s.w] { w->loop(); });
        }
    const int rc = fan_out(n, [&](int i) {
        nnn_node::Shard &s = n->shards[(size_t)i];
        char cpus[512];
        if (s.w && nnn_device_local_cpulist(s.device, cpus, sizeof(cpus)) == 0 && pin_to_cpulist(cpus)) s.cpus = cpus;
        const int cnt = s.hi - s.lo;
        s.b = nnn_batch_create_opts(model ? &model : nullptr, &cnt, 1, s.device, opts);
        return s.b ? 0 : 1;
    });
    if (rc) {
        std::string keep = nnn_last_error();
        nnn_node_destroy(n);
        nnn_set_error(keep.c_str());
        return nullptr;
    }
    return n;
}

extern "C" int nnn_node_num_streams(const nnn_node *n) { return n ? n->n_streams : 0; }
extern "C" int nnn_node_num_shards(const nnn_node *n) { return n ? (int)n->shards.size() : 0; }
extern "C" int nnn_node_shard(const nnn_node *n, int i, int *device, int *lo, int *hi)
{
    if (!n || i < 0 || i >= (int)n->shards.size()) return nnn_set_error("no such shard");
    if (device) *device = n->shards[(size_t)i].device;
    if (lo) *lo = n->shards[(size_t)i].lo;
    if (hi) *hi = n->shards[(size_t)i].hi;
    return 0;
}
extern "C" const char *nnn_node_shard_cpus(const nnn_node *n, int i)
{
    return (n && i >= 0 && i < (int)n->shards.size()) ? n->shards[(size_t)i].cpus.c_str() : "";
}
extern "C" nnn_batch *nnn_node_batch(nnn_node *n, int i) { return (n && i >= 0 && i < (int)n->shards.size()) ? n->shards[(size_t)i].b : nullptr; }

extern "C" int nnn_node_reset(nnn_node *n)
{
    if (!n) return nnn_set_error("null node");
    const int rc = fan_out(n, [n](int i) { return nnn_batch_reset(n->shards[(size_t)i].b); });
    if (!rc) {
        n->failed = false;
        n->fail_text.clear();
    }
    return rc;
}

// the shards' VAD blocks ([t][shard stream]) into the node's rows ([t][all streams])
static void spread_vad(nnn_node *n, float *vad, int n_frames)
{
    if (!vad) return;
    for (auto &s : n->shards) {
        const size_t cnt = (size_t)(s.hi - s.lo);
        for (int t = 0; t < n_frames; t++) memcpy(vad + (size_t)t * n->n_streams + s.lo, s.vad.data() + (size_t)t * cnt, cnt * sizeof(float));
    }
}

extern "C" int nnn_node_process_host(nnn_node *n, const float *in, float *out, float *vad, int n_frames, size_t stream_stride, size_t frame_stride)
{
    if (!n) return nnn_set_error("null node");
    if (n_frames <= 0) return 0;
    if (!in || !out) return nnn_set_error("null buffer");
    const int rc = process_call(n, [&](int i) {
        nnn_node::Shard &s = n->shards[(size_t)i];
        if (vad) s.vad.resize((size_t)n_frames * (size_t)(s.hi - s.lo));
        return nnn_batch_process_host(s.b, in + (size_t)s.lo * stream_stride, out + (size_t)s.lo * stream_stride, vad ? s.vad.data() : nullptr,
                                      n_frames, stream_stride, frame_stride);
    });
    if (!rc) spread_vad(n, vad, n_frames);
    return rc;
}

extern "C" int nnn_node_process_pcm_host(nnn_node *n, const void *in, void *out, float *vad, int n_frames, const nnn_pcm_layout *L)
{
    if (!n) return nnn_set_error("null node");
    if (n_frames <= 0) return 0;
    if (!in || !out || !L) return nnn_set_error("null argument");
    if (L->channels < 1) return nnn_set_error("channels must be positive");
    for (auto &s : n->shards)
        if (s.lo % L->channels || s.hi % L->channels) return nnn_set_error("the stream split cuts a channel group: n_streams / channels must divide evenly over the shards");
    const size_t e = L->format == NNN_PCM_I16 ? 2 : 4;
    const int rc = process_call(n, [&](int i) {
        nnn_node::Shard &s = n->shards[(size_t)i];
        if (vad) s.vad.resize((size_t)n_frames * (size_t)(s.hi - s.lo));
        const size_t off = (size_t)(s.lo / L->channels) * L->group_stride * e;   // the shard's first group
        return nnn_batch_process_pcm_host(s.b, (const char *)in + off, (char *)out + off, vad ? s.vad.data() : nullptr, n_frames, L);
    });
    if (!rc) spread_vad(n, vad, n_frames);
    return rc;
}

extern "C" int nnn_node_process_device_streams(nnn_node *n, const float *const *d_in, float *const *d_out, float *const *d_vad,
                                               void *const *hip_streams, int n_tables, int n_frames, size_t stream_stride, size_t frame_stride)
{
    if (!n) return nnn_set_error("null node");
    if (n_tables != (int)n->shards.size()) return nnn_set_error("the pointer tables must have one entry per shard");
    if (n_frames <= 0) return 0;
    if (!d_in || !d_out) return nnn_set_error("null buffer table");
    return process_call(n, [&](int i) {
        return nnn_batch_process_device(n->shards[(size_t)i].b, d_in[i], d_out[i], d_vad ? d_vad[i] : nullptr, n_frames, stream_stride, frame_stride,
                                        hip_streams ? hip_streams[i] : nullptr);
    });
}
extern "C" int nnn_node_process_device(nnn_node *n, const float *const *d_in, float *const *d_out, float *const *d_vad, int n_frames,
                                       size_t stream_stride, size_t frame_stride)
{
    return nnn_node_process_device_streams(n, d_in, d_out, d_vad, nullptr, n ? (int)n->shards.size() : 0, n_frames, stream_stride, frame_stride);
}

extern "C" int nnn_node_synchronize(nnn_node *n)
{
    if (!n) return nnn_set_error("null node");
    return fan_out(n, [n](int i) { return nnn_batch_synchronize(n->shards[(size_t)i].b); });
}

extern "C" int nnn_node_fault(const nnn_node *n)
{
    if (!n) return 0;
    for (auto &s : n->shards)
        if (nnn_batch_fault(s.b)) return 1;
    return 0;
}
