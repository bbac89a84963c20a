// SPDX-License-Identifier: BSD-3-Clause
// nnnoiseless.hpp -- C++ host-side mirror of the reference's public interface for the process_frame path
// (Rust is not available in the build image, so the host side above the C ABI is C++; the Rust façade a
// maintainer would compile is in bindings/rust/).  Same names, argument meaning and error behaviour:
//
//   reference (Rust)                                         here (C++)
//   RnnModel::from_bytes(&[u8]) -> Option<RnnModel>          RnnModel::from_bytes(ptr, len) -> std::optional<RnnModel>   src/rnn.rs:75-77
//   RnnModel::default()                                      RnnModel::default_model()                                  src/rnn.rs:235-240
//   DenoiseState::FRAME_SIZE                                 DenoiseState::FRAME_SIZE                                   src/denoise.rs:46
//   DenoiseState::new() / from_model / with_model            DenoiseState::create() / from_model / with_model           src/denoise.rs:53-74
//   process_frame(&mut self, &mut [f32], &[f32]) -> f32      process_frame(float* out, const float* in) -> float        src/denoise.rs:95-116
//   #[derive(Clone)] DenoiseState                            DenoiseState::clone() / BatchDenoiser::clone()             src/denoise.rs:36
//   for ch { states[ch].process_frame(..) }                  BatchDenoiser::process(...)                                src/signal.rs:102-104
//
// Everything runs in the HIP kernels behind include/nnn_batch.h; failures throw std::runtime_error.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "nnn_batch.h"
#include "nnn_node.h"
#include "nnn_resample.h"
#include "nnn_train.h"

namespace nnnoiseless {

class RnnModel {
  public:
    static std::optional<RnnModel> from_bytes(const uint8_t *bytes, size_t len)
    {
        RNNModel *m = nnn_model_from_bytes(bytes, len);
        if (!m) return std::nullopt;
        return RnnModel(m);
    }
    // an RNNoise / rnnoise-nu text model (train/convert_rnnoise.py + from_bytes in one step)
    static std::optional<RnnModel> from_rnnoise_text(const std::string &text)
    {
        RNNModel *m = nnn_model_from_rnnoise_text(text.data(), text.size());
        if (!m) return std::nullopt;
        return RnnModel(m);
    }
    static RnnModel default_model() { return RnnModel(nnn_model_default()); }
    const RNNModel *raw() const { return m_.get(); }
    // a copy with its own storage (nnn_model_clone); plain copies of this class share one immutable model
    RnnModel deep_clone() const
    {
        RNNModel *m = nnn_model_clone(m_.get());
        if (!m) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
        return RnnModel(m);
    }

  private:
    explicit RnnModel(RNNModel *m) : m_(m, nnn_model_free) {}
    std::shared_ptr<RNNModel> m_;  // Clone in the reference; sharing is equivalent for an immutable model
};

// n independent DenoiseStates advanced in lock-step on one GPU
// Page-locked host memory for the host-buffer calls (nnn_host_alloc): n elements of T.
template <class T> class PinnedBuffer {
  public:
    explicit PinnedBuffer(size_t n) : p_((T *)nnn_host_alloc(n * sizeof(T))), n_(n)
    {
        if (!p_) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    ~PinnedBuffer() { nnn_host_free(p_); }
    PinnedBuffer(const PinnedBuffer &) = delete;
    PinnedBuffer &operator=(const PinnedBuffer &) = delete;
    T *data() { return p_; }
    const T *data() const { return p_; }
    size_t size() const { return n_; }
    T &operator[](size_t i) { return p_[i]; }

  private:
    T *p_;
    size_t n_;
};

class BatchDenoiser {
  public:
    BatchDenoiser(int n_streams, const RnnModel *model = nullptr, int device = 0)
        : b_(nnn_batch_create(model ? model->raw() : nullptr, n_streams, device), nnn_batch_destroy)
    {
        if (!b_) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    // several models resident at once: streams [0, n0) run models[0], the next n1 models[1], ... (nullptr = built-in)
    BatchDenoiser(const std::vector<std::pair<const RnnModel *, int>> &groups, int device = 0)
    {
        std::vector<const RNNModel *> ms;
        std::vector<int> ns;
        for (const auto &g : groups) {
            ms.push_back(g.first ? g.first->raw() : nullptr);
            ns.push_back(g.second);
        }
        b_.reset(nnn_batch_create_grouped(ms.data(), ns.data(), (int)ns.size(), device), nnn_batch_destroy);
        if (!b_) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    // a batch sized for calls of at most `max_group_frames` frames (1 = a real-time host ticking one frame per call: 33 KB per
    // stream instead of 360; longer calls still work, cut into groups of that many frames)
    static BatchDenoiser sized(int n_streams, int max_group_frames, const RnnModel *model = nullptr, int device = 0)
    {
        const RNNModel *m = model ? model->raw() : nullptr;
        nnn_batch_opts o = {};
        o.max_group_frames = max_group_frames;
        BatchDenoiser d(nnn_batch_create_opts(&m, &n_streams, 1, device, &o));
        if (!d.b_) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
        return d;
    }
    int num_streams() const { return nnn_batch_num_streams(b_.get()); }
    size_t device_bytes() const { return nnn_batch_device_bytes(b_.get()); }
    int max_group_frames() const { return nnn_batch_max_group_frames(b_.get()); }
    // packed PCM in the reference callers' formats (int16 / unit floats, interleaved channels, first frame dropped)
    void process_pcm(const void *in, void *out, float *vad, int n_frames, const nnn_pcm_layout &layout)
    {
        check(nnn_batch_process_pcm_host(b_.get(), in, out, vad, n_frames, &layout));
    }
    void process_pcm_device(const void *d_in, void *d_out, float *d_vad, int n_frames, const nnn_pcm_layout &layout,
                            void *hip_stream = nullptr)
    {
        check(nnn_batch_process_pcm_device(b_.get(), d_in, d_out, d_vad, n_frames, &layout, hip_stream));
    }
    // host buffers: sample i of frame t of stream s at [s * stream_stride + t * frame_stride + i]; vad[t * n_streams + s]
    // (long calls cross the bus in chunks beside the kernels; PinnedBuffer memory goes by DMA, both directions at once)
    void process(const float *in, float *out, float *vad, int n_frames, size_t stream_stride, size_t frame_stride)
    {
        check(nnn_batch_process_host(b_.get(), in, out, vad, n_frames, stream_stride, frame_stride));
    }
    // promise that every process_device call's input is final when the call is made: consecutive calls may then overlap at
    // their boundary (include/nnn_batch.h); outputs stay ordered on the call's stream
    void set_inputs_ready(bool on) { check(nnn_batch_set_inputs_ready(b_.get(), on ? 1 : 0)); }
    // device buffers, asynchronous on `hip_stream` (nullptr = the batch's own stream)
    void process_device(const float *d_in, float *d_out, float *d_vad, int n_frames, size_t stream_stride, size_t frame_stride,
                        void *hip_stream = nullptr)
    {
        check(nnn_batch_process_device(b_.get(), d_in, d_out, d_vad, n_frames, stream_stride, frame_stride, hip_stream));
    }
    void synchronize() { check(nnn_batch_synchronize(b_.get())); }
    // true once a frame hand-off inside the pitch stage has failed (nnn_batch_fault): sticky until reset() / load_state(); for hosts
    // that synchronise their own HIP stream instead of calling synchronize()
    bool fault() const { return nnn_batch_fault(b_.get()) != 0; }
    void reset() { check(nnn_batch_reset(b_.get())); }
    // a second batch with the same models and a copy of every stream's state (DenoiseState: Clone)
    BatchDenoiser clone() const
    {
        nnn_batch *c = nnn_batch_clone(b_.get());
        if (!c) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
        return BatchDenoiser(c);
    }
    // the streams' state as bytes (a raw image: loads into a batch of the same shape made by the same build)
    std::vector<uint8_t> save_state() const
    {
        std::vector<uint8_t> buf(nnn_batch_state_bytes(b_.get()));
        check(nnn_batch_save_state(b_.get(), buf.data(), buf.size()));
        return buf;
    }
    void load_state(const std::vector<uint8_t> &buf) { check(nnn_batch_load_state(b_.get(), buf.data(), buf.size())); }
    nnn_batch *raw() { return b_.get(); }

  private:
    explicit BatchDenoiser(nnn_batch *owned) : b_(owned, nnn_batch_destroy) {}
    static void check(int rc)
    {
        if (rc) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    std::shared_ptr<nnn_batch> b_;
};

// the per-frame body of the reference's training-data generator (src/training.rs:113-160) for n_streams triples
class TrainingFeatures {
  public:
    static constexpr int ROW_WIDTH = NNN_TRAIN_COLS;
    explicit TrainingFeatures(int n_streams, int device = 0) : t_(nnn_train_create(n_streams, device), nnn_train_destroy)
    {
        if (!t_) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    // host buffers: signal / noise / combined [n_streams][n_frames][480]; cutoff, vad [n_frames][n_streams]; rows [..][87]
    void process(const float *signal, const float *noise, const float *combined, const int32_t *cutoff, const float *vad,
                 float *rows, int n_frames)
    {
        if (nnn_train_process_host(t_.get(), signal, noise, combined, cutoff, vad, rows, n_frames))
            throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    void reset() { nnn_train_reset(t_.get()); }

  private:
    std::shared_ptr<nnn_train> t_;
};

class DenoiseState {
  public:
    static constexpr size_t FRAME_SIZE = NNN_FRAME_SIZE;
    static DenoiseState create(int device = 0) { return DenoiseState(nullptr, device); }                    // DenoiseState::new()
    static DenoiseState from_model(const RnnModel &m, int device = 0) { return DenoiseState(&m, device); }
    static DenoiseState with_model(const RnnModel &m, int device = 0) { return DenoiseState(&m, device); }
    // `out` may alias `in`; returns the VAD probability
    float process_frame(float *out, const float *in)
    {
        float vad = 0.f;
        b_.process(in, out, &vad, 1, FRAME_SIZE, FRAME_SIZE);
        return vad;
    }
    DenoiseState clone() const { return DenoiseState(b_.clone()); }   // #[derive(Clone)], src/denoise.rs:36

  private:
    DenoiseState(const RnnModel *m, int device) : b_(BatchDenoiser::sized(1, 1, m, device)) {}   // one frame per call: sized for it
    explicit DenoiseState(BatchDenoiser b) : b_(std::move(b)) {}
    BatchDenoiser b_;
};

// n mono streams of one common sample rate -> 48 kHz with the CLI's 16-tap windowed sinc (src/nnnoiseless.rs:19-32, 106-131)
class Resampler {
  public:
    Resampler(int n_streams, double source_rate, int device = 0)
        : r_(nnn_resampler_create(n_streams, source_rate / 48000.0, device), nnn_resampler_destroy)
    {
        if (!r_) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    long max_output(long n_in) const { return nnn_resampler_max_output(r_.get(), n_in); }
    // host buffers, dense [n_streams][n_in] -> [n_streams][cap_out]; returns the samples produced per stream
    long process(const float *in, long n_in, float *out, long cap_out)
    {
        long n = 0;
        if (nnn_resampler_process_host(r_.get(), in, n_in, out, cap_out, &n)) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
        return n;
    }
    void reset() { nnn_resampler_reset(r_.get()); }

  private:
    std::shared_ptr<nnn_resampler> r_;
};


// All the GPUs of a node behind one object (include/nnn_node.h): contiguous stream shards, one batch and one host thread per device.
class NodeDenoiser {
  public:
    NodeDenoiser(int n_streams, const std::vector<int> &devices, const RnnModel *model = nullptr, int max_group_frames = 0)
    {
        nnn_batch_opts o = {};
        o.max_group_frames = max_group_frames;
        n_ = nnn_node_create(model ? model->raw() : nullptr, n_streams, devices.data(), (int)devices.size(), max_group_frames ? &o : nullptr);
        if (!n_) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    ~NodeDenoiser() { nnn_node_destroy(n_); }
    NodeDenoiser(const NodeDenoiser &) = delete;
    NodeDenoiser &operator=(const NodeDenoiser &) = delete;
    int num_streams() const { return nnn_node_num_streams(n_); }
    int num_shards() const { return nnn_node_num_shards(n_); }
    // in / out: [n_streams][n_frames][480], vad: [n_frames][n_streams] (may be null); every device works on its share at once
    void process(const float *in, float *out, float *vad, int n_frames)
    {
        if (nnn_node_process_host(n_, in, out, vad, n_frames, (size_t)n_frames * NNN_FRAME_SIZE, NNN_FRAME_SIZE))
            throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    // buffers resident on the shards' own devices, one pointer (and optionally one hipStream_t) per shard; asynchronous: synchronize()
    void process_device(const std::vector<const float *> &d_in, const std::vector<float *> &d_out, const std::vector<float *> &d_vad, int n_frames,
                        size_t stream_stride, size_t frame_stride, const std::vector<void *> &hip_streams = {})
    {
        const int n = num_shards();
        if ((int)d_in.size() != n || (int)d_out.size() != n || (!d_vad.empty() && (int)d_vad.size() != n) || (!hip_streams.empty() && (int)hip_streams.size() != n))
            throw std::invalid_argument("nnnoiseless: one table entry per shard");
        if (nnn_node_process_device_streams(n_, d_in.data(), d_out.data(), d_vad.empty() ? nullptr : d_vad.data(),
                                            hip_streams.empty() ? nullptr : hip_streams.data(), n, n_frames, stream_stride, frame_stride))
            throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    void synchronize()
    {
        if (nnn_node_synchronize(n_)) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    // the CPUs shard i's host thread is pinned to (its device's local CPUs), "" when not pinned
    std::string shard_cpus(int i) const { return nnn_node_shard_cpus(n_, i); }
    // after a call that failed on some shard the node refuses further calls until reset() (the shards sit at different frame counts)
    void reset()
    {
        if (nnn_node_reset(n_)) throw std::runtime_error(std::string("nnnoiseless: ") + nnn_last_error());
    }
    bool fault() const { return nnn_node_fault(n_) != 0; }
    nnn_node *raw() { return n_; }

  private:
    nnn_node *n_ = nullptr;
};

}  // namespace nnnoiseless
