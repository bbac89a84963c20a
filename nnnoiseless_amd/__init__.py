"""nnnoiseless_amd -- MI355X-native batched backend for nnnoiseless' DenoiseState::process_frame.

Host-side mirror of the reference's public interface for this path (jneem/nnnoiseless v0.5.1):

  RnnModel      from_bytes / from_static_bytes / default          (src/rnn.rs:72-94, 235-240)
  DenoiseState  FRAME_SIZE, new / from_model / with_model, process_frame(output, input) -> vad
                                                                    (src/denoise.rs:44-116)
  BatchDenoiser n independent DenoiseStates advanced in lock-step (the per-channel loop of
                src/signal.rs:102-104 and src/nnnoiseless.rs:318-320 as one call)

All arithmetic runs in hand-written HIP kernels (csrc/nnn_kernels.hip) behind the C ABI of
include/nnn_batch.h and include/rnnoise.h.  There is no CPU fallback: importing works anywhere,
but creating a state without the built library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

# (Several batches ticking side by side overlap only when their streams land on different hardware queues; the HIP runtime has four unless
# GPU_MAX_HW_QUEUES says otherwise when it initialises.  That is the HOST's setting -- INTEGRATION.md -- and neither this package nor the
# library touches the environment: until round 4 both exported it silently.  bench.py and the tick tools export it themselves.)

from . import _ffi
from .build import LIB_PATH, build_library

FRAME_SIZE = 480
FREQ_SIZE = 481
NB_BANDS = 22
NB_FEATURES = 42

_lib = None


def library():
    """The hipcc-built gfx950 library (loaded on first use)."""
    global _lib
    if _lib is None:
        import os
        import sys
        # A process that also uses PyTorch holds two HIP runtimes (torch ships its own): torch's does not come up once the
        # system one is running ("No HIP GPUs are available"), the other order works.  If torch is already imported, bring its
        # device up first; a host that imports torch later has to do the same itself (torch.cuda.init()) before its first batch.
        t = sys.modules.get("torch")
        if t is not None:
            try:
                if t.cuda.is_available():
                    t.cuda.init()
            except Exception:
                pass
        _lib = _ffi.Library(os.environ.get("NNN_LIBRARY", LIB_PATH))   # developer override: an experimental build
    return _lib


class _Pinned:
    """Owner of one nnn_host_alloc block (freed with the last array that views it)."""

    def __init__(self, lib, nbytes):
        self._lib, self.nbytes = lib, int(nbytes)
        self.ptr = lib.L.nnn_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("nnnoiseless_amd: " + lib.error())

    def __del__(self):
        if getattr(self, "ptr", None):
            self._lib.L.nnn_host_free(self.ptr)
            self.ptr = None


def pinned_empty(shape, dtype=np.float32, lib=None):
    """A numpy array in page-locked host memory (include/nnn_batch.h nnn_host_alloc): what BatchDenoiser.process /
    process_pcm transfer by DMA, uploads and downloads at the same time."""
    lib = lib or library()
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    own = _Pinned(lib, max(1, n * dt.itemsize))
    buf = (C.c_char * own.nbytes).from_address(own.ptr)
    buf._owner = own   # the ctypes view is the array's base: the block lives as long as any view of the array
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)


class RnnModel:
    """Model parameters (reference: src/rnn.rs:54-62).  Constructors return None on malformed input
    exactly where the reference returns None."""

    def __init__(self, handle, lib):
        self._h, self._lib = handle, lib

    @classmethod
    def from_bytes(cls, data, lib=None):
        lib = lib or library()
        data = bytes(data)
        h = lib.L.nnn_model_from_bytes(data, len(data))
        return cls(h, lib) if h else None

    from_static_bytes = from_bytes

    @classmethod
    def from_rnnoise_text(cls, text, lib=None):
        """An RNNoise / rnnoise-nu text model (what train/convert_rnnoise.py turns into a .rnn file first)."""
        lib = lib or library()
        data = text.encode() if isinstance(text, str) else bytes(text)
        h = lib.L.nnn_model_from_rnnoise_text(data, len(data))
        return cls(h, lib) if h else None

    @classmethod
    def default(cls, lib=None):
        lib = lib or library()
        return cls(lib.L.nnn_model_default(), lib)

    def clone(self):
        """`RnnModel: Clone` (src/rnn.rs:54): an independent copy."""
        h = self._lib.L.nnn_model_clone(self._h)
        if not h:
            raise RuntimeError(self._lib.error())
        return RnnModel(h, self._lib)

    __copy__ = clone

    def shape(self):
        s = (C.c_int32 * 12)()
        self._lib.L.nnn_model_shape(self._h, s)
        return list(s)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.L.nnn_model_free(self._h)
            self._h = None


class BatchDenoiser:
    """n_streams DenoiseStates in lock-step on one GPU."""

    def __init__(self, n_streams, model=None, device=0, lib=None, groups=None, taps=False, max_group_frames=None, _handle=None):
        """max_group_frames: the longest run of frames the kernels take at once (default 24).  A real-time host that ticks one frame
        per call passes 1: the batch then holds 33 KB per stream instead of 360 (`device_bytes()`); longer calls still work on it,
        cut into groups of that many frames.
        groups: [(model_or_None, n_streams), ...] keeps several models resident, one per run of streams (every run
        but the last a multiple of 64); it replaces `model` and must add up to n_streams.  taps=True also stores the
        intermediate quantities the kernels otherwise keep on chip (parity tests: everything inside the pitch kernel -- tap("xlp"),
        "xcorr1", "best1", "xcorr2c", "pitch_search" --, "P" beyond bin 399, "features")."""
        self._lib = lib or library()
        self._model = model
        self.n_streams = int(n_streams)
        self.frames_done = 0
        if _handle is not None:   # clone()
            self._h = _handle
            return
        if max_group_frames is not None and int(max_group_frames) < 1:
            raise ValueError("max_group_frames must be at least 1")
        if groups or max_group_frames is not None:
            groups_ = groups or [(model, self.n_streams)]
            if sum(n for _, n in groups_) != self.n_streams:
                raise ValueError("group sizes must add up to n_streams")
            if groups:
                self._model = [m for m, _ in groups_]
            hs = (C.c_void_p * len(groups_))(*[m._h if m is not None else None for m, _ in groups_])
            ns = (C.c_int * len(groups_))(*[int(n) for _, n in groups_])
            if max_group_frames is not None:
                opts = _ffi.BatchOpts(max_group_frames=int(max_group_frames))
                self._h = self._lib.L.nnn_batch_create_opts(hs, ns, len(groups_), device, C.byref(opts))
            else:
                self._h = self._lib.L.nnn_batch_create_grouped(hs, ns, len(groups_), device)
        else:
            self._h = self._lib.L.nnn_batch_create(model._h if model is not None else None, self.n_streams, device)
        if not self._h:
            raise RuntimeError("nnnoiseless_amd: " + self._lib.error())
        if taps:
            self.set_taps(True)

    def device_bytes(self):
        """Device memory the batch holds (state, scratch, tables, weights)."""
        return int(self._lib.L.nnn_batch_device_bytes(self._h))

    def max_group_frames(self):
        return int(self._lib.L.nnn_batch_max_group_frames(self._h))

    def clone(self):
        """A second batch with the same models and a copy of every stream's state (DenoiseState: Clone, src/denoise.rs:36)."""
        h = self._lib.L.nnn_batch_clone(self._h)
        if not h:
            raise RuntimeError("nnnoiseless_amd: " + self._lib.error())
        c = BatchDenoiser(self.n_streams, model=self._model, lib=self._lib, _handle=h)
        c.frames_done = self.frames_done
        return c

    def save_state(self):
        """The streams' state as bytes (a raw image: loads only into a batch of the same shape made by the same build)."""
        buf = np.empty(self._lib.L.nnn_batch_state_bytes(self._h), np.uint8)
        self._lib.check(self._lib.L.nnn_batch_save_state(self._h, _ffi.ptr(buf), buf.nbytes))
        return buf.tobytes()

    def load_state(self, data):
        buf = np.frombuffer(data, np.uint8)
        self._lib.check(self._lib.L.nnn_batch_load_state(self._h, _ffi.ptr(buf), buf.nbytes))

    def set_taps(self, on):
        self._lib.check(self._lib.L.nnn_batch_set_taps(self._h, int(on)))

    def set_schedule(self, mode, lanes=0):
        """mode: "seq" | "lanes" | "stages" (include/nnn_batch.h nnn_batch_set_schedule)."""
        self._lib.check(self._lib.L.nnn_batch_set_schedule(self._h, {"seq": 0, "lanes": 1, "stages": 2}[mode], int(lanes)))

    def process(self, x, out=None, vad=None):
        """x: float32 [n_streams, n_frames, 480] on the host -> (out same shape, vad [n_frames, n_streams]).  Arrays from
        pinned_empty (x, and out / vad handed in) cross the bus by DMA."""
        x = _ffi.as_f32(x)
        if x.ndim != 3 or x.shape[0] != self.n_streams or x.shape[2] != FRAME_SIZE:
            raise ValueError(f"process needs x of shape [{self.n_streams}, n_frames, {FRAME_SIZE}], got {x.shape}")
        S, T, F = x.shape
        out = np.empty_like(x) if out is None else out
        vad = np.empty((T, S), np.float32) if vad is None else vad
        # caller-supplied buffers are written by the C library: a wrong size or layout must never get that far
        if not (isinstance(out, np.ndarray) and out.shape == x.shape and out.dtype == np.float32 and out.flags.c_contiguous and out.flags.writeable):
            raise ValueError("process: `out` must be a writable C-contiguous float32 array of x's shape")
        if not (isinstance(vad, np.ndarray) and vad.shape == (T, S) and vad.dtype == np.float32 and vad.flags.c_contiguous and vad.flags.writeable):
            raise ValueError("process: `vad` must be a writable C-contiguous float32 array of shape [n_frames, n_streams]")
        self._lib.check(self._lib.L.nnn_batch_process_host(self._h, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(vad), T,
                                                           T * FRAME_SIZE, FRAME_SIZE))
        self.frames_done += T
        return out, vad

    def process_pcm(self, x, fmt, channels=1, discard_first=False):
        """Packed PCM in the reference callers' formats, x: [n_groups, n_frames * 480, channels] (int16 for PCM_I16,
        float32 otherwise), n_groups * channels == n_streams.  Returns (out [n_groups, n_out_frames * 480, channels],
        vad [n_frames, n_streams]); n_out_frames = n_frames - 1 if discard_first drops the first frame after a reset."""
        x = np.ascontiguousarray(x, dtype=_ffi.PCM_DTYPE[fmt])
        if x.ndim != 3 or x.shape[2] != channels or x.shape[0] * channels != self.n_streams or x.shape[1] % FRAME_SIZE:
            raise ValueError(f"process_pcm needs x of shape [n_streams / channels, n_frames * {FRAME_SIZE}, channels], got {x.shape}")
        G, N, Cc = x.shape
        T = N // FRAME_SIZE
        out = np.zeros_like(x)
        vad = np.empty((T, self.n_streams), np.float32)
        L = _ffi.PcmLayout(fmt, channels, int(bool(discard_first)), 0, N * channels, FRAME_SIZE * channels)
        fresh = self.frames_done == 0
        self._lib.check(self._lib.L.nnn_batch_process_pcm_host(self._h, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(vad), T, C.byref(L)))
        self.frames_done += T
        n_out = T - 1 if (discard_first and fresh) else T
        return out[:, :n_out * FRAME_SIZE], vad

    def process_device(self, d_in, d_out, d_vad, n_frames, stream_stride, frame_stride, hip_stream=0):
        """Raw device pointers (ints); asynchronous on hip_stream (0 = the batch's own stream)."""
        self._lib.check(self._lib.L.nnn_batch_process_device(self._h, d_in, d_out, d_vad, n_frames, stream_stride,
                                                             frame_stride, hip_stream))
        self.frames_done += n_frames

    def process_pcm_device(self, d_in, d_out, d_vad, n_frames, fmt, channels, group_stride, frame_stride, discard_first=False,
                           hip_stream=0):
        """Raw device pointers, strides in elements of the format (include/nnn_batch.h); asynchronous."""
        L = _ffi.PcmLayout(fmt, channels, int(bool(discard_first)), 0, group_stride, frame_stride)
        self._lib.check(self._lib.L.nnn_batch_process_pcm_device(self._h, d_in, d_out, d_vad, n_frames, C.byref(L), hip_stream))
        self.frames_done += n_frames

    def synchronize(self):
        self._lib.check(self._lib.L.nnn_batch_synchronize(self._h))

    def reset(self):
        self._lib.check(self._lib.L.nnn_batch_reset(self._h))
        self.frames_done = 0

    def tap(self, name):
        """Intermediate quantity of the most recent frame as [n_streams, len] (parity checks)."""
        t = _ffi.TAPS.index(name)
        ln, isint = C.c_int(), C.c_int()
        self._lib.check(self._lib.L.nnn_tap_info(t, C.byref(ln), C.byref(isint)))
        a = np.empty((self.n_streams, ln.value), np.int32 if isint.value else np.float32)
        self._lib.check(self._lib.L.nnn_batch_read_tap(self._h, t, _ffi.ptr(a), a.nbytes))
        return a

    def set_profiling(self, on):
        self._lib.check(self._lib.L.nnn_batch_set_profiling(self._h, int(on)))

    def set_graph(self, on):
        self._lib.check(self._lib.L.nnn_batch_set_graph(self._h, int(on)))

    def set_back_end(self, mode):
        """-1 (default): by batch size, as measured -- one-frame groups take the fused back end (k_back) up to 8192 streams and its RNN
        stretch alone above that or with other batches ticking on the same device, longer groups three launches; 0: transforms -> RNN ->
        synthesis as three launches; 1: one-frame groups take the fused back end; 2: every group; 3 / 4: the fused kernel's RNN stretch
        alone as the RNN kernel for one-frame / all groups (include/nnn_batch.h nnn_batch_set_back_end).  Every choice gives the same bits."""
        self._lib.check(self._lib.L.nnn_batch_set_back_end(self._h, int(mode)))

    def set_pipeline(self, on):
        self._lib.check(self._lib.L.nnn_batch_set_pipeline(self._h, int(on)))

    def set_inputs_ready(self, on):
        """Promise that the input of every process_device call is final when the call is made: consecutive calls may then
        overlap at their boundary (include/nnn_batch.h).  Outputs stay ordered on the caller's stream; same bits."""
        self._lib.check(self._lib.L.nnn_batch_set_inputs_ready(self._h, int(on)))

    def set_frame_log(self, d_log, frames):
        """Record (pitch index, branch mask, 22 smoothed gains) of the next `frames` frames into device memory at `d_log`
        (an int: [frames][n_streams][24] 32-bit words; include/nnn_batch.h nnn_batch_set_frame_log).  Parity tests."""
        self._lib.check(self._lib.L.nnn_batch_set_frame_log(self._h, d_log, int(frames)))

    def fault(self):
        """True once a pitch workgroup has given up waiting for the previous frame's hand-off (include/nnn_batch.h nnn_batch_fault):
        sticky until reset() / load_state().  A cheap host-memory read; no synchronisation."""
        fn = getattr(self._lib.L, "nnn_batch_fault", None)   # (absent from experimental builds of older sources: NNN_LIBRARY)
        return bool(fn(self._h)) if fn else False

    def kernel_times(self):
        """{kernel: (total_ms, launches)} accumulated while profiling; resets the counters."""
        n = self._lib.L.nnn_batch_num_kernels()
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        self._lib.check(self._lib.L.nnn_batch_read_kernel_times(self._h, ms, cnt, n))
        return {self._lib.L.nnn_batch_kernel_name(k).decode(): (ms[k], cnt[k]) for k in range(n)}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.L.nnn_batch_destroy(self._h)
            self._h = None

    __del__ = close


class DenoiseState:
    """One stream (reference: src/denoise.rs:36-116); a batch of one on the GPU."""

    FRAME_SIZE = FRAME_SIZE

    def __init__(self, model=None, device=0, lib=None, _batch=None):
        # process_frame takes one frame per call: a batch sized for one-frame groups
        self._b = _batch if _batch is not None else BatchDenoiser(1, model, device, lib, max_group_frames=1)

    def clone(self):
        """`impl Clone for DenoiseState` (src/denoise.rs:36)."""
        return DenoiseState(_batch=self._b.clone())

    @classmethod
    def new(cls, **kw):
        return cls(None, **kw)

    @classmethod
    def from_model(cls, model, **kw):
        return cls(model, **kw)

    with_model = from_model

    def process_frame(self, output, input):
        """Writes 480 samples into `output`, returns the VAD probability (src/denoise.rs:95-116)."""
        input = _ffi.as_f32(input)
        if input.shape != (FRAME_SIZE,) or output.shape != (FRAME_SIZE,):
            raise ValueError("process_frame needs slices of length DenoiseState.FRAME_SIZE")  # src/features.rs:98
        out, vad = self._b.process(input.reshape(1, 1, FRAME_SIZE))
        output[:] = out[0, 0]
        return float(vad[0, 0])


from .node import NodeDenoiser  # noqa: E402  (all the GPUs of a node behind one object, include/nnn_node.h)
