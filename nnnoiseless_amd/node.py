"""All the GPUs of a node behind one object (include/nnn_node.h): the streams cut into contiguous shards (shard.py's split), one batch and
one host thread per device, fan-out and join inside every call.  The reference's hosts walk a vector of independent states
(src/nnnoiseless.rs:305-320); this is that vector spread over devices."""
import ctypes as C

import numpy as np

from . import _ffi

FRAME_SIZE = _ffi.FRAME_SIZE


class NodeDenoiser:
    """n_streams DenoiseStates over `devices` (HIP ordinals; may repeat)."""

    def __init__(self, n_streams, devices, model=None, lib=None, max_group_frames=None):
        from . import library
        self._lib = lib or library()
        self._model = model
        self.n_streams = int(n_streams)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        opts = _ffi.BatchOpts(max_group_frames=int(max_group_frames)) if max_group_frames is not None else None
        self._h = self._lib.L.nnn_node_create(model._h if model is not None else None, self.n_streams, devs, len(devices),
                                              C.byref(opts) if opts is not None else None)
        if not self._h:
            raise RuntimeError("nnnoiseless_amd: " + self._lib.error())

    def shards(self):
        """[(device, lo, hi)] per shard."""
        out = []
        for i in range(self._lib.L.nnn_node_num_shards(self._h)):
            d, lo, hi = C.c_int(), C.c_int(), C.c_int()
            self._lib.check(self._lib.L.nnn_node_shard(self._h, i, C.byref(d), C.byref(lo), C.byref(hi)))
            out.append((d.value, lo.value, hi.value))
        return out

    def batch_handle(self, i):
        """The raw nnn_batch of shard i (owned by the node): for the per-device calls of include/nnn_batch.h."""
        return self._lib.L.nnn_node_batch(self._h, i)

    def process(self, x, out=None, vad=None):
        """x: float32 [n_streams, n_frames, 480] on the host -> (out, vad [n_frames, n_streams]); every device works on its share at once."""
        x = _ffi.as_f32(x)
        if x.ndim != 3 or x.shape[0] != self.n_streams or x.shape[2] != FRAME_SIZE:
            raise ValueError(f"process needs x of shape [{self.n_streams}, n_frames, {FRAME_SIZE}], got {x.shape}")
        S, T, _ = x.shape
        out = np.empty_like(x) if out is None else out
        vad = np.empty((T, S), np.float32) if vad is None else vad
        if not (isinstance(out, np.ndarray) and out.shape == x.shape and out.dtype == np.float32 and out.flags.c_contiguous and out.flags.writeable):
            raise ValueError("process: `out` must be a writable C-contiguous float32 array of x's shape")
        if not (isinstance(vad, np.ndarray) and vad.shape == (T, S) and vad.dtype == np.float32 and vad.flags.c_contiguous and vad.flags.writeable):
            raise ValueError("process: `vad` must be a writable C-contiguous float32 array of shape [n_frames, n_streams]")
        self._lib.check(self._lib.L.nnn_node_process_host(self._h, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(vad), T, T * FRAME_SIZE, FRAME_SIZE))
        return out, vad

    def process_pcm(self, x, fmt, channels=1, discard_first=False):
        """Packed PCM as BatchDenoiser.process_pcm over all the node's streams."""
        x = np.ascontiguousarray(x, dtype=_ffi.PCM_DTYPE[fmt])
        G, N, Cc = x.shape
        T = N // FRAME_SIZE
        out = np.zeros_like(x)
        vad = np.empty((T, self.n_streams), np.float32)
        L = _ffi.PcmLayout(fmt, channels, int(bool(discard_first)), 0, N * channels, FRAME_SIZE * channels)
        self._lib.check(self._lib.L.nnn_node_process_pcm_host(self._h, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(vad), T, C.byref(L)))
        return out, vad

    def shard_cpus(self, i):
        """The CPUs shard i's host thread is pinned to (its device's local_cpulist), "" when not pinned."""
        return (self._lib.L.nnn_node_shard_cpus(self._h, i) or b"").decode()

    def process_device(self, d_in, d_out, d_vad, n_frames, stream_stride, frame_stride, streams=None):
        """Per-shard device pointers (sequences of ints, one per shard, each on its shard's device) and, optionally, per-shard HIP
        streams to enqueue on (None / 0 entries: the shard's own stream); asynchronous, see synchronize()."""
        n = self._lib.L.nnn_node_num_shards(self._h)
        tables = {"d_in": d_in, "d_out": d_out}
        if d_vad is not None:
            tables["d_vad"] = d_vad
        if streams is not None:
            tables["streams"] = streams
        for name, v in tables.items():
            if len(v) != n:   # (the C side indexes every table by shard: a short one would be read past its end)
                raise ValueError(f"process_device: `{name}` has {len(v)} entries for {n} shards")
        arr = lambda v: (C.c_void_p * n)(*[int(p) if p else None for p in v])
        self._lib.check(self._lib.L.nnn_node_process_device_streams(self._h, arr(d_in), arr(d_out), arr(d_vad) if d_vad is not None else None,
                                                                    arr(streams) if streams is not None else None, n, n_frames,
                                                                    stream_stride, frame_stride))

    def synchronize(self):
        self._lib.check(self._lib.L.nnn_node_synchronize(self._h))

    def reset(self):
        self._lib.check(self._lib.L.nnn_node_reset(self._h))

    def fault(self):
        return bool(self._lib.L.nnn_node_fault(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.L.nnn_node_destroy(self._h)
            self._h = None

    __del__ = close
