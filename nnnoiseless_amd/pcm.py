"""The reference's multi-channel callers of process_frame, as batched calls on packed PCM.

  denoise_raw_i16   the CLI's raw-PCM loop: interleaved int16 in, interleaved int16 out, trailing partial frame
                    dropped, first frame not written (src/nnnoiseless.rs:301-331, RawFrameWriter :147-160)
  DenoiseSignal     the dasp adapter: unit-range float frames in and out, one state per channel, first frame
                    discarded (src/signal.rs:31-137)

Sample conversion, channel (de)interleave and the dropped first frame happen inside the first and last HIP kernels
(include/nnn_batch.h, nnn_batch_process_pcm_*); nothing here computes on the host beyond slicing.
"""
import numpy as np

from . import FRAME_SIZE, BatchDenoiser, _ffi

PCM_F32, PCM_I16, PCM_F32_UNIT = _ffi.PCM_F32, _ffi.PCM_I16, _ffi.PCM_F32_UNIT


def denoise_raw_i16(pcm, channels=1, model=None, block_frames=512, device=0, lib=None):
    """pcm: int16, either [n, channels] (one file) or [n_files, n, channels] (equal-length files denoised together).
    Returns int16 of the same rank with (n // 480 - 1) * 480 sample frames per file."""
    pcm = np.asarray(pcm, dtype=np.int16)
    single = pcm.ndim == 2
    if pcm.ndim == 1:
        pcm, single = pcm.reshape(-1, channels), True
    if single:
        pcm = pcm[None]
    G, n, C = pcm.shape
    if C != channels:
        raise ValueError("last axis must be the channel axis")
    T = n // FRAME_SIZE                                    # src/nnnoiseless.rs:303-311: a short read ends the loop
    bd = BatchDenoiser(G * channels, model, device, lib)
    outs = []
    for t0 in range(0, T, block_frames):
        t1 = min(T, t0 + block_frames)
        o, _ = bd.process_pcm(pcm[:, t0 * FRAME_SIZE:t1 * FRAME_SIZE], PCM_I16, channels, discard_first=True)
        outs.append(o)
    bd.close()
    out = np.concatenate(outs, axis=1) if outs else np.zeros((G, 0, channels), np.int16)
    return out[0] if single else out


class DenoiseSignal:
    """DenoiseSignal over an in-memory signal: `input` float32 [n, channels] in [-1, 1] (src/signal.rs:31-137).

    Iterating yields denoised sample frames (arrays of `channels` floats); `collect()` returns them all as [n_out,
    channels].  End of signal follows dasp_signal 0.11's `from_iter` source, which reports exhaustion as soon as its
    last frame has been taken: the refill that consumes the last input samples returns false, so its frame is never
    handed out (src/signal.rs:90-106, :129-134)."""

    def __init__(self, input, model=None, channels=None, device=0, lib=None):
        x = np.asarray(input, dtype=np.float32)
        if x.ndim == 1:
            x = x[:, None]
        self.channels = channels or x.shape[1]
        if x.shape[1] != self.channels:
            raise ValueError("last axis must be the channel axis")
        n = len(x)
        # refill k reads samples [480 k, 480 k + 480), zero ("equilibrium") padded, while any input is left
        n_proc = -(-n // FRAME_SIZE)
        pad = np.zeros((n_proc * FRAME_SIZE, self.channels), np.float32)
        pad[:n] = x
        if n_proc:
            bd = BatchDenoiser(self.channels, model, device, lib)
            y, _ = bd.process_pcm(pad[None], PCM_F32_UNIT, self.channels, discard_first=False)
            bd.close()
            y = y[0].reshape(n_proc, FRAME_SIZE, self.channels)
        else:
            y = np.zeros((0, FRAME_SIZE, self.channels), np.float32)
        zeros = np.zeros((FRAME_SIZE, self.channels), np.float32)

        def held(k):   # out_bufs after refills 0..k (a refill on an exhausted input leaves them untouched)
            k = min(k, n_proc - 1)
            return y[k] if k >= 0 else zeros

        frames = [held(1)]                                  # the constructor refills twice, :83-87
        k = 2
        while FRAME_SIZE * k < n and FRAME_SIZE * (k + 1) < n:   # refill k: input left before, and after
            frames.append(held(k))
            k += 1
        self._out = np.concatenate(frames, axis=0)

    def collect(self):
        return self._out

    def __iter__(self):
        return iter(self._out)

    def __len__(self):
        return len(self._out)


class Resampler:
    """n_streams mono streams of one common sample rate -> 48 kHz with the CLI's 16-tap windowed sinc
    (src/nnnoiseless.rs:19-32, 106-131), batched on the GPU (include/nnn_resample.h).  Feed the streams in chunks of any
    size: the output does not depend on the chunking."""

    def __init__(self, n_streams, source_rate, device=0, lib=None):
        from . import library
        self._lib = lib or library()
        self.n_streams = int(n_streams)
        self.ratio = float(source_rate) / 48000.0
        self._h = self._lib.L.nnn_resampler_create(self.n_streams, self.ratio, device)
        if not self._h:
            raise RuntimeError("nnnoiseless_amd: " + self._lib.error())

    def process(self, x):
        """x: float32 [n_streams, n] source samples -> float32 [n_streams, n_out]."""
        import ctypes as C
        x = _ffi.as_f32(x)
        S, n = x.shape
        assert S == self.n_streams
        cap = self._lib.L.nnn_resampler_max_output(self._h, n)
        out = np.zeros((S, cap), np.float32)
        n_out = C.c_long(0)
        self._lib.check(self._lib.L.nnn_resampler_process_host(self._h, _ffi.ptr(x), n, _ffi.ptr(out), cap, C.byref(n_out)))
        return np.ascontiguousarray(out[:, :n_out.value])

    def reset(self):
        self._lib.check(self._lib.L.nnn_resampler_reset(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.L.nnn_resampler_destroy(self._h)
            self._h = None

    __del__ = close
