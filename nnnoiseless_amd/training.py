"""Batched training-feature rows: the per-frame body of the reference's training-data generator
(src/training.rs:113-160) for many (clean, noise, mix) stream triples at once, on the denoiser's own HIP kernels.

The simulator that feeds it -- wav reading, random gains and filters, the energy-based VAD label and the band
cutoff (src/training.rs:263-422) -- is host-side I/O and random policy and stays with the caller.
"""
import numpy as np

from . import FRAME_SIZE, NB_BANDS, NB_FEATURES, _ffi, library

ROW_WIDTH = NB_FEATURES + 2 * NB_BANDS + 1          # 87, src/training.rs:89


class TrainingFeatures:
    """n_streams x (clean, noise, mix) DenoiseFeatures states."""

    def __init__(self, n_streams, device=0, lib=None):
        self._lib = lib or library()
        self.n_streams = int(n_streams)
        self._h = self._lib.L.nnn_train_create(self.n_streams, device)
        if not self._h:
            raise RuntimeError("nnnoiseless_amd: " + self._lib.error())

    def process(self, signal, noise, combined, band_gain_cutoff, vad, rows=None):
        """signal / noise / combined: float32 [n_streams, n_frames, 480] (i16 range); band_gain_cutoff int32 and vad
        float32 [n_frames, n_streams].  Returns rows float32 [n_frames, n_streams, 87]:
        42 features of the mix | 22 gains | 22 noise levels | vad.  Long calls cross the bus in 16-frame chunks beside the
        kernels; arrays from nnnoiseless_amd.pinned_empty (inputs, and `rows` handed in) go by DMA."""
        signal, noise, combined = (_ffi.as_f32(a) for a in (signal, noise, combined))
        if signal.ndim != 3 or signal.shape[0] != self.n_streams or signal.shape[2] != FRAME_SIZE or not (noise.shape == signal.shape == combined.shape):
            raise ValueError(f"process needs three arrays of shape [{self.n_streams}, n_frames, {FRAME_SIZE}]")
        S, T, F = signal.shape
        cut = np.ascontiguousarray(band_gain_cutoff, dtype=np.int32)
        vad = _ffi.as_f32(vad)
        if cut.shape != (T, S) or vad.shape != (T, S):
            raise ValueError("band_gain_cutoff and vad must have shape [n_frames, n_streams]")
        if rows is None:
            rows = np.empty((T, S, ROW_WIDTH), np.float32)
        # a caller-supplied row buffer is written by the C library: a wrong size or layout must never get that far
        if not (isinstance(rows, np.ndarray) and rows.shape == (T, S, ROW_WIDTH) and rows.dtype == np.float32 and rows.flags.c_contiguous and rows.flags.writeable):
            raise ValueError(f"rows must be a writable C-contiguous float32 array of shape [n_frames, n_streams, {ROW_WIDTH}]")
        self._lib.check(self._lib.L.nnn_train_process_host(self._h, _ffi.ptr(signal), _ffi.ptr(noise), _ffi.ptr(combined),
                                                           _ffi.ptr(cut), _ffi.ptr(vad), _ffi.ptr(rows), T))
        return rows

    def process_device(self, d_signal, d_noise, d_combined, d_cutoff, d_vad, d_rows, n_frames, stream_stride, frame_stride,
                       hip_stream=0):
        """Raw device pointers (include/nnn_train.h); asynchronous."""
        self._lib.check(self._lib.L.nnn_train_process_device(self._h, d_signal, d_noise, d_combined, d_cutoff, d_vad, d_rows,
                                                             n_frames, stream_stride, frame_stride, hip_stream))

    def reset(self):
        self._lib.check(self._lib.L.nnn_train_reset(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.L.nnn_train_destroy(self._h)
            self._h = None

    __del__ = close
