"""ctypes wrapper around the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing under nnnoiseless_amd/ does.  See oracle/nnn_oracle.h for what the oracle is and how
its parity is pinned (reference golden vectors, src/lib.rs:184-213).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

FRAME_SIZE = 480
FREQ_SIZE = 481
NB_BANDS = 22
NB_FEATURES = 42


class Taps(C.Structure):
    _fields_ = [
        ("filtered", C.c_float * 480),
        ("xlp", C.c_float * 864),
        ("ac", C.c_float * 5),
        ("lpc2", C.c_float * 5),
        ("xcorr1", C.c_float * 147),
        ("best1", C.c_int32 * 2),
        ("xcorr2", C.c_float * 294),
        ("pitch_search", C.c_int32),
        ("pitch_idx", C.c_int32),
        ("pitch_gain", C.c_float),
        ("X", C.c_float * 962),
        ("P", C.c_float * 962),
        ("ex", C.c_float * 22),
        ("ep", C.c_float * 22),
        ("exp_", C.c_float * 22),
        ("features", C.c_float * 42),
        ("silence", C.c_int32),
        ("g_raw", C.c_float * 22),
        ("g", C.c_float * 22),
        ("vad", C.c_float),
        ("out", C.c_float * 480),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = np.array(v) if hasattr(v, "__len__") else v
        return d


def build(force=False):
    """Compile oracle/nnn_oracle.c with the committed Makefile (gcc only)."""
    so = os.path.join(_BUILD, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(
            os.path.join(_HERE, "nnn_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return so


_libs = {}


def lib(f32_fft=False):
    key = "f32" if f32_fft else "f64"
    if key not in _libs:
        build()
        path = os.path.join(_BUILD, "liboracle_f32.so" if f32_fft else "liboracle.so")
        L = C.CDLL(path)
        L.nnno_model_from_bytes.restype = C.c_void_p
        L.nnno_model_from_bytes.argtypes = [C.c_char_p, C.c_size_t]
        L.nnno_model_free.argtypes = [C.c_void_p]
        L.nnno_model_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.nnno_create.restype = C.c_void_p
        L.nnno_create.argtypes = [C.c_void_p]
        L.nnno_destroy.argtypes = [C.c_void_p]
        L.nnno_process_frame.restype = C.c_float
        L.nnno_process_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nnno_get_taps.argtypes = [C.c_void_p, C.POINTER(Taps)]
        L.nnno_run_streams.restype = C.c_int
        L.nnno_run_streams.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int]
        L.nnno_rfft960.argtypes = [C.c_void_p, C.c_void_p]
        L.nnno_irfft960.argtypes = [C.c_void_p, C.c_void_p]
        L.nnno_get_tables.argtypes = [C.c_void_p] * 4
        _libs[key] = L
    return _libs[key]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Model:
    """Parsed .rnn model; `Model(bytes)` raises ValueError where the reference returns None."""

    def __init__(self, data: bytes, f32_fft=False):
        self._L = lib(f32_fft)
        self._h = self._L.nnno_model_from_bytes(data, len(data))
        if not self._h:
            raise ValueError("malformed .rnn model")

    def shape(self):
        s = (C.c_int32 * 12)()
        self._L.nnno_model_shape(self._h, s)
        return list(s)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.nnno_model_free(self._h)
            self._h = None


class State:
    """One DenoiseState (reference: src/denoise.rs:37-116)."""

    def __init__(self, model: Model):
        self._model = model
        self._L = model._L
        self._h = self._L.nnno_create(model._h)

    def process_frame(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.float32)
        assert frame.shape == (FRAME_SIZE,)
        out = np.empty(FRAME_SIZE, np.float32)
        vad = self._L.nnno_process_frame(self._h, _ptr(out), _ptr(frame))
        return out, float(vad)

    def taps(self):
        t = Taps()
        self._L.nnno_get_taps(self._h, C.byref(t))
        return t.as_dict()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.nnno_destroy(self._h)
            self._h = None


def run_streams(model: Model, x, n_threads=1, want=("out", "vad", "pitch", "gains", "feats", "cond")):
    """x: [S][T][480] float32.  Returns dict of arrays (fresh state per stream).  "cond" = per-frame conditioning of the
    reference's pitch-filter branch (nnno_frame_condition): tiny values mark frames the reference itself cannot pin;
    "branch" = nnno_frame_branch (22 `exp > g` bits + silence bit), "g_raw" / "exp" = the two sides of that comparison."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    S, Tn, F = x.shape
    assert F == FRAME_SIZE
    res = {
        "out": np.empty((S, Tn, 480), np.float32) if "out" in want else None,
        "vad": np.empty((S, Tn), np.float32) if "vad" in want else None,
        "pitch": np.empty((S, Tn), np.int32) if "pitch" in want else None,
        "gains": np.empty((S, Tn, 22), np.float32) if "gains" in want else None,
        "feats": np.empty((S, Tn, 42), np.float32) if "feats" in want else None,
        "cond": np.empty((S, Tn), np.float32) if "cond" in want else None,
        "branch": np.empty((S, Tn), np.int32) if "branch" in want else None,
        "g_raw": np.empty((S, Tn, 22), np.float32) if "g_raw" in want else None,
        "exp": np.empty((S, Tn, 22), np.float32) if "exp" in want else None,
    }
    fn = model._L.nnno_run_streams_full
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 10 + [C.c_int]
    used = fn(model._h, S, Tn, _ptr(x), _ptr(res["out"]), _ptr(res["vad"]), _ptr(res["pitch"]), _ptr(res["gains"]),
              _ptr(res["feats"]), _ptr(res["cond"]), _ptr(res["branch"]), _ptr(res["g_raw"]), _ptr(res["exp"]), int(n_threads))
    res["threads"] = used
    return res


def activation(x, kind):
    """tansig_approx (kind 0) / sigmoid_approx (kind 1) of src/util.rs:29-53, elementwise."""
    L = lib()
    fn = L.nnno_sigmoid if kind else L.nnno_tansig
    fn.restype = C.c_float
    fn.argtypes = [C.c_float]
    x = np.asarray(x, np.float32)
    return np.array([fn(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


def bench(model: Model, n_threads, iters, kind):
    """nnno_bench: (wall seconds, per-thread seconds) for n_threads x iters x 100 frames; kind 0 synthetic mix on a
    continuing state, kind 1 the reference's benches/sin.rs shape (fresh state per 100-frame iteration)."""
    fn = model._L.nnno_bench
    fn.restype = C.c_double
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    secs = np.zeros(max(1, n_threads), np.float64)
    wall = fn(model._h, int(n_threads), int(iters), int(kind), _ptr(secs))
    return float(wall), secs


def cli_raw_i16(model: Model, pcm, channels=1):
    """The CLI's raw-PCM loop (src/nnnoiseless.rs:301-331): pcm int16 [n, channels] -> int16 [n_out, channels]."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16).reshape(-1, channels)
    out = np.zeros_like(pcm)
    fn = model._L.nnno_cli_raw_i16
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    n = fn(model._h, _ptr(pcm), len(pcm), channels, _ptr(out))
    return out[:n]


def denoise_signal(model: Model, x, channels=1):
    """DenoiseSignal (src/signal.rs:83-137) over unit-range floats [n, channels] -> float32 [n_out, channels]."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, channels)
    out = np.zeros(((len(x) // FRAME_SIZE + 2) * FRAME_SIZE, channels), np.float32)
    fn = model._L.nnno_denoise_signal
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    n = fn(model._h, _ptr(x), len(x), channels, _ptr(out))
    return out[:n]


def resample(x, ratio, channels=1):
    """The CLI's 16-tap sinc resampler (src/nnnoiseless.rs:106-131): x float32 [n, channels] at the source rate,
    ratio = source_rate / 48000 -> float32 [n_out, channels]."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, channels)
    cap = int(len(x) / ratio) + 8
    out = np.zeros((cap, channels), np.float32)
    fn = lib().nnno_resample
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_void_p, C.c_long]
    n = fn(_ptr(x), len(x), channels, float(ratio), _ptr(out), cap)
    return out[:n]


def training_rows(model: Model, signal, noise, combined, cutoff, vad, n_threads=1):
    """src/training.rs:113-160 for [S][T][480] inputs and [T][S] cutoff / vad -> rows [T][S][87]."""
    signal, noise, combined = (np.ascontiguousarray(a, dtype=np.float32) for a in (signal, noise, combined))
    S, T, _ = signal.shape
    cutoff = np.ascontiguousarray(cutoff, dtype=np.int32)
    vad = np.ascontiguousarray(vad, dtype=np.float32)
    rows = np.empty((T, S, 87), np.float32)
    fn = model._L.nnno_training_rows
    fn.restype = None
    fn.argtypes = [C.c_void_p] + [C.c_int] * 2 + [C.c_void_p] * 6 + [C.c_int]
    fn(model._h, S, T, _ptr(signal), _ptr(noise), _ptr(combined), _ptr(cutoff), _ptr(vad), _ptr(rows), int(n_threads))
    return rows


def rfft960(x, f32_fft=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(962, np.float32)
    lib(f32_fft).nnno_rfft960(_ptr(x), _ptr(out))
    return out[0::2] + 1j * out[1::2]


def irfft960(X, f32_fft=False):
    buf = np.empty(962, np.float32)
    buf[0::2] = np.real(X)
    buf[1::2] = np.imag(X)
    out = np.empty(960, np.float32)
    lib(f32_fft).nnno_irfft960(_ptr(buf), _ptr(out))
    return out


def tables():
    w = np.empty(960, np.float32)
    d = np.empty(22 * 22, np.float32)
    n = np.empty(1, np.float32)
    t = np.empty(201, np.float32)
    lib().nnno_get_tables(_ptr(w), _ptr(d), _ptr(n), _ptr(t))
    return w, d.reshape(22, 22), float(n[0]), t
