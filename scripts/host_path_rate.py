#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points: frames/s when every call ships its audio from host
memory and brings the result back.  usage: host_path_rate.py [streams] [frames_per_call] [calls]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nnnoiseless_amd as nn
from nnnoiseless_amd import _ffi

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 48
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
lib = nn.library()
rng = np.random.default_rng(0)

def run(label, x, out, vad, call):
    call()   # staging buffers, first touch
    t0 = time.perf_counter()
    for _ in range(K):
        call()
    dt = (time.perf_counter() - t0) / K
    print(f"{label}: {S * T / dt / 1e6:.2f} M frames/s ({dt * 1e3:.1f} ms per call, {x.nbytes * 2 / dt / 1e9:.1f} GB/s both ways)", flush=True)

PAGEABLE = os.environ.get("PAGEABLE", "1") != "0"
for fmt, name in ((0, "f32"), (1, "i16")):
    bd = nn.BatchDenoiser(S)
    dt = np.float32 if fmt == 0 else np.int16
    x = (rng.standard_normal((S, T * 480)) * 3000).astype(dt)
    out = np.empty_like(x)
    vad = np.empty((T, S), np.float32)
    L = _ffi.PcmLayout(fmt, 1, 0, 0, T * 480, 480)
    if PAGEABLE:
        run(f"{name} pageable", x, out, vad,
            lambda: lib.check(lib.L.nnn_batch_process_pcm_host(bd._h, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(vad), T, C.byref(L))))
    px, po, pv = nn.pinned_empty(x.shape, dt), nn.pinned_empty(x.shape, dt), nn.pinned_empty((T, S))
    px[:] = x
    run(f"{name} pinned  ", x, out, vad,
        lambda: lib.check(lib.L.nnn_batch_process_pcm_host(bd._h, _ffi.ptr(px), _ffi.ptr(po), _ffi.ptr(pv), T, C.byref(L))))
    bd.close()
