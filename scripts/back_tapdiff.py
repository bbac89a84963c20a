#!/usr/bin/env python3
"""Where do the fused back end (mode 2) and the unfused kernels (mode 0) part on the GPU?  First frame of a fresh batch, taps on:
the quantities in pipeline order with their largest absolute difference (0 = bit-identical)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nnnoiseless_amd as nn
from nnnoiseless_amd.synthetic import make_streams
S = 256
x = make_streams(5, S, 3)
res = {}
for mode in (0, 2):
    bd = nn.BatchDenoiser(S, taps=True)
    bd.set_back_end(mode)
    out, vad = bd.process(x[:, :1])
    res[mode] = {t: bd.tap(t) for t in ("filtered", "pitch", "X", "P", "ex", "ep", "exp", "features", "g_raw", "g", "vad", "branch")}
    res[mode]["out"] = out
    bd.close()
for t in res[0]:
    a, b = res[0][t], res[2][t]
    same = np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    print(f"{t:10s} {'identical' if same else 'DIFFERS'}  max abs diff {d.max():.3e}  (max abs value {np.abs(a).max():.3e}) differing entries {int((d > 0).sum())} of {d.size}")
