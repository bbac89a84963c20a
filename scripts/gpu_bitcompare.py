#!/usr/bin/env python3
"""Two builds of the library on the same input, bit for bit: audio, VAD and the per-frame record (pitch, branch mask, gains) of every frame.
For changes that must not change a bit (LDS placement, DPP instead of shuffles, constants kept in registers).
usage: gpu_bitcompare.py other.so [streams] [frames]      (the first build is the product library)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import nnnoiseless_amd as nn
from nnnoiseless_amd import _ffi
from nnnoiseless_amd.synthetic import make_streams

other = _ffi.Library(sys.argv[1])
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T = int(sys.argv[3]) if len(sys.argv) > 3 else 30
x = make_streams(123, S, T)
res = []
for lib in (nn.library(), other):
    bd = nn.BatchDenoiser(S, lib=lib, taps=True)
    outs, vads, taps = [], [], []
    for n in (24, 1, T - 25):     # a full group, a one-frame call (the fused back end), the rest
        o, v = bd.process(x[:, sum(len(t) for t in taps) if False else sum(k.shape[1] for k in outs):][:, :n]) if n > 0 else (None, None)
        if o is not None:
            outs.append(o); vads.append(v)
            taps.append({k: bd.tap(k).copy() for k in ("X", "P", "ex", "ep", "exp", "features", "g")})
    res.append((np.concatenate(outs, 1), np.concatenate(vads, 0), taps))
    bd.close()
a, b = res
same = np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
for ta, tb in zip(a[2], b[2]):
    for k in ta:
        if not np.array_equal(ta[k].view(np.uint32), tb[k].view(np.uint32)):
            same = False
            print("tap", k, "differs: max abs", np.abs(ta[k] - tb[k]).max())
print(f"{S} streams x {T} frames:", "BIT-IDENTICAL" if same else "DIFFERENT", "| out max abs diff", float(np.abs(a[0] - b[0]).max()))
sys.exit(0 if same else 1)
