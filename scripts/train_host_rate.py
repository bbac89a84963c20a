#!/usr/bin/env python3
"""PCIe-inclusive rate of the training-row host entry point: rows/s when every call ships its three audio arrays and labels from
host memory and brings the rows back.  usage: train_host_rate.py [triples] [frames_per_call] [calls]"""
import sys, time
import numpy as np
import nnnoiseless_amd as nn
from nnnoiseless_amd.training import ROW_WIDTH, TrainingFeatures

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 48
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(0)
sig = (rng.standard_normal((S, T, 480)) * 3000).astype(np.float32)
noise = (rng.standard_normal((S, T, 480)) * 300).astype(np.float32)
comb = sig + noise
cut = np.full((T, S), 20, np.int32)
vad = np.ones((T, S), np.float32)
tf = TrainingFeatures(S)

def run(label, args, rows):
    tf.process(*args, rows=rows)
    t0 = time.perf_counter()
    for _ in range(K):
        tf.process(*args, rows=rows)
    dt = (time.perf_counter() - t0) / K
    nbytes = 3 * sig.nbytes + cut.nbytes + vad.nbytes + rows.nbytes
    print(f"{label}: {S * T / dt / 1e6:.2f} M rows/s ({dt * 1e3:.1f} ms per call, {nbytes / dt / 1e9:.1f} GB/s over the bus)", flush=True)

run("pageable", (sig, noise, comb, cut, vad), np.empty((T, S, ROW_WIDTH), np.float32))
pin = []
for a in (sig, noise, comb, cut, vad):
    p = nn.pinned_empty(a.shape, a.dtype)
    p[:] = a
    pin.append(p)
run("pinned  ", tuple(pin), nn.pinned_empty((T, S, ROW_WIDTH)))
