"""Synthetic .rnn models for tests: random i8 weights at chosen layer widths, in the file layout of
src/rnn.rs:116-232 (dense, 3 GRUs, 2 dense; [inputs, neurons, activation] headers; input-major weights)."""
import numpy as np


def make_model(nd=16, nv=20, nn=40, ndn=72, acts=(0, 2, 2, 2, 1, 1), seed=0, scale=24):
    rng = np.random.default_rng(seed)

    def w(*shape):
        return np.clip(np.round(rng.standard_normal(shape) * scale), -127, 127).astype(np.int8).tobytes()

    def dense(i, n, a):
        return bytes([i, n, a]) + w(i, n) + w(n)

    def gru(i, n, a):
        return bytes([i, n, a]) + w(i, 3 * n) + w(n, 3 * n) + w(3 * n)

    return (dense(42, nd, acts[0]) + gru(nd, nv, acts[1]) + gru(nd + nv + 42, nn, acts[2]) + gru(nv + nn + 42, ndn, acts[3])
            + dense(ndn, 22, acts[4]) + dense(nv, 1, acts[5]))
