"""Offline go / no-go study for a certified coarse pitch search (round 6, VERDICT r5 next #1b) -- CPU only, uses the oracle.

find_best_pitch over the 147 coarse lags (ref: src/pitch.rs:83-84, 372-405) only ever returns (best, second).  Question: if every
coarse cross-correlation is first computed APPROXIMATELY (two bf16 planes per operand on the matrix cores: hi*hi + hi*lo + lo*hi,
f32 accumulation) with a rigorous bound |approx - exact| <= eps * sqrt(xx * den_i), how many lags per frame cannot be ruled out of
the final pair and need the exact, reference-ordered sum?

    python scripts/pitch_survivors_study.py [streams] [frames]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle  # noqa: E402
from nnnoiseless_amd import synthetic  # noqa: E402

NLAG = 147


def pitch_bufs(model, x):
    """x [S][T][480] -> xlp [S*T][864], xcorr1 [S*T][147], best1 [S*T][2]"""
    S, T, _ = x.shape
    xlp = np.empty((S, T, 864), np.float32)
    xc = np.empty((S, T, NLAG), np.float32)
    b1 = np.empty((S, T, 2), np.int32)
    for s in range(S):
        st = oracle.State(model)
        for t in range(T):
            st.process_frame(x[s, t])
            tp = oracle.Taps()
            st._L.nnno_get_taps(st._h, oracle.C.byref(tp))
            xlp[s, t] = np.frombuffer(tp.xlp, np.float32)
            xc[s, t] = np.frombuffer(tp.xcorr1, np.float32)
            b1[s, t] = np.frombuffer(tp.best1, np.int32)
    return xlp.reshape(-1, 864), xc.reshape(-1, NLAG), b1.reshape(-1, 2)


def trunc_bf16(v):
    return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def den_sequence(y4):
    """the running energy every coarse lag sees, in the reference's order and arithmetic (f32)"""
    N = y4.shape[0]
    d = np.ones(N, np.float32)
    for j in range(240):
        d = d + y4[:, j] * y4[:, j]
    den = np.empty((N, NLAG), np.float32)
    for i in range(NLAG):
        den[:, i] = d
        d = d + (y4[:, i + 240] * y4[:, i + 240] - y4[:, i] * y4[:, i])
        d = np.maximum(d, np.float32(1.0))
    return den


def approx_xcorr(y4):
    """two bf16 planes per operand, three products, f32 accumulation over (k-step of 32 taps, product) in MFMA order"""
    N = y4.shape[0]
    x4 = y4[:, 192:432]
    hi = trunc_bf16(y4)
    lo = trunc_bf16(y4 - hi)
    xh, xl = hi[:, 192:432].astype(np.float64), lo[:, 192:432].astype(np.float64)
    yh, yl = hi.astype(np.float64), lo.astype(np.float64)
    acc = np.zeros((N, NLAG), np.float32)
    idx = np.arange(NLAG)[:, None] + np.arange(240)[None, :]      # [lag][tap]
    for k0 in range(0, 240, 32):
        k1 = min(240, k0 + 32)
        sl = idx[:, k0:k1]
        for a, b in ((xh, yh), (xh, yl), (xl, yh)):
            part = np.einsum("nk,nlk->nl", a[:, k0:k1], b[:, sl])
            acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc


def scan(xc, den, keep):
    """find_best_pitch over the lags where keep is set, reference arithmetic; returns best, second [N]"""
    N = xc.shape[0]
    f = np.float32
    bn = np.full(N, -1, f); sn = np.full(N, -1, f); bd = np.zeros(N, f); sd = np.zeros(N, f)
    b = np.zeros(N, np.int32); s = np.ones(N, np.int32)
    for i in range(NLAG):
        c = xc[:, i]
        y = den[:, i]
        num = c * c
        with np.errstate(over="ignore", invalid="ignore"):
            inn = keep[:, i] & (c > 0) & (num * sd > sn * y)
            top = inn & (num * bd > bn * y)
        mid = inn & ~top
        sn = np.where(top, bn, np.where(mid, num, sn)); sd = np.where(top, bd, np.where(mid, y, sd)); s = np.where(top, b, np.where(mid, i, s))
        bn = np.where(top, num, bn); bd = np.where(top, y, bd); b = np.where(top, i, b)
    return b, s


def study(name, xlp, xc, b1):
    N = xlp.shape[0]
    y4 = np.ascontiguousarray(xlp[:, 0::2])
    den = den_sequence(y4)
    full = np.ones((N, NLAG), bool)
    b, s = scan(xc, den, full)
    assert (b == b1[:, 0]).all() and (s == b1[:, 1]).all(), "the study's own scan disagrees with the oracle"
    ca = approx_xcorr(y4)
    xx = (y4[:, 192:432].astype(np.float64) ** 2).sum(1)
    norm = np.sqrt(xx[:, None] * den.astype(np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.where(norm > 0, np.abs(ca.astype(np.float64) - xc.astype(np.float64)) / norm, 0.0)
    print(f"== {name}: {N} frames; observed |approx - exact| / sqrt(xx den): max {rel.max():.3e} = 2^{np.log2(max(rel.max(), 1e-30)):.1f}, "
          f"99.9 % {np.quantile(rel, 0.999):.3e}")
    for log2eps in (-6, -7, -8, -10, -12, -14):
        eps = 2.0 ** log2eps
        e = eps * norm
        c_hi = ca.astype(np.float64) + e
        c_lo = ca.astype(np.float64) - e
        d64 = den.astype(np.float64)
        ub = np.where(c_hi > 0, c_hi * c_hi / d64, -1.0) * (1 + 2.0 ** -18)
        lb = np.where(c_lo > 0, c_lo * c_lo / d64, -np.inf) * (1 - 2.0 ** -18)
        srt = np.sort(lb, axis=1)
        thr = srt[:, -2]                                   # second-largest certain score (-inf: fewer than two certainly positive)
        keep = (c_hi > 0) & (ub >= thr[:, None])
        n = keep.sum(1)
        bb, ss = scan(xc, den, keep)
        bad = int(((bb != b1[:, 0]) | (ss != b1[:, 1])).sum())
        viol = int((rel > eps).sum())
        q = np.quantile(n, [0.5, 0.9, 0.99, 0.999])
        print(f"  eps 2^{log2eps}: survivors mean {n.mean():.2f}  median {q[0]:.0f}  90% {q[1]:.0f}  99% {q[2]:.0f}  99.9% {q[3]:.0f}  max {n.max()}"
              f"  frames > 32: {(n > 32).mean() * 100:.3f} %  > 8: {(n > 8).mean() * 100:.2f} %   wrong pairs {bad}   bound violations {viol}")


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    model = oracle.Model(open(os.path.join(os.path.dirname(oracle.__file__), "..", "nnnoiseless_amd", "data", "weights.rnn"), "rb").read())
    t0 = time.time()
    x = synthetic.make_streams(0, S, T)
    study(f"synthetic bench streams {S} x {T}", *pitch_bufs(model, x))
    print(f"   ({time.time() - t0:.0f} s)")
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "testing.raw")
    pcm = np.fromfile(g, np.int16).astype(np.float32)
    nfr = len(pcm) // 480
    rng = np.random.default_rng(7)
    real = np.stack([np.clip(pcm[:nfr * 480] * gain + rng.standard_normal(nfr * 480).astype(np.float32) * sig, -32768, 32767)
                     for gain, sig in ((1.0, 0.0), (0.25, 0.0), (1.0, 300.0), (0.05, 0.0), (0.5, 1000.0), (1.0, 30.0), (0.01, 0.0), (2.0, 0.0))])
    study("real audio (testing.raw at 8 gains / noise levels)", *pitch_bufs(model, real.reshape(len(real), nfr, 480)))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    try:
        import edge_streams
        ex = edge_streams.make_edge_streams(40)
        if ex is not None:
            study("edge streams", *pitch_bufs(model, np.asarray(ex, np.float32)))
    except Exception as e:   # noqa: BLE001
        print("edge streams skipped:", e)


if __name__ == "__main__":
    main()
