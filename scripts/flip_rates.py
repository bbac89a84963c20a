"""Flip statistics with a sample that can carry the claim -> profiles/r6_parity_flip_rates.json.

A frame's pitch-filter branch mask (`exp > g` per band, src/features.rs:227; bit 22: the silence gate) is a jump discontinuity decided by the
rounding of the FFT upstream of it; the reference's own FFT picks SIMD code at run time and is only defined to f32 rounding.  Three
arithmetics run the same inputs: the GPU, the oracle with its FFT in f64 (the checker) and the oracle with its FFT in f32 (the reference's
own rustfft arithmetic).  For the three pairings: flipped frames per million with exact Poisson 95 % intervals; and, where nothing flipped,
how far apart the two sides of the comparison -- exp and the smoothed gain g -- are (GPU vs f64 against f32 vs f64).

    python scripts/flip_rates.py [streams=8192] [frames=250] [real-audio seeds=20]       (GPU box; ~2 M + ~1 M frames)
"""
import json
import os
import sys
import time

import numpy as np
from scipy.stats import chi2

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch  # noqa: E402

torch.cuda.init()   # (torch's copy of the HIP runtime before the library's: tests/conftest.py)
import nnnoiseless_amd as nn  # noqa: E402
from nnnoiseless_amd.synthetic import make_streams  # noqa: E402
from oracle import oracle  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
SEEDS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
W = open(os.path.join(R, "nnnoiseless_amd", "data", "weights.rnn"), "rb").read()
NT = os.cpu_count() or 1
CH = 1024   # streams per chunk (host memory: three audio arrays of CH x T x 480 floats)


def poisson_ci(k, n, conf=0.95):
    a = 1.0 - conf
    lo = 0.0 if k == 0 else chi2.ppf(a / 2, 2 * k) / 2
    hi = chi2.ppf(1 - a / 2, 2 * (k + 1)) / 2
    return [1e6 * lo / n, 1e6 * hi / n]


def gpu_run(x):
    """x [n, T, 480] host -> out [n, T, 480], branch [n, T], pitch [n, T], gains [n, T, 22], exp of the last frame [n, 22]"""
    n, Tn = x.shape[:2]
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x).to(dev)
    y = torch.empty_like(xd)
    log = torch.zeros((Tn, n, 24), dtype=torch.int32, device=dev)
    bd = nn.BatchDenoiser(n)
    bd.set_frame_log(log.data_ptr(), Tn)
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    pos = 0
    while pos < Tn:
        k = min(48, Tn - pos)
        bd.process_device(xd.data_ptr() + pos * 480 * 4, y.data_ptr() + pos * 480 * 4, 0, k, Tn * 480, 480, st)
        pos += k
    torch.cuda.synchronize()
    assert not bd.fault()
    ex = bd.tap("exp")
    bd.close()
    lg = log.cpu().numpy()
    return (y.cpu().numpy(), np.ascontiguousarray(lg[:, :, 1]).T.copy(), np.ascontiguousarray(lg[:, :, 0]).T.copy(),
            np.ascontiguousarray(lg[:, :, 2:]).view(np.float32).transpose(1, 0, 2).copy(), ex)


class Acc:
    def __init__(self):
        self.frames = 0
        self.flips = {"gpu_vs_f64": 0, "gpu_vs_f32": 0, "f32_vs_f64": 0}
        self.num = {k: 0.0 for k in self.flips}      # unmasked squared error of the audio
        self.den = 0.0
        self.numm = {k: 0.0 for k in self.flips}     # ... outside flipped frames (and the frame behind one)
        self.denm = {k: 0.0 for k in self.flips}
        self.pitch_mismatches = 0
        self.gerr = {"gpu_vs_f64": [], "f32_vs_f64": []}
        self.eerr = {"gpu_vs_f64": [], "f32_vs_f64": []}

    def add(self, x):
        want = ("out", "pitch", "branch", "gains", "exp")
        r64 = oracle.run_streams(oracle.Model(W), x, n_threads=NT, want=want)
        r32 = oracle.run_streams(oracle.Model(W, f32_fft=True), x, n_threads=NT, want=want)
        out, br, pitch, gains, ex_last = gpu_run(x)
        self.pitch_mismatches += int((pitch != r64["pitch"]).sum())
        self.frames += br.size
        sides = {"gpu": (br, out), "f32": (r32["branch"], r32["out"]), "f64": (r64["branch"], r64["out"])}
        rr = r64["out"][:, 1:].astype(np.float64)
        self.den += float((rr ** 2).sum())
        for k in self.flips:
            a, b = k.split("_vs_")
            flip = sides[a][0] != sides[b][0]
            self.flips[k] += int(flip.sum())
            d = (sides[a][1][:, 1:].astype(np.float64) - sides[b][1][:, 1:])
            self.num[k] += float((d ** 2).sum())
            ok = ~(flip | np.roll(flip, 1, axis=1))[:, 1:]
            self.numm[k] += float((d[ok] ** 2).sum())
            self.denm[k] += float((rr[ok] ** 2).sum())
        # the two sides of the comparison on frames where no pairing flipped: smoothed gains of every frame, exp of the chunk's last frame
        calm = (br == r64["branch"]) & (r32["branch"] == r64["branch"])
        self.gerr["gpu_vs_f64"].append(np.abs(gains - r64["gains"]).max(axis=2)[calm])
        self.gerr["f32_vs_f64"].append(np.abs(r32["gains"] - r64["gains"]).max(axis=2)[calm])
        cl = calm[:, -1]
        self.eerr["gpu_vs_f64"].append(np.abs(ex_last - r64["exp"][:, -1]).max(axis=1)[cl])
        self.eerr["f32_vs_f64"].append(np.abs(r32["exp"][:, -1] - r64["exp"][:, -1]).max(axis=1)[cl])

    def report(self):
        def q(v):
            v = np.concatenate(v)
            return {"mean": float(v.mean()), "p50": float(np.quantile(v, 0.5)), "p99": float(np.quantile(v, 0.99)), "max": float(v.max()), "n": int(v.size)}
        g, e = {k: q(v) for k, v in self.gerr.items()}, {k: q(v) for k, v in self.eerr.items()}
        return {"frames": self.frames, "pitch_mismatches": self.pitch_mismatches,
                "flips": self.flips,
                "flips_per_million": {k: 1e6 * v / self.frames for k, v in self.flips.items()},
                "flips_per_million_95pct_poisson": {k: poisson_ci(v, self.frames) for k, v in self.flips.items()},
                "rel_rms_unmasked": {k: (v / self.den) ** 0.5 for k, v in self.num.items()},
                "rel_rms_outside_flipped_frames": {k: (self.numm[k] / self.denm[k]) ** 0.5 for k in self.flips},
                "gain_error_where_nothing_flipped": g, "exp_error_last_frame_where_nothing_flipped": e,
                "gpu_over_f32_distance_to_f64": {"gains_mean": g["gpu_vs_f64"]["mean"] / max(g["f32_vs_f64"]["mean"], 1e-30), "gains_p99": g["gpu_vs_f64"]["p99"] / max(g["f32_vs_f64"]["p99"], 1e-30),
                                                 "exp_mean": e["gpu_vs_f64"]["mean"] / max(e["f32_vs_f64"]["mean"], 1e-30), "exp_p99": e["gpu_vs_f64"]["p99"] / max(e["f32_vs_f64"]["p99"], 1e-30)}}


def real_audio(n_streams, n_frames, seed):
    from conftest import GOLDEN
    pcm = np.fromfile(os.path.join(GOLDEN, "testing.raw"), dtype="<i2").astype(np.float64)
    x = np.zeros((n_streams, n_frames * 480), np.float32)
    rms = np.sqrt((pcm ** 2).mean())
    for i in range(n_streams):
        rng = np.random.default_rng(1000003 * seed + 31000 + i)
        off = int(rng.integers(0, 480))
        g = 10.0 ** (rng.uniform(-30.0, 6.0) / 20.0)
        pol = -1.0 if (i >> 1) & 1 else 1.0
        seg = pcm[off:off + n_frames * 480] * (g * pol)
        snr = (None, 30.0, 20.0, 10.0, 0.0)[i % 5]
        if snr is not None:
            seg = seg + rng.standard_normal(seg.size) * (g * rms * 10.0 ** (-snr / 20.0))
        x[i] = np.clip(np.round(seg), -32768, 32767)
    return x.reshape(n_streams, n_frames, 480)


t0 = time.time()
syn = Acc()
for c0 in range(0, S, CH):
    syn.add(make_streams(20000 + c0, min(CH, S - c0), T))
    print(f"synthetic {c0 + CH}/{S} streams, {time.time() - t0:.0f} s", file=sys.stderr)
real = Acc()
for sd in range(SEEDS):
    real.add(real_audio(512, 99, sd))
print(f"real audio done, {time.time() - t0:.0f} s", file=sys.stderr)
rep = {"what": "branch-mask flips and the distances behind them, three arithmetics on the same inputs (scripts/flip_rates.py; ref: src/features.rs:227)",
       "synthetic_bench_streams": dict(streams=S, frames_per_stream=T, **syn.report()),
       "real_audio_512_streams_x_99_frames_x_seeds": dict(seeds=SEEDS, **real.report()),
       "seconds": time.time() - t0, "host_threads": NT}
out = os.path.join(R, "gpurun_out", "r6_parity_flip_rates.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rep, open(out, "w"), indent=1)
for k in ("synthetic_bench_streams", "real_audio_512_streams_x_99_frames_x_seeds"):
    r = rep[k]
    print(k, "frames", r["frames"], "pitch mismatches", r["pitch_mismatches"], "flips/M", {a: round(b, 2) for a, b in r["flips_per_million"].items()},
          "CI", {a: [round(c, 2) for c in b] for a, b in r["flips_per_million_95pct_poisson"].items()}, "ratio", {a: round(b, 2) for a, b in r["gpu_over_f32_distance_to_f64"].items()})
