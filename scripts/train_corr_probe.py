"""Developer probe: the six pitch-correlation columns of the training rows (34..39), per stream: worst error against the oracle
as a multiple of the oracle's own f32-FFT / f64-FFT spread on that stream (library named by NNN_LIBRARY)."""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from nnnoiseless_amd.training import TrainingFeatures
from oracle import oracle as O
from train_fixtures import make_training_inputs
wb = open("nnnoiseless_amd/data/weights.rnn", "rb").read()
sig, noise, comb, cutoff, vad = make_training_inputs(9, 1000, 30)
nt = os.cpu_count() or 1
ref = O.training_rows(O.Model(wb), sig, noise, comb, cutoff, vad, n_threads=nt)
ref32 = O.training_rows(O.Model(wb, f32_fft=True), sig, noise, comb, cutoff, vad, n_threads=nt)
rows = TrainingFeatures(1000).process(sig, noise, comb, cutoff, vad)
err = np.abs(rows[..., 34:40] - ref[..., 34:40]).max(axis=(0, -1))
spread = np.abs(ref32[..., 34:40] - ref[..., 34:40]).max(axis=(0, -1))
ratio = err / np.maximum(spread, 2e-4 / 3)
e32 = np.abs(ref32[..., 34:40] - ref[..., 34:40])
print(os.environ.get("NNN_LIBRARY", "default").split("/")[-1], "streams over 3x:", int((ratio > 3).sum()), "over 10x:", int((ratio > 10).sum()),
      "worst ratio %.1f" % ratio.max(), "worst err %.2e" % err.max(), "rms err %.2e" % np.sqrt(((rows[..., 34:40] - ref[..., 34:40]) ** 2).mean()),
      "| oracle f32 build: worst %.2e rms %.2e" % (e32.max(), np.sqrt((e32 ** 2).mean())))
