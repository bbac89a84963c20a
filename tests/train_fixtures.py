"""Inputs for the training-row tests: a deterministic stand-in for the reference's NoiseSimulator outputs
(src/training.rs:263-422) -- speech-like and noise-like streams with per-stream gains, silent stretches, the
energy-based VAD label (:364-384) and a band cutoff."""
import numpy as np


def make_training_inputs(seed, n_streams, n_frames):
    from nnnoiseless_amd.synthetic import make_streams
    rng = np.random.default_rng(seed)
    sig = make_streams(seed, n_streams, n_frames).astype(np.float32)
    noise = (rng.standard_normal((n_streams, n_frames, 480)) * rng.uniform(5, 2000, (n_streams, 1, 1))).astype(np.float32)
    sig[rng.random(n_streams) < 0.15] = 0.0                       # signal_gain = 0 (src/training.rs:345-347)
    quiet = rng.random((n_streams, n_frames)) < 0.1
    sig[quiet] = 0.0
    noise[rng.random((n_streams, n_frames)) < 0.1] = 0.0          # silent mix frames when both are quiet
    noise[quiet & (rng.random((n_streams, n_frames)) < 0.5)] = 0.0
    comb = sig + noise
    e = (sig.astype(np.float64) ** 2).sum(axis=2)                 # vad(): 0 / 0.5 / 1 from a counter; any label will do
    vad = np.where(e > 1e9, 1.0, np.where(e > 1e7, 0.5, 0.0)).astype(np.float32).T.copy()
    cutoff = rng.integers(0, 23, (n_frames, n_streams)).astype(np.int32)
    return sig, noise, comb, cutoff, vad


def check_rows(rows, ref, ref32=None):
    """Cepstral / delta / variability features within 2e-4 absolute (values of order 1..20; the oracle's own f32-FFT and
    f64-FFT builds differ by up to 4e-5 on them).  The six pitch-correlation features (columns 34..39: band correlation /
    sqrt(.001 + Ex Ep)) are ill-conditioned on bands that hold only rounding noise -- the two oracle builds differ by up
    to 1.1e-2 there, 8e-5 rms.  With `ref32` (the oracle's f32-FFT build on the same inputs) they are judged against that
    build's own error (see below); without it, 4e-2 worst case.  5e-4 rms either way.  Gains and log levels within 1e-4; the -1 markers,
    zeroed silent rows and the vad column exactly."""
    assert rows.shape == ref.shape
    assert np.array_equal(rows[..., 86], ref[..., 86])
    assert np.array_equal(rows[..., 42:64] == -1.0, ref[..., 42:64] == -1.0)
    assert np.array_equal((rows[..., :42] == 0).all(axis=-1), (ref[..., :42] == 0).all(axis=-1))
    d = np.abs(rows[..., :42] - ref[..., :42]).max(axis=tuple(range(rows.ndim - 1)))
    assert np.delete(d, slice(34, 40)).max() < 2e-4, d
    if ref32 is not None:
        # rows are [frame][stream][87].  The yardstick is what f32 rounding in the FFT alone does to these quotients of two
        # noise-floor energies: the oracle's own f32-FFT build against its f64-FFT build.  As a whole the device's error must
        # not exceed that build's (worst case and rms, 25 % margin).  Stream by stream (10 x the stream's own spread was
        # the loosest bar of the suite): within 2e-4 or FIVE times the stream's spread for all but 1 % of the streams -- one pair of
        # builds is a small sample of a chaotic quantity, and at three times 1 % of the streams fail for every build measured
        # (round 2's kernels, round 3's, the oracle's f32 build judged against a second f32 build: scripts/train_corr_probe.py) --
        # and those few within twenty times, under the global worst-case bound above.  Said plainly: against round 3's single
        # 10 x bar this is TIGHTER for 99 % of the streams (5 x) and LOOSER for the worst 1 % (20 x).
        e_dev, e_o32 = np.abs(rows[..., 34:40] - ref[..., 34:40]), np.abs(ref32[..., 34:40] - ref[..., 34:40])
        assert e_dev.max() <= 1.25 * e_o32.max(), (e_dev.max(), e_o32.max())
        assert np.sqrt((e_dev ** 2).mean()) <= 1.25 * np.sqrt((e_o32 ** 2).mean())
        err, spread = e_dev.max(axis=(0, -1)), e_o32.max(axis=(0, -1))
        over5 = np.argwhere(err > np.maximum(2e-4, 5.0 * spread))
        assert len(over5) <= max(1, err.size // 100), (len(over5), err.size, over5[:8].ravel())
        bad = np.argwhere(err > np.maximum(2e-4, 20.0 * spread))
        assert not len(bad), (len(bad), bad[:8].ravel(), err[bad[0]], spread[bad[0]])
        print(f"pitch-correlation features: worst error {e_dev.max():.2e} (oracle f32 build {e_o32.max():.2e}), rms {np.sqrt((e_dev ** 2).mean()):.2e} "
              f"({np.sqrt((e_o32 ** 2).mean()):.2e}); {len(over5)} of {err.size} streams beyond 5 x their own spread, worst at {(err / np.maximum(spread, 4e-5)).max():.1f} x")
    else:
        assert d[34:40].max() < 4e-2, d
    assert np.sqrt(((rows[..., 34:40] - ref[..., 34:40]) ** 2).mean()) < 5e-4
    assert np.abs(rows[..., 42:86] - ref[..., 42:86]).max() < 1e-4
