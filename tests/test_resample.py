"""the CLI's 16-tap sinc resampler to 48 kHz (src/nnnoiseless.rs:19-32, 106-131), batched.
Oracle: nnno_resample, a restatement of Resample<RS> around dasp_interpolate 0.11.0's Sinc<[f32; 16]> (a crate outside the
reference tree: its semantics are restated from the published source and are NOT pinned by any reference test)."""
import numpy as np
import pytest

RATES = [44100.0, 16000.0, 96000.0, 8000.0, 47999.0]


def _signal(S, n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    f = rng.uniform(100.0, 3000.0, (S, 1))
    return (8000.0 * np.sin(2 * np.pi * f * t / 44100.0) + 300.0 * rng.standard_normal((S, n))).astype(np.float32)


def _check(lib, oracle_mod, S, n, rate, cuts):
    from nnnoiseless_amd.pcm import Resampler
    x = _signal(S, n, seed=int(rate))
    ref = np.stack([oracle_mod.resample(x[s], rate / 48000.0)[:, 0] for s in range(S)])
    rs = Resampler(S, rate, lib=lib)
    parts, pos = [], 0
    for c in cuts:
        parts.append(rs.process(x[:, pos:pos + c]))
        pos += c
    parts.append(rs.process(x[:, pos:]))
    out = np.concatenate(parts, axis=1)
    # the oracle stops at the output whose source sample is missing; the batched resampler keeps the same outputs
    assert out.shape[1] == ref.shape[1], (rate, out.shape, ref.shape)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (rate, np.abs(out - ref).max())
    rs.close()


def test_oracle_resampler_sanity(oracle_mod):
    """The restated interpolator passes a 1 kHz tone through at the right rate, delayed by its 7-sample centre."""
    n = 4410
    x = (10000.0 * np.sin(2 * np.pi * 1000.0 * np.arange(n) / 44100.0)).astype(np.float32)
    y = oracle_mod.resample(x, 44100.0 / 48000.0)[:, 0]
    assert abs(len(y) - n * 48000 / 44100) <= 2
    t = (np.arange(len(y)) + 1) * 44100.0 / 48000.0 - 8.0          # source position of output m: 8 samples behind the newest
    want = 10000.0 * np.sin(2 * np.pi * 1000.0 * t / 44100.0)
    assert np.abs(y[100:] - want[100:]).max() < 150.0               # windowed 16-tap sinc: ~1 % ripple


@pytest.mark.parametrize("rate", RATES)
def test_resampler_matches_oracle_hostsim(hostsim_lib, oracle_mod, rate):
    _check(hostsim_lib, oracle_mod, 3, 700, rate, (1, 7, 250))


@pytest.mark.gpu
@pytest.mark.parametrize("rate", RATES)
def test_resampler_matches_oracle_gpu(gpu_lib, oracle_mod, rate):
    _check(gpu_lib, oracle_mod, 70, 44100, rate, (1, 480, 10000, 3))


@pytest.mark.gpu
def test_resample_then_denoise_gpu(gpu_lib, oracle_mod, weights_bytes):
    """44.1 kHz input -> resampler -> denoiser, against the oracle's resampler + process_frame chain (the CLI's order,
    src/nnnoiseless.rs:179-227 then :301-331)."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.pcm import Resampler
    x = _signal(8, 44100, seed=5)
    y = Resampler(8, 44100.0, lib=gpu_lib).process(x)
    T = y.shape[1] // 480
    out, _ = nn.BatchDenoiser(8).process(y[:, :T * 480].reshape(8, T, 480))
    ref_y = np.stack([oracle_mod.resample(x[s], 44100.0 / 48000.0)[:T * 480, 0] for s in range(8)])
    ref = oracle_mod.run_streams(oracle_mod.Model(weights_bytes), ref_y.reshape(8, T, 480))
    d = out[:, 1:].astype(np.float64) - ref["out"][:, 1:]
    assert np.sqrt((d ** 2).sum() / (ref["out"][:, 1:].astype(np.float64) ** 2).sum()) < 1e-4
