"""SURVEY.md 8(f) #1 on the CPU SIMT interpreter build of the product kernels: packed-PCM boundary formats (int16 /
unit float, channel interleave, dropped first frame) against the oracle's restatement of the reference's callers."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "hostsim"))
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def hostsim_lib():
    import build_hostsim
    from nnnoiseless_amd import _ffi
    return _ffi.Library(build_hostsim.build())


def _speech(n_frames, channels, seed=1):
    x = np.fromfile(os.path.join(GOLDEN, "testing.raw"), dtype="<i2")
    n = n_frames * 480
    cols = [np.roll(x, 977 * c)[:n] if c % 2 == 0 else np.roll(x, 977 * c)[:n][::-1] for c in range(channels)]
    return np.stack(cols, axis=1).astype(np.int16)


def test_interleaved_f32_matches_planar(hostsim_lib):
    """2 groups x 2 interleaved channels == the same 4 streams planar, bit for bit, with and without first-frame drop."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd import _ffi
    pcm = _speech(5, 4).astype(np.float32)                      # [n, 4 streams]
    planar = pcm.T.reshape(4, 5, 480)
    ref, vref = nn.BatchDenoiser(4, lib=hostsim_lib).process(planar)
    inter = pcm.reshape(-1, 2, 2).transpose(1, 0, 2)            # group g = streams 2g, 2g+1
    bd = nn.BatchDenoiser(4, lib=hostsim_lib)
    out, vad = bd.process_pcm(inter, _ffi.PCM_F32, 2)
    assert np.array_equal(vad, vref)
    for s in range(4):
        assert np.array_equal(out[s // 2, :, s % 2], ref[s].reshape(-1))
    bd.reset()
    o1, _ = bd.process_pcm(inter[:, :960], _ffi.PCM_F32, 2, discard_first=True)    # frames 0, 1 -> writes frame 1
    o2, _ = bd.process_pcm(inter[:, 960:], _ffi.PCM_F32, 2, discard_first=True)    # not fresh: nothing dropped
    assert o1.shape[1] == 480 and o2.shape[1] == 3 * 480
    assert np.array_equal(np.concatenate([o1, o2], axis=1), out[:, 480:])


@pytest.mark.parametrize("channels", [1, 2])
def test_cli_raw_i16(hostsim_lib, oracle_mod, weights_bytes, channels):
    """Interleaved int16 in and out as the CLI does it (src/nnnoiseless.rs:301-331), incl. a trailing partial frame."""
    from nnnoiseless_amd.pcm import denoise_raw_i16
    pcm = _speech(7, channels)[:-100]
    pcm[500:600] = 32767                                         # drive the clamp
    pcm[700:800, 0] = -32768
    ref = oracle_mod.cli_raw_i16(oracle_mod.Model(weights_bytes), pcm, channels)
    out = denoise_raw_i16(pcm, channels, lib=hostsim_lib, block_frames=4)
    assert out.shape == ref.shape == (5 * 480, channels) and out.dtype == np.int16
    d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3             # rounding ties only


def test_denoise_signal(hostsim_lib, oracle_mod, weights_bytes):
    """DenoiseSignal: x32768 in, /32768 + clamp out, first frame dropped, dasp end-of-signal rule (src/signal.rs)."""
    from nnnoiseless_amd.pcm import DenoiseSignal
    model = oracle_mod.Model(weights_bytes)
    x = _speech(6, 2).astype(np.float32) / 32768.0
    x[300:340] *= 40.0                                           # overdrive: the output clamp must engage
    for n in (0, 100, 480, 481, 960, 961, 1440, 2000, 2880):
        ref = oracle_mod.denoise_signal(model, x[:n], 2)
        out = DenoiseSignal(x[:n], lib=hostsim_lib).collect()
        assert out.shape == ref.shape, n
        assert np.abs(out - ref).max() <= 2e-5, n
    assert np.abs(out).max() <= 1.0
