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


def test_host_calls_in_chunks_are_bit_identical(hostsim_lib, monkeypatch):
    """A long host-buffer call crosses the bus in chunks of frames (include/nnn_batch.h nnn_batch_process_host); the chunks must
    not show: planar f32, and packed int16 stereo with the dropped first frame (outputs one frame ahead of their inputs)."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd import _ffi
    pcm = _speech(7, 4)
    planar = np.ascontiguousarray(pcm.T.reshape(4, 7, 480).astype(np.float32))
    inter = np.ascontiguousarray(pcm.reshape(-1, 2, 2).transpose(1, 0, 2))   # 2 groups x 2 channels
    res = {}
    for chunk in ("0", "1", "2", "3"):      # (1: what the default picks for short calls on large batches since round 5)
        monkeypatch.setenv("NNN_HOST_CHUNK", chunk)
        bd = nn.BatchDenoiser(4, lib=hostsim_lib)
        a, va = bd.process(planar)
        bd.reset()
        b, vb = bd.process_pcm(inter, _ffi.PCM_I16, 2, discard_first=True)
        assert b.shape == (2, 6 * 480, 2)
        res[chunk] = (a, va, b, vb)
    for chunk in ("1", "2", "3"):
        for u, v in zip(res["0"], res[chunk]):
            assert np.array_equal(u, v), chunk


def test_pinned_host_arrays(hostsim_lib):
    """pinned_empty arrays behave like any other numpy array (and free their block with their last view)."""
    import nnnoiseless_amd as nn
    x = nn.pinned_empty((3, 2, 480), lib=hostsim_lib)
    x[:] = _speech(2, 3).T.reshape(3, 2, 480)
    out, vad = nn.pinned_empty((3, 2, 480), lib=hostsim_lib), nn.pinned_empty((2, 3), lib=hostsim_lib)
    ref, vref = nn.BatchDenoiser(3, lib=hostsim_lib).process(np.array(x))
    o, v = nn.BatchDenoiser(3, lib=hostsim_lib).process(x, out=out, vad=vad)
    assert o is out and v is vad and np.array_equal(out, ref) and np.array_equal(vad, vref)
    view = out[1]
    del out, o
    assert np.array_equal(view, ref[1])
