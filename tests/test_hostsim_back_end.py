"""The fused back end (k_back, nnn_back.hip) under the test-only SIMT interpreter: every way of running the part of a frame behind
the pitch analysis gives the same bits -- here and on the GPU (tests/test_gpu_back_end.py): since round 4 a multiply fuses with an add
only where the source says fmaf, so the transforms round the same way in every kernel they are inlined into (include/nnn_batch.h promises
bit equality between the back ends)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _run(nn, lib, x, mode, chunks, model=None, taps=False, groups=None):
    S = x.shape[0]
    bd = nn.BatchDenoiser(S, model=model, lib=lib, taps=taps, groups=groups)
    bd.set_back_end(mode)
    outs, vads, t = [], [], 0
    for n in chunks:
        o, v = bd.process(x[:, t:t + n])
        outs.append(o)
        vads.append(v)
        t += n
    return bd, np.concatenate(outs, 1), np.concatenate(vads, 0)


def _bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("mode,chunks", [(2, (5,)), (1, (1, 1, 1, 1, 1)), (2, (2, 1, 2)), (4, (5,)), (3, (1, 1, 3))])
def test_back_ends_give_the_same_bits(hostsim_lib, mode, chunks):
    """nnn_batch_set_back_end: 0 = k_fft_xp -> k_rnn / k_rnn_wf -> k_synth; 1 / 2 = the fused kernel for one-frame / all groups; 3 / 4 =
    its RNN stretch alone as the RNN kernel.  70 streams (a full tile and a ragged one, the synthetic mix's silent and noise-only
    streams among them), groups and ticks mixed: audio, VAD and the taps the later kernels leave behind agree bit for bit."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 70, 5
    x = make_streams(3, S, T)
    ref, want, want_vad = _run(nn, hostsim_lib, x, 0, (T,))
    bd, got, vad = _run(nn, hostsim_lib, x, mode, chunks)
    assert np.array_equal(_bits(got), _bits(want)) and np.array_equal(_bits(vad), _bits(want_vad))
    for tap in ("g", "g_raw", "vad", "branch", "ex", "ep", "exp", "silence", "pitch"):
        assert np.array_equal(_bits(bd.tap(tap)), _bits(ref.tap(tap))), tap
    assert (ref.tap("silence")[:, 0] != 0).any() and (ref.tap("silence")[:, 0] == 0).any()


def test_ticks_with_and_without_the_riding_transform(hostsim_lib, monkeypatch):
    """NNN_X_RIDES=0: the fused kernel of a one-frame call computes X itself, as it does for groups; by default rider blocks of k_pitch's
    launch do.  Same bits, and the same as three launches."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 70, 4
    x = make_streams(8, S, T)
    _, want, want_vad = _run(nn, hostsim_lib, x, 0, (T,))
    for rides in ("0", "1"):
        monkeypatch.setenv("NNN_X_RIDES", rides)
        _, got, vad = _run(nn, hostsim_lib, x, 1, (1,) * T)
        assert np.array_equal(_bits(got), _bits(want)) and np.array_equal(_bits(vad), _bits(want_vad)), rides


def test_fused_back_end_taps_log_models_and_pcm(hostsim_lib):
    """The fused kernel with the parity taps on (both spectra and the 42 features reach memory only then), the per-frame record, a
    model of the same shape class with other activation kinds (sh.rnn), two models resident at once, and the packed int16 boundary
    with a dropped first frame: all as the unfused kernels do it."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd import _ffi
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 20, 4
    x = make_streams(21, S, T)
    ref, want, _ = _run(nn, hostsim_lib, x, 0, (T,), taps=True)
    bd, got, _ = _run(nn, hostsim_lib, x, 2, (T,), taps=True)
    assert np.array_equal(_bits(got), _bits(want))
    for tap in ("X", "P", "features", "xcorr1"):
        assert np.array_equal(_bits(bd.tap(tap)), _bits(ref.tap(tap))), tap
    # one frame per call: X comes from the rider blocks of k_pitch's launch (xt_rider) and is fetched by the fused kernel
    bd1, got1, _ = _run(nn, hostsim_lib, x, 1, (1,) * T, taps=True)
    assert np.array_equal(_bits(got1), _bits(want))
    for tap in ("X", "P", "features", "ex"):
        assert np.array_equal(_bits(bd1.tap(tap)), _bits(ref.tap(tap))), tap
    plain = nn.BatchDenoiser(S, lib=hostsim_lib)
    with pytest.raises(RuntimeError, match="set_taps"):
        plain.tap("X")
    # the per-frame record written by the fused kernel's synthesis
    logs = []
    for mode in (0, 2):
        b2 = nn.BatchDenoiser(S, lib=hostsim_lib)
        b2.set_back_end(mode)
        log = np.zeros((T, S, 24), np.uint32)
        b2.set_frame_log(log.ctypes.data, T)
        b2.process(x)
        logs.append(log)
    assert np.array_equal(logs[0], logs[1]) and logs[0][:, :, 0].min() >= 60
    # another model of the shape class; two models side by side (the second run of streams starts at a tile boundary)
    sh = nn.RnnModel.from_bytes(open(os.path.join(GOLDEN, "sh.rnn"), "rb").read())
    _, a, va = _run(nn, hostsim_lib, x, 0, (T,), model=sh)
    _, c, vc = _run(nn, hostsim_lib, x, 2, (T,), model=sh)
    assert np.array_equal(_bits(a), _bits(c)) and np.array_equal(_bits(va), _bits(vc)) and not np.array_equal(a, want)
    x2 = make_streams(5, 64 + 9, 3)
    _, a, va = _run(nn, hostsim_lib, x2, 0, (3,), groups=[(None, 64), (sh, 9)])
    _, c, vc = _run(nn, hostsim_lib, x2, 1, (1, 1, 1), groups=[(None, 64), (sh, 9)])
    assert np.array_equal(_bits(a), _bits(c)) and np.array_equal(_bits(va), _bits(vc))
    # packed int16, two interleaved channels, the callers' dropped first frame
    pcm = np.clip(np.rint(x[:, :3].reshape(S // 2, 2, 3 * 480).transpose(0, 2, 1)), -32768, 32767).astype(np.int16)
    outs = []
    for mode in (0, 2):
        b3 = nn.BatchDenoiser(S, lib=hostsim_lib)
        b3.set_back_end(mode)
        o, v = b3.process_pcm(pcm, _ffi.PCM_I16, channels=2, discard_first=True)
        outs.append((o, v))
    assert outs[0][0].shape[1] == 2 * 480 and np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(_bits(outs[0][1]), _bits(outs[1][1]))


def test_shape_class_plan_matches_the_packer(hostsim_lib):
    """k_back is compiled for the built-in model's layer sizes; a model of another shape must fall back to the unfused kernels (and
    still work) whatever back end is asked for."""
    import nnnoiseless_amd as nn
    from model_fixtures import make_model
    from nnnoiseless_amd.synthetic import make_streams
    blob = make_model(nd=16, nv=20, nn=40, ndn=60, seed=4)
    m = nn.RnnModel.from_bytes(blob)
    assert m is not None
    x = make_streams(2, 6, 3)
    _, a, va = _run(nn, hostsim_lib, x, 0, (3,), model=m)
    bd = nn.BatchDenoiser(6, model=m, lib=hostsim_lib)
    bd.set_back_end(2)
    bd.set_profiling(True)
    c, vc = bd.process(x)
    assert np.array_equal(_bits(a), _bits(c)) and np.array_equal(_bits(va), _bits(vc))
    assert bd.kernel_times()["k_back"][1] == 0            # not the fused kernel
    bi = nn.BatchDenoiser(6, lib=hostsim_lib)
    bi.set_back_end(2)
    bi.set_profiling(True)
    bi.process(x)
    kt = bi.kernel_times()
    assert kt["k_back"][1] == 1 and kt["k_fft_xp"][1] == 0 and kt["k_synth"][1] == 0 and kt["k_rnn"][1] == 0
