"""The fused back end (k_back, nnn_back.hip) on the MI355X: against the oracle at bench size, driven as ticks and as groups; the
five ways of running the part of a frame behind the pitch analysis against each other, bit for bit."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, rel_rms
from test_gpu_bench_parity import NDIST, check_against_oracle, oracle_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nn():
    import nnnoiseless_amd
    return nnnoiseless_amd


def run_device(nn, torch, S, model, x, calls, mode):
    """x: device tensor [S, T, 480]; calls back to back on torch's stream, no host synchronisation in between."""
    T = x.shape[1]
    assert sum(calls) == T
    bd = nn.BatchDenoiser(S, model=model)
    bd.set_back_end(mode)
    bd.set_inputs_ready(True)
    y = torch.empty_like(x)
    vad = torch.empty((T, S), dtype=torch.float32, device=x.device)
    log = torch.zeros((T, S, 24), dtype=torch.int32, device=x.device)
    bd.set_frame_log(log.data_ptr(), T)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    pos = 0
    for n in calls:
        bd.process_device(x.data_ptr() + pos * 480 * 4, y.data_ptr() + pos * 480 * 4, vad.data_ptr() + pos * S * 4, n, T * 480, 480, stream)
        pos += n
    torch.cuda.synchronize()
    assert not bd.fault()
    bd.close()
    return y, vad, log


@pytest.mark.parametrize("S,model_name,mode,calls", [
    (4096, "builtin", 1, (1,) * 48),                 # the real-time pattern: one frame per call, every call the fused kernel
    (4096, "builtin", 2, (24, 24)),                  # groups of 24 frames inside the fused kernel
    (4096, "sh", 2, (7, 1, 24, 16)),                 # another model of the shape class (other activation kinds), ragged groups
    (65536, "builtin", 1, (1,) * 12),
    (65536, "builtin", 2, (12,)),
])
def test_fused_back_end_against_the_oracle(nn, oracle_mod, weights_bytes, S, model_name, mode, calls):
    import torch
    from nnnoiseless_amd.synthetic import make_streams
    T = sum(calls)
    blob = weights_bytes if model_name == "builtin" else open(os.path.join(GOLDEN, "sh.rnn"), "rb").read()
    model = None if model_name == "builtin" else nn.RnnModel.from_bytes(blob)
    base = make_streams(9008 if model_name == "builtin" else 9504, NDIST, T)
    ref, gtol = oracle_reference(oracle_mod, blob, base)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(S + mode)
    idx = rng.permutation(S) % NDIST
    first = np.full(NDIST, -1, np.int64)
    for pos in range(S - 1, -1, -1):
        first[idx[pos]] = pos
    idx_d, first_d = torch.from_numpy(idx).to(dev), torch.from_numpy(first).to(dev)
    x = torch.from_numpy(base).to(dev)[idx_d]
    y, vad, log = run_device(nn, torch, S, model, x, calls, mode)
    del x
    ys, vs, ls = y[first_d], vad[:, first_d], log[:, first_d]
    # every copy of a stream agrees bit for bit, wherever in the batch / tile / workgroup it sits
    assert torch.equal(y, ys[idx_d]) and torch.equal(vad, vs[:, idx_d]) and torch.equal(log, ls[:, idx_d])
    out, v, lg = ys.cpu().numpy(), vs.cpu().numpy().T, ls.cpu().numpy().view(np.uint32)
    del y, vad, log
    torch.cuda.empty_cache()
    tag = f"fused_back_end_mode{mode}_{model_name}_{S}streams_{T}frames_{'ticks' if max(calls) == 1 else 'groups'}"
    rep = check_against_oracle(out, v, lg, ref, gtol, tag)
    sil = np.array([s % 16 == 7 for s in range(NDIST)])
    assert not out[sil].any() and not v[sil].any()           # silence in, exact zeros out
    assert rep["pitch_mismatches"] == 0


def test_back_ends_against_each_other(nn):
    """1100 streams (17 full tiles and a ragged one) x 30 frames in calls of mixed length: every way of running the part of a frame
    behind the pitch analysis -- three launches with either RNN kernel family, the fused kernel's RNN stretch alone, the fused kernel
    for ticks or for whole groups -- gives the same audio, VAD and per-frame record, bit for bit."""
    import torch
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 1100, 30
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(make_streams(77, S, T)).to(dev)
    calls = {0: (30,), 3: (1, 1, 5, 1, 22), 4: (30,), 1: (1, 7, 1, 1, 20), 2: (13, 1, 16)}
    res = {m: run_device(nn, torch, S, None, x, calls[m], m) for m in calls}
    y0, v0, l0 = res[0]
    for m in (3, 4):
        y, v, lg = res[m]
        assert torch.equal(y, y0) and torch.equal(v, v0) and torch.equal(lg, l0), m
    # the fused kernel runs the transforms and the synthesis too: since multiply-adds fuse only where the source says so, the same bits
    for m in (1, 2):
        y, v, lg = res[m]
        assert torch.equal(lg, l0) and torch.equal(v, v0), m
        assert torch.equal(y, y0), (m, rel_rms(y.cpu().numpy(), y0.cpu().numpy()))
    # ticks and groups of the fused kernel itself: the same bits
    assert torch.equal(res[1][0], res[2][0]) and torch.equal(res[1][1], res[2][1]) and torch.equal(res[1][2], res[2][2])


def test_fused_back_end_boundary_formats_and_models(nn):
    """Packed int16 with two interleaved channels and the dropped first frame through the fused kernel's output conversion (within
    1 LSB of the unfused path); two models resident at once; state clone in the middle of a run of ticks."""
    from nnnoiseless_amd import _ffi
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 256, 6
    x = make_streams(31, S, T)
    pcm = np.clip(np.rint(x.reshape(S // 2, 2, T * 480).transpose(0, 2, 1)), -32768, 32767).astype(np.int16)
    outs = []
    for mode, chunks in ((0, (T,)), (1, (1,) * T)):
        bd = nn.BatchDenoiser(S)
        bd.set_back_end(mode)
        os_, t = [], 0
        for n in chunks:
            o, v = bd.process_pcm(pcm[:, t * 480:(t + n) * 480], _ffi.PCM_I16, channels=2, discard_first=True)
            os_.append(o)
            t += n
        outs.append(np.concatenate(os_, 1))
    assert outs[0].shape == outs[1].shape == (S // 2, (T - 1) * 480, 2)
    assert np.array_equal(outs[0], outs[1])
    sh = nn.RnnModel.from_bytes(open(os.path.join(GOLDEN, "sh.rnn"), "rb").read())
    a = nn.BatchDenoiser(S, groups=[(None, 128), (sh, 128)])
    a.set_back_end(0)
    want, want_vad = a.process(x)
    c = nn.BatchDenoiser(S, groups=[(None, 128), (sh, 128)])
    got = np.zeros_like(want)
    vad = np.zeros_like(want_vad)
    for t in range(T):
        if t == 3:
            c = c.clone()
        got[:, t:t + 1], vad[t:t + 1] = c.process(x[:, t:t + 1])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(vad.view(np.uint32), want_vad.view(np.uint32))


def test_high_pass_on_one_wave_or_two(nn, monkeypatch):
    """k_hp (one wave per 64 streams) and k_hp2 (its recurrence on one wave, everything else on a second: launches of up to 256
    tiles) at 4096 and 20 000 streams (the latter above the automatic switch), in groups and ticks, with and without the LPC sums'
    head start in k_hp2's launch: same audio, VAD and per-frame record, bit for bit."""
    import torch
    from nnnoiseless_amd.synthetic import make_streams
    dev = torch.device("cuda", 0)
    for S, T, calls in ((4096, 27, (1, 1, 24, 1)), (20000, 9, (8, 1))):
        x = torch.from_numpy(make_streams(5, S, T)).to(dev)
        res = []
        for split, head in (("0", "1"), ("1", "0"), ("1", "1")):   # (the LPC sums' head start rides in k_hp2's launch, one-frame calls only)
            monkeypatch.setenv("NNN_HP_SPLIT", split)
            monkeypatch.setenv("NNN_LPC_HEAD", head)
            res.append(run_device(nn, torch, S, None, x, calls, -1))
        for r in res[1:]:
            for a, b in zip(res[0], r):
                assert torch.equal(a, b), S
        assert res[0][0].abs().max() > 1.0
