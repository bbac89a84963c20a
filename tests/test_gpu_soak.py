"""Long runs at bench size: the reference's contract is a continuous stream (src/denoise.rs:14-35), and a race in round
3 (k_lpc results overwritten at 32 768 / 65 536 streams) only showed with size x duration.

* 4096 streams x 3000 frames (30 s of audio per stream) in calls of mixed length -- 1, 7, 24, 31, 32, 48, 100 frames, one-frame ticks
  between 48-frame calls, the schedule flipped between `lanes` and `seq` on the way, so that every kernel choice (fused back end for
  the ticks, layer-pipelined RNN for the groups, chained and looped pitch frames, pipelined and in-order calls) meets every ring
  position: every copy of a stream bit-identical, 64 distinct streams against the oracle on EVERY frame (pitch index exact, VAD and
  gains to 1e-4 / spread, audio to 1e-4 relative RMS outside the listed frames whose pitch-filter branch flipped);
* 65 536 streams x 480 frames in 48-frame calls with the inputs-ready promise: every copy of a stream bit-identical, no hand-off fault.
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from test_gpu_bench_parity import check_against_oracle, oracle_reference

pytestmark = pytest.mark.gpu


def _scatter(S, ndist, seed):
    rng = np.random.default_rng(seed)
    idx = rng.permutation(S) % ndist
    first = np.full(ndist, -1, np.int64)
    for pos in range(S - 1, -1, -1):
        first[idx[pos]] = pos
    return idx, first


def test_soak_4096_streams_3000_frames_mixed_calls(oracle_mod, weights_bytes):
    import torch
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T, ND = 4096, 3000, 64
    base = make_streams(4800, ND, T)                           # [64, 3000, 480]; stream i of the mix is silent when i % 16 == 7
    ref, gtol = oracle_reference(oracle_mod, weights_bytes, base)
    dev = torch.device("cuda", 0)
    idx, first = _scatter(S, ND, 11)
    idx_d, first_d = torch.from_numpy(idx).to(dev), torch.from_numpy(first).to(dev)
    base_d = torch.from_numpy(base).to(dev)
    bd = nn.BatchDenoiser(S)
    bd.set_inputs_ready(True)
    log = torch.zeros((T, S, 24), dtype=torch.int32, device=dev)
    bd.set_frame_log(log.data_ptr(), T)
    stream = torch.cuda.current_stream().cuda_stream
    out = np.empty((ND, T, 480), np.float32)
    vad = np.empty((ND, T), np.float32)
    pattern = [1, 7, 24, 31, 32, 48, 100, 1, 1, 48, 1, 48, 1, 1, 1, 33, 64, 1, 96, 5]
    t, k, n_calls, ticks = 0, 0, 0, 0
    while t < T:
        n = min(pattern[k % len(pattern)], T - t)
        if k % 7 == 3:
            bd.set_schedule("seq")                              # every call in order on the caller's stream
        elif k % 7 == 5:
            bd.set_schedule("lanes", 1)                         # the high-pass chain on its own stream for calls of 32 frames or more
        x = base_d[:, t:t + n][idx_d].contiguous()              # [S, n, 480]
        y = torch.empty_like(x)
        v = torch.empty((n, S), dtype=torch.float32, device=dev)
        # the input is final before the call -- the promise made above, and torch's null stream (`stream` = 0 selects the batch's own
        # non-blocking stream) is not ordered with the library's otherwise
        torch.cuda.synchronize()
        bd.process_device(x.data_ptr(), y.data_ptr(), v.data_ptr(), n, n * 480, 480, stream)
        torch.cuda.synchronize()
        ys, vs = y[first_d], v[:, first_d]
        assert torch.equal(y, ys[idx_d]) and torch.equal(v, vs[:, idx_d]), f"copies of a stream differ in the call at frame {t} ({n} frames)"
        out[:, t:t + n], vad[:, t:t + n] = ys.cpu().numpy(), vs.cpu().numpy().T
        t += n
        k += 1
        n_calls += 1
        ticks += n == 1
    assert not bd.fault()
    ls = log[:, first_d]
    assert torch.equal(log, ls[:, idx_d])
    rep = check_against_oracle(out, vad, ls.cpu().numpy().view(np.uint32), ref, gtol, "soak_4096streams_3000frames_mixed_calls")
    rep.update({"streams": S, "calls": n_calls, "one_frame_calls": int(ticks), "call_lengths": sorted(set(pattern))})
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "parity_soak_4096streams_3000frames_mixed_calls.json"), "w"), indent=1)
    sil = np.array([s % 16 == 7 for s in range(ND)])
    assert not out[sil].any() and not vad[sil].any()
    bd.close()


def test_soak_65536_streams_480_frames(oracle_mod):
    import torch
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T, ND, C = 65536, 480, 256, 48
    base_d = torch.from_numpy(make_streams(6100, ND, T)).to(torch.device("cuda", 0))
    idx, first = _scatter(S, ND, 5)
    idx_d, first_d = torch.from_numpy(idx).to(base_d.device), torch.from_numpy(first).to(base_d.device)
    bd = nn.BatchDenoiser(S)
    bd.set_inputs_ready(True)
    stream = torch.cuda.current_stream().cuda_stream
    bufs = [(torch.empty((S, C, 480), dtype=torch.float32, device=base_d.device), torch.empty((S, C, 480), dtype=torch.float32, device=base_d.device),
             torch.empty((C, S), dtype=torch.float32, device=base_d.device)) for _ in range(2)]
    energy = 0.0
    for c in range(T // C):
        x, y, v = bufs[c & 1]
        x.copy_(base_d[:, c * C:(c + 1) * C][idx_d])
        torch.cuda.synchronize()                                    # (the input is final before the call: the promise made above)
        bd.process_device(x.data_ptr(), y.data_ptr(), v.data_ptr(), C, C * 480, 480, stream)
        if c:                                                      # check the previous call while this one runs
            xp, yp, vp = bufs[(c - 1) & 1]
            ok = torch.equal(yp, yp[first_d][idx_d]) and torch.equal(vp, vp[:, first_d][:, idx_d])
            assert ok, f"copies of a stream differ in call {c - 1}"
            energy += float((yp[first_d] ** 2).sum())
    torch.cuda.synchronize()
    xp, yp, vp = bufs[(T // C - 1) & 1]
    assert torch.equal(yp, yp[first_d][idx_d]) and torch.equal(vp, vp[:, first_d][:, idx_d])
    assert not bd.fault() and energy > 0 and bool(torch.isfinite(yp).all())
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"case": "soak_65536streams_480frames", "streams": S, "frames": T, "distinct_streams": ND, "copies_agree": True, "fault": False},
              open(os.path.join(ROOT, "gpurun_out", "parity_soak_65536streams_480frames.json"), "w"), indent=1)
    bd.close()
