"""The certified coarse pitch search of k_pitch (round 6) against the oracle, frame by frame, through the taps of mode 2.

find_best_pitch (ref: src/pitch.rs:372-405, call site :83-84) only returns the two best coarse lags; the kernel computes all 147
cross-correlations approximately on the matrix cores, rules out every lag that provably cannot be in the pair, and sums the survivors
exactly in the reference's order.  Checked here: the pair is the oracle's on every frame, every survivor's sum is the oracle's bit for
bit, the pitch index follows, the search really rules lags out (few survivors), and streams it must not certify take the full search.
Under the SIMT interpreter where there is no GPU, on the GPU at size."""
import numpy as np
import pytest

from edge_streams import make_edge_streams


def _run(nn, oracle_mod, weights_bytes, x, lib=None, full_ok=False):
    S, T = x.shape[:2]
    om = oracle_mod.Model(weights_bytes)
    states = [oracle_mod.State(om) for _ in range(S)]
    bd = nn.BatchDenoiser(S, lib=lib)
    bd.set_taps(2)
    counts = np.zeros((S, T), np.int32)
    for t in range(T):
        bd.process(x[:, t:t + 1])
        xc, b1, ps, pi = bd.tap("xcorr1"), bd.tap("best1"), bd.tap("pitch_search"), bd.tap("pitch")
        for s in range(S):
            states[s].process_frame(x[s, t])
            ot = states[s].taps()
            keep = ~np.isnan(xc[s])
            counts[s, t] = keep.sum()
            assert np.array_equal(b1[s], ot["best1"]), ("best1", s, t, b1[s], ot["best1"])
            assert np.array_equal(xc[s][keep].view(np.uint32), ot["xcorr1"][keep].view(np.uint32)), ("xcorr1", s, t)
            assert ps[s, 0] == ot["pitch_search"] and pi[s, 0] == ot["pitch_idx"], ("pitch", s, t)
            # the pair itself always survives (where it is a real lag: find_best_pitch's initial 0 / 1 are not)
            for L in ot["best1"]:
                assert keep[L] or not ot["xcorr1"][L] > 0, ("the returned lag was ruled out", s, t, L)
    return counts


def test_certified_search_matches_the_oracle_hostsim(hostsim_lib, oracle_mod, weights_bytes, golden_io):
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 48, 5
    x = make_streams(0, S, T)
    x[0] = golden_io[0][:T]
    counts = _run(nn, oracle_mod, weights_bytes, x, lib=hostsim_lib)
    assert counts.max() < 147, "no block should need the full search on the synthetic streams"
    assert counts[:, 1:].mean() < 10, counts.mean()    # typically three to five lags per stream-frame


def test_certified_search_edge_streams_hostsim(hostsim_lib, oracle_mod, weights_bytes):
    """Full scale, DC, impulses, +-1 LSB noise, onsets, pitch-range ends, chirp, clipped noise, silence: some of these the search must
    refuse to certify (fewer than two certain lags and many candidates) -- the block then takes the full search, all 147 sums exact."""
    import nnnoiseless_amd as nn
    x = make_edge_streams(6)
    counts = _run(nn, oracle_mod, weights_bytes, x, lib=hostsim_lib)
    assert (counts == 147).any(), "the edge streams are expected to force the full search at least once"


def test_full_and_certified_search_give_the_same_frames_hostsim(hostsim_lib):
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    x = make_streams(100, 32, 4)
    a, b = nn.BatchDenoiser(32, lib=hostsim_lib), nn.BatchDenoiser(32, lib=hostsim_lib)
    a.set_taps(1)   # full search
    b.set_taps(2)   # certified search
    for t in range(4):
        oa, va = a.process(x[:, t:t + 1])
        ob, vb = b.process(x[:, t:t + 1])
        assert np.array_equal(oa.view(np.uint32), ob.view(np.uint32)) and np.array_equal(va, vb)
        for k in ("best1", "pitch_search", "pitch", "pitch_gain"):
            assert np.array_equal(a.tap(k), b.tap(k)), (k, t)


@pytest.mark.gpu
def test_certified_search_matches_the_oracle_gpu(oracle_mod, weights_bytes, golden_io):
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 1024, 12
    x = make_streams(0, S, T)
    x[0] = golden_io[0][:T]
    counts = _run(nn, oracle_mod, weights_bytes, x)
    assert counts.max() < 147
    assert counts[:, 2:].mean() < 8, counts.mean()


@pytest.mark.gpu
def test_certified_search_edge_streams_gpu(oracle_mod, weights_bytes):
    import nnnoiseless_amd as nn
    x = make_edge_streams(40)
    counts = _run(nn, oracle_mod, weights_bytes, x)
    assert (counts == 147).any()


@pytest.mark.gpu
def test_full_and_certified_search_give_the_same_frames_gpu():
    """65 536 streams x 24 frames through both searches, one frame per call and 24 frames per call: outputs, pitch and the coarse pair
    bit for bit (a race inside the certified search -- the first waves of a block listing survivors while others still read the
    planes -- would show here; the interpreter runs a block's waves one after the other and cannot see it)."""
    import nnnoiseless_amd as nn
    import torch
    from nnnoiseless_amd.synthetic import make_streams_device
    S, T = 65536, 24
    dev = torch.device("cuda", 0)
    x = make_streams_device(torch, dev, S, T, seed=5)
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for mode, per_call in ((1, 1), (2, 1), (0, T)):   # full search (taps 1), certified with taps (2), certified as production runs it
        bd = nn.BatchDenoiser(S)
        if mode:
            bd.set_taps(mode)
        y = torch.empty_like(x)
        pitch = []
        for t in range(0, T, per_call):
            bd.process_device(x.data_ptr() + t * 480 * 4, y.data_ptr() + t * 480 * 4, 0, per_call, T * 480, 480, stream)
            torch.cuda.synchronize()
            if mode:
                pitch.append((bd.tap("best1").copy(), bd.tap("pitch_search").copy(), bd.tap("pitch").copy()))
        assert not bd.fault()
        bd.close()
        outs.append((y, pitch))
    assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32))
    assert torch.equal(outs[0][0].view(torch.int32), outs[2][0].view(torch.int32))
    for t in range(T):
        for a, b in zip(outs[0][1][t], outs[1][1][t]):
            assert np.array_equal(a, b), t
