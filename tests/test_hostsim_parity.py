"""Kernel LOGIC against the oracle where there is no GPU: the unmodified product sources run under the
test-only SIMT interpreter (tests/hostsim).  Same assertions as the GPU parity tests, small sizes."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_metric, rel_rms

INT_TAPS = {"best1": "best1", "pitch_search": "pitch_search", "pitch": "pitch_idx", "silence": "silence"}
EXACT_TAPS = {"filtered": "filtered", "xlp": "xlp", "ac": "ac", "lpc2": "lpc2", "xcorr1": "xcorr1", "pitch_gain": "pitch_gain"}
TOL_TAPS = {"X": "X", "P": "P", "ex": "ex", "ep": "ep", "exp": "exp_", "features": "features", "g_raw": "g_raw", "g": "g", "vad": "vad"}


def test_every_stage_against_oracle_taps(hostsim_lib, oracle_mod, weights_bytes, golden_io):
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 66, 5  # two tiles, the second one mostly padding
    x = make_streams(0, S, T)
    x[0] = golden_io[0][:T]
    om = oracle_mod.Model(weights_bytes)
    states = [oracle_mod.State(om) for _ in range(S)]
    bd = nn.BatchDenoiser(S, lib=hostsim_lib, taps=True)
    for t in range(T):
        out, vad = bd.process(x[:, t:t + 1])
        taps = {k: bd.tap(k) for k in list(INT_TAPS) + list(EXACT_TAPS) + list(TOL_TAPS)}
        for s in range(S):
            o, v = states[s].process_frame(x[s, t])
            ot = states[s].taps()
            for k, ok in INT_TAPS.items():
                assert np.array_equal(taps[k][s], np.atleast_1d(ot[ok])), (k, s, t)
            for k, ok in EXACT_TAPS.items():   # everything upstream of the pitch index: bit-identical
                assert np.array_equal(taps[k][s].view(np.uint32), np.atleast_1d(ot[ok]).astype(np.float32).view(np.uint32)), (k, s, t)
            for k, ok in TOL_TAPS.items():
                ref = np.atleast_1d(ot[ok]).astype(np.float64)
                err = np.abs(taps[k][s] - ref).max()
                assert err <= 2e-5 * max(np.abs(ref).max(), 1.0), (k, s, t, err)
            assert np.abs(out[s, 0] - o).max() <= 2e-5 * max(np.abs(o).max(), 1.0)
            assert abs(vad[0, s] - v) < 1e-5


def test_golden_vectors_through_the_kernels(hostsim_lib, golden_io):
    import nnnoiseless_amd as nn
    frames, ref = golden_io
    bd = nn.BatchDenoiser(1, lib=hostsim_lib)
    out, vad = bd.process(frames[None])                      # 100 frames in one call
    assert golden_metric(out[0, 1:].reshape(-1), ref) < 1e-5
    kat = json.load(open(os.path.join(GOLDEN, "pitch_kat.json")))
    assert bd.tap("pitch")[0, 0] == kat["testing_raw"][-1]


def test_chunking_and_reset_are_bit_identical(hostsim_lib):
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    x = make_streams(20, 5, 6)
    bd = nn.BatchDenoiser(5, lib=hostsim_lib)
    a, va = bd.process(x)
    bd.reset()
    parts = [bd.process(x[:, t:t + 2]) for t in range(0, 6, 2)]
    b = np.concatenate([p[0] for p in parts], axis=1)
    vb = np.concatenate([p[1] for p in parts], axis=0)
    assert np.array_equal(a, b) and np.array_equal(va, vb)


def test_custom_model(hostsim_lib, oracle_mod):
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    sh = open(os.path.join(GOLDEN, "sh.rnn"), "rb").read()
    x = make_streams(0, 4, 6)
    ref = oracle_mod.run_streams(oracle_mod.Model(sh), x)
    bd = nn.BatchDenoiser(4, model=nn.RnnModel.from_bytes(sh, lib=hostsim_lib), lib=hostsim_lib)
    out, vad = bd.process(x)
    assert np.array_equal(bd.tap("pitch")[:, 0], ref["pitch"][:, -1])
    assert rel_rms(out, ref["out"]) < 1e-5
    assert np.abs(vad.T - ref["vad"]).max() < 1e-4


def test_rnnoise_c_abi_flow(hostsim_lib, golden_io):
    """The call sequence of the reference's rnnoise_demo.c (create, in-place process_frame, destroy)."""
    frames, ref = golden_io
    L = hostsim_lib.L
    st = L.rnnoise_create(None)
    assert st
    outs = []
    for f in frames[:12]:
        buf = np.array(f, np.float32)
        vad = L.rnnoise_process_frame(st, buf.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p))  # out aliases in
        assert 0.0 <= vad <= 1.0
        outs.append(buf)
    L.rnnoise_destroy(st)
    got = np.concatenate(outs[1:])
    assert golden_metric(got, ref[: got.size]) < 1e-4


def test_edge_case_inputs(hostsim_lib, oracle_mod, weights_bytes):
    """Full scale, DC, impulses, +-1 LSB noise, onsets, pitch-range ends, chirp, clipped noise, silence."""
    import nnnoiseless_amd as nn
    from edge_streams import make_edge_streams, oracle_reference
    x = make_edge_streams(12)
    ref = oracle_reference(oracle_mod, weights_bytes, x)
    bd = nn.BatchDenoiser(x.shape[0], lib=hostsim_lib)
    out = np.empty_like(x)
    for t in range(x.shape[1]):
        o, v = bd.process(x[:, t:t + 1])
        out[:, t] = o[:, 0]
        assert np.array_equal(bd.tap("pitch")[:, 0], ref["pitch"][:, t]), t
        assert np.abs(v[0] - ref["vad"][:, t]).max() < 1e-4
        assert np.abs(bd.tap("g") - ref["g"][:, t]).max() < 1e-3   # oracle f32-vs-f64 FFT spread here is ~2e-4
    scale = np.maximum(np.abs(ref["out"]).max(axis=(1, 2)), 1.0)[:, None]
    err = np.abs(out - ref["out"]).max(axis=2) / scale
    # 1e-4 of the stream's peak on every frame the reference itself is well conditioned on (see oracle_reference);
    # a sanity bound on the few frames where its pitch-filter branch is decided by rounding noise
    assert err[~ref["ill"]].max() <= 1e-4, np.argwhere((err > 1e-4) & ~ref["ill"])
    assert err.max() <= 5e-2
    assert ref["ill"].mean() < 0.2


def test_long_run_ring_wrap(hostsim_lib, oracle_mod, weights_bytes):
    """120 frames (the 52-slot history rings wrap twice, the scratch sets five times) in uneven multi-frame calls."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    x = make_streams(200, 3, 120)
    ref = oracle_mod.run_streams(oracle_mod.Model(weights_bytes), x)
    bd = nn.BatchDenoiser(3, lib=hostsim_lib)
    outs, pos = [], 0
    for n in (1, 7, 13, 2, 40, 5, 52):
        outs.append(bd.process(x[:, pos:pos + n])[0])
        pos += n
    out = np.concatenate(outs, axis=1)
    assert np.array_equal(bd.tap("pitch")[:, 0], ref["pitch"][:, -1])
    assert rel_rms(out[:, 1:], ref["out"][:, 1:]) < 1e-5


def test_nonfinite_inputs_stay_contained(hostsim_lib, oracle_mod, weights_bytes):
    """NaN / Inf / 1e30 samples in some streams: no crash, neighbours bit-identical to a clean run, pitch indices and
    the extent of NaN propagation as in the oracle (the reference has no input validation either, src/denoise.rs:95)."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    x = make_streams(3, 8, 6)
    bad = x.copy()
    bad[1, 2, 100] = np.nan
    bad[3, 1, :] = np.inf
    bad[5, :, :] = 1e30
    bad[6, 3, 7] = -np.inf
    clean, _ = nn.BatchDenoiser(8, lib=hostsim_lib).process(x)
    bd = nn.BatchDenoiser(8, lib=hostsim_lib)
    out, _ = bd.process(bad)
    for s in (0, 2, 4, 7):
        assert np.array_equal(out[s], clean[s]), s
    ref = oracle_mod.run_streams(oracle_mod.Model(weights_bytes), bad)
    assert np.array_equal(bd.tap("pitch")[:, 0], ref["pitch"][:, -1])
    assert np.array_equal(np.isnan(out), np.isnan(ref["out"]))
    ok = np.isfinite(out) & np.isfinite(ref["out"])
    # f32 rounding of the transforms is ~1e-6 of a stream's peak, so the absolute slack follows the peak (1.2e4 here) instead of the
    # fixed 0.01 this test had until round 4.  Measured at round 5's HEAD: worst |error| 0.0186 on a stream peaking at 10 188 (1.8e-6 of
    # its peak, 0.016 beyond rtol) -- the old fixed bar fails by a factor 1.6, this one (3e-6 of the peak = 0.035) holds with 2x to spare.
    assert np.allclose(out[ok], ref["out"][ok], rtol=1e-4, atol=3e-6 * float(np.abs(clean).max()))


def test_call_length_patterns(hostsim_lib):
    """Group rotation and ramped group sizes: any way of cutting 37 frames into calls gives the same bits."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    x = make_streams(60, 3, 37)
    bd = nn.BatchDenoiser(3, lib=hostsim_lib)
    want, want_vad = bd.process(x)
    for cuts in ((1, 2, 5, 7, 11, 4, 7), (2, 2, 13, 1, 1, 18)):
        bd.reset()
        outs, vads, pos = [], [], 0
        for n in cuts:
            o, v = bd.process(x[:, pos:pos + n])
            outs.append(o)
            vads.append(v)
            pos += n
        assert np.array_equal(np.concatenate(outs, axis=1), want), cuts
        assert np.array_equal(np.concatenate(vads, axis=0), want_vad), cuts


def test_rnnoise_c_abi_through_the_kernels(hostsim_lib, oracle_mod, weights_bytes):
    """The RNNoise-compatible C ABI (include/rnnoise.h, ref: src/capi.rs:16-113) end to end on the interpreter build:
    create with the built-in model, in-place process_frame (out aliases in, as test_data/rnnoise_demo.c:52 does), destroy."""
    import ctypes as C
    L = hostsim_lib.L
    assert L.rnnoise_get_frame_size() == 480
    st = L.rnnoise_create(None)
    assert st
    pcm = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "testing.raw"), dtype="<i2").astype(np.float32)
    frames = pcm[:12 * 480].reshape(12, 480)
    ref = oracle_mod.run_streams(oracle_mod.Model(weights_bytes), frames[None])
    buf = np.empty(480, np.float32)
    for t in range(12):
        buf[:] = frames[t]
        vad = L.rnnoise_process_frame(st, buf.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p))
        assert abs(vad - ref["vad"][0, t]) < 1e-4
        if t:
            assert rel_rms(buf, ref["out"][0, t]) < 1e-5
    L.rnnoise_destroy(st)


def test_clone_and_snapshot(hostsim_lib):
    """DenoiseState: Clone (src/denoise.rs:36) on the interpreter build: clone / saved snapshot continue bit-identically."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    x = make_streams(90, 5, 11)
    a = nn.BatchDenoiser(5, lib=hostsim_lib)
    a.process(x[:, :5])
    snap = a.save_state()
    b = a.clone()
    oa, va = a.process(x[:, 5:])
    ob, vb = b.process(x[:, 5:])
    assert np.array_equal(oa, ob) and np.array_equal(va, vb)
    c = nn.BatchDenoiser(5, lib=hostsim_lib)
    c.load_state(snap)
    oc, vc = c.process(x[:, 5:])
    assert np.array_equal(oa, oc) and np.array_equal(va, vc)
    with pytest.raises(RuntimeError):
        nn.BatchDenoiser(6, lib=hostsim_lib).load_state(snap)          # another shape: refused


def test_activation_functions_known_answers(hostsim_lib, oracle_mod):
    """tansig_approx / sigmoid_approx / relu of the kernels (src/util.rs:29-53) bit for bit against the oracle."""
    knots = (np.arange(201, dtype=np.float64) * 0.04).astype(np.float32)
    base = np.concatenate([knots, np.float32([8.0, 7.9999995, 8.000001, 16.0, 100.0, 1e30, 1e-30, 0.0])])
    xs = np.concatenate([base, np.nextafter(base, np.float32(np.inf)), np.nextafter(base, np.float32(-np.inf))])
    xs = np.concatenate([xs, -xs, np.float32([np.inf, -np.inf, np.nan]), np.random.default_rng(3).uniform(-20, 20, 500).astype(np.float32)])
    for act in (0, 1):
        y = np.empty_like(xs)
        hostsim_lib.check(hostsim_lib.L.nnn_debug_activations(0, act, xs.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(xs)))
        ref = oracle_mod.activation(xs, act)
        assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), act


@pytest.mark.parametrize("rows", ["32", "16"])
def test_wide_dense_layers(hostsim_lib, oracle_mod, rows, monkeypatch):
    """An input dense layer wider than one round of the RNN block's waves: nd = 80, both block shapes."""
    import nnnoiseless_amd as nn
    from model_fixtures import make_model
    from nnnoiseless_amd.synthetic import make_streams
    blob = make_model(80, 4, 32, 64, seed=80)       # 80 + 4 + 42 inputs to the noise GRU: the format's limit is 127
    x = make_streams(17, 3, 4)
    ref = oracle_mod.run_streams(oracle_mod.Model(blob), x)
    monkeypatch.setenv("NNN_RNN_ROWS", rows)
    bd = nn.BatchDenoiser(3, model=nn.RnnModel.from_bytes(blob, lib=hostsim_lib), lib=hostsim_lib)
    out, vad = bd.process(x)
    assert rel_rms(out[:, 1:], ref["out"][:, 1:]) <= 1e-5
    assert np.abs(vad.T - ref["vad"]).max() <= 1e-4


def test_quarter_tile_block_remap(hostsim_lib, oracle_mod, weights_bytes):
    """512 streams = 32 quarter-tile blocks of k_pitch: the XCD-aware block remap is active (a permutation of the blocks);
    every stream still gets its own result."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    base = make_streams(0, 16, 2)
    x = np.tile(base, (32, 1, 1))                       # 512 streams, 16 distinct ones
    bd = nn.BatchDenoiser(512, lib=hostsim_lib)
    out, vad = bd.process(x)
    ref = oracle_mod.run_streams(oracle_mod.Model(weights_bytes), base)
    pitch = bd.tap("pitch")[:, 0].reshape(32, 16)
    assert all(np.array_equal(pitch[i], ref["pitch"][:, -1]) for i in range(32))
    o = out.reshape(32, 16, 2, 480)
    assert all(np.array_equal(o[0], o[i]) for i in range(1, 32))
    assert rel_rms(o[0][:, 1:], ref["out"][:, 1:]) < 1e-5


def test_pitch_frames_chained_and_looped_agree(hostsim_lib, oracle_mod, weights_bytes, monkeypatch):
    """k_pitch runs the frames of a group either side by side (one workgroup per frame and quarter tile, the previous frame's
    pitch handed over through a flag) or in a loop inside one workgroup: the same bits either way, pitch equal to the oracle's."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 20, 11
    x = make_streams(123, S, T)
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("NNN_PITCH_CHAIN", mode)
        bd = nn.BatchDenoiser(S, lib=hostsim_lib)
        out, vad = bd.process(x)
        res[mode] = (out, vad, bd.tap("pitch").copy(), bd.tap("pitch_gain").copy())
        bd.close()
    for a, b in zip(res["0"], res["2"]):
        assert np.array_equal(a, b)
    ref = oracle_mod.run_streams(oracle_mod.Model(weights_bytes), x)
    assert np.array_equal(res["2"][2][:, 0], ref["pitch"][:, -1])


def test_pitch_taps_exist_only_after_set_taps(hostsim_lib):
    """Nothing inside the pitch analysis leaves LDS in production: its taps are an error until the taps are switched on (which
    also allocates their arrays), and switching them on afterwards works mid-stream."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    x = make_streams(5, 3, 3)
    bd = nn.BatchDenoiser(3, lib=hostsim_lib)
    bd.process(x[:, :1])
    assert bd.tap("pitch").shape == (3, 1)            # the pitch index itself is always there (the transforms need it)
    assert bd.tap("ac").shape == (3, 5) and bd.tap("lpc2").shape == (3, 5)   # k_lpc's results travel through memory: always there
    for name in ("xlp", "xcorr1", "best1", "xcorr2c", "pitch_search", "features"):
        with pytest.raises(RuntimeError):
            bd.tap(name)
    bd.set_taps(True)
    bd.process(x[:, 1:2])
    assert bd.tap("xlp").shape == (3, 864) and np.isfinite(bd.tap("xlp")).all()
    ref = nn.BatchDenoiser(3, lib=hostsim_lib, taps=True)
    ref.process(x[:, :2])
    for name in ("xlp", "ac", "xcorr1", "best1", "pitch_search", "pitch"):
        assert np.array_equal(bd.tap(name), ref.tap(name)), name


def test_rnn_kernels_agree_bit_for_bit(hostsim_lib, monkeypatch):
    """One-frame groups on large batches run k_rnn, longer groups the layer-pipelined k_rnn_wf: a stream that changes kernel from
    call to call (forced here through NNN_RNN_WF_MIN_G) must not notice."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    x = make_streams(31, 5, 7)
    monkeypatch.setenv("NNN_RNN_WF_MIN_G", "1")
    ref, vref = nn.BatchDenoiser(5, lib=hostsim_lib).process(x)
    monkeypatch.setenv("NNN_RNN_WF_MIN_G", "3")
    bd = nn.BatchDenoiser(5, lib=hostsim_lib)
    parts = [bd.process(x[:, a:b]) for a, b in ((0, 1), (1, 4), (4, 6), (6, 7))]   # k_rnn, k_rnn_wf, k_rnn, k_rnn
    assert np.array_equal(np.concatenate([p[0] for p in parts], axis=1), ref)
    assert np.array_equal(np.concatenate([p[1] for p in parts], axis=0), vref)


def test_high_pass_on_one_wave_or_two(hostsim_lib, monkeypatch):
    """k_hp (one wave per tile) and k_hp2 (recurrence on one wave; stores, decimation and the trip through LDS on a second: what
    launches of up to 256 tiles take) run the same arithmetic: same bits for every boundary format, in groups and one frame per
    call.  (Every other test of this file runs k_hp2 against the oracle: the batches here are small.)"""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd import _ffi
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 70, 5                                  # two tiles, the second one ragged
    x = make_streams(77, S, T)
    xi = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    inter = np.ascontiguousarray(xi.reshape(S // 2, 2, T * 480).transpose(0, 2, 1))   # [group][sample][channel]
    res = {}
    for split, tpb in (("0", "0"), ("1", "1"), ("1", "2")):   # (two tiles per block: what groups of larger batches run; here its last block is ragged)
        monkeypatch.setenv("NNN_HP_SPLIT", split)
        monkeypatch.setenv("NNN_HP_TPB", tpb)
        bd = nn.BatchDenoiser(S, lib=hostsim_lib, taps=True)
        o1, v1 = bd.process(x[:, :3])
        o2, v2 = bd.process(x[:, 3:4])
        filt = bd.tap("filtered").copy()
        o3, v3 = bd.process(x[:, 4:])
        pcm = nn.BatchDenoiser(S, lib=hostsim_lib)
        p1, w1 = pcm.process_pcm(inter[:, :960], _ffi.PCM_I16, channels=2, discard_first=True)
        p2, w2 = pcm.process_pcm(inter[:, 960:] / np.float32(32768.0), _ffi.PCM_F32_UNIT, channels=2)
        res[split + tpb] = (o1, o2, o3, v1, v2, v3, filt, p1, p2, w1, w2)
        bd.close(); pcm.close()
    for k in ("11", "12"):
        for a, b in zip(res["00"], res[k]):
            assert np.array_equal(a, b), k
    assert np.abs(res["11"][8]).max() > 1e-3


def test_lpc_sums_started_beside_the_high_pass(hostsim_lib, oracle_mod, weights_bytes, monkeypatch):
    """One-frame calls on small batches: the first 608 steps of the five autocorrelation sums -- rows older than the frame -- are taken
    by extra blocks of k_hp2's launch and k_pitch carries on from there.  Same order, same bits as without the head start and as the
    grouped path (k_lpc); the autocorrelation and the FIR taps equal the oracle's bit for bit."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 70, 7
    x = make_streams(404, S, T)
    res = {}
    for head in ("0", "1"):
        monkeypatch.setenv("NNN_LPC_HEAD", head)
        bd = nn.BatchDenoiser(S, lib=hostsim_lib, taps=True)
        outs = [bd.process(x[:, t:t + 1]) for t in range(T)]
        res[head] = (np.concatenate([o for o, _ in outs], axis=1), np.concatenate([v for _, v in outs], axis=0),
                     bd.tap("ac").copy(), bd.tap("lpc2").copy(), bd.tap("pitch").copy())
        bd.close()
    for a, b in zip(res["0"], res["1"]):
        assert np.array_equal(a, b)
    monkeypatch.delenv("NNN_LPC_HEAD")
    grouped = nn.BatchDenoiser(S, lib=hostsim_lib)
    og, vg = grouped.process(x)
    assert np.array_equal(og, res["1"][0]) and np.array_equal(vg, res["1"][1])
    ref = oracle_mod.run_streams(oracle_mod.Model(weights_bytes), x)
    assert np.array_equal(res["1"][4][:, 0], ref["pitch"][:, -1])


def test_high_pass_two_tiles_per_block_with_an_odd_tile_count(hostsim_lib, monkeypatch):
    """k_hp2<2> on three tiles: the second block's spare pair of waves only keeps the barrier count."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 130, 2
    x = make_streams(78, S, T)
    res = []
    for tpb in ("1", "2"):
        monkeypatch.setenv("NNN_HP_TPB", tpb)
        bd = nn.BatchDenoiser(S, lib=hostsim_lib, taps=True)
        o, v = bd.process(x)
        res.append((o, v, bd.tap("filtered").copy()))
        bd.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
