"""Round-3 host logic under the test-only SIMT interpreter (no GPU): the per-frame parity record, the hand-off fault and its
test hook, the resampler's output-capacity precondition, argument validation of caller-supplied buffers."""
import ctypes as C

import numpy as np
import pytest

from nnnoiseless_amd import _ffi


def test_frame_log_records_every_frame(hostsim_lib):
    """nnn_batch_set_frame_log: (pitch, branch, 22 gains) of every frame of multi-frame calls equal the taps of the same frames
    processed one at a time; frames beyond the record's capacity are not written."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 9, 7
    x = make_streams(3, S, T)
    ref = nn.BatchDenoiser(S, lib=hostsim_lib)
    want = np.zeros((T, S, 24), np.uint32)
    for t in range(T):
        ref.process(x[:, t:t + 1])
        want[t, :, 0] = ref.tap("pitch")[:, 0].view(np.uint32)
        want[t, :, 1] = ref.tap("branch")[:, 0].view(np.uint32)
        want[t, :, 2:] = ref.tap("g").view(np.uint32)
    bd = nn.BatchDenoiser(S, lib=hostsim_lib)
    log = np.full((T, S, 24), 0xDEADBEEF, np.uint32)
    bd.set_frame_log(log.ctypes.data, T - 1)          # room for all frames but the last
    bd.process(x[:, :4])
    bd.process(x[:, 4:])
    assert np.array_equal(log[:T - 1], want[:T - 1])
    assert (log[T - 1] == 0xDEADBEEF).all()
    silent = want[:, :, 1] == 1 << 22
    assert silent[:, 7 - 3].all() and not want[:, :, 2:][silent].any()   # stream 7 of the synthetic mix is silence


def test_withheld_handoff_flag_raises_a_sticky_fault(hostsim_lib):
    """The frames of a group run side by side in k_pitch and hand the last pitch from workgroup to workgroup; a flag that never
    arrives must not hang and must not pass silently: the waiting workgroup times out, the batch reports the fault from
    synchronize() and from every later call, and reset() clears it."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 5, 6
    x = make_streams(0, S, T)
    bd = nn.BatchDenoiser(S, lib=hostsim_lib)
    clean, _ = bd.process(x)
    assert not bd.fault()
    bd.reset()
    hostsim_lib.check(hostsim_lib.L.nnn_batch_debug_withhold_flag(bd._h, 2))    # frame 2's flag is never published
    with pytest.raises(RuntimeError, match="hand-off"):
        bd.process(x)                                                            # (the host call ends in a synchronize)
    assert bd.fault()
    with pytest.raises(RuntimeError, match="hand-off"):
        bd.process(x[:, :1])
    with pytest.raises(RuntimeError, match="hand-off"):
        bd.synchronize()
    bd.reset()
    hostsim_lib.check(hostsim_lib.L.nnn_batch_debug_withhold_flag(bd._h, -1))
    assert not bd.fault()
    again, _ = bd.process(x)
    assert np.array_equal(again, clean)


def test_resampler_refuses_a_short_output_buffer(hostsim_lib):
    """cap_out below nnn_resampler_max_output would drop outputs while the ring moves on: refused, nothing consumed."""
    L = hostsim_lib.L
    r = L.nnn_resampler_create(2, 44100.0 / 48000.0, 0)
    assert r
    x = np.random.default_rng(0).standard_normal((2, 500)).astype(np.float32)
    cap = L.nnn_resampler_max_output(r, 500)
    out = np.zeros((2, cap), np.float32)
    n = C.c_long(-1)
    assert L.nnn_resampler_process_host(r, _ffi.ptr(x), 500, _ffi.ptr(out), cap - 200, C.byref(n)) != 0
    assert "cap_out" in hostsim_lib.error() and n.value == 0
    assert L.nnn_resampler_process_host(r, _ffi.ptr(x), 500, _ffi.ptr(out), cap, C.byref(n)) == 0 and n.value > 500
    L.nnn_resampler_destroy(r)


def test_caller_supplied_buffers_are_validated(hostsim_lib):
    """Wrong-sized `out` / `vad` / `rows` never reach the C library (explicit ValueError, also under python -O)."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.training import ROW_WIDTH, TrainingFeatures
    bd = nn.BatchDenoiser(3, lib=hostsim_lib)
    x = np.zeros((3, 2, 480), np.float32)
    with pytest.raises(ValueError):
        bd.process(x, out=np.zeros((3, 1, 480), np.float32))
    with pytest.raises(ValueError):
        bd.process(x, vad=np.zeros((3, 2), np.float32))
    with pytest.raises(ValueError):
        bd.process(x, out=np.zeros((3, 2, 480), np.float64))
    with pytest.raises(ValueError):
        bd.process(np.zeros((2, 2, 480), np.float32))
    tf = TrainingFeatures(3, lib=hostsim_lib)
    cut, vad = np.zeros((2, 3), np.int32), np.zeros((2, 3), np.float32)
    with pytest.raises(ValueError):
        tf.process(x, x, x, cut, vad, rows=np.zeros((2, 3, ROW_WIDTH - 1), np.float32))
    with pytest.raises(ValueError):
        tf.process(x, x, x, cut[:1], vad)


def test_lpc_kernel_variants_agree_bit_for_bit(hostsim_lib, oracle_mod, weights_bytes, monkeypatch):
    """k_lpc (every lane carries its stream's five autocorrelation chains, a wave walking the windows of 1, 2, 4 or 8 consecutive
    frames at once, their sums in frame pairs) and k_lpc_wide (one lag per wave, what launches too small to fill the GPU use) give the same bits, and both the
    oracle's: autocorrelation, FIR taps and everything downstream."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 7, 7
    x = make_streams(40, S, T)
    res = {}
    for wide, fc in (("0", "1"), ("0", "2"), ("0", "4"), ("0", "8"), ("1", "0")):
        monkeypatch.setenv("NNN_LPC_WIDE", wide)
        monkeypatch.setenv("NNN_LPC_FC", fc)
        bd = nn.BatchDenoiser(S, lib=hostsim_lib, taps=True)
        out, vad = bd.process(x)                     # one group of 7 frames: with four frames per wave, a wave of 4 and a wave of 3; with eight, one wave of 7
        res[wide + fc] = (out, vad, bd.tap("ac").copy(), bd.tap("lpc2").copy(), bd.tap("xlp").copy(), bd.tap("pitch").copy())
    for key in ("02", "04", "08", "10"):
        for a, b in zip(res["01"], res[key]):
            assert np.array_equal(a, b), key
    om = oracle_mod.Model(weights_bytes)
    for s in range(S):
        st = oracle_mod.State(om)
        for t in range(T):
            st.process_frame(x[s, t])
        ot = st.taps()
        for k, i in (("ac", 2), ("lpc2", 3), ("xlp", 4)):
            assert np.array_equal(res["04"][i][s].view(np.uint32), np.atleast_1d(ot[k]).astype(np.float32).view(np.uint32)), (k, s)


@pytest.mark.parametrize("gmax", [1, 2, 5])
def test_batches_sized_for_short_groups(hostsim_lib, gmax):
    """nnn_batch_create_opts(max_group_frames): scratch sets and history rings sized for groups of that many frames.  One-frame ticks
    and longer calls (cut into groups of <= gmax frames, ring wrapping many times) give the default batch's bits; the batch holds a
    fraction of the memory; clone keeps the size; a snapshot of a differently sized batch is refused."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 6, 23
    x = make_streams(11, S, T)
    ref = nn.BatchDenoiser(S, lib=hostsim_lib)
    want, want_vad = ref.process(x)
    bd = nn.BatchDenoiser(S, lib=hostsim_lib, max_group_frames=gmax)
    assert bd.max_group_frames() == gmax and ref.max_group_frames() == 24
    got = np.zeros_like(want)
    vad = np.zeros_like(want_vad)
    t = 0
    for n in (1, 1, 3, 1, 7, 1, 1, 8):            # 23 frames in calls of mixed length
        o, v = bd.process(x[:, t:t + n])
        got[:, t:t + n], vad[t:t + n] = o, v
        t += n
        if t == 5:
            bd = bd.clone()
            assert bd.max_group_frames() == gmax
    assert np.array_equal(got, want) and np.array_equal(vad, want_vad)
    assert bd.device_bytes() < ref.device_bytes() * (0.3 if gmax <= 2 else 0.5)
    with pytest.raises(RuntimeError, match="does not match"):
        bd.load_state(ref.save_state())
    with pytest.raises(ValueError):
        nn.BatchDenoiser(S, lib=hostsim_lib, max_group_frames=0)


@pytest.mark.parametrize("n_frames", [35])
def test_long_odd_calls_on_a_batch_sized_for_ticks(hostsim_lib, n_frames):
    """a pipelined call (32 frames or more) with an ODD frame count on a max_group_frames = 1 batch used to be
    cut into an even number of groups, one more than there are frames -- an empty group, launches with an empty grid and scratch
    set -1.  `Longer calls still work on such a batch` (include/nnn_batch.h): same bits as the default batch, taps readable."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S = 3
    x = make_streams(5, S, n_frames)
    want, want_vad = nn.BatchDenoiser(S, lib=hostsim_lib).process(x)
    bd = nn.BatchDenoiser(S, lib=hostsim_lib, max_group_frames=1)
    got, vad = bd.process(x)
    assert np.array_equal(got, want) and np.array_equal(vad, want_vad)
    assert bd.tap("pitch").shape == (S, 1) and not bd.fault()


def test_oversize_group_request_is_refused(hostsim_lib):
    """max_group_frames above the kernels' longest group (24) used to be clamped silently."""
    import nnnoiseless_amd as nn
    with pytest.raises(RuntimeError, match="must not exceed 24"):
        nn.BatchDenoiser(6, lib=hostsim_lib, max_group_frames=25)
    assert nn.BatchDenoiser(6, lib=hostsim_lib, max_group_frames=24).max_group_frames() == 24


def test_xcd_tile_mapping_changes_no_bits(hostsim_lib):
    """The tiles of a batch that come in eights are dealt to the XCDs -- tile t's blocks to XCD t mod 8 in k_lpc, k_pitch, k_fft_xp,
    k_rnn / k_rnn_wf and k_synth, i.e. another block -> stream mapping --, the last few keep block order.  The same streams give the
    same bits whichever way their batch is cut: 10 tiles (8 dealt + 2), 8 tiles (all dealt), 7 tiles (none)."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    T = 2
    x = make_streams(9, 600, T)
    a, va = nn.BatchDenoiser(600, lib=hostsim_lib).process(x)
    b, vb = nn.BatchDenoiser(512, lib=hostsim_lib).process(x[:512])
    c, vc = nn.BatchDenoiser(448, lib=hostsim_lib).process(x[:448])
    assert np.array_equal(a[:512], b) and np.array_equal(va[:, :512], vb)
    assert np.array_equal(a[:448], c) and np.array_equal(va[:, :448], vc)
    assert np.abs(a[512:]).max() > 0
    one = nn.BatchDenoiser(600, lib=hostsim_lib)                          # ... and frame by frame (one-frame launches)
    for t in range(T):
        o, v = one.process(x[:, t:t + 1])
        assert np.array_equal(o[:, 0], a[:, t]) and np.array_equal(v[0], va[t])
