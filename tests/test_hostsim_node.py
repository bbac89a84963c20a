"""The node-level object (include/nnn_node.h) under the test-only SIMT interpreter: the streams of a node cut into contiguous shards,
one batch per shard -- here several shards on the interpreter's one "device", run one after the other (NNN_NODE_THREADS=0: the
interpreter runs kernels on the calling thread and is not re-entrant) -- give, stream for stream, the bits of one unsharded batch."""
import os

import numpy as np
import pytest


@pytest.fixture()
def inline_shards(monkeypatch):
    monkeypatch.setenv("NNN_NODE_THREADS", "0")


@pytest.mark.parametrize("S,devices", [(70, (0, 0)), (7, (0, 0, 0))])
def test_sharded_node_equals_one_batch(hostsim_lib, inline_shards, S, devices):
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.shard import shard_range
    from nnnoiseless_amd.synthetic import make_streams
    T = 5
    x = make_streams(17, S, T)
    want, want_vad = nn.BatchDenoiser(S, lib=hostsim_lib).process(x)
    node = nn.NodeDenoiser(S, devices, lib=hostsim_lib)
    assert node.shards() == [(d,) + shard_range(S, i, len(devices)) for i, d in enumerate(devices)]      # shard.py's split
    got = np.zeros_like(want)
    vad = np.zeros_like(want_vad)
    got[:, :2], vad[:2] = node.process(x[:, :2])             # calls of mixed length: state carries over per shard
    got[:, 2:3], vad[2:3] = node.process(x[:, 2:3])
    got[:, 3:], vad[3:] = node.process(x[:, 3:])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(vad.view(np.uint32), want_vad.view(np.uint32))
    assert not node.fault()
    node.reset()
    again, _ = node.process(x)
    assert np.array_equal(again.view(np.uint32), want.view(np.uint32))
    node.close()


def test_node_pcm_and_errors(hostsim_lib, inline_shards):
    import nnnoiseless_amd as nn
    from nnnoiseless_amd import _ffi
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 8, 3
    x = make_streams(3, S, T)
    pcm = np.clip(np.rint(x.reshape(S // 2, 2, T * 480).transpose(0, 2, 1)), -32768, 32767).astype(np.int16)
    want, want_vad = nn.BatchDenoiser(S, lib=hostsim_lib).process_pcm(pcm, _ffi.PCM_I16, channels=2)
    node = nn.NodeDenoiser(S, (0, 0), lib=hostsim_lib)
    got, vad = node.process_pcm(pcm, _ffi.PCM_I16, channels=2)
    assert np.array_equal(got, want) and np.array_equal(vad.view(np.uint32), want_vad.view(np.uint32))
    odd = nn.NodeDenoiser(6, (0, 0), lib=hostsim_lib)            # 3 + 3 streams: a stereo pair would straddle the cut
    with pytest.raises(RuntimeError, match="channel group"):
        odd.process_pcm(pcm[:3], _ffi.PCM_I16, channels=2)
    with pytest.raises(RuntimeError, match="fewer streams"):
        nn.NodeDenoiser(1, (0, 0), lib=hostsim_lib)
    with pytest.raises(RuntimeError):
        nn.NodeDenoiser(8, (0, 5), lib=hostsim_lib)                 # no such device


def test_eight_shards_uneven_split_channel_groups_and_a_failing_shard(hostsim_lib, inline_shards):
    """The node as a host of eight GPUs would hold it, on the interpreter's one "device": 8 shards, 26 three-channel
    groups (78 streams) cut unevenly -- 10, 10, 10, 10, 10, 10, 9, 9 streams: the first six shards one more, and the cut falls
    inside channel groups -- plus an even 8 x 9 = 72-stream node whose shards are whole groups; a shard that fails mid-call: its
    text reaches the caller, the other shards are joined (they advanced), the node refuses further calls until reset."""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd import _ffi
    from nnnoiseless_amd.shard import shard_range
    from nnnoiseless_amd.synthetic import make_streams
    devices = (0,) * 8
    # uneven split, mono layout: stream for stream the bits of one batch
    S, T = 78, 2
    x = make_streams(23, S, T)
    want, want_vad = nn.BatchDenoiser(S, lib=hostsim_lib).process(x)
    node = nn.NodeDenoiser(S, devices, lib=hostsim_lib)
    sizes = [hi - lo for _, lo, hi in node.shards()]
    assert sizes == [10, 10, 10, 10, 10, 10, 9, 9] and node.shards() == [(0,) + shard_range(S, i, 8) for i in range(8)]
    got, vad = node.process(x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(vad.view(np.uint32), want_vad.view(np.uint32))
    pcm3 = np.zeros((S // 3, T * 480, 3), np.int16)
    with pytest.raises(RuntimeError, match="channel group"):       # 10 streams per shard: a three-channel group straddles the cut
        node.process_pcm(pcm3, _ffi.PCM_I16, channels=3)
    assert [node.shard_cpus(i) for i in range(8)] == [""] * 8      # (shards on the caller's thread: nothing pinned)
    node.close()
    # three-channel groups, 24 groups over 8 shards: every shard three whole groups
    S = 72
    x = make_streams(29, S, T)
    pcm = np.clip(np.rint(x.reshape(S // 3, 3, T * 480).transpose(0, 2, 1)), -32768, 32767).astype(np.int16)     # [group][sample][channel]
    want, want_vad = nn.BatchDenoiser(S, lib=hostsim_lib).process_pcm(pcm, _ffi.PCM_I16, channels=3)
    node = nn.NodeDenoiser(S, devices, lib=hostsim_lib)
    got, vad = node.process_pcm(pcm, _ffi.PCM_I16, channels=3)
    assert np.array_equal(got, want) and np.array_equal(vad.view(np.uint32), want_vad.view(np.uint32))
    # shard 5 loses a frame hand-off in the next call (test hook of its batch): the call fails with that shard's text ...
    hostsim_lib.check(hostsim_lib.L.nnn_batch_debug_withhold_flag(node.batch_handle(5), 1))
    x2 = make_streams(31, S, 3)
    with pytest.raises(RuntimeError, match="hand-off"):
        node.process(x2)
    assert node.fault()
    # ... every other shard ran to the end of the call (joined, not abandoned): its own batch continues from there
    lone = nn.BatchDenoiser(9, lib=hostsim_lib)
    lone.process_pcm(pcm[:3], _ffi.PCM_I16, channels=3)            # shard 0's history: the PCM call, then the three frames
    lone.process(x2[:9])
    tail = make_streams(37, 9, 1)
    want_tail, _ = lone.process(tail)
    out0 = np.empty_like(tail)
    L = hostsim_lib.L
    hostsim_lib.check(L.nnn_batch_process_host(node.batch_handle(0), _ffi.ptr(tail), _ffi.ptr(out0), None, 1, 480, 480))
    assert np.array_equal(out0.view(np.uint32), want_tail.view(np.uint32))
    # ... and the node refuses to go on with shards at different frame counts, until reset
    with pytest.raises(RuntimeError, match="node failed earlier.*hand-off"):
        node.process(x2)
    with pytest.raises(ValueError, match="8 shards"):
        node.process_device([0] * 7, [0] * 8, None, 1, 480, 480)   # a short pointer table is refused before it reaches the C side
    node.reset()
    hostsim_lib.check(L.nnn_batch_debug_withhold_flag(node.batch_handle(5), -1))
    assert not node.fault()
    got, vad = node.process_pcm(pcm, _ffi.PCM_I16, channels=3)
    assert np.array_equal(got, want)
    node.close()


def test_shards_on_other_devices_are_not_beside_each_other(hostsim_lib, inline_shards):
    """the "other batches are ticking beside this one" heuristic (which trades the fused tick kernel for one that shares the
    GPU better) counted batches on OTHER devices -- a node's shards all tick at once, so a node never took the fused kernel.  The mark is
    per device now: two shards on two devices each run the fused back end (no k_fft_xp launch).  (That two shards on ONE device do see
    each other needs calls within 5 ms of each other -- the GPU's pace, tests/test_gpu_node.py; the interpreter takes seconds per call.)"""
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 8, 4
    x = make_streams(3, S, T)
    L = hostsim_lib.L
    n_k = L.nnn_batch_num_kernels()
    names = [L.nnn_batch_kernel_name(k).decode() for k in range(n_k)]

    def launches(devices):
        import ctypes as C
        node = nn.NodeDenoiser(S, devices, lib=hostsim_lib, max_group_frames=1)
        for i in range(len(devices)):
            hostsim_lib.check(L.nnn_batch_set_profiling(node.batch_handle(i), 1))
        outs = [node.process(x[:, t:t + 1])[0] for t in range(T)]           # one frame per call: the real-time tick
        per = []
        for i in range(len(devices)):
            ms, cnt = (C.c_double * n_k)(), (C.c_int64 * n_k)()
            hostsim_lib.check(L.nnn_batch_read_kernel_times(node.batch_handle(i), ms, cnt, n_k))
            per.append(dict(zip(names, list(cnt))))
        node.close()
        return np.concatenate(outs, axis=1), per

    apart, per_apart = launches((0, 1))
    together, per_together = launches((0, 0))
    assert np.array_equal(apart.view(np.uint32), together.view(np.uint32))      # (either kernel choice gives the same bits)
    for p in per_apart:
        assert p["k_back"] == T and p["k_fft_xp"] == 0 and p["k_synth"] == 0, p   # the fused back end on every tick
    assert all(p["k_back"] == T for p in per_together)
