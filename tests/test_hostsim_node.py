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
