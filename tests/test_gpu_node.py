"""The node-level object (include/nnn_node.h) on a GPU box: this pool's boxes have ONE MI355X, so the shards here sit on the same device
(ordinals may repeat), each driven by its own host thread as on a node of eight -- fan-out, join, the stream split and the VAD rows
are what is tested; the N > 1 scaling run is the driver's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_node_of_two_shards_equals_one_batch():
    import torch
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 1030, 30
    x = make_streams(41, S, T)
    want, want_vad = nn.BatchDenoiser(S).process(x)
    node = nn.NodeDenoiser(S, (0, 0))
    assert node.shards() == [(0, 0, 515), (0, 515, 1030)]
    got = np.zeros_like(want)
    vad = np.zeros_like(want_vad)
    t = 0
    for n in (1, 24, 1, 4):                                # host buffers: every shard uploads, runs and downloads on its own thread
        got[:, t:t + n], vad[t:t + n] = node.process(x[:, t:t + n])
        t += n
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(vad.view(np.uint32), want_vad.view(np.uint32))
    # buffers resident on the shards' devices: per-shard pointers, asynchronous, one synchronize for the node
    node.reset()
    dev = torch.device("cuda", 0)
    parts = [(torch.from_numpy(x[lo:hi]).to(dev).contiguous(), torch.empty((hi - lo, T, 480), dtype=torch.float32, device=dev),
              torch.empty((T, hi - lo), dtype=torch.float32, device=dev)) for _, lo, hi in node.shards()]
    torch.cuda.synchronize()
    node.process_device([p[0].data_ptr() for p in parts], [p[1].data_ptr() for p in parts], [p[2].data_ptr() for p in parts], T, T * 480, 480)
    node.synchronize()
    assert not node.fault()
    y = np.concatenate([p[1].cpu().numpy() for p in parts], 0)
    v = np.concatenate([p[2].cpu().numpy() for p in parts], 1)
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32)) and np.array_equal(v.view(np.uint32), want_vad.view(np.uint32))
    node.close()
