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


def test_tick_kernel_choice_counts_batches_on_the_same_device_only():
    """One-frame ticks: a batch alone on its device takes the fused back end (k_back, no k_fft_xp / k_synth launch); shards ticking side
    by side on ONE device see each other within a few calls and take the RNN stretch alone between k_fft_xp and k_synth (it shares the
    GPU better).  ADVICE r4: the mark used to be process-wide, so shards on different GPUs counted as "beside" too and a node never took
    the tick kernels; with per-device marks a one-shard node -- which is what every device of a real node looks like to itself --
    stays on the fused kernel while another node ticks on ... the same device here (one GPU per box), so both directions are shown on
    device 0: alone = fused, together = not."""
    import ctypes as C
    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams
    lib = nn.library()
    L = lib.L
    n_k = L.nnn_batch_num_kernels()
    names = [L.nnn_batch_kernel_name(k).decode() for k in range(n_k)]
    S, T = 2048, 40
    x = make_streams(77, S, T)

    def run(devices):
        node = nn.NodeDenoiser(S, devices, max_group_frames=1)
        for i in range(len(devices)):
            lib.check(L.nnn_batch_set_profiling(node.batch_handle(i), 1))
        outs = [node.process(x[:, t:t + 1])[0] for t in range(T)]
        per = []
        for i in range(len(devices)):
            ms, cnt = (C.c_double * n_k)(), (C.c_int64 * n_k)()
            lib.check(L.nnn_batch_read_kernel_times(node.batch_handle(i), ms, cnt, n_k))
            per.append(dict(zip(names, list(cnt))))
        cpus = [node.shard_cpus(i) for i in range(len(devices))]
        node.close()
        return np.concatenate(outs, axis=1), per, cpus

    alone, per_alone, _ = run((0,))
    assert per_alone[0]["k_back"] == T and per_alone[0]["k_fft_xp"] == 0 and per_alone[0]["k_synth"] == 0, per_alone
    both, per_both, cpus = run((0, 0))
    assert np.array_equal(alone.view(np.uint32), both.view(np.uint32))                       # same bits whichever kernels ran
    assert sum(p["k_fft_xp"] for p in per_both) >= T, per_both                              # the shards saw each other
    print("worker CPUs:", cpus)
    assert cpus[0] == cpus[1]                                                              # both pinned (or both not) to device 0's CPUs


def test_rnnoise_surface_honours_nnn_device(monkeypatch):
    """NNN_DEVICE picks the HIP device of the rnnoise_* single-stream surface: 0 works, an ordinal the box has not got fails loudly
    (rnnoise_create returns NULL with the reason on stderr), nothing falls back."""
    import ctypes as C
    import nnnoiseless_amd as nn
    L = nn.library().L
    L.rnnoise_create.restype = C.c_void_p
    L.rnnoise_create.argtypes = [C.c_void_p]
    L.rnnoise_destroy.argtypes = [C.c_void_p]
    monkeypatch.setenv("NNN_DEVICE", "0")
    st = L.rnnoise_create(None)
    assert st
    L.rnnoise_destroy(st)
    monkeypatch.setenv("NNN_DEVICE", "63")
    assert not L.rnnoise_create(None)
    assert b"no HIP device 63" in L.nnn_last_error()
