"""Parity on the configurations bench.py times, driven the way bench.py drives them:

* BASELINE configs[1] / [3] (per-GPU share) / [2] / [4] at their own stream counts, built-in model and the custom `sh.rnn`:
  device-resident input, 48-frame `nnn_batch_process_device` calls with `nnn_batch_set_inputs_ready(1)` and no host
  synchronisation between them, 96 frames (frame groups in flight across the call boundary), 256 distinct streams
  against the oracle on EVERY frame through the per-frame record (`nnn_batch_set_frame_log`): pitch index bit for bit,
  VAD <= 1e-4, band gains <= 1e-4 (or three times the oracle's own f32-vs-f64 FFT spread for the stream), audio <= 1e-4
  relative RMS outside the frames whose pitch-filter branch mask differs from the oracle's (listed, < 0.1 %);
* a real-audio batch: 512 streams derived from the reference's own recording `testing.raw` (gain, sample offset,
  polarity, added noise at several SNRs) x 99 frames, same bars;
* the reference's three 44.1 kHz `.wav` files through resampler -> denoiser -> int16, against the oracle chain.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_flips_in_line, flip_stats, rel_rms

pytestmark = pytest.mark.gpu

NDIST = 256      # distinct streams compared with the oracle at every size


@pytest.fixture(scope="module")
def nn():
    import nnnoiseless_amd
    return nnnoiseless_amd


def oracle_reference(oracle_mod, blob, x):
    """The oracle on x [n, T, 480] and the per-stream gain tolerance max(1e-4, 3 x its own f32-FFT vs f64-FFT spread)."""
    nt = os.cpu_count() or 1
    ref = oracle_mod.run_streams(oracle_mod.Model(blob), x, n_threads=nt, want=("out", "pitch", "branch", "vad", "gains"))
    ref32 = oracle_mod.run_streams(oracle_mod.Model(blob, f32_fft=True), x, n_threads=nt, want=("gains", "out", "branch"))
    gtol = np.maximum(1e-4, 3.0 * np.abs(ref["gains"] - ref32["gains"]).max(axis=(1, 2)))
    ref["f32"] = ref32        # the oracle with the reference's own f32 FFT arithmetic: what flips and differs without any GPU (flip_stats)
    return ref, gtol


def check_against_oracle(out, vad, log, ref, gtol, tag):
    """out [n, T, 480], vad [n, T], log [T, n, 24] uint32 (nnn_batch_set_frame_log).  Returns the report that is also
    written to gpurun_out/ (the flipped-frame list)."""
    n, T = vad.shape
    pitch = np.ascontiguousarray(log[:, :, 0]).view(np.int32).T
    branch = np.ascontiguousarray(log[:, :, 1]).view(np.int32).T
    gains = np.ascontiguousarray(log[:, :, 2:]).view(np.float32).transpose(1, 0, 2)
    mism = int((pitch != ref["pitch"]).sum())
    assert mism == 0, f"{tag}: {mism} of {pitch.size} pitch indices differ: {np.argwhere(pitch != ref['pitch'])[:8]}"
    verr = np.abs(vad - ref["vad"]).max()
    assert verr <= 1e-4, (tag, verr)
    gerr = np.abs(gains - ref["gains"]).max(axis=(1, 2))
    assert (gerr <= gtol).all(), (tag, float(gerr.max()), np.argwhere(gerr > gtol)[:8])
    flip = branch != ref["branch"]
    excused = flip.copy()
    excused[:, 1:] |= flip[:, :-1]            # the frame after a flipped one carries its overlap-add memory
    lst = [(int(s), int(t), int(branch[s, t] ^ ref["branch"][s, t])) for s, t in np.argwhere(flip)]
    d = (out[:, 1:] - ref["out"][:, 1:]).astype(np.float64)
    rr = ref["out"][:, 1:].astype(np.float64)
    ok = ~excused[:, 1:]
    den = max((rr[ok] ** 2).sum(), 1e-30)
    r = float(np.sqrt((d[ok] ** 2).sum() / den))
    r_all = float(np.sqrt((d ** 2).sum() / max((rr ** 2).sum(), 1e-30)))
    per_stream = np.sqrt(((d * ok[..., None]) ** 2).sum(axis=(1, 2)) / np.maximum((rr ** 2).sum(axis=(1, 2)), 1e-9))
    report = {"case": tag, "streams_compared": n, "frames": T, "pitch_mismatches": mism, "vad_max_err": float(verr),
              "gain_max_err": float(gerr.max()), "gain_tolerance_max": float(gtol.max()), "flipped": lst,
              "flipped_fraction": len(lst) / (n * T), "excused_fraction": float(excused.mean()), "rel_rms": r,
              "rel_rms_unmasked": r_all, "per_stream_rel_rms_max": float(per_stream.max())}
    if "f32" in ref:
        report.update(flip_stats(branch, out, ref, ref["f32"]))
    print(json.dumps(report))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", f"parity_{tag}.json"), "w"), indent=1)
    assert excused.mean() < 1e-3, (tag, report["excused_fraction"])
    if "f32" in ref:
        assert_flips_in_line(report, tag)
    assert r <= 1e-4, (tag, r)
    assert per_stream.max() <= 1e-4, (tag, float(per_stream.max()), int(per_stream.argmax()))
    assert np.abs(d).max() <= 0.05 * max(np.abs(rr).max(), 1.0)        # flipped frames stay sane
    return report


def run_bench_style(nn, torch, S, model, x, calls):
    """x: device tensor [S, T, 480].  Calls of `calls` frames back to back on torch's stream, inputs declared final, no host
    synchronisation in between -- bench.py's timed loop.  Returns (y, vad [T, S], log [T, S, 24] int32) on the device."""
    T = x.shape[1]
    assert sum(calls) == T
    bd = nn.BatchDenoiser(S, model=model)
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


@pytest.mark.parametrize("S", [4096, 32768, 65536])
@pytest.mark.parametrize("model_name", ["builtin", "sh"])
def test_bench_configs_at_their_own_size(nn, oracle_mod, weights_bytes, S, model_name):
    import torch
    from nnnoiseless_amd.synthetic import make_streams
    T, calls = 96, (48, 48)
    blob = weights_bytes if model_name == "builtin" else open(os.path.join(GOLDEN, "sh.rnn"), "rb").read()
    model = None if model_name == "builtin" else nn.RnnModel.from_bytes(blob)
    first_id = 9008 if model_name == "builtin" else 9504        # (multiples of 16: stream i of the mix is silent when i % 16 == 7)
    base = make_streams(first_id, NDIST, T)                    # [256, 96, 480]
    ref, gtol = oracle_reference(oracle_mod, blob, base)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(S + len(model_name))
    idx = rng.permutation(S) % NDIST                        # every distinct stream S / 256 times, at scattered positions
    first = np.full(NDIST, -1, np.int64)
    for pos in range(S - 1, -1, -1):
        first[idx[pos]] = pos
    idx_d, first_d = torch.from_numpy(idx).to(dev), torch.from_numpy(first).to(dev)
    x = torch.from_numpy(base).to(dev)[idx_d]               # [S, T, 480] resident
    y, vad, log = run_bench_style(nn, torch, S, model, x, calls)
    del x
    ys, vs, ls = y[first_d], vad[:, first_d], log[:, first_d]
    # results do not depend on where in the batch / tile / workgroup a stream sits: every copy agrees bit for bit
    assert torch.equal(y, ys[idx_d]) and torch.equal(vad, vs[:, idx_d]) and torch.equal(log, ls[:, idx_d])
    out, v, lg = ys.cpu().numpy(), vs.cpu().numpy().T, ls.cpu().numpy().view(np.uint32)
    del y, vad, log
    torch.cuda.empty_cache()
    rep = check_against_oracle(out, v, lg, ref, gtol, f"bench_style_{model_name}_{S}streams_96frames")
    sil = np.array([s % 16 == 7 for s in range(NDIST)])
    assert not out[sil].any() and not v[sil].any()          # silence stays exactly zero
    assert rep["pitch_mismatches"] == 0


def real_audio_batch(n_streams, n_frames=99):
    """Streams derived from the reference's own recording testing.raw (src/lib.rs:196-213): gain -30 .. +6 dB, start
    offset 0 .. 479 samples (every alignment of the speech against the frame grid), polarity, white noise at SNR
    inf / 30 / 20 / 10 / 0 dB; rounded and clipped to int16 like any PCM source."""
    pcm = np.fromfile(os.path.join(GOLDEN, "testing.raw"), dtype="<i2").astype(np.float64)
    x = np.zeros((n_streams, n_frames * 480), np.float32)
    rms = np.sqrt((pcm ** 2).mean())
    for i in range(n_streams):
        rng = np.random.default_rng(31000 + i)
        off = i % 480 if i < 480 else int(rng.integers(0, 480))
        g = 10.0 ** (rng.uniform(-30.0, 6.0) / 20.0)
        pol = -1.0 if (i >> 1) & 1 else 1.0
        seg = pcm[off:off + n_frames * 480] * (g * pol)
        snr = (None, 30.0, 20.0, 10.0, 0.0)[i % 5]
        if snr is not None:
            seg = seg + rng.standard_normal(seg.size) * (g * rms * 10.0 ** (-snr / 20.0))
        x[i] = np.clip(np.round(seg), -32768, 32767)
    x[0] = pcm[:n_frames * 480]                               # stream 0: the recording as it is
    return x.reshape(n_streams, n_frames, 480)


def test_real_audio_batch(nn, oracle_mod, weights_bytes):
    import torch
    S, T = 512, 99
    x = real_audio_batch(S, T)
    ref, gtol = oracle_reference(oracle_mod, weights_bytes, x)
    y, vad, log = run_bench_style(nn, torch, S, None, torch.from_numpy(x).cuda(), (48, 48, 3))
    rep = check_against_oracle(y.cpu().numpy(), vad.cpu().numpy().T, log.cpu().numpy().view(np.uint32), ref, gtol,
                               "real_audio_512streams_99frames")
    # stream 0 is the golden recording itself: the reference's own acceptance metric on it (src/lib.rs:184-194)
    from conftest import golden_metric
    gold = np.fromfile(os.path.join(GOLDEN, "reference_output.raw"), dtype="<i2")
    assert golden_metric(y[0, 1:].cpu().numpy().reshape(-1), gold[:98 * 480]) < 1e-5
    assert rep["streams_compared"] == S


def _read_wav(path):
    """(rate, samples [n, channels] as the CLI hands them to the resampler): 16-bit ints as they are, floats x 32767
    (src/nnnoiseless.rs:179-227)."""
    import warnings
    from scipy.io import wavfile
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rate, d = wavfile.read(path)
    d = d.reshape(len(d), -1)
    if d.dtype == np.int16:
        return rate, d.astype(np.float32)
    assert d.dtype == np.float32
    return rate, d * np.float32(32767.0)


def _to_i16(v):
    """RawFrameWriter / WavFrameWriter: clamp, round half away from zero (src/nnnoiseless.rs:147-177)."""
    v = np.clip(v.astype(np.float64), -32768.0, 32767.0)
    return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.int16)


@pytest.mark.parametrize("name", ["mono", "stereo", "mono-float"])
def test_reference_wav_files_through_resampler_and_denoiser(nn, oracle_mod, weights_bytes, name):
    """The reference's bundled 44.1 kHz recordings through the CLI's chain: per-channel 16-tap sinc resampling to 48 kHz,
    480-sample frames while input lasts, process_frame per channel, first frame dropped, int16 out."""
    import torch
    from nnnoiseless_amd.pcm import Resampler
    rate, x = _read_wav(os.path.join(GOLDEN, name + ".wav"))            # [n, C]
    assert rate == 44100
    C_ = x.shape[1]
    ratio = rate / 48000.0
    ref_y = oracle_mod.resample(x, ratio, C_)                            # [n48, C]
    rs = Resampler(C_, rate)
    y = np.concatenate([rs.process(np.ascontiguousarray(x[a:a + 50000].T)) for a in range(0, len(x), 50000)], axis=1)   # [C, n48]
    assert y.shape[1] == ref_y.shape[0]
    assert np.array_equal(y.view(np.uint32), np.ascontiguousarray(ref_y.T).view(np.uint32))   # the resampler is bit-exact
    T = y.shape[1] // 480
    frames = np.ascontiguousarray(y[:, :T * 480]).reshape(C_, T, 480)
    ref, gtol = oracle_reference(oracle_mod, weights_bytes, frames)
    calls = [48] * (T // 48) + ([T % 48] if T % 48 else [])
    yd, vad, log = run_bench_style(nn, torch, C_, None, torch.from_numpy(frames).cuda(), tuple(calls))
    out = yd.cpu().numpy()
    rep = check_against_oracle(out, vad.cpu().numpy().T, log.cpu().numpy().view(np.uint32), ref, gtol, f"wav_{name}_{T}frames")
    got16, ref16 = _to_i16(out[:, 1:]), _to_i16(ref["out"][:, 1:])
    d = np.abs(got16.astype(np.int32) - ref16.astype(np.int32))
    ok = np.ones(d.shape[:2], bool)
    for s, t, _ in rep["flipped"]:
        ok[s, max(t - 1, 0):t + 1] = False                               # (frame t and its successor, in 1-based frame numbering)
    assert d[ok].max() <= 1 and (d[ok] != 0).mean() < 5e-3, (int(d[ok].max()), float((d[ok] != 0).mean()))


def test_host_calls_with_inputs_ready_do_not_race_their_upload(nn):
    """with set_inputs_ready(1), a pipelined host-buffer call must not start its high-pass before its own
    upload has landed.  Small batch, T >= 32, page-locked buffers, two consecutive calls: same bits as without the flag."""
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 30, 64
    x = make_streams(77, S, 2 * T)
    ref_bd = nn.BatchDenoiser(S)
    r1, v1 = ref_bd.process(x[:, :T])
    r2, v2 = ref_bd.process(x[:, T:])
    for rep in range(6):
        bd = nn.BatchDenoiser(S)
        bd.set_inputs_ready(True)
        px, po, pv = nn.pinned_empty((S, T, 480)), nn.pinned_empty((S, T, 480)), nn.pinned_empty((T, S))
        px[:] = x[:, :T]
        bd.process(px, out=po, vad=pv)
        a, va = po.copy(), pv.copy()
        px[:] = x[:, T:]                         # the staging area still holds the first call's audio when this one starts
        bd.process(px, out=po, vad=pv)
        assert np.array_equal(a, r1) and np.array_equal(va, v1), rep
        assert np.array_equal(po, r2) and np.array_equal(pv, v2), rep
        bd.close()


@pytest.mark.parametrize("gmax", [1, 4])
def test_batch_sized_for_real_time_ticks(nn, oracle_mod, weights_bytes, gmax):
    """a batch created with max_group_frames = 1 (4) holds a small fraction of the default batch's memory and
    gives the same bits, tick after tick (rings wrapping a dozen times) and on a longer call cut into short groups; 64 of its
    streams against the oracle."""
    import torch
    from nnnoiseless_amd.synthetic import make_streams
    S, T = 4096, 60
    dlog = torch.zeros((T, S, 24), dtype=torch.int32, device="cuda")   # (torch's context first: it does not come up behind the library's)
    x = np.tile(make_streams(5, 128, T), (S // 128, 1, 1))
    ref_bd = nn.BatchDenoiser(S)
    want, want_vad = ref_bd.process(x)
    bd = nn.BatchDenoiser(S, max_group_frames=gmax)
    got, vad = np.zeros_like(want), np.zeros_like(want_vad)
    for t in range(40):
        got[:, t:t + 1], vad[t:t + 1] = bd.process(x[:, t:t + 1])
    got[:, 40:], vad[40:] = bd.process(x[:, 40:])          # 20 frames on a batch sized for groups of gmax
    assert np.array_equal(got, want) and np.array_equal(vad, want_vad)
    per_stream, per_stream_default = bd.device_bytes() / S, ref_bd.device_bytes() / S
    print(f"max_group_frames={gmax}: {per_stream / 1024:.1f} KB per stream (default {per_stream_default / 1024:.1f} KB)")
    assert per_stream < (48 if gmax == 1 else 120) * 1024 and per_stream_default > 300 * 1024
    ref = oracle_mod.run_streams(oracle_mod.Model(weights_bytes), x[:64], n_threads=os.cpu_count() or 1, want=("out", "pitch", "vad"))
    bd.reset()
    bd.set_frame_log(dlog.data_ptr(), T)
    for t in range(T):
        bd.process(x[:, t:t + 1])
    bd.synchronize()
    log = dlog.cpu().numpy().view(np.uint32)
    assert np.array_equal(log[:, :64, 0].T.astype(np.int32), ref["pitch"])
    assert rel_rms(got[:64, 1:], ref["out"][:, 1:]) < 1e-4
    bd.close()
    ref_bd.close()


def test_withheld_handoff_flag_raises_a_sticky_fault(nn, gpu_lib):
    """a hand-off flag that never arrives (test hook) times out instead of hanging, and the failure reaches a
    caller that never calls nnn_batch_synchronize: nnn_batch_fault() reads it, the next process call refuses, reset clears."""
    import torch
    from nnnoiseless_amd.synthetic import make_streams_device
    S, T = 256, 32
    dev = torch.device("cuda", 0)
    x = make_streams_device(torch, dev, S, T, seed=3)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    bd = nn.BatchDenoiser(S)
    bd.process_device(x.data_ptr(), y.data_ptr(), 0, T, T * 480, 480, stream)
    torch.cuda.synchronize()
    clean = y.clone()
    assert not bd.fault()
    bd.reset()
    gpu_lib.check(gpu_lib.L.nnn_batch_debug_withhold_flag(bd._h, 5))
    bd.process_device(x.data_ptr(), y.data_ptr(), 0, T, T * 480, 480, stream)
    torch.cuda.synchronize()                      # the caller synchronises its own stream, as bench.py does
    assert bd.fault()
    with pytest.raises(RuntimeError, match="hand-off"):
        bd.process_device(x.data_ptr(), y.data_ptr(), 0, 1, T * 480, 480, stream)
    with pytest.raises(RuntimeError, match="hand-off"):
        bd.synchronize()
    gpu_lib.check(gpu_lib.L.nnn_batch_debug_withhold_flag(bd._h, -1))
    bd.reset()
    assert not bd.fault()
    bd.process_device(x.data_ptr(), y.data_ptr(), 0, T, T * 480, 480, stream)
    torch.cuda.synchronize()
    assert torch.equal(y, clean)


def test_rccl_single_rank_aggregate():
    """the first SCALE run must not be the first time RCCL is touched.  One rank, backend "nccl" (= RCCL on
    ROCm): init, the rank-count all-reduce of ones and bench.py's result aggregation (sum of frames, max of elapsed)."""
    import subprocess
    import sys
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from nnnoiseless_amd.shard import aggregate\n"
        "os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')\n"
        "os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(0)\n"
        "dist.init_process_group(backend='nccl', device_id=dev)\n"
        "one = torch.ones(1, dtype=torch.int32, device=dev); dist.all_reduce(one); assert int(one.item()) == 1\n"
        "f, t = aggregate(dist, 12345, 0.5, dev); assert f == 12345 and abs(t - 0.5) < 1e-6, (f, t)\n"
        "dist.barrier(); dist.destroy_process_group(); print('rccl ok', dist.Backend.NCCL)\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    txt = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=600).decode()
    assert "rccl ok" in txt
