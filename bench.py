#!/usr/bin/env python
"""bench.py -- 480-sample frames/sec of the batched process_frame path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch: every stream of this rank's shard advances by
`--frames-per-step` frames (default 48 = 0.48 s of audio per stream per call; the library runs them as groups
of 4 frames per launch, three groups in flight).  The single-frame tick (`--frames-per-step 1`, what a live 10 ms cadence would
use) is measured as well and reported in `tick`.  The workload is BASELINE.json configs[1]: 4096 concurrent
mono streams per GPU, built-in model, synthetic 48 kHz sine + noise, inputs resident
in HBM before the timed region.  Streams are
independent, so ranks shard them with no data-path collective (weak scaling: 4096 streams per GPU);
the only collective is the aggregation of the result.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel: algorithmic bytes per launch / its average duration measured live
                with HIP events on the launch stream (a second, event-instrumented pass over the same
                workload: the timed pass replays a hipGraph, which cannot carry per-kernel events)
  cpu_baseline  the CPU oracle (scalar C port of the reference, f32 FFT) on the host's cores, bounded
                sample; the genuine Rust reference cannot be built here (no cargo), hence kind "port"
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3
BYTES_PER_FRAME_FUSED = 16948  # I/O + resident-state touch of one process_frame
FLOPS_PER_FRAME = 0.42e6

# Algorithmic HBM bytes per stream-frame of each kernel (its own inputs + outputs, each counted once;
# derivation in DESIGN.md "Kernels")
KERNEL_BYTES = {
    "k_hp": 1920 + 1920 + 16 + 2 * 960 + 8,          # input, history slot, biquad state, 240 decimated values (stored twice)
    "k_lpc": 3456 + 40 + 2 * 3456,                    # decimated window in; taps + pitch_buf (TI + SM) out
    "k_xcorr": 3456 + 588 + 1544,                     # pitch_buf in; 147 coarse lags + xx / yy_lookup (386) out
    "k_best1": 1548 + 588 + 8,
    "k_refine": 3456 + 8 + 40,
    "k_best2": 3456 + 40 + 8 + 4 + 2 * 1176,
    "k_doubling": 3456 + 116 + 24,
    "k_fft_x": 3840 + 3848 + 88,
    "k_fft_p": 3840 + 4 + 3848 + 3848 + 176,
    "k_rnn": 264 + 88 + 704 + 88 + 168 + 16 + 2 * 672 + 2 * 88 + 180,   # features stage + RNN state/gains (weights amortised)
    "k_synth": 7696 + 440 + 3840 + 1920 + 4,
    "k_advance": 0,
}


def cpu_baseline(budget_s=12.0):
    """Time the CPU oracle (f32-FFT build) on all host cores over a bounded sample of the same workload."""
    from oracle import oracle as O
    from nnnoiseless_amd.synthetic import make_streams_fast
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    model = O.Model(open(os.path.join(ROOT, "nnnoiseless_amd", "data", "weights.rnn"), "rb").read(), f32_fft=True)
    frames = 100
    x = make_streams_fast(cores, frames, seed=99)
    t0 = time.perf_counter()
    O.run_streams(model, x[:1], n_threads=1, want=("out",))
    per_frame = (time.perf_counter() - t0) / frames
    per_thread = max(1, min(64, int(budget_s / (per_frame * frames))))
    x = make_streams_fast(cores * per_thread, frames, seed=99)
    t0 = time.perf_counter()
    used = O.run_streams(model, x, n_threads=cores, want=("out",))["threads"]
    dt = time.perf_counter() - t0
    total = x.shape[0] * frames
    return {"value": total / dt, "unit": "frames/s", "cores": used, "kind": "port",
            "sample": f"{x.shape[0]} synthetic streams x {frames} frames, {used} threads, "
                      f"oracle/nnn_oracle.c -O3 f32 FFT; single-thread {1.0 / per_frame:.0f} frames/s",
            "note": "genuine Rust reference not buildable here (no cargo/rustc)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--streams", type=int, default=4096, help="concurrent streams PER GPU")
    ap.add_argument("--frames-per-step", type=int, default=48,
                    help="frames per stream per call (0.48 s of audio by default); the same JSON line also reports the one-frame-per-call rate (`tick`)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--workload", choices=["denoise", "train"], default="denoise",
                    help="denoise = process_frame (the headline); train = 87-column training rows (SURVEY 8(f) #3)")
    ap.add_argument("--pcm", choices=["f32", "i16", "unit"], default="f32",
                    help="boundary sample format (SURVEY 8(f) #1): f32 = process_frame's own (headline), i16 = the CLI's "
                         "packed int16, unit = DenoiseSignal's [-1, 1] floats")
    ap.add_argument("--channels", type=int, default=1, help="interleaved channels per group (with --pcm i16/unit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import nnnoiseless_amd as nn
    from nnnoiseless_amd.synthetic import make_streams_fast
    from nnnoiseless_amd.shard import aggregate

    if args.workload == "train":
        return bench_train(args, rank, world, dev, local_rank, dist)
    S, fps, K, W = args.streams, args.frames_per_step, args.steps, args.warmup
    total_frames = (K + W) * fps
    # distinct audio for every step while it fits ~6 GB per GPU, else cycle through a pool of frames
    pool = total_frames
    while S * pool * 480 * 4 > 6e9 and pool > 8:
        pool //= 2
    x_host = make_streams_fast(S, pool, seed=rank)
    fmt = {"f32": 0, "i16": 1, "unit": 2}[args.pcm]
    Cc = args.channels
    assert S % Cc == 0
    if fmt or Cc > 1:   # packed PCM: [groups][pool * 480][channels], channel-interleaved
        x_host = x_host.reshape(S // Cc, Cc, pool * 480).transpose(0, 2, 1)
        x_host = np.ascontiguousarray(x_host.astype(np.int16) if fmt == 1 else (x_host / 32768.0 if fmt == 2 else x_host))
    x = torch.from_numpy(x_host).to(dev)              # resident in HBM
    y = torch.empty_like(x)
    vad = torch.empty((pool, S), dtype=torch.float32, device=dev)
    del x_host
    esz = x.element_size()
    bd = nn.BatchDenoiser(S, device=local_rank)
    if args.no_graph:
        bd.set_graph(False)
    stream = torch.cuda.current_stream().cuda_stream

    def run(f0, n):   # n frames of every stream starting at frame f0 of the pool
        off = f0 * 480 * Cc * esz
        if fmt or Cc > 1:
            bd.process_pcm_device(x.data_ptr() + off, y.data_ptr() + off, vad.data_ptr() + f0 * S * 4, n, fmt, Cc,
                                  pool * 480 * Cc, 480 * Cc, False, stream)
        else:
            bd.process_device(x.data_ptr() + off, y.data_ptr() + off, vad.data_ptr() + f0 * S * 4, n, pool * 480, 480, stream)

    def step(i):
        f0 = (i * fps) % pool
        n = min(fps, pool - f0)  # a step never wraps inside the pool unless fps does not divide it
        run(f0, n)
        if n < fps:
            run(0, fps - n)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(W):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        step(i)
    t_enq = time.perf_counter() - t0          # host time to enqueue everything (the calls are asynchronous)
    barrier()
    elapsed = time.perf_counter() - t0
    frames_done, elapsed_max = aggregate(dist if world > 1 else None, S * fps * K, elapsed, dev)
    ms_per_step = elapsed_max * 1e3 / K
    value = frames_done / elapsed_max
    finite = bool(torch.isfinite(y.float()).all().item())

    # the same workload at one frame per call (live 10 ms tick): no frames in flight, one graph replay per frame
    tick = None
    if fps != 1:
        kt = min(200, max(20, K * fps // 4))
        pos = (W + K) * fps
        def tick_step(j):
            run((pos + j) % pool, 1)
        for j in range(10):
            tick_step(j)
        barrier()
        t1 = time.perf_counter()
        for j in range(10, 10 + kt):
            tick_step(j)
        barrier()
        tt = time.perf_counter() - t1
        tf, tmax = aggregate(dist if world > 1 else None, S * kt, tt, dev)
        tick = {"frames_per_step": 1, "value": tf / tmax, "unit": "frames/s", "ms_per_step": tmax * 1e3 / kt, "steps": kt}

    roofline = None
    kern = {}
    if rank == 0 and not args.no_roofline:
        # second pass over the same workload with HIP events around every launch (eager launches)
        bd.set_profiling(True)
        kp = min(K, 50)
        t1 = time.perf_counter()
        for i in range(W + K, W + K + kp):
            run((i * fps) % pool, 1)
        torch.cuda.synchronize()
        prof_ms_per_step = (time.perf_counter() - t1) * 1e3 / kp   # per FRAME: the instrumented pass runs single frames
        times = bd.kernel_times()
        bd.set_profiling(False)
        kern = {k: {"avg_us": 1e3 * ms / max(n, 1), "launches": n} for k, (ms, n) in times.items()}
        dom = max(times, key=lambda k: times[k][0])
        avg_s = times[dom][0] / times[dom][1] * 1e-3
        achieved = KERNEL_BYTES[dom] * S / avg_s / 1e9
        sum_us = sum(v["avg_us"] for v in kern.values())
        traffic = None
        try:   # HBM-side bytes per launch of that kernel from the committed rocprofv3 --pmc passes (same workload only)
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_4096streams.json")))
            if pm.get("streams") == S and dom in pm["kernels"]:
                traffic = pm["kernels"][dom]["hbm_bytes_per_launch"]
        except Exception:
            pass
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "avg_kernel_us": avg_s * 1e6, "bytes_per_launch": KERNEL_BYTES[dom] * S,
                    "sum_kernel_us_per_frame": sum_us, "profiled_ms_per_frame": prof_ms_per_step,
                    "pipeline_fused_bytes_GBs": value / args.gpus * BYTES_PER_FRAME_FUSED / 1e9,
                    "pipeline_hbm_frac": value / args.gpus * BYTES_PER_FRAME_FUSED / 1e9 / HBM_PEAK_GBS,
                    "pipeline_fp32_frac": value / args.gpus * FLOPS_PER_FRAME / 1e12 / FP32_PEAK_TFLOPS,
                    "note": "path is FP32-VALU/latency bound (23-100 FLOP/B, SURVEY 8d); HBM fraction reported as north_star asks"}

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        line = {
            "metric": "480-sample frames/sec (whole node) at N concurrent streams", "value": value, "unit": "frames/s",
            "n_gpus": args.gpus, "steps": K, "warmup": W, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{S} concurrent mono streams per GPU, built-in weights.rnn, synthetic 48 kHz sine+noise "
                                   f"(BASELINE.json configs[1]), {fps} frame(s) per stream per step",
                       "boundary_format": {"f32": "f32 planar (process_frame's own)", "i16": "packed int16",
                                           "unit": "unit-range f32"}[args.pcm] + (f", {Cc} interleaved channels" if Cc > 1 else ""),
                       "streams_per_gpu": S, "streams_total": S * world, "frames_per_step": fps,
                       "launch": ("groups of 4 frames per launch for the kernels without cross-frame state, three groups in flight on three HIP streams, each group's pitch-front segment replayed as a hipGraph"
                                  if fps > 1 else "eager launches, one frame per call"),
                       "parallelism": f"streams sharded x{world}"},
            "tick": tick,
            "host_enqueue_ms_per_step": t_enq * 1e3 / K,
            "outputs_finite": finite,
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kern,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_train(args, rank, world, dev, local_rank, dist):
    """Training-feature rows per second: three feature states per stream, no RNN, no synthesis."""
    import torch
    from nnnoiseless_amd.shard import aggregate
    from nnnoiseless_amd.synthetic import make_streams_fast
    from nnnoiseless_amd.training import ROW_WIDTH, TrainingFeatures
    S, fps, K, W = args.streams, args.frames_per_step, args.steps, args.warmup
    pool = max(fps, 16)
    sig = torch.from_numpy(make_streams_fast(S, pool, seed=rank)).to(dev)
    noise = torch.from_numpy(make_streams_fast(S, pool, seed=1000 + rank) * 0.3).to(dev)
    comb = sig + noise
    cutoff = torch.full((pool, S), 20, dtype=torch.int32, device=dev)
    vad = torch.ones((pool, S), dtype=torch.float32, device=dev)
    rows = torch.empty((pool, S, ROW_WIDTH), dtype=torch.float32, device=dev)
    tf = TrainingFeatures(S, device=local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        tf.process_device(sig.data_ptr(), noise.data_ptr(), comb.data_ptr(), cutoff.data_ptr(), vad.data_ptr(), rows.data_ptr(),
                          fps, pool * 480, 480, stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(W):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    done, tmax = aggregate(dist if world > 1 else None, S * fps * K, elapsed, dev)
    if rank == 0:
        print(json.dumps({
            "metric": "training rows/sec (87 columns; 3 feature states per row)", "value": done / tmax, "unit": "rows/s",
            "n_gpus": args.gpus, "steps": K, "warmup": W, "ms_per_step": tmax * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{S} (clean, noise, mix) stream triples per GPU, {fps} frame(s) per step (src/training.rs:113-160)",
                       "streams_per_gpu": S, "frames_per_step": fps},
            "outputs_finite": bool(torch.isfinite(rows).all().item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
