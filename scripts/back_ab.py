#!/usr/bin/env python3
"""Same-run A/B of the back ends (nnn_batch_set_back_end) at one batch size: one frame per call (the real-time tick) and 48 frames
per call, device-resident buffers, one process.  usage: back_ab.py [streams] [reps]
  mode 0  k_fft_xp -> k_rnn / k_rnn_wf -> k_synth          mode 1 / 2  the fused kernel (k_back) for one-frame / all groups
  mode 3 / 4  the fused kernel's RNN stretch alone (k_back<false>) as the RNN kernel for one-frame / all groups"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nnnoiseless_amd as nn
from nnnoiseless_amd.synthetic import make_streams_device

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
pool = 96
x = make_streams_device(torch, dev, S, pool, seed=0)
y = torch.empty_like(x)
vad = torch.empty((pool, S), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def call(bd, f0, n):
    off = f0 * 480 * 4
    bd.process_device(x.data_ptr() + off, y.data_ptr() + off, vad.data_ptr() + f0 * S * 4, n, pool * 480, 480, stream)


def tick_rate(mode, gmax=None, n=300):
    bd = nn.BatchDenoiser(S, max_group_frames=gmax)
    bd.set_back_end(mode)
    bd.set_inputs_ready(True)
    for j in range(20):
        call(bd, j % pool, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(n):
        call(bd, (20 + j) % pool, 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    # per-kernel HIP-event times of the same calls (adds some microseconds per launch)
    bd.set_profiling(True)
    for j in range(40):
        call(bd, j % pool, 1)
    torch.cuda.synchronize()
    kt = {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in bd.kernel_times().items() if c}
    bd.close()
    return dt, kt


def group_rate(mode, fps=48, n=None):
    n = n or max(6, int(2.0e8 / (S * fps)))
    bd = nn.BatchDenoiser(S)
    bd.set_back_end(mode)
    bd.set_inputs_ready(True)
    for j in range(3):
        call(bd, (j * fps) % (pool - fps + 1), fps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(n):
        call(bd, (j % 2) * fps, fps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    bd.set_profiling(True)
    for j in range(2):
        call(bd, (j % 2) * fps, fps)
    torch.cuda.synchronize()
    kt = {k: round(1e3 * ms / fps / 2, 2) for k, (ms, c) in bd.kernel_times().items() if c}
    bd.close()
    return dt, kt


for rep in range(REPS):
    for mode in (0, 1, 3):
        dt, kt = tick_rate(mode)
        print(f"S={S} tick  mode {mode}: {dt * 1e6:7.1f} us per call = {S / dt / 1e6:6.2f} M frames/s   per-launch us {kt}", flush=True)
    dt, kt = tick_rate(1, gmax=1)
    print(f"S={S} tick  mode 1 (batch sized for ticks): {dt * 1e6:7.1f} us per call = {S / dt / 1e6:6.2f} M frames/s", flush=True)
    for mode in (0, 2, 4):
        dt, kt = group_rate(mode)
        print(f"S={S} 48 fr mode {mode}: {dt * 1e3:7.3f} ms per call = {S * 48 / dt / 1e6:6.2f} M frames/s   us per frame {kt}", flush=True)
