#!/usr/bin/env python3
"""One 10 ms frame per call, the way a real-time server is driven: H independent batches (handles) of S streams each, every batch
on its own HIP stream, called round-robin.  Calls of different batches overlap on the GPU; a single batch called back to back
(bench.py's `tick`) cannot overlap with itself.  usage: tick_capacity.py [streams_per_batch] [batches] [rounds] [host_threads] [max_group_frames: 0 = default batch, 1 = sized for ticks]"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # (the host's setting: batches ticking side by side want eight hardware queues)
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nnnoiseless_amd as nn
from nnnoiseless_amd.synthetic import make_streams_device

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
pool = 16
xs = [make_streams_device(torch, dev, S, pool, seed=h) for h in range(H)]
ys = [torch.empty_like(x) for x in xs]
vs = [torch.empty((pool, S), dtype=torch.float32, device=dev) for _ in range(H)]
GM = int(sys.argv[5]) if len(sys.argv) > 5 else 0
bds = [nn.BatchDenoiser(S, max_group_frames=GM or None) for _ in range(H)]
streams = [torch.cuda.Stream() for _ in range(H)]
for b in bds:
    b.set_inputs_ready(True)

def rounds(n, f0):
    for r in range(n):
        f = (f0 + r) % pool
        for h in range(H):
            off = f * 480 * 4
            bds[h].process_device(xs[h].data_ptr() + off, ys[h].data_ptr() + off, vs[h].data_ptr() + f * S * 4, 1, pool * 480, 480,
                                  streams[h].cuda_stream)

NT = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # host threads, each driving its share of the batches (ctypes calls release the GIL)

def rounds_threaded(n, f0):
    import threading
    def work(t):
        for r in range(n):
            f = (f0 + r) % pool
            for h in range(t, H, NT):
                off = f * 480 * 4
                bds[h].process_device(xs[h].data_ptr() + off, ys[h].data_ptr() + off, vs[h].data_ptr() + f * S * 4, 1, pool * 480, 480,
                                      streams[h].cuda_stream)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(NT)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()

run = rounds if NT == 1 else rounds_threaded
run(10, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(R, 10)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{NT} host thread(s), {H} batches x {S} streams, one frame per call: {H * S * R / dt / 1e6:.2f} M frames/s "
      f"({dt / R * 1e6:.0f} us per round of {H} calls = {H * S} streams served; {dt / R / H * 1e6:.0f} us per call); "
      f"max_group_frames {bds[0].max_group_frames()}, {bds[0].device_bytes() / S / 1024:.1f} KB per stream", flush=True)
