"""Developer probe (round 6): frames per group of a big batch under the overlapping schedules -- 65 536 streams, 48-frame calls, inputs resident.
usage: python scripts/gpu_group_len.py [streams]  -> lines 'gmax schedule: M frames/s'"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import nnnoiseless_amd as nn
from nnnoiseless_amd.synthetic import make_streams_device
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = 48
dev = torch.device("cuda", 0)
x = make_streams_device(torch, dev, S, T, seed=0)
y = torch.empty_like(x)
v = torch.empty((T, S), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for gmax in (24, 16, 12, 8):
        for mode in ("auto", "stages", "lanes2", "lanes3"):
            bd = nn.BatchDenoiser(S, max_group_frames=gmax)
            if mode == "stages": bd.set_schedule("stages")
            elif mode.startswith("lanes"): bd.set_schedule("lanes", int(mode[5:]))
            bd.set_inputs_ready(True)
            for _ in range(2): bd.process_device(x.data_ptr(), y.data_ptr(), v.data_ptr(), T, T * 480, 480, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 8
            for _ in range(K): bd.process_device(x.data_ptr(), y.data_ptr(), v.data_ptr(), T, T * 480, 480, stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"S={S} gmax={gmax} {mode}: {S * T * K / dt / 1e6:.2f} M frames/s  ({bd.device_bytes() / S / 1024:.0f} KB per stream)", flush=True)
            bd.close()
