import os, sys, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # the host's setting
sys.path.insert(0, os.getcwd())
import torch
import nnnoiseless_amd as nn
from nnnoiseless_amd.synthetic import make_streams_device
S = 4096
dev = torch.device("cuda", 0)
x = make_streams_device(torch, dev, S, 64, seed=0)
y = torch.empty_like(x); vad = torch.empty((64, S), dtype=torch.float32, device=dev)
bd = nn.BatchDenoiser(S, max_group_frames=1)
st = torch.cuda.current_stream().cuda_stream
for t in range(20):
    bd.process_device(x.data_ptr() + t * 480 * 4, y.data_ptr() + t * 480 * 4, vad.data_ptr() + t * S * 4, 1, 64 * 480, 480, st)
torch.cuda.synchronize()
bd.set_profiling(True)
for t in range(20, 60):
    bd.process_device(x.data_ptr() + t * 480 * 4, y.data_ptr() + t * 480 * 4, vad.data_ptr() + t * S * 4, 1, 64 * 480, 480, st)
torch.cuda.synchronize()
print({k: round(ms * 1e3 / max(n, 1), 1) for k, (ms, n) in bd.kernel_times().items() if n})
