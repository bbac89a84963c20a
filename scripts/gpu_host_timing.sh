#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python - <<'PY'
import numpy as np, sys, torch, time
sys.path.insert(0, '.')
torch.cuda.init()
import nnnoiseless_amd as nn
from nnnoiseless_amd.synthetic import make_streams_fast
for S in (1024, 4096):
    T = 64
    xd = torch.from_numpy(make_streams_fast(S, T)).cuda()
    vd = torch.zeros((T, S), device="cuda")
    bd = nn.BatchDenoiser(S)
    for mode in ("pipelined", "graph-per-frame", "eager-per-frame"):
        bd.set_pipeline(mode == "pipelined")
        bd.set_graph(mode != "eager-per-frame")
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "pipelined":
                bd.process_device(xd.data_ptr(), xd.data_ptr(), vd.data_ptr(), T, T*480, 480, 0)
            else:
                for t in range(T):
                    bd.process_device(xd.data_ptr() + t*480*4, xd.data_ptr() + t*480*4, vd.data_ptr() + t*S*4, 1, T*480, 480, 0)
            t1 = time.perf_counter()
            bd.synchronize(); torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"S={S} {mode:16s} host enqueue {1e6*(t1-t0)/T:7.1f} us/frame   total {1e6*(t2-t0)/T:7.1f} us/frame", flush=True)
PY
