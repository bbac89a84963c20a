#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for MODE in nosync nosync_ownstream; do
timeout 300 python -X faulthandler - $MODE <<'PY'
import numpy as np, sys, torch
sys.path.insert(0, '.')
torch.cuda.init()
import nnnoiseless_amd as nn
from nnnoiseless_amd.synthetic import make_streams
mode = sys.argv[1]
S = 256
x = make_streams(0, S, 8)
xd = torch.from_numpy(np.tile(x, (1, 40, 1))).cuda()
vd = torch.zeros((320, S), device="cuda")
torch.cuda.synchronize()
bd = nn.BatchDenoiser(S)
st = 0 if mode == "nosync_ownstream" else torch.cuda.current_stream().cuda_stream
n = 0
for it in range(40):
    bd.process_device(xd.data_ptr() + it*8*480*4, xd.data_ptr() + it*8*480*4, vd.data_ptr() + it*8*S*4, 8, 320*480, 480, st)
    n += 1
    print(n, end=" ", flush=True)
bd.synchronize(); torch.cuda.synchronize()
print(mode, "completed", n, flush=True)
PY
echo "exit $?"
done
