#!/bin/bash
# Round 4, first GPU session of the fused back end: its parity tests, then same-run A/Bs of the back ends at 4096 and 65536 streams and
# eight 4096-stream batches ticking side by side.
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_back_end.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r4_back_tests.txt
for S in 4096 65536; do
  timeout 600 python scripts/back_ab.py $S 2 2>&1 | grep -v Warning
done | tee gpurun_out/r4_back_ab.txt
for B in 0 1; do
  echo "NNN_BACK=$B"; NNN_BACK=$B timeout 300 python scripts/tick_capacity.py 4096 8 200 1 1 2>&1 | tail -1
done | tee gpurun_out/r4_back_tick_capacity.txt
