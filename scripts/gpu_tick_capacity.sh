#!/bin/bash
# Real-time serving pattern: several independent batches ticking one frame per call on their own streams
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "4096 1" "4096 2" "4096 4" "4096 8" "4096 16" "1024 16" "16384 4"; do
  set -- $cfg
  PYTHONPATH=. timeout 300 python scripts/tick_capacity.py $1 $2 200 2>&1 | grep "frames/s"
done | tee gpurun_out/tick_capacity.txt
