#!/bin/bash
# Training-row path: parity on the GPU, then rows/s at 4096 and 16384 (clean, noise, mix) triples
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -x -q -k "rows or train" 2>&1 | tail -3
for S in 4096 16384; do
  timeout 300 python bench.py --workload train --streams $S --steps 20 --warmup 3 > gpurun_out/train_$S.json 2> gpurun_out/train_$S.err
  python -c "import json,sys; d=json.load(open('gpurun_out/train_$S.json')); print($S, round(d['value']/1e6,2), 'M rows/s')" || tail -5 gpurun_out/train_$S.err
done
