#!/bin/bash
# Lane-count experiment for pipelined calls (frames in flight on separate HIP streams).
set -u
mkdir -p gpurun_out
for L in 2 3 4; do for F in 16 64; do
  NNN_LANES=$L timeout 300 python bench.py --frames-per-step $F --steps $((960 / F)) --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/lanes_${L}_$F.json 2>gpurun_out/lanes_${L}_$F.err
  python -c "import json; d=json.load(open('gpurun_out/lanes_${L}_$F.json')); print('lanes=$L frames/step=$F value=%.3e ms/step=%.3f' % (d['value'], d['ms_per_step']))"
done; done
