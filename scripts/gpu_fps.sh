#!/bin/bash
# frames-per-call sweep of the pipelined path
set -u
mkdir -p gpurun_out
for S in ${STREAMS:-4096}; do for F in ${FPS_LIST:-8 12 16 24 36 48 96}; do
  timeout 300 python bench.py --streams $S --frames-per-step $F --steps $((1440 / F)) --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/fps_${S}_$F.json 2>gpurun_out/fps_${S}_$F.err
  python -c "import json; d=json.load(open('gpurun_out/fps_${S}_$F.json')); print('S=$S frames/step=$F value=%.3e ms/step=%.3f' % (d['value'], d['ms_per_step']))"
done; done
