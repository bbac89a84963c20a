#!/bin/bash
# Frames-per-call sweep at the headline batch: how much of a call is pipeline fill and drain
set -u
mkdir -p gpurun_out
for F in ${FPSS:-48 96 192 384}; do
  timeout 300 python bench.py --frames-per-step $F --steps $((1920 / F)) --warmup 3 --no-cpu-baseline --no-roofline --no-also --no-tick ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frames/step=$F: %.2f M  (%.3f ms per step)' % (d['value']/1e6, d['ms_per_step']))"
done
