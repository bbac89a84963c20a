#!/bin/bash
# Quick perf check: parity suite, headline bench, tick benches at scale with per-kernel times.
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/q_bench.json 2>gpurun_out/q_bench.err
python -c "
import json; d=json.load(open('gpurun_out/q_bench.json'))
print('4096x16: %.2f M  tick %.2f M' % (d['value']/1e6, d['tick']['value']/1e6), {k[2:]: round(v['avg_us'],1) for k,v in d['kernels'].items()})"
for S in ${SCALE:-16384 65536}; do
  timeout 300 python bench.py --streams $S --frames-per-step 1 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/q_scale_$S.json 2>gpurun_out/q_scale_$S.err
  python -c "
import json; d=json.load(open('gpurun_out/q_scale_$S.json'))
print('S=$S tick: %.2f M' % (d['value']/1e6), {k[2:]: round(v['avg_us']) for k,v in d['kernels'].items()})"
done
