#!/bin/bash
# RNN rows-per-block experiment: tick (1 frame/call) and pipelined throughput at several batch sizes.
set -u
mkdir -p gpurun_out
for S in ${STREAMS:-1024 4096 16384}; do for R in 64 32 16; do
  NNN_RNN_ROWS=$R timeout 300 python bench.py --streams $S --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/rows_${S}_$R.json 2>gpurun_out/rows_${S}_$R.err
  python -c "import json; d=json.load(open('gpurun_out/rows_${S}_$R.json')); print('S=$S rows=$R value=%.3e tick=%.3e k_rnn=%.1f us' % (d['value'], d['tick']['value'], d['kernels']['k_rnn']['avg_us']))"
done; done
