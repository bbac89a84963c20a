#!/bin/bash
# One frame per call (`tick`) at 4096 and 65536 streams: the layer-pipelined RNN kernel against k_rnn for one-frame groups
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
for G in 1 2; do
  for C in 1 2; do
    NNN_RNN_WF_MIN_G=$G timeout 300 python bench.py --config $C --steps 4 --warmup 1 --no-cpu-baseline --no-also --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wf_min_g=$G config $C: %.2f M, tick %.2f M (%.1f us per call)' % (d['value']/1e6, d['tick']['value']/1e6, d['tick']['ms_per_step']*1e3))"
  done
done
done 2>&1 | tee gpurun_out/tick.txt
