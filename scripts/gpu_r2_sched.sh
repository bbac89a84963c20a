#!/bin/bash
# scheduling experiments at 4096 streams x 48 frames + kernel trace with timeline
set -u
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-also --no-roofline --steps 20 --warmup 4"
for L in 1 2 3 4; do
  NNN_LANES=$L timeout 200 $B > gpurun_out/s_l$L.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/s_l$L.json')); print('lanes=$L: %.2f M (enq %.2f ms/step of %.2f)' % (d['value']/1e6, d['host_enqueue_ms_per_step'], d['ms_per_step']))"
done
for Q in 8; do
  GPU_MAX_HW_QUEUES=$Q timeout 200 $B > gpurun_out/s_q$Q.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/s_q$Q.json')); print('hwq=$Q lanes=3: %.2f M' % (d['value']/1e6))"
done
for L in 2 3; do
NNN_LIBRARY=$PWD/nnnoiseless_amd/lib/variants/libnnn_g8.so NNN_LANES=$L timeout 200 $B > gpurun_out/s_g8.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/s_g8.json')); print('group=8 lanes=$L: %.2f M' % (d['value']/1e6))"
done
TIMELINE=70 bash scripts/gpu_trace.sh 2>&1 | tail -90
