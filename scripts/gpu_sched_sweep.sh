#!/bin/bash
# Schedule variants at the headline batch (same box, back to back): lanes, stages, hardware queues
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
one() {
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-also --no-tick ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*: %.2f M  (%.3f ms per step)' % (d['value']/1e6, d['ms_per_step']))"
}
for rep in 1 2; do
one A=1
one NNN_LANES=1
one NNN_LANES=3
one NNN_SCHED=stages
one NNN_SCHED=seq
one GPU_MAX_HW_QUEUES=8
one GPU_MAX_HW_QUEUES=8 NNN_LANES=3
one GPU_MAX_HW_QUEUES=8 NNN_SCHED=stages
done 2>&1 | tee gpurun_out/sched_sweep.txt
