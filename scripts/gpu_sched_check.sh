#!/bin/bash
# The automatic schedule of a big batch against the explicit ones, interleaved on one box -> gpurun_out/r6_sched_check.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
one() { local tag="$1"; shift; env "$@" timeout 300 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-also --no-tick --no-host 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag: %.2f M (%.3f ms per step)' % (d['value']/1e6, d['ms_per_step']))"; }
for rep in 1 2 3 4; do
one auto A=1
one seq NNN_SCHED=seq
one stages NNN_SCHED=stages
done 2>&1 | tee gpurun_out/r6_sched_check.txt
