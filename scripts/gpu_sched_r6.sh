#!/bin/bash
# Round 6: which schedule the big batches want now that k_pitch waits more and issues less (NNN_SCHED / NNN_LANES are read at batch creation; same bits
# either way).  Interleaved repetitions on one box -> gpurun_out/r6_sched_sweep.txt
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
one() {
  local tag="$1"; shift
  env "$@" timeout 300 python bench.py --streams $S --frames-per-step $F --steps $ST --warmup 2 --no-cpu-baseline --no-roofline --no-also --no-tick --no-host 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$S F=$F $tag: %.2f M  (%.3f ms per step)' % (d['value']/1e6, d['ms_per_step']))"
}
for rep in 1 2 3; do
  for S in ${SIZES:-65536 32768}; do
    for F in ${FPS:-48 96}; do
      ST=$((480 / F))
      one seq NNN_SCHED=seq
      one lanes2 NNN_SCHED=lanes NNN_LANES=2
      one lanes3 NNN_SCHED=lanes NNN_LANES=3
      one stages NNN_SCHED=stages
    done
  done
done 2>&1 | tee gpurun_out/r6_sched_sweep.txt
