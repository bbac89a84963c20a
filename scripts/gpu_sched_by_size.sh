#!/bin/bash
# Schedule by batch size: NNN_LANES=2 (default) / 1 / NNN_SCHED=seq at several stream counts, 48-frame calls (M frames/s)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for S in ${SIZES:-4096 16384 32768 65536}; do
  ST=$((12 * 65536 / S)); [ $ST -gt 120 ] && ST=120
  for V in "NNN_LANES=2" "NNN_LANES=1" "NNN_SCHED=seq"; do
    env $V python bench.py --streams $S --steps $ST --warmup 3 --no-cpu-baseline --no-also --no-tick --no-host --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$S streams $V: %.2f M' % (d['value']/1e6))"
  done
done
