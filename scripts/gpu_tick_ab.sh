#!/bin/bash
# One-frame-per-call rate (bench.py's `tick` entry: the default batch and a batch created with max_group_frames = 1) for library
# variants, configs[1] and configs[2].   usage (GPU box): VARIANTS="default prev" bash scripts/gpu_tick_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for V in ${VARIANTS:-default}; do
  if [ "$V" = default ]; then unset NNN_LIBRARY; else export NNN_LIBRARY=$R/nnnoiseless_amd/lib/variants/$V.so; fi
  for C in 1 2; do
  python bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-host --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V config $C tick', json.dumps(d.get('tick')))"
  done
done
