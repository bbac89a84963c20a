#!/bin/bash
# Same build, one environment variable, side by side: ENVVAR=NNN_LPC_FC VALUES="1 0" bash scripts/gpu_ab_env.sh   (configs 2 and 1, twice)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for L in ${VALUES:-0 1}; do for C in ${CONFIGS:-2 1}; do
  ST=12; [ "$C" = 1 ] && ST=120
  env ${ENVVAR}=$L python bench.py --config $C --steps $ST --warmup 3 --no-cpu-baseline --no-also --no-tick --no-host 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${ENVVAR}=$L config $C: %.2f M' % (d['value']/1e6), {k[2:]: round(v['us_per_frame'],1) for k,v in d['kernels'].items()})"
done; done; done
