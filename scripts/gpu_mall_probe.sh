#!/bin/bash
# does the X / P round trip between k_fft_xp and k_synth stay inside the 256 MB Infinity Cache when a group is small enough?
# 8192 streams: a 4-frame group's spectra are 236 MB, a 24-frame group's 1.4 GB.  Kernel times per frame (HIP events) and HBM bytes per
# stream-frame (FETCH_SIZE / WRITE_SIZE passes) at both group lengths -> gpurun_out/r6_mall_probe.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out; mkdir -p $O
{
for S in 8192; do
  for F in 2 4 8 24; do
    ST=$((960 / F))
    timeout 300 python bench.py --streams $S --frames-per-step $F --steps $ST --warmup 4 --no-cpu-baseline --no-also --no-tick --no-host > $O/mall.json 2> $O/mall.err
    python - <<PY
import json
try:
    d = json.load(open('$O/mall.json'))
    print('$S streams, $F frames per call: %.2f M frames/s' % (d['value'] / 1e6), {k[2:]: round(v['us_per_frame'], 2) for k, v in d['kernels'].items()})
except Exception as e: print('parse fail', e); print(open('$O/mall.err').read()[-600:])
PY
  done
  for F in 4 24; do
    echo "-- HBM bytes per stream-frame by the counters, $F-frame launches"
    STREAMS=$S FPS=$F SUFFIX=_${F}fpl PMC_STEPS=$((96 / F)) bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -9
  done
done
} 2>&1 | tee $O/r6_mall_probe.txt
