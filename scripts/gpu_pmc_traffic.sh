#!/bin/bash
# HBM traffic per kernel launch: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (never with other trace domains).
# Calls of 24 frames: every launch covers one full frame group (the production group length) of every stream.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
S=${STREAMS:-4096}
F=${FPS:-24}      # frames per call = frames per launch (24: the production group length)
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export NNN_PMC_CALIB=1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmct_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmct_$C -o pmc -- python $R/bench.py --streams $S --frames-per-step $F --steps ${PMC_STEPS:-6} --warmup 2 --no-cpu-baseline --no-roofline --no-also --no-tick > $O/pmct_$C.json 2> $O/pmct_$C.err
  tail -1 $O/pmct_$C.err | cut -c1-120
done
cd $R
python scripts/rocpd_pmc_traffic.py $(find $O/pmct_FETCH_SIZE -name '*_results.db' | head -1) $(find $O/pmct_WRITE_SIZE -name '*_results.db' | head -1) $S $F > $O/pmc_traffic_${S}streams${SUFFIX:-}.json
python -c "
import json; d=json.load(open('$O/pmc_traffic_${S}streams${SUFFIX:-}.json'))
print(d['calibration'])
tot = 0
for k,v in d['kernels'].items():
    print(k, round(v['hbm_bytes_per_stream_frame'])); tot += v['hbm_bytes_per_stream_frame']
print('sum', round(tot))
"
find $O/pmct_FETCH_SIZE $O/pmct_WRITE_SIZE -name '*.db' -size +20M -delete
