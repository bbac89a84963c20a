#!/bin/bash
# Stream-count sweep + HBM counter passes (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 runs).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for S in 1024 4096 16384 32768 65536; do
  timeout 600 python bench.py --streams $S --steps 60 --warmup 10 --no-cpu-baseline > $O/sweep_$S.json 2> $O/sweep_$S.err
  python - <<PY
import json
d=json.load(open("$O/sweep_$S.json"))
k=d["kernels"]
print("S=$S value=%.3e ms/step=%.3f prof_ms=%.3f" % (d["value"], d["ms_per_step"], d["roofline"]["profiled_ms_per_step"]), " ".join(f"{n[2:]}={v['avg_us']:.0f}" for n,v in k.items()))
PY
done
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o pmc -- python $R/bench.py --streams 4096 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_$C.json 2> $O/pmc_$C.err
  tail -1 $O/pmc_$C.err; ls $O/pmc_$C | head
done
