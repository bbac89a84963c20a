#!/bin/bash
# Per-kernel times and SQ counters at scale (single-frame calls, so every launch is one frame of all streams).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for S in ${SWEEP:-16384 65536}; do
  timeout 600 python bench.py --streams $S --frames-per-step 1 --steps 60 --warmup 6 --no-cpu-baseline > $O/scale_$S.json 2> $O/scale_$S.err
  python - <<PY
import json
d=json.load(open("$O/scale_$S.json"))
k=d["kernels"]
print("S=$S value=%.3e ms/step=%.3f prof_ms=%.3f" % (d["value"], d["ms_per_step"], d["roofline"]["profiled_ms_per_frame"]), " ".join(f"{n[2:]}={v['avg_us']:.0f}" for n,v in k.items()))
PY
done
cd /tmp && export TMPDIR=/tmp
PS=${PMC_STREAMS:-65536}
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $C | cut -d' ' -f1)
  rm -rf $O/pmcs_$tag
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmcs_$tag -o pmc -- python $R/bench.py --streams $PS --frames-per-step 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmcs_$tag.json 2> $O/pmcs_$tag.err
  tail -1 $O/pmcs_$tag.err | cut -c1-150
done
