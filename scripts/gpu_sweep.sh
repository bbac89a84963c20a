#!/bin/bash
# Stream-count sweep + counter passes (separate rocprofv3 runs per counter set).  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for S in ${SWEEP:-4096 16384 65536}; do
  timeout 600 python bench.py --streams $S --frames-per-step 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/sweep_$S.json 2> $O/sweep_$S.err
  python - <<PY
import json
d=json.load(open("$O/sweep_$S.json"))
k=d["kernels"]
print("S=$S value=%.3e ms/step=%.3f prof_ms=%.3f" % (d["value"], d["ms_per_step"], d["roofline"]["profiled_ms_per_frame"]), " ".join(f"{n[2:]}={v['avg_us']:.0f}" for n,v in k.items()))
PY
done
for F in ${FPS_SWEEP:-8 32}; do
  for S in ${FPS_STREAMS:-4096 16384}; do
    timeout 600 python bench.py --streams $S --frames-per-step $F --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/fps_${S}_$F.json 2> $O/fps_${S}_$F.err
    python -c "import json; d=json.load(open('$O/fps_${S}_$F.json')); print('S=$S frames/step=$F value=%.3e ms/step=%.3f' % (d['value'], d['ms_per_step']))"
  done
done
cd /tmp && export TMPDIR=/tmp
PS=${PMC_STREAMS:-16384}
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$tag -o pmc -- python $R/bench.py --streams $PS --frames-per-step 1 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_$tag.json 2> $O/pmc_$tag.err
  tail -1 $O/pmc_$tag.err | cut -c1-150
done
