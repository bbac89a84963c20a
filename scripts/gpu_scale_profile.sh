#!/bin/bash
# SQ counters per kernel at scale (two --pmc passes, kernel trace only); frame groups of 16, so every launch covers 16 frames.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PS=${PMC_STREAMS:-65536}
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $C | cut -d' ' -f1)
  rm -rf $O/pmcs_$tag
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmcs_$tag -o pmc -- python $R/bench.py --streams $PS --frames-per-step 16 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also > $O/pmcs_$tag.json 2> $O/pmcs_$tag.err
  tail -1 $O/pmcs_$tag.err | cut -c1-150
done
cd $R
python scripts/rocpd_pmc_table.py $(find $O/pmcs_SQ_WAVES $O/pmcs_SQ_WAIT_ANY -name '*_results.db') | tee $O/pmc_sq_${PS}streams.md
find $O/pmcs_SQ_WAVES $O/pmcs_SQ_WAIT_ANY -name '*.db' -size +20M -delete
