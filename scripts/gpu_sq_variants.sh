#!/bin/bash
# SQ counters (instruction counts, busy fractions) per kernel for several library variants in one session, 65536 streams.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
S=${PMC_STREAMS:-65536}
for V in ${VARIANTS:-default}; do
  if [ "$V" = default ]; then unset NNN_LIBRARY; else export NNN_LIBRARY=$R/nnnoiseless_amd/lib/variants/$V.so; fi
  cd /tmp && export TMPDIR=/tmp
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $C | cut -d' ' -f1)
    rm -rf $O/pmcs_$tag
    timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmcs_$tag -o pmc -- python $R/bench.py --streams $S --frames-per-step 24 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also --no-tick --no-host > $O/pmcs_$tag.json 2> $O/pmcs_$tag.err
  done
  cd $R
  python scripts/rocpd_pmc_sq.py $S $(find $O/pmcs_SQ_WAVES $O/pmcs_SQ_WAIT_ANY -name '*_results.db') > $O/pmc_sq_${S}streams_$V.json
  python -c "
import json; d=json.load(open('$O/pmc_sq_${S}streams_$V.json'))
for k,v in d['kernels'].items():
    c=v['counters']
    print('$V', k, 'valu_busy %.2f lds_busy %.2f' % (v.get('valu_busy',0), v.get('lds_busy',0)), 'valu %.3g salu %.3g lds %.3g vmem_rd %.3g vmem_wr %.3g wait_any/wave_cycles %.2f' % (c.get('SQ_INSTS_VALU',0), c.get('SQ_INSTS_SALU',0), c.get('SQ_INSTS_LDS',0), c.get('SQ_INSTS_VMEM_RD',0), c.get('SQ_INSTS_VMEM_WR',0), c.get('SQ_WAIT_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1)))
"
  find $O/pmcs_SQ_WAVES $O/pmcs_SQ_WAIT_ANY -name '*.db' -size +20M -delete
done
