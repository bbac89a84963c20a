#!/bin/bash
# Round-3 evidence session (one gpurun call): the GPU parity suite (incl. the bench-style / real-audio / wav parity tests), the
# driver-style default bench line, same-box A/B of library variants (nnnoiseless_amd/lib/variants/*.so), rocprofv3 kernel stats
# (pipelined and sequential), PMC traffic and SQ counter passes at 65536 and 4096 streams.  Sections: PARTS="tests bench ab stats pmc rows"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
PARTS=${PARTS:-"tests bench ab stats pmc rows"}
TAG=${TAG:-r3_a}
for P in $PARTS; do
case $P in
tests)
  timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -${TAIL:-25} $O/${TAG}_pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  ;;
bench)
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/${TAG}_bench_default.json
  python - <<PY
import json
try:
    d = json.load(open('$O/${TAG}_bench_default.json'))
    print('value %.2f M  (timed %.2f s)' % (d['value'] / 1e6, d['timed_s']))
    print('kernels us/frame', {k[2:]: round(v['us_per_frame'], 1) for k, v in d['kernels'].items()})
    print('frac of roof', {k[2:]: round(v['frac_of_roof'], 3) for k, v in d['kernels'].items() if v.get('frac_of_roof')})
    for k, v in (d.get('also') or {}).items():
        print(k, '%.2f M' % (v['value'] / 1e6), 'timed %.2f s' % v['timed_s'], v['kernels_us_per_frame'], 'tick', (v.get('tick') or {}).get('ms_per_step'))
    print('tick', d.get('tick'))
    print('default semantics', d.get('default_semantics'))
    print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
    print('host', {k: round(v['value'] / 1e6, 1) for k, v in d['host_boundary'].items() if isinstance(v, dict)})
except Exception as e:
    print('parse fail', e); print(open('$O/${TAG}_bench_default.err').read()[-1500:])
PY
  ;;
ab)
  for V in ${VARIANTS:-default}; do
    if [ "$V" = default ]; then unset NNN_LIBRARY; else export NNN_LIBRARY=$R/nnnoiseless_amd/lib/variants/$V.so; fi
    for C in ${CONFIGS:-2 1}; do
      ST=12; [ "$C" = 1 ] && ST=120
      timeout 300 python bench.py --config $C --steps $ST --warmup 3 --no-cpu-baseline --no-also --no-tick --no-host > $O/ab.json 2>$O/ab.err
      python - <<PY
import json
try:
    d = json.load(open('$O/ab.json'))
    print('$V config $C: %.2f M' % (d['value'] / 1e6), {k[2:]: round(v['us_per_frame'], 1) for k, v in d['kernels'].items()})
except Exception as e: print('$V config $C: parse fail', e); print(open('$O/ab.err').read()[-800:])
PY
    done
  done
  unset NNN_LIBRARY
  ;;
stats)
  cd /tmp && export TMPDIR=/tmp
  for C in 2 1; do
    ST=10; [ "$C" = 1 ] && ST=40
    rm -rf $O/prof_c$C $O/prof_seq_c$C
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c$C -- python $R/bench.py --config $C --steps $ST --warmup 2 --no-cpu-baseline --no-also --no-tick --no-host --no-roofline > $O/prof_c$C.log 2>&1; echo "rocprof config $C rc=$?"
    NNN_SCHED=seq timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_seq_c$C -- python $R/bench.py --config $C --steps $ST --warmup 2 --no-cpu-baseline --no-also --no-tick --no-host > $O/prof_seq_c$C.log 2>&1; echo "rocprof seq config $C rc=$?"
    for K in prof_c$C prof_seq_c$C; do
      DB=$(find $O/$K -name '*_results.db' | head -1)
      python $R/scripts/rocpd_kernel_stats.py "$DB" > $O/${TAG}_kernel_stats_${K#prof_}.md; head -9 $O/${TAG}_kernel_stats_${K#prof_}.md
      find $O/$K -name '*.db' -delete
    done
    grep -o '"avg_kernel_us": [0-9.]*' $O/prof_seq_c$C.log | head -1
  done
  cd $R
  ;;
rows)
  for A in "--workload train --streams 4096" "--workload train --streams 16384" "--workload resample --streams 4096" "--pcm i16" "--pcm i16 --channels 2" "--config 1 --streams 16384"; do
    timeout 300 python bench.py $A --config 1 --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-tick --no-host --no-roofline > $O/${TAG}_row.json 2>$O/${TAG}_row.err
    python - <<PY
import json
try:
    d = json.load(open('$O/${TAG}_row.json')); print('$A:', '%.2f M %s' % (d['value'] / 1e6, d['unit']), (d.get('roofline') or {}).get('frac'))
    open('$O/${TAG}_rows.jsonl', 'a').write(json.dumps({'args': '$A', 'line': d}) + chr(10))
except Exception as e: print('$A: parse fail', e); print(open('$O/${TAG}_row.err').read()[-600:])
PY
  done
  ;;
pmc)
  for S in ${PMC_SIZES:-65536 4096}; do
    STREAMS=$S bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -8
    cd /tmp && export TMPDIR=/tmp
    for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
      tag=$(echo $C | cut -d' ' -f1)
      rm -rf $O/pmcs_$tag
      timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmcs_$tag -o pmc -- python $R/bench.py --streams $S --frames-per-step 24 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also --no-tick --no-host > $O/pmcs_$tag.json 2> $O/pmcs_$tag.err
      tail -1 $O/pmcs_$tag.err | cut -c1-150
    done
    cd $R
    python scripts/rocpd_pmc_sq.py $S $(find $O/pmcs_SQ_WAVES $O/pmcs_SQ_WAIT_ANY -name '*_results.db') > $O/pmc_sq_${S}streams.json
    python -c "
import json; d=json.load(open('$O/pmc_sq_${S}streams.json'))
for k,v in d['kernels'].items(): print('$S', k, 'valu_busy %.2f lds_busy %.2f conflicts %.2f' % (v.get('valu_busy',0), v.get('lds_busy',0), v.get('lds_conflict_share',0)), 'insts_valu %.3g' % v['counters'].get('SQ_INSTS_VALU',0))
"
    find $O/pmcs_SQ_WAVES $O/pmcs_SQ_WAIT_ANY -name '*.db' -size +20M -delete
  done
  ;;
esac
done
