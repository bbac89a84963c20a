#!/bin/bash
# One GPU-box session: smoke, parity tests, bench, rocprofv3 kernel trace.  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
nproc > $O/host.txt; lscpu | grep "Model name" >> $O/host.txt; rocm-smi --showproductname 2>/dev/null | head -8 >> $O/host.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 | tee $O/pytest_gpu.txt
echo "== bench"; timeout 900 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cat $O/bench.json
if [ "${SKIP_PROF:-0}" != "1" ]; then
  echo "== rocprofv3"; cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/prof_bench.json 2> $O/prof.err
  tail -2 $O/prof.err; cat $O/prof_bench.json
  find $O/prof -name "*kernel_stats*" | head -3
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
fi
