#!/bin/bash
# Kernel trace of the headline bench + overlap analysis (printed; the database itself stays on the box unless small).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/trace
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-also --no-tick ${BENCH_ARGS:-} > $O/trace_bench.json 2> $O/trace.err
cd $R
DB=$(find $O/trace -name '*_results.db' | head -1)
python scripts/rocpd_trace_overlap.py $DB ${FPS:-48} ${TIMELINE:-} | tee $O/trace_overlap.txt
python scripts/rocpd_kernel_stats.py $DB > $O/trace_kernel_stats.md
find $O/trace -name '*.db' -size +20M -delete
