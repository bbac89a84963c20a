#!/bin/bash
# rocprofv3 kernel stats of the headline workload with every launch sequential on one stream (NNN_SCHED=seq, no one-frame calls):
# the per-launch durations bench.py's roofline pass measures with HIP events (its launches are sequential too), for comparison
# with the pipelined run's stats, where overlapping kernels share the machine and each takes longer.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/prof_seq
cd /tmp && export TMPDIR=/tmp
NNN_SCHED=seq timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_seq" -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-tick > "$R/gpurun_out/prof_seq.log" 2>&1; echo "rocprof rc=$?"
cd "$R"
DB=$(find gpurun_out/prof_seq -name '*_results.db' | head -1)
python scripts/rocpd_kernel_stats.py "$DB" > gpurun_out/kernel_stats_seq.md; head -8 gpurun_out/kernel_stats_seq.md
grep -o '"avg_kernel_us": [0-9.]*' gpurun_out/prof_seq.log | head -2
find gpurun_out/prof_seq -name '*.db' -delete
