#!/bin/bash
# Copies the summaries of a round-5 evidence session (scripts/gpu_r3_session.sh, TAG=$1) from gpurun_out/ (scratch) into profiles/.
set -eu
T=${1:-r5_a}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out; P=$R/profiles
cp $O/${T}_bench_default.json $P/r5_bench_default.json
cp $O/${T}_kernel_stats_c2.md $P/r5_kernel_stats_65536streams_48fps.md
cp $O/${T}_kernel_stats_seq_c2.md $P/r5_kernel_stats_65536streams_48fps_sequential.md
cp $O/${T}_kernel_stats_c1.md $P/r5_kernel_stats_4096streams_48fps.md
cp $O/${T}_kernel_stats_seq_c1.md $P/r5_kernel_stats_4096streams_48fps_sequential.md
for S in 65536 4096; do cp $O/pmc_traffic_${S}streams.json $O/pmc_sq_${S}streams.json $P/; done
for f in $O/parity_*.json; do cp $f $P/r5_$(basename $f); done
[ -f $O/${T}_rows.jsonl ] && cp $O/${T}_rows.jsonl $P/r5_rows_bench_lines.jsonl
ls $P | grep "r5_\|pmc_" | wc -l
