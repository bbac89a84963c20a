#!/bin/bash
# Copies the summaries of a round-6 evidence session (scripts/gpu_r3_session.sh, TAG=$1) from gpurun_out/ (scratch) into profiles/.
set -eu
T=${1:-r6_b}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out; P=$R/profiles
cp $O/${T}_bench_default.json $P/r6_bench_default.json
cp $O/${T}_kernel_stats_c2.md $P/r6_kernel_stats_65536streams_48fps.md
cp $O/${T}_kernel_stats_seq_c2.md $P/r6_kernel_stats_65536streams_48fps_sequential.md
cp $O/${T}_kernel_stats_c1.md $P/r6_kernel_stats_4096streams_48fps.md
cp $O/${T}_kernel_stats_seq_c1.md $P/r6_kernel_stats_4096streams_48fps_sequential.md
for S in 65536 4096; do
  [ -f $O/pmc_traffic_${S}streams.json ] && cp $O/pmc_traffic_${S}streams.json $P/r6_pmc_traffic_${S}streams.json
  [ -f $O/pmc_sq_${S}streams.json ] && cp $O/pmc_sq_${S}streams.json $P/r6_pmc_sq_${S}streams.json
done
for f in $O/parity_*.json; do cp $f $P/r6_$(basename $f); done
[ -f $O/${T}_rows.jsonl ] && cp $O/${T}_rows.jsonl $P/r6_rows_bench_lines.jsonl
[ -f $O/${T}_session.txt ] && cp $O/${T}_session.txt $P/r6_session.txt
[ -f $O/r6_k_pitch_phase_stamps.txt ] && cp $O/r6_k_pitch_phase_stamps.txt $P/
ls $P | grep "r6_" | wc -l
