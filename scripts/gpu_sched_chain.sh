export HSA_ENABLE_IPC_MODE_LEGACY=0
one() { local tag="$1"; shift; env "$@" timeout 300 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-tick --no-host 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag: %.2f M' % (d['value']/1e6), {k[2:]: round(v['us_per_frame'],1) for k,v in d['kernels'].items()})"; }
for rep in 1 2 3; do
one seq NNN_SCHED=seq
one seq_chain1 NNN_SCHED=seq NNN_PITCH_CHAIN=1
one stages NNN_SCHED=stages
one stages_chain1 NNN_SCHED=stages NNN_PITCH_CHAIN=1
done 2>&1 | tee gpurun_out/r6_sched_chain.txt
