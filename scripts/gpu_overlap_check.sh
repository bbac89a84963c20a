set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "boundary or repeat or two_frames or patterns" 2>&1 | tail -4
for O in "" "--no-overlap" "" "--no-overlap"; do
python bench.py --no-cpu-baseline --no-also --no-roofline --no-tick $O 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap[$O]: %.2f M  (%.3f ms per step)' % (d['value']/1e6, d['ms_per_step']))"
done
