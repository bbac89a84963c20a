R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for V in default prev default prev; do
  if [ "$V" = default ]; then unset NNN_LIBRARY; else export NNN_LIBRARY=$R/nnnoiseless_amd/lib/variants/$V.so; fi
  for C in 1 2; do
  python bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-host --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('tick',{})
print('$V config $C tick', {k: (round(v,4) if isinstance(v,float) else v) for k,v in t.items() if not isinstance(v,dict)})"
  done
done
