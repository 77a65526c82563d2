cd $GRAFT_REPO_ROOT
for st in 1 2 3 4; do
  timeout -k 10 120 python bench.py --streams $st --steps 10 --warmup 3 --no-cpu-baseline --no-temporal 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$st', d['value'], d['ms_per_step'], d['config'])" >> gpurun_out/streams_ab.txt || exit 1
done
