# same-box A/B of conv2 + conv3 of layer2's identity blocks in one launch (MT4_NO_EXPAND_FUSE=1: two launches)
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export MT4_NO_EXPAND_FUSE=1; else unset MT4_NO_EXPAND_FUSE; fi
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-temporal 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_expand_fuse=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches_per_step'])" >> gpurun_out/expand_ab.txt || exit 1
done
