#!/bin/bash
# A/B of the environment switches that steer the headline step (GPU box, same process order each time): bash tools/headline_knobs.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo -n "$1: "; env $1 timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-temporal 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; }
run MT4_DUMMY=0
run MT4_NT_MIN_MB=100
run MT4_NT_MIN_MB=400
run MT4_NO_NT=1
run MT4_CHAIN=3
run MT4_PATCH3X3=2
run MT4_DUMMY=1
