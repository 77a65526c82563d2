#!/bin/bash
# same-box A/B of the chained conv3 + next-conv1 launches in the ResNet-50 bf16 extractor: MT4_CHAIN = 0 (off) / 3 (layer3) / 23 (layers 2 and 3),
# alternating, bench headline only
for rep in 1 2; do
  for c in 0 3 23; do
    MT4_CHAIN=$c python bench.py --no-cpu-baseline --no-temporal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MT4_CHAIN=$c', d['value'], 'frames/s', d['ms_per_step'], 'ms/step  conv', d['roofline']['conv_ms_per_step'], 'ms  launches', d['roofline']['launches_per_step'], ' frac', d['roofline']['frac'])"
  done
done
