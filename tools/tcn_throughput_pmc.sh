#!/bin/bash
# HBM traffic of the temporal head's throughput mode (bf16, 64 videos of 256 frames per forward): two launches per DilatedResidualLayer against one
# (GPU box):  bash tools/tcn_throughput_pmc.sh   ->  gpurun_out/r04_tcn_throughput_traffic.json
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for layer in two fused; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_tp_${layer}_$c
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_tp_${layer}_$c -- python3 $R/tools/tenco_infer_prof.py --dtype bf16 --videos 64 --layer $layer --replays 3 > $O/pmc_tp_${layer}_$c.log 2>&1
  done
  python3 $R/tools/collect_tcn_traffic.py $O/pmc_tp_${layer}_FETCH_SIZE $O/pmc_tp_${layer}_WRITE_SIZE tenco4_bf16_64x256_$layer $O/r04_tcn_throughput_traffic.json
  rm -rf $O/pmc_tp_${layer}_FETCH_SIZE $O/pmc_tp_${layer}_WRITE_SIZE
done
