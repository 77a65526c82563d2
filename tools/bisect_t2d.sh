#!/bin/bash
# which earlier test changes the outcome of the resnet18 fixture test?
T='test_train_step_vs_reference_autograd and cnn_train_resnet18'
for pre in "nothing_matches_this" "batchnorm" "conv2d_weight" "pool_backward" "kd_mix"; do
  echo "== predecessor: $pre"
  timeout -k 10 200 python -m pytest tests/test_gpu_train2d.py -q -k "($pre) or ($T)" 2>&1 | grep -E "^E  +Assert|passed|failed"
done
