#!/usr/bin/env bash
# Student evaluation on MI355X: frame features (ResNet-18 student) -> per-video TCN.  Flag recipe follows the
# reference's Scripts/test_fold1.sh; NGPU>1 shards whole videos over ranks with torchrun.
set -euo pipefail
FOLD=${KFOLD:-1}; RUN=${VERSION:-SwinL2Res18}; NGPU=${NGPU:-1}; DTYPE=${DTYPE:-fp32}
here="$(cd "$(dirname "$0")" && pwd)"
launch() { if [ "$NGPU" -gt 1 ]; then python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 "$@"; else python "$@"; fi; }
( cd "$here/../Spatial_cnn" && launch test.py -e --rates 1 1 1 --temp 4 --soft_type KL_T --network resnet18 --student_dim 512 \
    --loss_type all --dataset_variant=cholect45-crossval --kfold "$FOLD" --batch=8 --version="$RUN" --dtype "$DTYPE" "$@" )
( cd "$here/../Temporal_tenco" && python run.py -e --seed 19991111 --mask --input_dim 512 --loss_type all --fpn \
    --dataset_variant=cholect45-crossval --kfold="$FOLD" --version="${RUN}_TCN" --version1="$RUN" \
    --test_ckpt "./__checkpoint__/run_${RUN}_TCN/rendezvous_l8_cholectcholect45-crossval_k${FOLD}_batchnorm_lowres_latest.pth" "$@" )
