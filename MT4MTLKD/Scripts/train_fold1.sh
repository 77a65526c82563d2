#!/usr/bin/env bash
# The four-stage recipe of the reference's Scripts/train_fold1.sh (lines 12-28), flag for flag, on MI355X:
#   Swin + Q2L teacher (one task) -> its frame features -> MS-TCT teacher -> its features + raw predictions ->
#   ResNet-18 student (hard + soft + feature distillation) -> its frame features -> TCN student.
# NGPU>1 runs the data-parallel stages under torchrun (one process per GPU, RCCL).  All four trainings and every extraction /
# evaluation stage run here; SKIP_TEACHER_TRAIN=1 starts from existing teacher checkpoints instead.
set -euo pipefail
GPU=${GPU:-0}; KFOLD=${KFOLD:-1}; NGPU=${NGPU:-1}
VERSION=${VERSION:-SwinL}; TASK=${TASK:-i}; IN_DIM=${IN_DIM:-1536}; IM_SIZE=${IM_SIZE:-384}; BACKBONE=${BACKBONE:-swin_L_${IM_SIZE}_22k}
VERSION_S=${VERSION_S:-SwinL2Res18}
SKIP_TEACHER_TRAIN=${SKIP_TEACHER_TRAIN:-0}   # 1: teacher checkpoints already exist (e.g. the authors' released weights)
here="$(cd "$(dirname "$0")" && pwd)"
launch() { if [ "$NGPU" -gt 1 ]; then python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 "$@"; else python "$@"; fi; }

# Teacher training
cd "$here/../Spatial_transformer"
if [ "$SKIP_TEACHER_TRAIN" != 1 ]; then
launch run.py -t -e --img_size ${IM_SIZE} --backbone ${BACKBONE} --hidden_dim ${IN_DIM} --loss_type ${TASK} --dataset_variant=cholect45-crossval --kfold ${KFOLD} --epochs=100 --batch=16 -l 1e-2 5e-3 1e-5 --version=${VERSION} --gpu ${GPU} --val_interval 5 "$@"
fi
python test.py -e --img_size ${IM_SIZE} --backbone ${BACKBONE} --hidden_dim ${IN_DIM} --loss_type ${TASK} --dataset_variant=cholect45-crossval --kfold ${KFOLD} --epochs=100 --batch=16 -l 1e-2 5e-3 1e-5 --version=${VERSION} --gpu ${GPU} --val_interval 5 "$@"
cd "$here/../Temporal_mstct"
if [ "$SKIP_TEACHER_TRAIN" != 1 ]; then
launch run.py -t -e --loss_type ${TASK} --input_dim ${IN_DIM} --dataset_variant=cholect45-crossval --kfold=${KFOLD} --epochs=2000 --batch=31 -l 1e-2 5e-3 1e-2 -w 9 18 500 --decay_rate 0.999 --version=${VERSION}_MSTCT --version1=${VERSION} --gpu ${GPU} --val_interval 20 "$@"
fi
python test.py -e --loss_type ${TASK} --input_dim ${IN_DIM} --dataset_variant=cholect45-crossval --kfold=${KFOLD} --epochs=2000 --batch=31 -l 1e-2 5e-3 1e-2 -w 9 18 500 --decay_rate 0.999 --version=${VERSION}_MSTCT --version1=${VERSION} --gpu ${GPU} --val_interval 20 "$@"

# Student training
cd "$here/../Spatial_cnn"
launch run.py -t -e --rates 1 1 1 --temp 4 --network resnet18 --teacher_feat_version ${VERSION} --teacher_pred_version ${VERSION}_MSTCT --student_dim 512 --loss_type all --dataset_variant=cholect45-crossval --kfold ${KFOLD} --epochs=200 --batch=8 -l 1e-2 5e-3 1e-3 --version=${VERSION_S} --gpu ${GPU} --val_interval 10 "$@"
launch test.py -e --rates 1 1 1 --temp 4 --network resnet18 --student_dim 512 --loss_type all --dataset_variant=cholect45-crossval --kfold ${KFOLD} --epochs=200 --batch=8 -l 1e-2 5e-3 1e-3 --version=${VERSION_S} --gpu ${GPU} --val_interval 10 "$@"
cd "$here/../Temporal_tenco"
launch run.py -t -e --seed 19991111 --mask --input_dim 512 --loss_type all --fpn --dataset_variant=cholect45-crossval --kfold=${KFOLD} --epochs=1000 --batch=31 -l 1e-2 5e-3 1e-2 -w 9 18 200 --version=${VERSION_S}_TCN --version1=${VERSION_S} --gpu ${GPU} --val_interval 20 "$@"
