#!/usr/bin/env bash
# Teacher extraction on MI355X (train_fold1.sh's two `test.py` lines): Swin + Q2L frame features for one task, then the
# MS-TCT teacher's features + raw predictions that the student's KD losses read.
set -euo pipefail
FOLD=${KFOLD:-1}; RUN=${VERSION:-SwinL}; TASK=${TASK:-i}; DIM=${IN_DIM:-1536}; SIZE=${IM_SIZE:-384}; BACKBONE=${BACKBONE:-swin_L_${SIZE}_22k}
here="$(cd "$(dirname "$0")" && pwd)"
( cd "$here/../Spatial_transformer" && python test.py -e --img_size "$SIZE" --backbone "$BACKBONE" --hidden_dim "$DIM" --loss_type "$TASK" \
    --dataset_variant=cholect45-crossval --kfold "$FOLD" --batch=16 --version="$RUN" "$@" )
( cd "$here/../Temporal_mstct" && python test.py -e --loss_type "$TASK" --input_dim "$DIM" --dataset_variant=cholect45-crossval \
    --kfold="$FOLD" --version="${RUN}_MSTCT" --version1="$RUN" "$@" )
