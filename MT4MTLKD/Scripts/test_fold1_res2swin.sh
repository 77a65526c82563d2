#!/usr/bin/env bash
# The reference's Scripts/test_fold1_res2swin.sh (lines 7-10): the Res18 -> SwinL direction.  Its first line hands Swin flags to
# Spatial_cnn/test.py, which ignores them (parse_known_args) and extracts with the ResNet student of run_${VERSION} -- kept as is
# (SURVEY 8(b)); the TCN then evaluates on those features.
set -euo pipefail
GPU=${GPU:-0}; KFOLD=${KFOLD:-1}; VERSION=${VERSION:-Res182SwinL}; NGPU=${NGPU:-1}
here="$(cd "$(dirname "$0")" && pwd)"
launch() { if [ "$NGPU" -gt 1 ]; then python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 "$@"; else python "$@"; fi; }
cd "$here/../Spatial_cnn"
launch test.py -e --rates 1 1 1 --temp 4 --soft_type KL_T --img_size 384 --backbone swin_L_384_22k --hidden_dim 1536 --spatialKD --loss_type all --dataset_variant=cholect45-crossval --kfold ${KFOLD} --epochs=100 --batch=8 -l 1e-2 5e-3 1e-3 -w 9 18 38 --version=${VERSION} --gpu ${GPU} --val_interval 5 "$@"
cd "$here/../Temporal_tenco"
python run.py -e --seed 19991111 --mask --input_dim 512 --loss_type all --fpn --dataset_variant=cholect45-crossval --kfold=${KFOLD} --epochs=1000 --batch=31 -l 1e-2 5e-3 1e-2 -w 9 18 200 --version=${VERSION}_TCN --version1=${VERSION} --gpu ${GPU} --val_interval 20 --test_ckpt ./__checkpoint__/run_${VERSION}_TCN/rendezvous_l8_cholectcholect45-crossval_k${KFOLD}_batchnorm_lowres_latest.pth "$@"
