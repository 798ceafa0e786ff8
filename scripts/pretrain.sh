#!/bin/bash
# scripts/pretrain.sh <gpu> [train.py flags] -- same entry point as the reference's
# scripts/pretrain.sh:5-10 (README.md:69-83), e.g.
#   bash scripts/pretrain.sh 0 --moco --nce-k 16384 --synthetic 1000000,10000000
gpu=$1
ARGS=${@:2}
python train.py \
  --exp Pretrain \
  --model-path saved \
  --tb-path tensorboard \
  --gpu $gpu \
  $ARGS
