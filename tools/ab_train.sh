#!/bin/bash
# A/B of the training step between library builds on one device: tools/ab_train.sh lib_dir... (under nerf_and_dietnerf_amd/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for d in "$@"; do
    for pol in fp32 mixed; do
      NERF_MI355_LIB=$R/nerf_and_dietnerf_amd/$d/libnerf_mi355.so timeout -k 10 120 python3 $R/tools/train_bench.py 30 4096 $pol 2>&1 | grep train_step | sed "s|^|[$d] |" || exit 1
    done
  done
done
