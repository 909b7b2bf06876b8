#!/bin/bash
# kernel traces of two training steps for several library builds: tools/trace_variants.sh policy lib_dir...
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
pol=$1; shift
for d in "$@"; do
  export NERF_MI355_LIB=$R/nerf_and_dietnerf_amd/$d/libnerf_mi355.so
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr -o v_$d -- python3 $R/tools/train_bench.py 2 4096 $pol > $R/gpurun_out/tr_$d.log 2>&1 || exit 1
done
