#!/bin/bash
# kernel trace of one training step of the xyz-only network under both policies
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for pol in mixed fp32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trx -o x_$pol -- python3 $R/tools/train_bench.py 2 4096 $pol xyz > $R/gpurun_out/trx_$pol.log 2>&1 || exit 1
  echo "== xyz-only $pol"; python3 $R/tools/step_trace.py $(find $R/gpurun_out/trx -name "x_${pol}_kernel_trace.csv" | head -1)
done
