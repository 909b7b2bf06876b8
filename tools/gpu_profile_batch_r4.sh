#!/bin/bash
# Round-4 evidence batch (tools/pmc_summary_r4.py turns gpurun_out/* into profiles/r4_*).  Two parts, one gpurun call each:
#   bash tools/gpu_profile_batch_r4.sh a   default bench line, rocprofv3 kernel stats of the bench and of the training step under
#                                          both policies, three PMC passes (SQ/GRBM set, FETCH_SIZE, WRITE_SIZE: separate passes as
#                                          MI355X_MICROARCH.md prescribes) for the two render modes
#   bash tools/gpu_profile_batch_r4.sh b   the same three PMC passes for the trainer under both policies + an LDS pass
#                                          (SQ_LDS_IDX_ACTIVE, conflicts) + the issue-side breakdowns (tools/pmc_issue_breakdown.sh)
# Every step prints a progress line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
part=${1:-a}
PMCSET="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
LDSSET="GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT"
if [ $part = a ]; then
  cd $R
  timeout -k 10 500 python bench.py > $O/r4_bench_default.json 2> $O/r4_bench_default.err || exit 1
  tail -c 300 $O/r4_bench_default.json; echo
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_r4_bench -o b --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --quick > $O/prof_r4_bench.log 2>&1 || exit 1
  echo "bench trace ok"
  # per-kernel times of a training step on ONE stream (NERF_TRAIN_OVERLAP=0: with the fine pass's weight-gradient launch on its
  # second stream the overlapped kernels' durations stretch and their sum says nothing); the step itself is timed both ways below
  export NERF_TRAIN_OVERLAP=0
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_r4_train -o t --output-format csv -- python3 $R/tools/train_bench.py 8 > $O/prof_r4_train.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_r4_mixed -o m --output-format csv -- python3 $R/tools/train_bench.py 8 4096 mixed > $O/prof_r4_mixed.log 2>&1 || exit 1
  grep train_step $O/prof_r4_train.log $O/prof_r4_mixed.log
  for ov in 0 1; do for pol in fp32 mixed; do
    NERF_TRAIN_OVERLAP=$ov timeout -k 10 120 python3 $R/tools/train_bench.py 30 4096 $pol 2>&1 | grep train_step | sed "s|^|[second stream: $ov] |"
  done; done | tee $O/r4_train_step_both_ways.txt
  unset NERF_TRAIN_OVERLAP
  for mode in f16x3 f16; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMCSET -d $O/pmc4_${mode}a -o a --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick --precision $mode > $O/pmc4_${mode}a.log 2>&1 || exit 1
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc4_${mode}b -o b --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick --precision $mode > $O/pmc4_${mode}b.log 2>&1 || exit 1
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc4_${mode}c -o c --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick --precision $mode > $O/pmc4_${mode}c.log 2>&1 || exit 1
    echo "pmc $mode ok"
  done
else
  cd /tmp && export TMPDIR=/tmp
  for pol in train mixed; do
    extra=""; [ $pol = mixed ] && extra="4096 mixed"
    timeout -k 10 250 rocprofv3 --kernel-trace --pmc $PMCSET -d $O/pmc4_${pol}a -o a --output-format csv -- python3 $R/tools/train_bench.py 2 $extra > $O/pmc4_${pol}a.log 2>&1 || exit 1
    timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc4_${pol}b -o b --output-format csv -- python3 $R/tools/train_bench.py 2 $extra > $O/pmc4_${pol}b.log 2>&1 || exit 1
    timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc4_${pol}c -o c --output-format csv -- python3 $R/tools/train_bench.py 2 $extra > $O/pmc4_${pol}c.log 2>&1 || exit 1
    timeout -k 10 250 rocprofv3 --kernel-trace --pmc $LDSSET -d $O/pmc4_${pol}d -o d --output-format csv -- python3 $R/tools/train_bench.py 2 $extra > $O/pmc4_${pol}d.log 2>&1 || exit 1
    echo "pmc $pol ok"
  done
  cd $R
  bash tools/pmc_issue_breakdown.sh mixed || exit 1
  bash tools/pmc_issue_breakdown.sh fp32 || exit 1
  bash tools/pmc_issue_breakdown.sh render || exit 1
fi
