#!/bin/bash
# Round-3 evidence batch, one GPU-box call (tools/pmc_summary_r3.py turns gpurun_out/* into profiles/r3_*):
# default bench line, rocprofv3 kernel stats of the bench and of the training step under both policies, three PMC passes
# each (SQ/GRBM set, FETCH_SIZE, WRITE_SIZE: separate passes as MI355X_MICROARCH.md prescribes) for the render modes and
# the trainer.  Every step prints a progress line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err || exit 1
tail -c 300 $O/r3_bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_r3_bench -o b --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --quick > $O/prof_r3_bench.log 2>&1 || exit 1
echo "bench trace ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_r3_train -o t --output-format csv -- python3 $R/tools/train_bench.py 8 > $O/prof_r3_train.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_r3_mixed -o m --output-format csv -- python3 $R/tools/train_bench.py 8 4096 mixed > $O/prof_r3_mixed.log 2>&1 || exit 1
grep train_step $O/prof_r3_train.log $O/prof_r3_mixed.log
PMCSET="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
for mode in f16x3 f16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMCSET -d $O/pmc3_${mode}a -o a --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick --precision $mode > $O/pmc3_${mode}a.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc3_${mode}b -o b --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick --precision $mode > $O/pmc3_${mode}b.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc3_${mode}c -o c --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick --precision $mode > $O/pmc3_${mode}c.log 2>&1 || exit 1
  echo "pmc $mode ok"
done
for pol in train mixed; do
  extra=""; [ $pol = mixed ] && extra="4096 mixed"
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $PMCSET -d $O/pmc3_${pol}a -o a --output-format csv -- python3 $R/tools/train_bench.py 2 $extra > $O/pmc3_${pol}a.log 2>&1 || exit 1
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc3_${pol}b -o b --output-format csv -- python3 $R/tools/train_bench.py 2 $extra > $O/pmc3_${pol}b.log 2>&1 || exit 1
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc3_${pol}c -o c --output-format csv -- python3 $R/tools/train_bench.py 2 $extra > $O/pmc3_${pol}c.log 2>&1 || exit 1
  echo "pmc $pol ok"
done
