#!/bin/bash
# Round-2 evidence batch, one GPU-box call (tools/pmc_summary_r2.py turns gpurun_out/* into profiles/r2_*):
# default bench line, rocprofv3 kernel stats of the bench and of the training step, three PMC passes each
# (SQ/GRBM set, FETCH_SIZE, WRITE_SIZE: separate passes as MI355X_MICROARCH.md prescribes) for the render modes and the trainer.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err || exit 1
tail -c 400 gpurun_out/r2_bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
R=/root/repo/gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/prof_r2_bench -o b --output-format csv -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --quick > $R/prof_r2_bench.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/prof_r2_train -o t --output-format csv -- python3 /root/repo/tools/train_bench.py 8 > $R/prof_r2_train.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/prof_r2_mixed -o m --output-format csv -- python3 /root/repo/tools/train_bench.py 8 4096 mixed > $R/prof_r2_mixed.log 2>&1 || exit 1
grep train_step $R/prof_r2_train.log $R/prof_r2_mixed.log
PMCSET="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
for mode in f16x3 f16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMCSET -d $R/pmc2_${mode}a -o a --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick --precision $mode > $R/pmc2_${mode}a.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/pmc2_${mode}b -o b --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick --precision $mode > $R/pmc2_${mode}b.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/pmc2_${mode}c -o c --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick --precision $mode > $R/pmc2_${mode}c.log 2>&1 || exit 1
  echo "pmc $mode ok"
done
timeout -k 10 250 rocprofv3 --kernel-trace --pmc $PMCSET -d $R/pmc2_traina -o a --output-format csv -- python3 /root/repo/tools/train_bench.py 2 > $R/pmc2_traina.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/pmc2_trainb -o b --output-format csv -- python3 /root/repo/tools/train_bench.py 2 > $R/pmc2_trainb.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/pmc2_trainc -o c --output-format csv -- python3 /root/repo/tools/train_bench.py 2 > $R/pmc2_trainc.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc $PMCSET -d $R/pmc2_mixeda -o a --output-format csv -- python3 /root/repo/tools/train_bench.py 2 4096 mixed > $R/pmc2_mixeda.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/pmc2_mixedb -o b --output-format csv -- python3 /root/repo/tools/train_bench.py 2 4096 mixed > $R/pmc2_mixedb.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/pmc2_mixedc -o c --output-format csv -- python3 /root/repo/tools/train_bench.py 2 4096 mixed > $R/pmc2_mixedc.log 2>&1 || exit 1
echo "pmc train ok"
