#!/bin/bash
# One GPU-box call: parity tests, the default bench line, an N=8 per-rank rehearsal, per-phase stamps,
# rocprofv3 kernel stats and the PMC passes (separate --pmc passes, as MI355X_MICROARCH.md prescribes).
set -o pipefail
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"; cat gpurun_out/bench_default.json
for k in 2 4 8; do timeout -k 10 200 python bench.py --rehearse-world $k --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rehearse world $k: est rays/s', d['value'], 'ms/step', d['ms_per_step'])"; done
if [ -f nerf_and_dietnerf_amd/lib/libnerf_st.so ]; then NERF_MI355_LIB=$PWD/nerf_and_dietnerf_amd/lib/libnerf_st.so timeout -k 10 100 python tools/stamps.py f16x3 2>&1 | tail -12; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r1f -o f16 --output-format csv -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/prof_r1f.log 2>&1; echo "prof rc $?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_train -o train --output-format csv -- python3 /root/repo/tools/train_bench.py 5 > /root/repo/gpurun_out/prof_train.log 2>&1; echo "prof train rc $?"; tail -1 /root/repo/gpurun_out/prof_train.log
if [ -z "$SKIP_PMC" ]; then
for mode in f16x3 fp32; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d /root/repo/gpurun_out/pmc_${mode}a -o a --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --precision $mode > /root/repo/gpurun_out/pmc_${mode}a.log 2>&1; echo "pmcA $mode rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /root/repo/gpurun_out/pmc_${mode}b -o b --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --precision $mode > /root/repo/gpurun_out/pmc_${mode}b.log 2>&1; echo "pmcB $mode rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /root/repo/gpurun_out/pmc_${mode}c -o c --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --precision $mode > /root/repo/gpurun_out/pmc_${mode}c.log 2>&1; echo "pmcC $mode rc $?"
done
PMCSET="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
timeout -k 10 250 rocprofv3 --kernel-trace --pmc $PMCSET -d /root/repo/gpurun_out/pmc_traina -o a --output-format csv -- python3 /root/repo/tools/train_bench.py 2 > /root/repo/gpurun_out/pmc_traina.log 2>&1; echo "pmcA train rc $?"
timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /root/repo/gpurun_out/pmc_trainb -o b --output-format csv -- python3 /root/repo/tools/train_bench.py 2 > /root/repo/gpurun_out/pmc_trainb.log 2>&1; echo "pmcB train rc $?"
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /root/repo/gpurun_out/pmc_trainc -o c --output-format csv -- python3 /root/repo/tools/train_bench.py 2 > /root/repo/gpurun_out/pmc_trainc.log 2>&1; echo "pmcC train rc $?"
fi
