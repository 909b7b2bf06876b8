set -o pipefail
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"; cat gpurun_out/bench_default.json
BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err; echo "gloo2 rc $?"; tail -c 600 gpurun_out/bench_gloo2.json; tail -3 gpurun_out/bench_gloo2.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r1f -o f16 --output-format csv -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/prof_r1f.log 2>&1; echo "prof rc $?"
for mode in f16x3 fp32; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d /root/repo/gpurun_out/pmc_${mode}a -o a --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision $mode > /root/repo/gpurun_out/pmc_${mode}a.log 2>&1; echo "pmcA $mode rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /root/repo/gpurun_out/pmc_${mode}b -o b --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision $mode > /root/repo/gpurun_out/pmc_${mode}b.log 2>&1; echo "pmcB $mode rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /root/repo/gpurun_out/pmc_${mode}c -o c --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision $mode > /root/repo/gpurun_out/pmc_${mode}c.log 2>&1; echo "pmcC $mode rc $?"
done
