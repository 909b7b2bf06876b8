#!/bin/bash
# The N > 1 bench when the library's own communicator never comes back (BENCH_C_CHECK_TEST_HANG=1 makes the worker thread of
# the cross-check block forever): the check runs AFTER the measurements, so the run must still exit 0 with its one JSON line,
# per_rank of length 2 and the time-out said in gather.check.  One-GPU rehearsal (gloo + the RCCL stand-in), self-launched.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
gcc -O2 -shared -fPIC -I/opt/rocm/include tests/stub_rccl.c -o /tmp/libstub_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt || exit 1
export NERF_RCCL_LIB=/tmp/libstub_rccl.so BENCH_BACKEND=gloo BENCH_LAUNCH_TIMEOUT=280 BENCH_C_CHECK_TEST_HANG=1 BENCH_C_CHECK_TIMEOUT=5
unset WORLD_SIZE RANK LOCAL_RANK
t0=$(date +%s)
timeout -k 10 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 --quick --no-train > gpurun_out/r4_bench_hang2.json 2> gpurun_out/r4_bench_hang2.err
rc=$?
echo "simulated hang: rc $rc after $(( $(date +%s) - t0 )) s, $(wc -l < gpurun_out/r4_bench_hang2.json) line(s) on stdout"
[ $rc -eq 0 ] || { tail -20 gpurun_out/r4_bench_hang2.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r4_bench_hang2.json').read())
print('n_gpus', d['n_gpus'], 'value', round(d['value']), 'per_rank', len(d['per_rank']), 'gather.check:', d['gather']['check'])
assert d['n_gpus'] == 2 and len(d['per_rank']) == 2 and 'did not return within the time limit' in d['gather']['check'] and d['value'] > 0"
