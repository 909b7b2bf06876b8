#!/bin/bash
# `python3 bench.py --gpus 2` WITHOUT a launcher on a one-GPU box: bench.py starts its own two ranks (gloo carries torch's
# collectives, the library's ncclAllGather goes through the test-only stand-in tests/stub_rccl.c).  A rehearsal of the
# command line the scaling harness may use, not a scaling result.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=${1:-r4}
gcc -O2 -shared -fPIC -I/opt/rocm/include tests/stub_rccl.c -o /tmp/libstub_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt || exit 1
export NERF_RCCL_LIB=/tmp/libstub_rccl.so BENCH_BACKEND=gloo BENCH_LAUNCH_TIMEOUT=280
unset WORLD_SIZE RANK LOCAL_RANK
timeout -k 10 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 --quick --no-train \
    > gpurun_out/${TAG}_bench_selflaunch2.json 2> gpurun_out/${TAG}_bench_selflaunch2.err
rc=$?
echo "self-launched bench.py --gpus 2: rc $rc, $(wc -l < gpurun_out/${TAG}_bench_selflaunch2.json) line(s) on stdout"
[ $rc -eq 0 ] || { tail -20 gpurun_out/${TAG}_bench_selflaunch2.err; exit $rc; }
python3 -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench_selflaunch2.json').read())
print('n_gpus', d['n_gpus'], 'per_rank', len(d['per_rank']), 'gather.check:', d['gather']['check'])
assert d['n_gpus'] == 2 and len(d['per_rank']) == 2 and 'bit for bit' in d['gather']['check']"
