#!/bin/bash
# Two ranks of bench.py on ONE GPU: gloo carries torch's collectives, the library's own ncclAllGather goes through the
# test-only stand-in (tests/stub_rccl.c) -- a rehearsal of the N > 1 code paths and of their output format, not a scaling result.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
gcc -O2 -shared -fPIC -I/opt/rocm/include tests/stub_rccl.c -o /tmp/libstub_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt || exit 1
export NERF_RCCL_LIB=/tmp/libstub_rccl.so BENCH_BACKEND=gloo
for extra in "" "--c-gather"; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 3 --warmup 1 --quick --no-train --no-cpu-baseline $extra 2> gpurun_out/r3_rehearsal$extra.err | tail -1 > "gpurun_out/r3_bench_gloo2_stub${extra}.json" || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r3_bench_gloo2_stub${extra}.json').read()); print('${extra:-torch-gather}', d['n_gpus'], d['gather'], [ (r['rank'], r['slab_rays'], round(r['render_ms_median'],2), round(r['gather_ms_median'],3)) for r in d['per_rank']])"
done
