#!/bin/bash
# A/B of libnerf_mi355 builds on the fp16 single-pass frame time (bench.py --quick): usage tools/ab_f16.sh lib_dir...
cd "$(dirname "$0")/.."
for d in "$@"; do
  NERF_MI355_LIB=$PWD/nerf_and_dietnerf_amd/$d/libnerf_mi355.so timeout -k 10 200 python bench.py --quick --no-cpu-baseline --no-train > gpurun_out/ab_$d.json 2>/dev/null || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/ab_$d.json').read().strip().splitlines()[-1]); m=d['fp16_single_pass_mode']; print('$d', 'f16 frac %.4f  %.3f ms/frame | f16x3 frac %.4f' % (m['roofline']['frac'], m['ms_per_step'], d['roofline']['frac']))"
done
