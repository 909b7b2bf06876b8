#!/bin/bash
# Issue-side breakdown of the trainer kernels (three PMC passes, 8 SQ counters each): where a wave's cycles go.
# Usage (GPU box): bash tools/pmc_issue_breakdown.sh [mixed|fp32|render]   -> gpurun_out/issue_<pol>{a,b,c}/
# (render: bench.py --quick in the headline mode and the single-pass mode instead of the training step)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; pol=${1:-mixed}
CMD="python3 $R/tools/train_bench.py 2 4096 $pol"
[ "$pol" = render ] && CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --quick"
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES"
timeout -k 10 250 rocprofv3 --kernel-trace --pmc $A -d $O/issue_${pol}a -o a --output-format csv -- $CMD > $O/issue_${pol}a.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc $B -d $O/issue_${pol}b -o b --output-format csv -- $CMD > $O/issue_${pol}b.log 2>&1 || exit 1
C="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
timeout -k 10 250 rocprofv3 --kernel-trace --pmc $C -d $O/issue_${pol}c -o c --output-format csv -- $CMD > $O/issue_${pol}c.log 2>&1 || exit 1
echo "issue breakdown $pol ok"
