#!/usr/bin/env python3
"""Per-kernel issue-side breakdown from tools/pmc_issue_breakdown.sh's three passes.
Usage: python tools/pmc_issue_summary.py [mixed|fp32] -> fractions of SQ_WAVE_CYCLES (quad-cycles) and per-MFMA figures"""
import csv, collections, sys
pol = sys.argv[1] if len(sys.argv) > 1 else "mixed"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for ps in "abc":
    try:
        rows = csv.DictReader(open(f"gpurun_out/issue_{pol}{ps}/{ps}_counter_collection.csv"))
    except FileNotFoundError:
        continue
    for r in rows:
        k = r["Kernel_Name"].replace("void ", "").replace("nerf::", "").split("(")[0][:34]
        agg[k][ps + ":" + r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if "mlp_" not in k and "gemm_atb" not in k:
        continue
    wa, wb, wc = v["a:SQ_WAVE_CYCLES"], v["a:SQ_WAVE_CYCLES"], v["c:SQ_WAVE_CYCLES"]
    if wa == 0:
        continue
    mf = max(v["b:SQ_INSTS_MFMA"], 1.0)
    f = lambda n: v[n] / wa
    print(f"{k:34s} active {f('a:SQ_ACTIVE_INST_ANY'):.2f} (valu {f('a:SQ_ACTIVE_INST_VALU'):.2f} lds {f('a:SQ_ACTIVE_INST_LDS'):.2f} "
          f"vmem {f('a:SQ_ACTIVE_INST_VMEM'):.2f} flat {f('a:SQ_ACTIVE_INST_FLAT'):.2f} sca {f('a:SQ_ACTIVE_INST_SCA'):.2f} misc {f('a:SQ_ACTIVE_INST_MISC'):.2f}) "
          f"wait_any {f('b:SQ_WAIT_ANY'):.2f} wait_inst {f('b:SQ_WAIT_INST_ANY'):.2f} (lds {v['c:SQ_WAIT_INST_LDS'] / max(wc, 1):.2f}) | "
          f"cycles/mfma {4 * wa / mf:.1f} valu/mfma {v['b:SQ_INSTS_VALU'] / mf:.2f} coexec/mfma_busy {v['b:SQ_VALU_MFMA_COEXEC_CYCLES'] / max(v['b:SQ_VALU_MFMA_BUSY_CYCLES'], 1):.2f} "
          f"lds_level {v['c:SQ_INST_LEVEL_LDS'] / max(wc, 1):.2f} vmem_level {v['c:SQ_INST_LEVEL_VMEM'] / max(wc, 1):.2f} "
          f"lds_busy/wave {v['c:SQ_LDS_IDX_ACTIVE'] / max(wc, 1):.2f}")
