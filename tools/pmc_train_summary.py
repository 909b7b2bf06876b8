#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of the training step (gpurun_out/pmc_train[abc]) per GEMM kernel kind into
profiles/r1_train_pmc_summary.json: MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / GRBM_GUI_ACTIVE/8),
wait fractions, LDS bank-conflict share, HBM bytes per launch ((2*FETCH_SIZE + WRITE_SIZE) KiB, the gfx950
correction of MI355X_MICROARCH.md) -- averaged over all launches of the kind (coarse and fine passes, all layers)."""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = (("mlp_f16x3_stash_kernel", "fused forward with activation stash (3-pass split fp16)"),
         ("gemm_abt_h_kernel", "gemm_abt_h data gradient (3-pass split fp16)"),
         ("gemm_atb_h_kernel", "gemm_atb_h weight gradient (3-pass split fp16)"),
         ("gemm_abt_kernel<2, 2, 2, 2, 0>", "gemm_abt forward, exact fp32 (xyz-only network / NERF_TRAIN_FORWARD=gemm)"),
         ("gemm_abt_kernel<2, 2, 2, 2, 2>", "gemm_abt data gradient, exact fp32 (NERF_TRAIN_DGRAD=fp32)"),
         ("gemm_atb_kernel", "gemm_atb weight gradient, exact fp32 (NERF_TRAIN_WGRAD=fp32)"))


def main():
    vals = {k: collections.defaultdict(list) for _, k in KINDS}
    durs = {k: [] for _, k in KINDS}
    for tag in "abc":
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_train{tag}", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                for sub, kind in KINDS:
                    if sub in r["Kernel_Name"]:
                        vals[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_traina", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            for sub, kind in KINDS:
                if sub in r["Kernel_Name"]:
                    durs[kind].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    out = {}
    for _, kind in KINDS:
        a = {c: sum(v) / len(v) for c, v in vals[kind].items()}
        if not a:
            continue
        t = sum(durs[kind]) / max(1, len(durs[kind]))
        cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        o = {"launches": len(durs[kind]), "avg_launch_us": t * 1e6}
        if cyc:
            o["clock_ghz"] = cyc / t / 1e9 if t else None
            o["mfma_util"] = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc   # 1024 SIMDs
        if a.get("SQ_WAVE_CYCLES"):
            o["wait_any_frac"] = a.get("SQ_WAIT_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
            o["wait_inst_frac"] = a.get("SQ_WAIT_INST_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
        if a.get("SQ_BUSY_CYCLES"):
            o["lds_bank_conflict_per_busy_cycle"] = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_BUSY_CYCLES"]
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            o["hbm_bytes_per_launch"] = (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0
        out[kind] = o
    with open(os.path.join(ROOT, "profiles", "r1_train_pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
