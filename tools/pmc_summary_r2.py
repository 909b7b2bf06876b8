#!/usr/bin/env python3
"""profiles/r2_pmc_summary.json from the PMC passes of tools/gpu_profile_batch_r2.sh (gpurun_out/pmc2_<tag>[abc]):
per kernel kind the per-launch averages -- clock (GRBM_GUI_ACTIVE / 8 / t), MFMA utilisation
(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles), wait fractions, LDS bank conflicts per busy cycle and HBM bytes
((2 * FETCH_SIZE + WRITE_SIZE) KiB: the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md) -- and refreshes
profiles/pmc_traffic.json (read by bench.py into roofline.traffic)."""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = {  # tag -> [(kernel-name substring, label)]
    "f16x3": [("mlp_f16x3_kernel", "mlp_f16x3_kernel (render, 3-pass split fp16)")],
    "f16": [("mlp_f16_2t_kernel", "mlp_f16_2t_kernel (render, single-pass fp16, two tiles per wave)")],
    "train": [("mlp_f16x3_stash_kernel", "mlp_f16x3_stash_kernel (training forward with stash + mask records)"),
              ("mlp_bwd_f16x3_dx_kernel", "mlp_bwd_f16x3_dx_kernel (fused data-gradient chain, fine pass, with encoding tiles)"),
              ("mlp_bwd_f16x3_kernel", "mlp_bwd_f16x3_kernel (fused data-gradient chain, coarse pass)"),
              ("gemm_atb_h_kernel<256>", "gemm_atb_h<256> (weight gradient, 3-pass split fp16)"),
              ("gemm_atb_h_kernel<128>", "gemm_atb_h<128> (weight gradient of layer 8)")],
    "mixed": [("mlp_f16_stash_kernel", "mixed_float16 policy: mlp_f16_stash_kernel (single-pass forward, fp16 stash)"),
              ("mlp_bwd_f16_dx_kernel", "mixed_float16 policy: mlp_bwd_f16_dx_kernel (single-pass backward chain, fine pass)"),
              ("mlp_bwd_f16_kernel", "mixed_float16 policy: mlp_bwd_f16_kernel (single-pass backward chain, coarse pass)"),
              ("gemm_atb_f16_kernel<256>", "mixed_float16 policy: gemm_atb_f16<256> (weight gradient on fp16 rows, one pass)"),
              ("gemm_atb_f16_kernel<128>", "mixed_float16 policy: gemm_atb_f16<128> (weight gradient of layer 8)"),
              ("head_wgrad_frag_kernel", "mixed_float16 policy: head_wgrad_frag (the two heads' weight gradients)"),
              ("reduce_grad_vec_kernel", "mixed_float16 policy: reduce_grad_vec (sum of the 256 row-slab partials of a wide layer)")],
}


def main():
    out = {}
    traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    for tag, kinds in KINDS.items():
        for sub, label in kinds:
            vals = collections.defaultdict(list)
            durs = []
            for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc2_{tag}[abc]", "*_counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    if sub in r["Kernel_Name"]:
                        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc2_{tag}a", "*_kernel_trace.csv")):
                for r in csv.DictReader(open(f)):
                    if sub in r["Kernel_Name"]:
                        durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
            if not durs or not vals:
                continue
            a = {k: sum(v) / len(v) for k, v in vals.items()}
            t = sum(durs) / len(durs)
            cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            o = {"launches": len(durs), "avg_launch_us": t * 1e6}
            if cyc:
                o["clock_ghz"] = cyc / t / 1e9
                o["mfma_util"] = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc
            if a.get("SQ_WAVE_CYCLES"):
                o["wait_any_frac"] = a.get("SQ_WAIT_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
                o["wait_inst_frac"] = a.get("SQ_WAIT_INST_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
            if a.get("SQ_BUSY_CYCLES"):
                o["lds_bank_conflict_per_busy_cycle"] = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_BUSY_CYCLES"]
            if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
                o["hbm_bytes_per_launch"] = (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0
                if tag in ("f16x3", "f16"):
                    traffic[tag] = o["hbm_bytes_per_launch"]
            out[label] = o
    json.dump(out, open(os.path.join(ROOT, "profiles", "r2_pmc_summary.json"), "w"), indent=1)
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
