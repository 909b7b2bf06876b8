#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_*/ *_counter_collection.csv) for the MLP kernel
into profiles/: per-launch averages, the HBM traffic figure bench.py reports, MFMA utilisation.

HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts 64 B per 128-B request for wide streaming reads, so it is doubled; WRITE_SIZE is exact.
(Our reads are 4-B z values, broadcast ray rows and L2-resident weights, i.e. not the calibrated
16-B/lane pattern: the doubled figure is an upper estimate.)"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, dtype, kernel_substr):
    vals = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}[abc]", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs = []
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}a", "*_kernel_trace.csv"))):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    avg = {k: sum(v) / len(v) for k, v in vals.items()}
    out = {"kernel": kernel_substr, "launches_sampled": {k: len(v) for k, v in vals.items()}, "avg_per_launch": avg}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        out["hbm_bytes_per_launch"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    if durs and "GRBM_GUI_ACTIVE" in avg:
        t = sum(durs) / len(durs)
        out["avg_launch_s"] = t
        out["clock_ghz"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / t / 1e9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            # busy cycles are summed over the 1024 SIMDs; one wave per SIMD
            out["mfma_util"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (avg["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_WAVE_CYCLES" in avg and "SQ_WAIT_ANY" in avg:
            out["wait_any_frac"] = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", f"r1_{dtype}_pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(tf)) if os.path.exists(tf) else {}
    if "hbm_bytes_per_launch" in out:
        traffic[dtype] = out["hbm_bytes_per_launch"]
        json.dump(traffic, open(tf, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "mlp_fp32")
