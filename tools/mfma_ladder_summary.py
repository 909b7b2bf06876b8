#!/usr/bin/env python3
"""profiles/r2_mfma_energy_ladder.json from one GPU-box call of
    tools/microbench/mfma_energy_ladder > gpurun_out/ladder.txt
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/ladder_pmc -o l --output-format csv -- tools/microbench/mfma_energy_ladder
    tools/microbench/mfma_shape_power > gpurun_out/shape_power.txt
: per rung the sustained rate of fp16 MFMA work (wall clock, random operands), the clock it holds (GRBM_GUI_ACTIVE / 8 / t)
and MFMA utilisation -- the evidence behind DESIGN.md's power-wall statement for mlp_f16x3_kernel."""
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"rungs": [], "shape_power": []}
for line in open(os.path.join(ROOT, "gpurun_out", "ladder.txt")):
    m = re.match(r"rung (\d) \((.*)\): ([\d.]+) ms, ([\d.]+) PFLOP/s", line)
    if m:
        out["rungs"].append({"rung": int(m.group(1)), "what": m.group(2), "pflops_fp16_mfma": float(m.group(4)),
                             "frac_of_2.5PF": float(m.group(4)) / 2.5})
sp = os.path.join(ROOT, "gpurun_out", "shape_power.txt")
if os.path.exists(sp):
    out["shape_power"] = [l.strip() for l in open(sp) if "TFLOP/s" in l]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "ladder_pmc", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"ladder(?:<|ILi)(\d)", r["Kernel_Name"])
        if m:
            vals[int(m.group(1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "ladder_pmc", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"ladder(?:<|ILi)(\d)", r["Kernel_Name"])
        if m:
            durs[int(m.group(1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
for rung, v in sorted(vals.items()):
    a = {k: sum(x) / len(x) for k, x in v.items()}
    t = sum(durs[rung]) / len(durs[rung])
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    out.setdefault("pmc", []).append({"rung": rung, "launches": len(durs[rung]), "avg_launch_ms": t * 1e3,
                                      "clock_ghz": cyc / t / 1e9,
                                      "mfma_util": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc})
json.dump(out, open(os.path.join(ROOT, "profiles", "r2_mfma_energy_ladder.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
