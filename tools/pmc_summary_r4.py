#!/usr/bin/env python3
"""profiles/r4_pmc_summary.json from the PMC passes of tools/gpu_profile_batch_r4.sh (gpurun_out/pmc4_<tag>[abcd]):
per kernel kind the per-launch averages -- clock (GRBM_GUI_ACTIVE / 8 / t), MFMA utilisation
(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles), wait fractions, LDS bank conflicts per busy cycle and HBM bytes
((2 * FETCH_SIZE + WRITE_SIZE) KiB: the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).

Round 4: the trainer's pass d (SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_WAIT_INST_LDS) gives the LDS array's busy share of a
launch next to the conflict count; profiles/pmc_traffic.json records the csrc hash the counters were collected on (bench.py
marks roofline.traffic stale when the running sources differ) and the issue-side evidence of the bound.

Round-3 hygiene: the GRBM_GUI_ACTIVE quotient reads high for launches under ~0.3 ms (MI355X_MICROARCH.md; round 2 printed
3.57 GHz for a 14 us kernel on a 2.4 GHz part), so `clock_ghz` and the `mfma_util` that divides by the same cycle count
are given only for launches of at least 0.3 ms; shorter ones carry "clock_ghz": null and a note.

Also sums the HBM bytes of ONE training step under each policy (bytes per launch x launches per step, steps counted by
opt_tick_kernel launches) and writes them, with the render kernels' bytes per launch, to profiles/pmc_traffic.json, which
bench.py reads into roofline.traffic (labelled offline)."""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIN_S_FOR_CLOCK = 0.3e-3
KINDS = {  # tag -> [(kernel-name substring, label)]
    "f16x3": [("mlp_f16x3_kernel", "mlp_f16x3_kernel (render, 3-pass split fp16)")],
    "f16": [("mlp_f16_2t_kernel", "mlp_f16_2t_kernel (render, single-pass fp16, two tiles per wave)")],
    "train": [("mlp_f16x3_stash_kernel", "mlp_f16x3_stash_kernel (training forward with stash + mask records)"),
              ("mlp_bwd_f16x3_dx_kernel", "mlp_bwd_f16x3_dx_kernel (fused data-gradient chain, fine pass, with encoding tiles)"),
              ("mlp_bwd_f16x3_kernel", "mlp_bwd_f16x3_kernel (fused data-gradient chain, coarse pass)"),
              ("gemm_atb_p_kernel<256", "gemm_atb_p<256> (weight gradients of the eight 256-wide layers of a pass, ONE batched launch; pair16 gradient operand)"),
              ("gemm_atb_p_kernel<128", "gemm_atb_p<128, sigma> (weight gradient of layer 8 + the sigma head's as a by-product)"),
              ("head_wgrad_frag_kernel", "head_wgrad_frag (the rgb head's weight gradient)"),
              ("reduce_grad_vec_kernel", "reduce_grad_vec (fixed-order sum of the row-slab partials; batched per pass)")],
    "mixed": [("mlp_f16_stash_kernel", "mixed_float16 policy: mlp_f16_stash_kernel (single-pass forward, fp16 stash)"),
              ("mlp_bwd_f16_dx_kernel", "mixed_float16 policy: mlp_bwd_f16_dx_kernel (single-pass backward chain, fine pass)"),
              ("mlp_bwd_f16_kernel", "mixed_float16 policy: mlp_bwd_f16_kernel (single-pass backward chain, coarse pass)"),
              ("gemm_atb_f16_kernel<256", "mixed_float16 policy: gemm_atb_f16<256> (weight gradients of the eight 256-wide layers of a pass, ONE batched launch)"),
              ("gemm_atb_f16_kernel<128", "mixed_float16 policy: gemm_atb_f16<128, sigma> (weight gradient of layer 8 + the sigma head's as a by-product)"),
              ("head_wgrad_frag_kernel", "mixed_float16 policy: head_wgrad_frag (the rgb head's weight gradient)"),
              ("reduce_grad_vec_kernel", "mixed_float16 policy: reduce_grad_vec (fixed-order sum of the row-slab partials; batched per pass)")],
}


def rows_of(pattern):
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True):
        yield from csv.DictReader(open(f))


def step_bytes(tag):
    """HBM bytes of one training step: every dispatch's (2 FETCH + WRITE) KiB summed, divided by the steps of the run."""
    fetch = sum(float(r["Counter_Value"]) for r in rows_of(f"pmc4_{tag}b/**/*_counter_collection.csv") if r["Counter_Name"] == "FETCH_SIZE")
    write = sum(float(r["Counter_Value"]) for r in rows_of(f"pmc4_{tag}c/**/*_counter_collection.csv") if r["Counter_Name"] == "WRITE_SIZE")
    steps = sum(1 for r in rows_of(f"pmc4_{tag}a/**/*_kernel_trace.csv") if "opt_tick_kernel" in r["Kernel_Name"])
    if not steps or not fetch:
        return None
    return (2.0 * fetch + write) * 1024.0 / steps


def main():
    out = {}
    traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    for tag, kinds in KINDS.items():
        for sub, label in kinds:
            vals = collections.defaultdict(list)
            durs = []
            for r in rows_of(f"pmc4_{tag}[abc]/**/*_counter_collection.csv"):
                if sub in r["Kernel_Name"]:
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            lds = collections.defaultdict(list)       # pass d has its own GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES: keep it apart
            for r in rows_of(f"pmc4_{tag}d/**/*_counter_collection.csv"):
                if sub in r["Kernel_Name"]:
                    lds[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for r in rows_of(f"pmc4_{tag}a/**/*_kernel_trace.csv"):
                if sub in r["Kernel_Name"]:
                    durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
            if not durs or not vals:
                continue
            a = {k: sum(v) / len(v) for k, v in vals.items()}
            t = sum(durs) / len(durs)
            cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            o = {"launches": len(durs), "avg_launch_us": t * 1e6}
            if cyc and t >= MIN_S_FOR_CLOCK:
                o["clock_ghz"] = cyc / t / 1e9
                o["mfma_util"] = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc
            elif cyc:
                o["clock_ghz"] = None
                o["clock_note"] = "launch shorter than 0.3 ms: the GRBM_GUI_ACTIVE quotient is not a clock there"
            if a.get("SQ_WAVE_CYCLES"):
                o["wait_any_frac"] = a.get("SQ_WAIT_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
                o["wait_inst_frac"] = a.get("SQ_WAIT_INST_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
            if a.get("SQ_BUSY_CYCLES"):
                o["lds_bank_conflict_per_sq_busy_cycle_r3_quotient"] = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_BUSY_CYCLES"]
            if lds.get("SQ_LDS_IDX_ACTIVE") and lds.get("GRBM_GUI_ACTIVE"):
                d = {k: sum(v) / len(v) for k, v in lds.items()}
                # SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT are LDS-array cycles summed over the chip's 256 CUs (checked against the
                # instruction mix of gemm_atb_p: 12 ds_read_b128 + 8 ds_write_b64 per wave and step); GRBM_GUI_ACTIVE / 8 is the
                # launch's length in cycles.  SQ_BUSY_CYCLES sums over ~31 units, NOT 256: round 3's "conflicts per busy cycle"
                # (kept below under its old name for comparison) overstates the per-CU share by 8x.
                dc = d["GRBM_GUI_ACTIVE"] / 8.0
                o["lds_array_busy_frac_per_cu"] = d["SQ_LDS_IDX_ACTIVE"] / 256.0 / dc
                o["lds_bank_conflict_frac_of_cycles_per_cu"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / 256.0 / dc
                o["sq_busy_cycles_units"] = d["SQ_BUSY_CYCLES"] / dc
            if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
                o["hbm_bytes_per_launch"] = (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0
                o["hbm_tb_per_s"] = o["hbm_bytes_per_launch"] / t / 1e12
                if tag in ("f16x3", "f16"):
                    traffic[tag] = o["hbm_bytes_per_launch"]
            out[label] = o
    for tag, key in (("train", "train_step_f32"), ("mixed", "train_step_mixed")):
        b = step_bytes(tag)
        if b:
            traffic[key] = b
            out[f"{key}: HBM bytes of one 4096-ray training step (all kernels)"] = b
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nerf_and_dietnerf_amd", "csrc")
    for name in sorted(os.listdir(d)):          # = bench.py::csrc_sha16
        if name.endswith((".hip", ".h")):
            h.update(name.encode() + b"\0" + open(os.path.join(d, name), "rb").read())
    traffic["csrc_sha16"] = h.hexdigest()[:16]
    issue = os.path.join(ROOT, "profiles", "r4_issue_breakdown.txt")
    traffic["issue_side"] = ("profiles/r4_issue_breakdown.txt: the fused stash / backward kernels issue vector instructions in "
                             "0.3-0.4 of their wave cycles and are parked or issue-stalled in 0.45-0.5; the weight-gradient GEMMs "
                             "stream at 0.8-0.95 of the achievable read rate (profiles/r4_pmc_summary.json)") if os.path.exists(issue) else None
    json.dump(out, open(os.path.join(ROOT, "profiles", "r4_pmc_summary.json"), "w"), indent=1)
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
