#!/usr/bin/env python3
"""Summarises gpurun_out/pmc_<tag>/ (tools/pmc_passes.sh) per kernel: prints a table and writes
profiles/<round>/pmc_<tag>.json; with --traffic also writes profiles/traffic.json (HBM bytes per launch of the two
traversal kernels, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE) for bench.py."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}", "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if "k_trace_wide_both" in name:  # path mode's one launch per bounce (extension + shadow rays)
            name = "k_trace_wide_both" + ("_count" if "<true>" in name else "")
        elif "k_trace" in name:
            base = "k_trace_wide" if "k_trace_wide" in name else "k_trace"
            name = base + ("_occluded" if "<true" in name else "_closest") + ("_count" if ", true>" in name else "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name][r["Counter_Name"]] += 1
out = {}
for k in sorted(agg):
    v, n = agg[k], calls[k]
    d = {c: v[c] for c in v}
    d["launches"] = max(n.values())
    if v.get("SQ_ACTIVE_INST_VALU"):
        d["lanes_per_valu_inst"] = v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"]
        d["valu_insts_per_wave"] = v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1)
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            d["frac_" + c] = v[c] / max(v["SQ_WAVE_CYCLES"], 1)
    if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
        d["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        d["l1_hit_rate_est"] = 1.0 - v.get("TCP_TCC_READ_REQ_sum", 0) / v["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if "FETCH_SIZE" in v:
        # rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B: doubled
        d["hbm_read_bytes_per_launch"] = 2.0 * v["FETCH_SIZE"] * 1024.0 / n["FETCH_SIZE"]
    if "WRITE_SIZE" in v:
        d["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024.0 / n["WRITE_SIZE"]
    out[k] = d
os.makedirs(os.path.join(root, "profiles", rnd), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", rnd, f"pmc_{tag}.json"), "w"), indent=1, sort_keys=True)
for k, d in out.items():
    if not k.startswith("k_"):
        continue
    print(k, d["launches"])
    for c in ("lanes_per_valu_inst", "valu_insts_per_wave", "frac_SQ_WAIT_ANY", "frac_SQ_WAIT_INST_ANY", "frac_SQ_ACTIVE_INST_VALU",
              "l2_hit_rate", "l1_hit_rate_est", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "SQ_INSTS_VMEM_RD",
              "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "TA_BUSY_avr", "SQ_WAVES"):
        if c in d:
            print(f"    {c}: {d[c]:.4g}")
if "--traffic" in sys.argv:
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tools"))
    import bench  # kernel_source_hash(): bench.py reports these counters only for the sources they were recorded for
    import valu_cost
    costs = valu_cost.kernel_costs()
    t = {"kernel_source_hash": bench.kernel_source_hash(), "recorded_as": f"profiles/{rnd}/pmc_{tag}.json"}
    for k in ("k_trace_closest", "k_trace_occluded", "k_trace_wide_closest", "k_trace_wide_occluded"):
        if k in out and "hbm_read_bytes_per_launch" in out[k]:
            t[k] = {"hbm_bytes_per_launch": out[k]["hbm_read_bytes_per_launch"] + out[k].get("hbm_write_bytes_per_launch", 0.0),
                    "source": f"profiles/{rnd}/pmc_{tag}.json: 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes of "
                              "bench.py --steps 2 --warmup 1, averaged over that kernel's launches"}
            t[k]["l2_hit_rate"] = out[k].get("l2_hit_rate")
            if out[k].get("SQ_INSTS_VALU") and out[k].get("GRBM_GUI_ACTIVE"):
                # Vector-ALU issue-slot occupancy.  SQ_ACTIVE_INST_VALU ticks once per VALU instruction on this part
                # whatever the instruction costs (profiles/README.md, "unit of SQ_ACTIVE_INST_VALU"), so it is not a
                # busy time; instead: executed instructions x the mean issue cost of the kernel's instructions
                # (tools/valu_cost.py: the compiled kernel's mix priced with tools/micro/issue_rate.hip's cycles at
                # 4 wavefronts per SIMD) over SIMDs x kernel cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024
                # SIMDs on the chip).  lo / hi: every instruction at 2.8 / at 4.5 cycles.
                o = out[k]
                mean = costs.get(k, {}).get("mean_issue_cycles", 4.5)
                per_simd = o["SQ_INSTS_VALU"] / 1024.0
                cyc = o["GRBM_GUI_ACTIVE"] / 8.0
                t[k]["valu_issue_occupancy"] = per_simd * mean / cyc
                t[k]["valu_issue_occupancy_range"] = [per_simd * 2.8 / cyc, per_simd * 4.5 / cyc]
                t[k]["valu_mean_issue_cycles"] = mean
                if o.get("lanes_per_valu_inst"):
                    t[k]["lanes_per_valu_inst"] = o["lanes_per_valu_inst"]
                    t[k]["useful_lane_frac"] = t[k]["valu_issue_occupancy"] * o["lanes_per_valu_inst"] / 64.0
    if "k_shade_path" in out and "hbm_read_bytes_per_launch" in out["k_shade_path"]:
        o = out["k_shade_path"]
        t["k_shade_path"] = {"hbm_read_bytes_per_launch": o["hbm_read_bytes_per_launch"],
                             "hbm_write_bytes_per_launch": o.get("hbm_write_bytes_per_launch", 0.0),
                             "hbm_bytes_per_launch": o["hbm_read_bytes_per_launch"] + o.get("hbm_write_bytes_per_launch", 0.0),
                             "launches": o["launches"], "l2_hit_rate": o.get("l2_hit_rate"),
                             "source": "as above; averaged over ALL launches of k_shade_path in the run (a finals-only launch is "
                                       "among them only where a frame issues one: more than one light, or a cap above five bounces)"}
    json.dump(t, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
