#!/usr/bin/env python3
"""per-kernel sums of the counters under gpurun_out/pmcd_<tag>/ (tools/pmc_diag.sh), as a table"""
import collections, csv, glob, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tag in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(os.path.join(root, "gpurun_out", f"pmcd_{tag}", "*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].split("::")[-1]
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[name][r["Counter_Name"]] += 1
    for k in sorted(agg):
        if not ("k_trace_wide<" in k and ", false>" in k): continue
        print(tag, k)
        for c in sorted(agg[k]): print(f"    {c:44s} {agg[k][c] / n[k][c]:14.5g} per launch ({n[k][c]} launches)")
