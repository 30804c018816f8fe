#!/bin/bash
# One SQ counter pass over a short bench.py for a VARIANT library (build_variants/libmipt_<name>.so, tools/tune_trace.sh):
#   bash tools/pmc_variant.sh <name>     -> gpurun_out/pmcv_<name>/  (dynamic instruction counts of two builds side by side)
name=$1
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmcv_$name
mkdir -p $out
export MIPT_LIB_PATH=$root/build_variants/libmipt_$name.so
[ -f "$MIPT_LIB_PATH" ] || { echo "no such variant: $name"; exit 1; }
cd /tmp && export TMPDIR=/tmp
# optional second argument: the counters of the pass (default: instruction counts)
counters=${2:-"SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"}
rm -rf $out/*
timeout -s KILL 150 rocprofv3 --pmc $counters \
  --output-format csv -d $out -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > $out/run.log 2>&1 || { echo "pass failed"; tail -3 $out/run.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_trace_wide" not in k: continue
    k = ("occluded" if "<true" in k else "closest") + ("_count" if ", true>" in k else "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    first = globals().setdefault("first", r["Counter_Name"])
    if r["Counter_Name"] == first: n[k] += 1
for k in sorted(agg):
    print(k, n[k], {a: f"{b / n[k]:.4g}" for a, b in sorted(agg[k].items())})
PY
