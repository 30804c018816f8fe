#!/bin/bash
# Two SQ passes that tell what the traversal kernels' wavefronts wait for: average latency of vector-memory and LDS
# instructions (SQ_INST_LEVEL_* / SQ_INSTS_*), time in waits, LDS conflict and FIFO stalls.  -> gpurun_out/pmcl_<tag>/
tag=${1:-run}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmcl_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }
}
run_pass lat SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES &&
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_LDS &&
python3 - $out <<'PY'
import csv, glob, sys, collections
for p in ("lat", "lds"):
    f = glob.glob(sys.argv[1] + f"/{p}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    first = None
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_trace_wide" not in k or ", true>" in k: continue
        k = "occluded" if "<true" in k else "closest"
        first = first or r["Counter_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first: n[k] += 1
    for k in sorted(agg):
        print(p, k, n[k], {a: f"{b / n[k]:.4g}" for a, b in sorted(agg[k].items())})
PY
