#!/bin/bash
# Benches variants of libmipt.so that differ in the shading kernel (build_variants/libmipt_<name>.so, MIPT_LIB_PATH), one
# process each under its own time limit, and prints step time and the shading kernel's average launch:
#   bash tools/shade_ab.sh base name1 name2 ...      ("base" = the in-tree library)
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/gpurun_out/shade_ab
for name in "$@"; do
  lib=$root/build_variants/libmipt_$name.so
  [ "$name" = base ] && lib=$root/path-tracer_amd/libmipt.so
  [ -f "$lib" ] || { echo "no such variant: $name"; exit 1; }
  MIPT_LIB_PATH=$lib timeout -k 10 200 python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $root/gpurun_out/shade_ab/$name.json 2> $root/gpurun_out/shade_ab/$name.err || { echo "$name failed"; tail -3 $root/gpurun_out/shade_ab/$name.err; exit 1; }
  python3 - "$name" $root/gpurun_out/shade_ab/$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); pk = d["roofline"]["per_kernel"]
print(f'{sys.argv[1]:12s} {d["value"]:8.1f} Mrays/s  step {d["ms_per_step"]:7.3f} ms  serialised {d["config"]["ms_per_step_serialised_with_events"]:7.3f}  shade {pk["k_shade_path"]["avg_launch_ms"]:6.3f} ms  closest {pk["k_trace_closest"]["avg_launch_ms"]:6.3f}  any-hit {pk["k_trace_occluded"]["avg_launch_ms"]:6.3f}', flush=True)
PY
done
