#!/bin/bash
# Builds variants of libmipt.so that differ in the traversal kernels' scheduling constants (-DPT_TUNE_..., kernels_trace*.hip) into
# build_variants/ (CPU side, before the GPU call):    bash tools/tune_trace.sh build name1 "-DPT_TUNE_REFILL=8" name2 "..."
# and benches ONE built variant on the GPU box:    bash tools/tune_trace.sh bench <name>   (one variant per gpurun call)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/path-tracer_amd/csrc
out=$root/build_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
if [ "$1" = build ]; then
  shift
  mkdir -p $out
  ( cd $src && make -j8 all >/dev/null )
  while [ $# -ge 2 ]; do
    name=$1; defs=$2; shift 2
    /opt/rocm/bin/hipcc $FLAGS $defs -c $src/device/kernels_trace.hip -o $out/kt_$name.o
    /opt/rocm/bin/hipcc $FLAGS $defs -c $src/device/kernels_trace_wide.hip -o $out/kw_$name.o
    /opt/rocm/bin/hipcc $FLAGS $defs -c $src/api/api_device.hip -o $out/api_$name.o   # kTopNodes / kWideTopNodes are used on the host side too
    /opt/rocm/bin/hipcc $FLAGS $defs -c $src/device/kernels_integrate.hip -o $out/ki_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libmipt_$name.so $out/kt_$name.o $out/kw_$name.o $out/ki_$name.o $src/device/kernels_trace_bittrail.o $src/device/kernels_tiles.o \
        $out/api_$name.o $src/api/api_node.o $src/api/api_host.o $src/host/obj_loader.o $src/host/bvh_build.o $src/host/image_io.o $src/host/scene_cache.o -lz -ldl -lpthread
    rm -f $out/kt_$name.o $out/kw_$name.o $out/api_$name.o $out/ki_$name.o
    echo "built $name ($defs)"
  done
elif [ "$1" = benchall ]; then
  # several variants, each its own process under its own time limit; stops at the first failure
  shift
  for name in "$@"; do bash "$0" bench $name || exit 1; done
elif [ "$1" = bench ]; then
  # ONE variant per invocation: bash tools/tune_trace.sh bench <name>
  name=$2
  lib=$out/libmipt_$name.so
  [ -f "$lib" ] || { echo "no such variant: $name"; exit 1; }
  mkdir -p $root/gpurun_out
  MIPT_LIB_PATH=$lib timeout -k 10 200 python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $root/gpurun_out/tune_$name.json 2> $root/gpurun_out/tune_$name.err || { echo "$name failed"; exit 1; }
  python3 - "$name" $root/gpurun_out/tune_$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); pk = d["roofline"]["per_kernel"]
print(f'{sys.argv[1]:28s} {d["value"]:8.1f} Mrays/s  step {d["ms_per_step"]:7.2f} ms  closest {pk["k_trace_closest"]["avg_launch_ms"]:6.3f}  occluded {pk["k_trace_occluded"]["avg_launch_ms"]:6.3f}')
for k in ("k_trace_closest", "k_trace_occluded"):  # counters per ray (their meaning changes in -DPT_DIAG_UTIL builds)
    print(f'    {k:18s} counter0/ray {pk[k]["node_visits_per_ray"]:9.3f}  counter1/ray {pk[k]["tri_tests_per_ray"]:9.3f}')
PY
fi
