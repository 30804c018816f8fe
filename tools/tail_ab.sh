#!/bin/bash
# A/B of launch tails: small launches and the 1/1 and 1/8 shares of the bench frame, for the in-tree library and for
# each variant of build_variants/ named on the command line (tools/tune_trace.sh build ...).  One process per measurement.
root=$(cd "$(dirname "$0")/.." && pwd)
for name in tree "$@"; do
  if [ "$name" = tree ]; then unset MIPT_LIB_PATH; else export MIPT_LIB_PATH=$root/build_variants/libmipt_$name.so; fi
  echo "== $name"
  timeout -k 10 100 python3 $root/tools/small_launch.py 2>&1 | grep -v amdgpu.ids || exit 1
  for p in 1 8; do timeout -k 10 100 python3 $root/tools/rehearse_rank.py --strong --parts $p --part 0 2>&1 | grep -v amdgpu.ids || exit 1; done
done
