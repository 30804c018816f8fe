#!/bin/bash
# the 1/1 and 1/8 shares of the bench frame (tools/rehearse_rank.py --strong), three runs each, for the in-tree library and
# for each variant of build_variants/ named on the command line
root=$(cd "$(dirname "$0")/.." && pwd)
for name in tree "$@"; do
  if [ "$name" = tree ]; then unset MIPT_LIB_PATH; else export MIPT_LIB_PATH=$root/build_variants/libmipt_$name.so; fi
  for p in 1 8; do for k in 1 2; do echo -n "$name: "; timeout -k 10 100 python3 $root/tools/rehearse_rank.py --strong --parts $p --part 0 --steps 6 2>&1 | grep -v amdgpu.ids || exit 1; done; done
done
