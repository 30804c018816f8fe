#!/bin/bash
# resident wavefronts of the traversal kernels for one library variant: bash tools/pmc_occ.sh <tag> [lib]
tag=${1:-occ}
root=/root/repo
[ -n "$2" ] && export MIPT_LIB_PATH=$root/$2
out=$root/gpurun_out/pmcd_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/occ -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/occ.log 2>&1 || { echo "pass failed"; tail -3 $out/occ.log; exit 1; }
echo ok $tag
