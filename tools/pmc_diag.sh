#!/bin/bash
# Diagnostic PMC passes (LDS conflicts, memory-pipeline latency and stalls) over a short bench.py; optional variant library:
#   bash tools/pmc_diag.sh <tag> [build_variants/libmipt_<name>.so]
tag=${1:-diag}
root=/root/repo
[ -n "$2" ] && export MIPT_LIB_PATH=$root/$2
out=$root/gpurun_out/pmcd_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }
  echo "pass $name ok"
}
run_pass d1 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD &&
run_pass d2 TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_avr GRBM_GUI_ACTIVE &&
run_pass d3 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
