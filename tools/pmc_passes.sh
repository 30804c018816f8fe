#!/bin/bash
# Runs the rocprofv3 PMC passes (one counter group per run: the hardware cannot multiplex) over a short bench.py
# and leaves the CSVs under gpurun_out/pmc_<tag>/<pass>/.  Use on the GPU box:  bash tools/pmc_passes.sh <tag>
# Each pass is its own process under a hard timeout; no tracing domains are combined with --pmc.
tag=${1:-run}
root=/root/repo
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }
  echo "pass $name ok"
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY &&
run_pass sq3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS &&
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT &&
run_pass mem1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE &&
run_pass fetch FETCH_SIZE TCC_REQ_sum &&
run_pass write WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
