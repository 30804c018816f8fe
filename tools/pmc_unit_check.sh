#!/bin/bash
# What does SQ_ACTIVE_INST_VALU count on gfx950?  Runs the issue-rate micro-benchmark (known instruction counts and
# mixes: 20,000 iterations x 32 instructions per wavefront, v_fma_f32 = 2 cycles, v_max3_f32 / v_pk_fma_f32 = 4) under
# rocprofv3 --pmc, program directly after `--`, and leaves the per-dispatch counters under gpurun_out/pmc_unit/.
out=/root/repo/gpurun_out/pmc_unit
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for which in 0 1 2; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d $out/run$which -- /root/repo/tools/micro/issue_rate $which > $out/run$which.log 2>&1 || { echo "pass $which failed"; tail -3 $out/run$which.log; exit 1; }
done
# second question: how do the per-class counters (SQ_INSTS_VALU_ADD/MUL/FMA_F32 ...) count packed and other instructions?
for which in 0 1 2; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 \
    --output-format csv -d $out/mix$which -- /root/repo/tools/micro/issue_rate $which > $out/mix$which.log 2>&1 || { echo "mix pass $which failed"; tail -3 $out/mix$which.log; exit 1; }
done
echo ok
