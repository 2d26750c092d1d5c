#!/bin/bash
# Runs ON THE GPU BOX: per-kernel vector-ALU and matrix-core issue counters of one mnist.prms step
# (is a kernel issue-bound?  fp32 MFMA cycles and VALU cycles of a SIMD add up, tools/probe/mfma_valu.hip).
OUT=$GRAFT_REPO_ROOT/gpurun_out/valu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --sequential --steps 5 --warmup 2"
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o p -- $B "$@" > $OUT/a.log 2>&1 || echo FAILED a
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -o p -- $B "$@" > $OUT/b.log 2>&1 || echo FAILED b
python $GRAFT_REPO_ROOT/tools/condense_profiles.py $OUT
cat $OUT/a/counters.csv $OUT/b/counters.csv | grep -v "^kernel" | sort | head -150
tail -3 $OUT/a.log
