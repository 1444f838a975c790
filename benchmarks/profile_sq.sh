#!/bin/bash
# SQ counters of the final build over a short sampling run (separate --pmc passes, --kernel-trace only): MFMA busy, VALU
# instruction counts, wave cycles, per kernel.  Output: gpurun_out/sq_a/, gpurun_out/sq_b/ (summarised by profiles/summarize_sq.py)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --sampler-steps 3 --no-cpu-baseline --no-kernel-events --no-other-configs"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/sq_a -o a -- python $R/bench.py $ARGS > $OUT/sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/sq_b -o b -- python $R/bench.py $ARGS > $OUT/sq_b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq_c -o c -- python $R/bench.py $ARGS > $OUT/sq_c.log 2>&1
ls -la $OUT/sq_a $OUT/sq_b $OUT/sq_c
tail -3 $OUT/sq_c.log
