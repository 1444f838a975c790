#!/bin/bash
# SQ counters over a short sampling run (separate --pmc passes, --kernel-trace only), per kernel.  MODE=bf16|split3|fp8|exact
# Output: gpurun_out/sq_a/, sq_b/, sq_c/ (summarised by profiles/summarize_sq.py)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
MODE=${MODE:-bf16}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/sq_a $OUT/sq_b $OUT/sq_c
ARGS="--mode $MODE --steps 1 --warmup 0 --sampler-steps 3 --no-cpu-baseline --no-kernel-events --no-other-modes --no-parity --detail-file /tmp/x.json"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq_a -o a -- python $R/bench.py $ARGS > $OUT/sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq_b -o b -- python $R/bench.py $ARGS > $OUT/sq_b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq_c -o c -- python $R/bench.py $ARGS > $OUT/sq_c.log 2>&1
find $OUT/sq_a $OUT/sq_b $OUT/sq_c -name "*counter_collection.csv" | xargs ls -la
tail -3 $OUT/sq_a.log $OUT/sq_b.log $OUT/sq_c.log
