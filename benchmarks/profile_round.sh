#!/bin/bash
# Collects the per-round evidence on a GPU box (run through gpurun from the repo root):
#   1. bench.py (full contract line, with cpu_baseline and the other modes)  -> gpurun_out/bench_full.json
#   2. rocprofv3 --kernel-trace --stats of a short bench run (csv)           -> gpurun_out/prof_stats/
#   3. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes      -> gpurun_out/pmc_fetch/, gpurun_out/pmc_write/
# (PMC passes never combine with trace domains other than --kernel-trace.)   MODE=bf16|split3|exact (default bf16)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
MODE=${MODE:-bf16}
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --mode $MODE --steps 3 --warmup 1 --kernel-table $OUT/kt_full.json > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -c 6000 $OUT/bench_full.json
rm -rf $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o st -- python $R/bench.py --mode $MODE --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-other-configs --no-other-modes > $OUT/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --mode $MODE --steps 1 --warmup 0 --sampler-steps 3 --no-cpu-baseline --no-kernel-events --no-other-configs --no-other-modes > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --mode $MODE --steps 1 --warmup 0 --sampler-steps 3 --no-cpu-baseline --no-kernel-events --no-other-configs --no-other-modes > $OUT/pmc_write.log 2>&1
find $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write -name "*.csv" | xargs ls -la
