#!/bin/bash
# Round-5 evidence on a GPU box (run through gpurun from the repo root), everything under gpurun_out/r5prof/:
#   1. bench.py as the driver runs it (all modes, five BASELINE configs, job block, power block, cpu baseline) + per-mode kernel tables
#   2. rocprofv3 --kernel-trace --stats of a short run, fp32-parity mode (headline) and bf16 mode
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, both modes (PMC passes combine with --kernel-trace only)
#   4. the fused global-attention block: one launch against two / three, in-kernel time line, batch sweep
#   5. what the board says about power / clocks / throttling while the path runs (rocm-smi, amd-smi)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 --kernel-table $OUT/kt.json > $OUT/bench_line.json 2> $OUT/bench_line.err
tail -c 300 $OUT/bench_line.err
SHORT="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-other-configs --no-other-modes --no-power --no-parity --no-small-batch --no-job"
TINY="--steps 1 --warmup 0 --sampler-steps 3 --no-cpu-baseline --no-kernel-events --no-other-configs --no-other-modes --no-power --no-parity --no-small-batch --no-job"
for MODE in split3 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$MODE -o st -- python $R/bench.py --mode $MODE $SHORT > $OUT/stats_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$MODE -o f -- python $R/bench.py --mode $MODE $TINY > $OUT/pmc_fetch_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$MODE -o w -- python $R/bench.py --mode $MODE $TINY > $OUT/pmc_write_$MODE.log 2>&1
done
# (benchmarks/attn_block_bench.py: removed in round 6 with the rendezvous form it compared; its log is profiles/r05_attn_block_bench.log)
# power management while the headline pass runs: a background loop of passes, the tools beside it
( python $R/bench.py --mode split3 --steps 400 --warmup 2 --no-cpu-baseline --no-kernel-events --no-other-configs --no-other-modes --no-power --no-parity --no-small-batch --no-job > /dev/null 2>&1 ) &
LOAD=$!
sleep 25
{ echo "== rocm-smi under load"; rocm-smi --showpower --showclocks --showmaxpower --showperflevel --showtemp 2>&1 | grep -v "^$";
  echo "== amd-smi metric under load"; amd-smi metric --power --clock --temperature 2>&1 | head -80;
  echo "== amd-smi metric --help (which throttle / violation fields exist)"; amd-smi metric --help 2>&1 | head -60;
  echo "== amd-smi metric --violation (twice, 10 s apart: accumulator deltas under load)"; amd-smi metric --violation 2>&1 | head -8; sleep 10; amd-smi metric --violation 2>&1 | head -8; } > $OUT/power_tools.log 2>&1
kill $LOAD 2>/dev/null; wait $LOAD
find $OUT -name "*kernel_trace.csv" -path "*stats_*" -size +20M -delete
du -sh $OUT; find $OUT -name "*.csv" | xargs ls -la | head -30
