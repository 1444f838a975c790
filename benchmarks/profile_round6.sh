#!/bin/bash
# Round-6 evidence on a GPU box (run through gpurun from the repo root), everything under gpurun_out/r6prof/:
#   1. bench.py as the driver runs it (the one stdout line + the detail file), then once more with --detail (all opt-in blocks)
#   2. rocprofv3 --kernel-trace --stats of a short run per mode (fp32-parity headline, bf16, fp8) + the launch-gap table from the traces
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per mode (PMC passes combine with --kernel-trace only)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( time python $R/bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $OUT/bench_detail.json --kernel-table $OUT/kt.json > $OUT/bench_line.json 2> $OUT/bench_line.err ) 2> $OUT/bench_time.txt
wc -c $OUT/bench_line.json $OUT/bench_line.err; cat $OUT/bench_time.txt
( time python $R/bench.py --gpus 1 --steps 20 --warmup 5 --detail --detail-file $OUT/bench_detail_full.json > $OUT/bench_line_full.json 2> $OUT/bench_line_full.err ) 2> $OUT/bench_time_full.txt
cat $OUT/bench_time_full.txt
SHORT="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-other-modes --no-parity --detail-file /tmp/x.json"
TINY="--steps 1 --warmup 0 --sampler-steps 3 --no-cpu-baseline --no-kernel-events --no-other-modes --no-parity --detail-file /tmp/x.json"
for MODE in split3 bf16 fp8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$MODE -o st -- python $R/bench.py --mode $MODE $SHORT > $OUT/stats_$MODE.log 2>&1
  python $R/benchmarks/launch_gaps.py "$OUT/stats_$MODE/**/*kernel_trace.csv" > $OUT/launch_gaps_$MODE.txt 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$MODE -o f -- python $R/bench.py --mode $MODE $TINY > $OUT/pmc_fetch_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$MODE -o w -- python $R/bench.py --mode $MODE $TINY > $OUT/pmc_write_$MODE.log 2>&1
  F=$(find $OUT/pmc_fetch_$MODE -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write_$MODE -name "*counter_collection.csv" | head -1)
  python $R/profiles/summarize_pmc.py $F $W $OUT/pmc_summary_$MODE.json $OUT/pmc_traffic_$MODE.json > $OUT/pmc_summary_$MODE.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
du -sh $OUT; ls $OUT
