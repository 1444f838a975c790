#!/bin/bash
# Round-4 evidence on a GPU box (run through gpurun from the repo root), everything under gpurun_out/r4prof/:
#   1. bench.py, the contract line the driver runs (--steps 20 --warmup 5; all modes, other configs, cpu baseline) + per-mode kernel tables
#   2. rocprofv3 --kernel-trace --stats of a short run, fp32-parity mode (headline) and bf16 mode
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, both modes (PMC passes combine with --kernel-trace only)
#   4. SQ counters of the headline mode (three passes)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 --kernel-table $OUT/kt.json > $OUT/bench_line.json 2> $OUT/bench_line.err
tail -c 400 $OUT/bench_line.err
SHORT="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-other-configs --no-other-modes --no-power --no-parity --no-small-batch"
TINY="--steps 1 --warmup 0 --sampler-steps 3 --no-cpu-baseline --no-kernel-events --no-other-configs --no-other-modes --no-power --no-parity --no-small-batch"
for MODE in split3 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$MODE -o st -- python $R/bench.py --mode $MODE $SHORT > $OUT/stats_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$MODE -o f -- python $R/bench.py --mode $MODE $TINY > $OUT/pmc_fetch_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$MODE -o w -- python $R/bench.py --mode $MODE $TINY > $OUT/pmc_write_$MODE.log 2>&1
done
MODE=split3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $OUT/sq_a -o a -- python $R/bench.py --mode $MODE $TINY > $OUT/sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq_b -o b -- python $R/bench.py --mode $MODE $TINY > $OUT/sq_b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq_c -o c -- python $R/bench.py --mode $MODE $TINY > $OUT/sq_c.log 2>&1
# 5. round-4 micro-benchmarks: projections without a norm (loader waves), per-workgroup time lines of the heavy kernels, bf16 tiled kernel
python $R/benchmarks/x3r_bench.py > $OUT/x3r_bench.log 2>&1
python $R/benchmarks/wg_timeline.py > $OUT/wg_timeline.log 2>&1
python $R/benchmarks/tiled_bf16_bench.py > $OUT/tiled_bf16_bench.log 2>&1
python $R/benchmarks/attn_x3_bench.py 30 > $OUT/attn_x3_bench.log 2>&1
# (removed in round 6 with hipGraph replay; bench.py --small-batch) python $R/benchmarks/small_batch.py configs/config_oxford_flowers.json 1 sample_dpmpp_2m 50 configs/config_oxford_flowers.json 4 sample_dpmpp_2m 50 > $OUT/small_batch_bf16.log 2>&1
# (removed) KDIFF_GEMM=split3 python $R/benchmarks/small_batch.py configs/config_oxford_flowers.json 1 sample_dpmpp_2m 50 configs/config_oxford_flowers.json 4 sample_dpmpp_2m 50 > $OUT/small_batch_split3.log 2>&1
# 6. socket power / shader clock while the path runs (is it power-limited?)
for m in split3 bf16 exact; do $R/benchmarks/power_watch.sh $m $([ $m = exact ] && echo 15 || echo 60) > $OUT/power_$m.log 2>&1; done
# keep what travels back small: the per-dispatch traces of the stats runs are large, the stats tables are not
find $OUT -name "*kernel_trace.csv" -path "*stats_*" -size +20M -delete
du -sh $OUT; find $OUT -name "*.csv" | xargs ls -la | head -40
