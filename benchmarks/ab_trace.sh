#!/bin/bash
# Same-box A/B of two environment settings under rocprofv3 --kernel-trace --stats: per-kernel averages side by side.
#   bash benchmarks/ab_trace.sh "X=1" "KDIFF_COND_SCHEDULE=0"
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  rm -rf $OUT/ab_$i
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab_$i -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-other-configs --no-other-modes > $OUT/ab_$i.log 2>&1
  tail -1 $OUT/ab_$i.log | cut -c1-120
  i=$((i+1))
done
