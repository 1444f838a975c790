#!/bin/bash
# Batch-1 (or $B) latency of the headline config, fp32-parity mode: ms per forward and the per-kernel event table for a few KDIFF_OPTIONS settings.
#   [MODE=bf16] bash benchmarks/batch1_table.sh [batch] ["opt list" ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1}; shift
[ $# -eq 0 ] && set -- "" "x3_min_rows=256" "x3_min_rows=128"
cd /tmp
for OPT in "$@"; do
  KDIFF_OPTIONS=$OPT python $R/bench.py --mode ${MODE:-split3} --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-other-modes --no-power --no-parity --no-small-batch --kernel-table /tmp/kt.json > /tmp/b1.json 2>/dev/null
  python - "$OPT" <<'P'
import json, sys
line = [l for l in open('/tmp/b1.json') if l.startswith('{')][-1]
d = json.loads(line)
print(f"opts=[{sys.argv[1]}] batch {d['config'].get('batch_per_gpu', '?')}: {d['ms_per_step'] / 50:.4f} ms per forward, {d['value']:.2f} images/s")
k = json.load(open('/tmp/kt.json'))['kernels']
tot = sum(v['ms'] for v in k.values())
for n, v in sorted(k.items(), key=lambda kv: -kv[1]['ms'])[:18]:
    print(f"   {100 * v['ms'] / tot:5.1f}%  {1e3 * v['ms'] / v['launches']:6.1f} us x {v['launches']:4d}  {n}")
P
done
