#!/bin/bash
# Small-batch A/B of the kernel-selection thresholds (fp32-parity mode): ms per forward at batch 1 .. 16 for a few KDIFF_OPTIONS settings.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for OPT in ${OPTS:-"" "x3s_max_rows=0"}; do
  for B in 1 2 4 8 16; do
    KDIFF_OPTIONS=$OPT KDIFF_GEMM=split3 python $R/benchmarks/small_batch.py $R/configs/config_oxford_flowers.json $B sample_dpmpp_2m 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('opts=[$OPT] batch', d['batch'], 'ms/forward', d['direct_ms_per_forward'], 'images/s', d['direct_images_per_s'])"
  done
done
