#!/bin/bash
# Round 6, first GPU call (through gpurun from the repo root; everything under gpurun_out/r6a/):
#   1. the new / changed GPU tests first (fp8 matrix-instruction kernel, the reference-pinned demo path, the 50-step batch-32 headline golden),
#      then the whole -m gpu suite
#   2. bench.py as the driver runs it: the one stdout line + the detail file
#   3. write-through result stores (libkdiff_hip_wt.so, kd_common.h: st16) against the default build, same box, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6a
rm -rf $OUT && mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q -x -k "mx8 or fp8_mode or demo_path or cfg_wrapper or headline_50" > $OUT/new_tests.log 2>&1
tail -25 $OUT/new_tests.log
python -m pytest tests -m gpu -q --maxfail=40 > $OUT/gpu_tests.log 2>&1
tail -15 $OUT/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $OUT/bench_detail.json > $OUT/bench_line.json 2> $OUT/bench_line.err
echo "bench rc=$? stdout bytes: $(wc -c < $OUT/bench_line.json) stderr bytes: $(wc -c < $OUT/bench_line.err)"
cat $OUT/bench_line.json
AB="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-kernel-events --modes split3,bf16"
for i in 1 2; do
  python bench.py $AB --detail-file $OUT/ab_default_$i.json > $OUT/ab_default_$i.line 2>/dev/null
  KDIFF_HIP_LIB=$R/k-diffusion_amd/csrc/libkdiff_hip_wt.so python bench.py $AB --detail-file $OUT/ab_wt_$i.json > $OUT/ab_wt_$i.line 2>/dev/null
done
python - <<EOF
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.line")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["mode_values"])
    except Exception as e:
        print(f, "unreadable", e)
EOF
