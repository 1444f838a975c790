#!/bin/bash
# Sustained job: sample.py over N images of the headline workload (256 x 256, DPM++2M x 50, batch 32, host noise ahead of the sampler), with the
# device's power / clock / memory sampled beside it every 10 s.  Output: gpurun_out/soak.log (rate of the whole job, rocm-smi samples).
#   bash benchmarks/soak.sh [N = 48000]        (~225 s at 213 images/s)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-48000}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
( while true; do echo "$(date +%s) $(rocm-smi --showpower --showclocks --showmemuse --showtemp 2>/dev/null | grep -i 'power\|sclk\|mclk\|GPU Memory Allocated\|Temperature (Sensor junction)' | tr -s ' ' | tr '\n' '|')"; sleep 10; done ) > $OUT/soak_smi.log 2>&1 &
SMI=$!
START=$(date +%s)
python $R/sample.py --config $R/configs/config_oxford_flowers.json --random-weights --seed 0 --class-cond -1 --sampler dpmpp_2m --steps 50 \
  -n $N --batch-size 32 --no-png --gather-uint8 > $OUT/soak_job.log 2>&1
RC=$?
END=$(date +%s)
kill $SMI
{ echo "# benchmarks/soak.sh $N: sample.py, config_oxford_flowers.json, dpmpp_2m x 50, batch 32, --seed 0 (host noise), uint8 gather; rc=$RC, $((END-START)) s wall incl. start-up";
  grep -v "it/s\|s/it\|amdgpu.ids" $OUT/soak_job.log | tail -5;
  echo "# rocm-smi every 10 s (first, middle, last samples):"; L=$(wc -l < $OUT/soak_smi.log); sed -n "2p;$((L/2))p;$((L-1))p" $OUT/soak_smi.log; } > $OUT/soak.log
cat $OUT/soak.log
