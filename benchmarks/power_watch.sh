#!/bin/bash
# Socket power / shader clock samples (rocm-smi, every 0.25 s) while bench.py runs one arithmetic mode: is the path power-limited?
#    benchmarks/power_watch.sh split3 > gpurun_out/power_split3.log
MODE=${1:-split3}
STEPS=${2:-40}
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/bench.py --steps $STEPS --warmup 3 --mode $MODE --no-cpu-baseline --no-other-modes --no-other-configs --no-parity --no-small-batch --no-power --no-kernel-events > /tmp/pw_bench.json 2>/dev/null &
BP=$!
sleep 4
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Current Socket Graphics Package Power|sclk|junction" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  sleep 0.25
done
wait $BP
python -c "import json; d=json.load(open('/tmp/pw_bench.json')); print('$MODE', d['value'], 'images/s')"
