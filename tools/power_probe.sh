#!/bin/bash
# usage (GPU box): tools/power_probe.sh <inflight>   -- samples socket power and shader clock (rocm-smi) while bench.py's timed loop runs
N=${1:-4}
python bench.py --inflight $N --steps 4000 --warmup 5 --repeats 1 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --autotune-cache /tmp/at_probe.json > /tmp/probe_$N.log 2>&1 &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|Average Graphics Package Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 1
done
wait $BP
tail -1 /tmp/probe_$N.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('inflight $N ms_per_step', d['ms_per_step'])"
