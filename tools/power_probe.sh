#!/bin/bash
# usage (GPU box): tools/power_probe.sh <inflight>   -- samples socket power and shader clock (rocm-smi) while bench.py's timed loop runs
N=${1:-4}
python bench.py --inflight $N --steps 20000 --warmup 5 --repeats 1 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --autotune-cache /tmp/at_probe.json > /tmp/probe_$N.log 2>&1 &
BP=$!
for i in $(seq 1 16); do
  sleep 4
  echo -n "t=$((4*i))s "
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Package Power|sclk|GPU use" | sed 's/.*: //' | tr '\n' ';'
  echo
done
wait $BP
tail -1 /tmp/probe_$N.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('inflight $N ms_per_step', d['ms_per_step'])"
