# same-lease A/B of bench.py's two kernel tables (time-tuned live vs the shipped throughput table), alternating
for i in 1 2 3; do for t in time throughput; do
  python bench.py --tuning $t --steps 60 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-time-table-pass 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); p=d['power'] or {}
print('--tuning %-10s %8.1f images/s  %.3f ms/step  regions %s | one in flight %.3f ms | %s W %s MHz %s J/step | bit-identical tables: %s' % ('$t', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'], d['config']['ms_per_forward_one_in_flight'], p.get('socket_w'), p.get('sclk_mhz'), p.get('joules_per_step'), d['config']['tunings_bit_identical']))"
done; done
