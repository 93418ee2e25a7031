export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_h}
mkdir -p $out
cd $R
for n in 3 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$((n > 4 ? n + 2 : 8)) timeout 300 python bench.py --inflight $n --steps 200 --repeats 3 --no-cpu-baseline --no-fp32-mode --no-train --no-proj-feat-variant --no-power --no-time-table-pass --no-config5 --no-ceiling-probe --no-other-half --no-pgcn --detail-out /tmp/d.json 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('inflight $n: %.1f img/s %.3f ms %s' % (d['value'], d['ms_per_step'], d['config']['region_ms_per_step']))" >> $out/inflight_sweep.txt
done
timeout 600 python tools/train_glue.py 32 > $out/train_glue.txt 2>&1
