cd $GRAFT_REPO_ROOT
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
echo "=== u8 alone"
SOURCES=none python tools/bench_fromdisk.py 131072 256 8 2>&1 | grep -E "u8 shards, 4"
echo "=== u8 with 8 busy processes"
pids=""
for i in 1 2 3 4 5 6 7 8; do python -c "
while True: pass" & pids="$pids $!"; done
SOURCES=none python tools/bench_fromdisk.py 131072 256 8 2>&1 | grep -E "u8 shards, 4"
kill $pids
sleep 1
echo "=== jpeg, workers niced"
DIR_RING_NICE=15 SOURCES=jpeg python tools/bench_fromdisk.py 131072 256 8 2>&1 | grep -E "jpeg  "
