cd $GRAFT_REPO_ROOT
SOURCES=jpeg python tools/bench_fromdisk.py 131072 256 8,8,8,6 2>&1 | grep -E "jpeg  |u8 shards"
