cd $GRAFT_REPO_ROOT
BACKBONE=hrnet_w48 NSTEP=4 python tools/bench_train_graphed.py 32 2>&1 | tail -3
