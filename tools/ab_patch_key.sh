#!/bin/bash
# usage (via gpurun): tools/ab_patch_key.sh  -- the halo-reuse kernel (DIR_CONV_VARIANT 12) on the 3x3 layers, plain patch key (lib/libdir_hip_plainkey.so:
# DIR_BUILD_TAG=plainkey DIR_HIPCC_EXTRA=-DDIR_PATCH_PLAIN_KEY=1 python -m dir_amd.build) against the round-5 key, alternating, plus the f16 / bf16 step
for i in 1 2; do
  for v in plainkey new; do
    lib=""; [ $v = plainkey ] && lib="$GRAFT_REPO_ROOT/dir_amd/lib/libdir_hip_plainkey.so"
    echo "== $v"
    DIR_LIB_PATH=$lib VARIANT=12 ONLY=3x3 python tools/bench_layers.py 2>&1 | grep -v "^count"
  done
done
