#!/bin/bash
# usage (via gpurun): tools/ab_layers.sh <lib tag> "<variants>" [ONLY pattern]  -- tools/bench_layers.py per DIR_CONV_VARIANT, lib/libdir_hip_<tag>.so against the
# product library, alternating twice (DIR_BUILD_TAG=<tag> python -m dir_amd.build makes the second library from the tree as it stands)
tag=$1; vars=$2; only=${3:-3x3}
for i in 1 2; do for v in $vars; do for l in $tag new; do
  lib=""; [ $l != new ] && lib="$GRAFT_REPO_ROOT/dir_amd/lib/libdir_hip_$l.so"
  echo "== variant $v lib $l"
  DIR_LIB_PATH=$lib VARIANT=$v ONLY="$only" python tools/bench_layers.py 2>&1 | grep -v "^count\|amdgpu.ids"
done; done; done
