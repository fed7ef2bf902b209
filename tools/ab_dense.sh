#!/bin/bash
# Run ON THE GPU BOX: A/B of prebuilt libraries ab_libs/*.so on the dense regime (tools/dense_case.py), interleaved, same box.
#   tools/ab_dense.sh "<libs>" "<kinds>" [rounds]
set -u
cd $GRAFT_REPO_ROOT
cp dagl_amd/csrc/libdagl_ce.so /tmp/keep.so
R=${3:-2}
for r in $(seq 1 $R); do
  for v in $1; do
    cp ab_libs/$v.so dagl_amd/csrc/libdagl_ce.so
    for k in $2; do echo -n "$v: "; python tools/dense_case.py $k 30 2>/dev/null | tail -1; done
  done
done
cp /tmp/keep.so dagl_amd/csrc/libdagl_ce.so
