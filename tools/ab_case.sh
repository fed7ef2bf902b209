#!/bin/bash
# Run ON THE GPU BOX: A/B of prebuilt libraries ab_libs/*.so on any one-line timing command, interleaved, same box.
#   tools/ab_case.sh "<libs>" "<command>" [rounds]
set -u
cd $GRAFT_REPO_ROOT
cp dagl_amd/csrc/libdagl_ce.so /tmp/keep.so
R=${3:-2}
for r in $(seq 1 $R); do
  for v in $1; do
    cp ab_libs/$v.so dagl_amd/csrc/libdagl_ce.so
    echo -n "$v: "; $2 2>/dev/null | tail -1
  done
done
cp /tmp/keep.so dagl_amd/csrc/libdagl_ce.so
