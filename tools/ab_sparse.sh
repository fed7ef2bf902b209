#!/bin/bash
# Run ON THE GPU BOX: A/B of prebuilt libraries ab_libs/*.so on the adaptive regimes (mean degree 7.7 / 55 no-wait, dense synthetic), interleaved.
#   tools/ab_sparse.sh "<libs>" [rounds]
set -u
cd $GRAFT_REPO_ROOT
cp dagl_amd/csrc/libdagl_ce.so /tmp/keep.so
R=${2:-2}
for r in $(seq 1 $R); do
  for v in $1; do
    cp ab_libs/$v.so dagl_amd/csrc/libdagl_ce.so
    echo -n "$v: "; python tools/sparse_case.py 1.95 300 auto 2>/dev/null | tail -1
    echo -n "$v: "; python tools/sparse_case.py 1.8 300 auto 2>/dev/null | tail -1
    echo -n "$v: "; python tools/dense_case.py synth 30 2>/dev/null | tail -1
  done
done
cp /tmp/keep.so dagl_amd/csrc/libdagl_ce.so
