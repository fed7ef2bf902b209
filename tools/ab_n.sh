#!/bin/bash
# Run ON THE GPU BOX: A/B of any number of prebuilt libraries ab_libs/*.so on the default benchmark, interleaved, same box.
#   tools/ab_n.sh "<lib names>" [rounds]
set -u
cd $GRAFT_REPO_ROOT
cp dagl_amd/csrc/libdagl_ce.so /tmp/keep.so
R=${2:-3}
for r in $(seq 1 $R); do
  for v in $1; do
    cp ab_libs/$v.so dagl_amd/csrc/libdagl_ce.so
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), 'parity', d.get('parity_err'), {k: round(x*1e3,1) for k,x in d['stage_ms'].items()})"
  done
done
cp /tmp/keep.so dagl_amd/csrc/libdagl_ce.so
