#!/bin/bash
# Run ON THE GPU BOX: A/B of bench.py argument sets on the default benchmark, interleaved, same box, same library.
#   tools/ab_flag.sh "<args A>" "<args B>" [rounds]
set -u
cd $GRAFT_REPO_ROOT
R=${3:-3}
for r in $(seq 1 $R); do
  for v in "$1" "$2"; do
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-quality --no-extra $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', round(d['ms_per_step'],4), {k: round(x*1e3,1) for k,x in d['stage_ms'].items()})"
  done
done
