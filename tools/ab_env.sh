#!/bin/bash
# Run ON THE GPU BOX: A/B of one library under an environment switch, interleaved on the same box.
#   tools/ab_env.sh VAR valA valB [rounds] [extra bench args]
set -u
cd $GRAFT_REPO_ROOT
VAR=$1; A=$2; B=$3; R=${4:-3}; shift 4 || true
for r in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-quality --no-extra "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['ms_per_step'],4), {k: round(x*1e3,1) for k,x in d['stage_ms'].items()})"
  done
done
