#!/bin/bash
# Run ON THE GPU BOX (ablation build): time the default benchmark under environment settings, one run per argument; several
# variables for one run are joined by commas.
#   tools/ablate_env.sh DAGL_SCREEN_SAMPLE=8 DAGL_SCREEN_QBLOCK=256,DAGL_SCREEN_VARIANT=8
set -u
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > /dev/null 2>&1
for kv in "$@"; do
  env ${kv//,/ } python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', round(d['ms_per_step'],4), {k: round(x*1e3,1) for k,x in d['stage_ms'].items()})"
done
