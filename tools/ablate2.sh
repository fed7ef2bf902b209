#!/bin/bash
# Run ON THE GPU BOX (ablation build): default benchmark under several "ENV=val,ENV2=val2" settings (comma separated).
set -u
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > /dev/null 2>&1
for kv in "$@"; do
  env ${kv//,/ } python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', round(d['ms_per_step'],4))"
done
