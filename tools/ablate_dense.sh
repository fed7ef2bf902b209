#!/bin/bash
# Run ON THE GPU BOX (ablation build): the dense regime (adaptive, default-initialised heads) under DAGL_DENSE_VARIANT values
# (1 no A V, 2 no S, 4 no value staging; results wrong by construction).
set -u
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > /dev/null 2>&1
for v in "$@"; do
  DAGL_DENSE_VARIANT=$v python bench.py --mode adaptive --variant default --steps 10 --warmup 3 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense variant $v', round(d['ms_per_step'],4))"
done
