#!/bin/bash
# Run ON THE GPU BOX: rebuild the library with the ablation variants compiled in (-DDAGL_ABLATION; they give wrong results
# by construction and are absent from the release build) and time the default benchmark's stages under each variant.
#   tools/ablate.sh P16 0 3 7 1 5 6      -> DAGL_P16_VARIANT in {0,3,7,1,5,6}
set -u
WHICH=$1; shift
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > /dev/null 2>&1
for v in "$@"; do
  env DAGL_${WHICH}_VARIANT=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WHICH variant $v', round(d['ms_per_step'],4), {k: round(x*1e3,1) for k,x in d['stage_ms'].items()})"
done
