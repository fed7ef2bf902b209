#!/bin/bash
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force 2>&1 | grep -i " error" 
for v in 0 256 0 256; do
  rm -f /tmp/times_$v.txt
  DAGL_SCREEN_RING=1 DAGL_TIMES_FILE=/tmp/times_$v.txt DAGL_SCREEN_VARIANT=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', round(d['ms_per_step'],4), {k: round(x*1e3,1) for k,x in d['stage_ms'].items()})"
  python tools/block_times.py /tmp/times_$v.txt 4 | grep -A3 'screen_ring' | grep -E 'span|prologue'
done
