#!/bin/bash
# one line: the graded gather's roofline block of the default bench.py run
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['roofline_gather']; print('gather', round(g['ms_per_launch']*1e3,2), 'us frac', round(g['frac'],3), 'same-set', round(g['same_set_ms_per_launch']*1e3,2), 'step', round(d['ms_per_step'],4))"
