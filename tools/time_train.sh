#!/bin/bash
# one line: ms per training step of bench.py --train --mode $1
python bench.py --train --mode $1 --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train $1', round(d['ms_per_step'],2), 'ms')"
