#!/bin/bash
# one line: ms per step of bench.py --mode topk --k $1 (no extras)
python bench.py --mode topk --k $1 --steps 20 --warmup 5 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=$1', round(d['ms_per_step'],4))"
