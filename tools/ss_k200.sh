#!/bin/bash
python tools/scale_sweep.py 2>/dev/null | grep "k=200" | tr '\n' ';' | sed 's/normwise//g; s/range_fallback 0  scan screened//g; s/topk           default  k=200 input//g'
echo
