#!/bin/bash
# A/B: base vs cur on inference headline + training steps + gemm tests
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in base cur; do
    cp ab_libs/$v.so dagl_amd/csrc/libdagl_ce.so
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), {k: round(x*1e3,1) for k,x in d['stage_ms'].items()})"
    echo -n "$v "; bash tools/time_train.sh topk
    echo -n "$v "; bash tools/time_train.sh adaptive
  done
done
cp ab_libs/cur.so dagl_amd/csrc/libdagl_ce.so
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_stages.py tests/test_gpu_block.py tests/test_gpu_backward.py -x -q 2>&1 | tail -3
