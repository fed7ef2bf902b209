#!/bin/bash
# on the GPU box: the dense backward on split-fp16 products -- gradient tests, then the adaptive-mode training step A/B
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_configs.py tests/test_gpu_round3.py tests/test_gpu_gemm.py -x -q -m gpu -k "dense or grad or train or default or gemm or fc_grad" 2>&1 | tail -8
for v in fp32 f16; do
  timeout 600 python bench.py --train --dense-backward $v --mode adaptive --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],2))"
done
