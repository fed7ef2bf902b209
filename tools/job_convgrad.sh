#!/bin/bash
# on the GPU box: the direct g / theta backward -- tests, then the training step A/B (unfold route vs direct)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "conv_pair or prologue_backward" -rP 2>&1 | grep -E "conv_pair_backward|passed|failed|Error|error" | tail -40
for r in 1 2; do
  for v in unfold direct; do
    timeout 600 python bench.py --train --prologue-backward $v --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],2))"
  done
done
