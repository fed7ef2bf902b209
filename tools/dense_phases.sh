#!/bin/bash
# Run ON THE GPU BOX: per-role phase clocks of dense_attend_kernel (ablation build, DAGL_DENSE_VARIANT=64; s_memtime perturbs ~10 %).
#   tools/dense_phases.sh <tag> "<kinds>"
set -u
TAG=$1; KINDS=${2:-"real synth"}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/dense_phases_$TAG.log; : > $OUT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > /dev/null 2>&1
for kind in $KINDS; do
  echo "== $kind" >> $OUT
  DAGL_DENSE_VARIANT=64 DAGL_TIMES_FILE=$OUT python tools/dense_case.py $kind 20 2>/dev/null | tail -1 >> $OUT
done
python -m dagl_amd.build --force > /dev/null 2>&1
cat $OUT
