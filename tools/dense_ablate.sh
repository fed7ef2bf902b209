#!/bin/bash
# Run ON THE GPU BOX: ablation ladder of dense_attend_kernel (ablation build; results wrong by construction).
#   tools/dense_ablate.sh "<kinds>" "<variants>"     variants: 0 as shipped, 1 no A V, 16 no weights' arithmetic, 32 no zero-granule skip, sums of them
set -u
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > /dev/null 2>&1
for kind in $1; do
  for v in $2; do
    echo -n "variant $v: "; DAGL_DENSE_VARIANT=$v python tools/dense_case.py $kind 20 2>/dev/null | tail -1
  done
done
python -m dagl_amd.build --force > /dev/null 2>&1
