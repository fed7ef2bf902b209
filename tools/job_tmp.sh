#!/bin/bash
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force 2>&1 | grep -i " error" 
rm -f /tmp/times.txt
DAGL_TIMES_FILE=/tmp/times.txt python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-quality --no-extra > /dev/null 2>&1
python tools/block_times.py /tmp/times.txt 4
