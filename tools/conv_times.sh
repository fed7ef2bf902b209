#!/bin/bash
# Run ON THE GPU BOX: block timeline of conv_pair16_kernel (ablation build, DAGL_TIMES_FILE), headline configuration.
set -u
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > /dev/null 2>&1
rm -f /tmp/conv_times.txt
DAGL_TIMES_FILE=/tmp/conv_times.txt DAGL_TIMES_SKIP=40 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-quality --no-extra > /dev/null 2>&1
grep "^conv_pair16" /tmp/conv_times.txt > /tmp/conv_only.txt
python tools/block_times.py /tmp/conv_only.txt 2
python -m dagl_amd.build --force > /dev/null 2>&1
