#!/bin/bash
# Run ON THE GPU BOX (ablation build): clock and matrix-pipe occupancy of the screen kernel under DAGL_SCREEN_VARIANT values.
#   tools/ablate_pmc.sh 0 57 121
set -u
cd $GRAFT_REPO_ROOT
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > /dev/null 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/abl_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  DAGL_SCREEN_VARIANT=$v timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/v$v -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-quality --no-extra > $OUT/v$v.log 2>&1
  python - $OUT/v$v $v <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'screen_kernel<1' in r['Kernel_Name']:
            cnt[r['Dispatch_Id']][r['Counter_Name']].append(float(r['Counter_Value']))
dur = {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'screen_kernel<1' in r['Kernel_Name']:
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
import statistics
rows = []
for k, c in cnt.items():
    if k in dur and 'GRBM_GUI_ACTIVE' in c:
        gui = sum(c['GRBM_GUI_ACTIVE']) / 8; mf = sum(c['SQ_VALU_MFMA_BUSY_CYCLES'])
        rows.append((dur[k], gui / dur[k] / 1e3, mf / (1024 * gui)))
rows = rows[3:]
print('variant', v, 'us', round(statistics.mean(r[0] for r in rows), 1), 'clock GHz', round(statistics.mean(r[1] for r in rows), 3), 'mfma busy', round(statistics.mean(r[2] for r in rows), 3))
PY
done
