#!/bin/bash
# on the GPU box: per-kernel averages of the training step (rocprofv3 --kernel-trace --stats), top 40 rows
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --train --mode topk --steps 12 --warmup 4 > $OUT/train.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot / 1e6)
for r in rows[:40]:
    print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), round(float(r["TotalDurationNs"]) / 1e6, 2), r["Percentage"])
PY
tail -1 $OUT/train.log | cut -c1-300
