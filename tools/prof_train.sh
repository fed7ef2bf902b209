#!/bin/bash
# on the GPU box: per-kernel averages of the training step (rocprofv3 --kernel-trace --stats), top rows
#   tools/prof_train.sh [bench.py arguments after --train; default: --mode topk]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
ARGS=${@:---mode topk}
MIOPEN_FIND_MODE=FAST timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --train $ARGS --steps 12 --warmup 4 > $OUT/train.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot / 1e6, "(16 steps)")
for r in rows[:34]:
    print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), round(float(r["TotalDurationNs"]) / 1e6 / 16, 2), r["Percentage"])
# the split-fp16 GEMM by launch shape (grid): which product is which
t = glob.glob("$OUT/**/k_kernel_trace.csv", recursive=True)[0]
by = {}
for r in csv.DictReader(open(t)):
    if "gemm16s_kernel" in r["Kernel_Name"]:
        key = (r["Kernel_Name"][12:34], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        by.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("gemm16s", k, len(v), "calls", round(sum(v) / len(v), 1), "us avg")
PY
grep -o '"ms_per_step": [0-9.]*' $OUT/train.log | head -2
