#!/bin/bash
# Run ON THE GPU BOX: round-6 baseline evidence for project16_kernel before touching it:
#   FETCH_SIZE / WRITE_SIZE apart, the SQ issue counters, and the block timeline of an ablation build.
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_base
mkdir -p $OUT
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-quality --no-extra > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-quality --no-extra"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o k -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o k -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/insts -o k -- $CMD > $OUT/insts.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq -o k -- $CMD > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq3 -o k -- $CMD > $OUT/sq3.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ("fetch", "write", "insts", "sq", "sq3"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f: print(sub, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        acc[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(sub, k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
cd $GRAFT_REPO_ROOT
# block timeline (ablation build: stamps compiled in)
DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > $OUT/build.log 2>&1
DAGL_TIMES_FILE=$OUT/times.txt python bench.py --steps 10 --warmup 3 --prewarm 0.1 --no-cpu-baseline --no-quality --no-extra > /dev/null 2>&1
python tools/block_times.py $OUT/times.txt 4 | tee $OUT/timeline.txt
rm -f $OUT/times.txt
