#!/bin/bash
# Run ON THE GPU BOX: dense regime -- kernel stats and SQ counters of dense_attend_kernel (release build), then the ablation
# ladder (DAGL_DENSE_VARIANT: 1 no A V, 2 no S, 4 no staging, 16 constant weights, 32 no zero-granule skip).
#   tools/dense_study.sh <tag> "<kinds>" "<variants>" ("none": no ablation build)
set -u
TAG=$1; KINDS=${2:-"real synth"}; VARS=${3:-"0 1 2 3 4 32"}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/dense_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kind in $KINDS; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$kind -o k -- python $GRAFT_REPO_ROOT/tools/dense_case.py $kind 30 > $OUT/stats_$kind.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq_$kind -o k -- python $GRAFT_REPO_ROOT/tools/dense_case.py $kind 12 > $OUT/sq_$kind.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq2_$kind -o k -- python $GRAFT_REPO_ROOT/tools/dense_case.py $kind 12 > $OUT/sq2_$kind.log 2>&1
  python - $OUT $kind <<'PY'
import csv, glob, sys, collections, json
out, kind = sys.argv[1], sys.argv[2]
res = {"case": kind}
# (dense_attend_kernel / dense_combine_kernel are launched twice per call, the second time gated: launches that exit at once are left out)
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/stats_{kind}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "dense_attend" in n or "dense_combine" in n or "feat_split" in n or "split_map" in n:
            dur[n.split("(")[0].split("::")[-1]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for n, v in dur.items():
    big = [x for x in v if x >= 0.1 * max(v)]
    res[n + "_avg_us"] = round(sum(big) / len(big), 2)
    res[n + "_launches_counted_of"] = [len(big), len(v)]
for sub in ("sq", "sq2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{sub}_{kind}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "dense_attend" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        big = [x for x in v if x >= 0.1 * max(v)] if max(v) > 0 else v
        res[c] = round(sum(big) / len(big))
if "GRBM_GUI_ACTIVE" in res and "SQ_VALU_MFMA_BUSY_CYCLES" in res:
    gui = res["GRBM_GUI_ACTIVE"] / 8                       # per-XCD counter summed over 8 XCDs
    res["mfma_busy_frac_of_simd_cycles"] = round(res["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui), 3)
    res["lds_active_frac_of_cu_cycles"] = round(res.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * gui), 3)
    res["lds_bank_conflict_frac_of_lds_active"] = round(res.get("SQ_LDS_BANK_CONFLICT", 0) / max(res.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4)
print(json.dumps(res))
json.dump(res, open(f"{out}/pmc_{kind}.json", "w"), indent=1)
PY
done
cd $GRAFT_REPO_ROOT
if [ -n "$VARS" ] && [ "$VARS" != "none" ]; then
  DAGL_EXTRA_FLAGS=-DDAGL_ABLATION python -m dagl_amd.build --force > /dev/null 2>&1
  for kind in $KINDS; do
    for v in $VARS; do
      echo -n "variant $v: "; DAGL_DENSE_VARIANT=$v python tools/dense_case.py $kind 20 2>/dev/null | tail -1
    done
  done | tee $OUT/ablation.log
  python -m dagl_amd.build --force > /dev/null 2>&1
fi
