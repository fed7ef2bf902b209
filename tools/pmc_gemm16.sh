#!/bin/bash
# Run ON THE GPU BOX: HBM traffic + SQ counters of gemm16s_kernel on the fc2-backward shape (tools/time_fc_grad.py: B = 8, 128 x 128,
# n = 131 072 patch rows; d W = <4,2,3>-style 256 x 128 tiles with split-K, d rows = 128 x 128 tiles).  Separate rocprofv3 --pmc passes
# (FETCH_SIZE / WRITE_SIZE / two SQ sets), kernel trace only; per-instantiation means + a JSON summary under gpurun_out/pmc_gemm16/.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm16
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/time_fc_grad.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $CMD > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o k -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o k -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq -o k -- $CMD > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq2 -o k -- $CMD > $OUT/sq2.log 2>&1
python - <<PY
import csv, glob, json, collections
out = {"note": "gemm16s_kernel on the fc2-backward shape (tools/time_fc_grad.py, n = 131072): rocprofv3 --pmc passes (one counter set per pass), "
               "mean per launch per instantiation; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md prescribes for gfx950; "
               "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), lds_util = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8) as in tools/summarize_profiles.py"}
res = collections.defaultdict(dict)
for sub in ("fetch", "write", "sq", "sq2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        if "gemm16s_kernel" in row["Kernel_Name"]:
            key = row["Kernel_Name"].split("(")[0].replace("void dagl::", "") + " grid " + row.get("Grid_Size", "?")
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res[k][c] = sum(v) / len(v)
        res[k]["launches"] = len(next(iter(cs.values())))
f = glob.glob("$OUT/stats/**/k_kernel_trace.csv", recursive=True)
if f:
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gemm16s_kernel" in r["Kernel_Name"]:
            key = r["Kernel_Name"].split("(")[0].replace("void dagl::", "") + " grid " + str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out["duration_us"] = {k: sum(v) / len(v) for k, v in dur.items()}
for k, c in res.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    if c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        c["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        if "SQ_LDS_IDX_ACTIVE" in c: c["lds_util"] = c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
out["kernels"] = res
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
tail -2 $OUT/stats.log
