#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 passes of the default benchmark command, output under gpurun_out/prof_$1/.
#   1. --kernel-trace --stats                       per-kernel durations (the CSV the judge compares with bench.py's events)
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE            HBM traffic per kernel (separate passes: TCC slots; no trace domains)
#   3. --pmc SQ_* counters                           MFMA utilisation / wait breakdown of the matrix-core kernels
# tools/summarize_profiles.py turns the CSVs into the small JSON / CSV files kept under profiles/.
set -u
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-quality --no-extra"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $CMD > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o k -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o k -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq -o k -- $CMD > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq2 -o k -- $CMD > $OUT/sq2.log 2>&1
# the dense regime (shipped semantics at default init) and the mean-degree-8 regime: kernel stats only
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dense -o k -- python $GRAFT_REPO_ROOT/bench.py --mode adaptive --variant default --steps 10 --warmup 3 --no-cpu-baseline --no-quality --no-extra > $OUT/dense.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/md8 -o k -- python $GRAFT_REPO_ROOT/bench.py --mode adaptive --variant sparse --sparse-gain 1.95 --wseed 41 --fseed 41 --steps 20 --warmup 5 --no-cpu-baseline --no-quality --no-extra > $OUT/md8.log 2>&1
# (MIOPEN_FIND_MODE=FAST: without it MIOpen's find phase -- 32 calls x 38 ms of naive_conv_* and Tensile benchmarking inside the
# first step -- is two thirds of the traced time and the CSV does not show the steady-state step)
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o k -- python $GRAFT_REPO_ROOT/bench.py --train --mode topk --steps 12 --warmup 4 > $OUT/train.log 2>&1
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_adaptive -o k -- python $GRAFT_REPO_ROOT/bench.py --train --mode adaptive --steps 8 --warmup 3 > $OUT/train_adaptive.log 2>&1
# (round 6) BASELINE configs[2] / configs[3]: kernel by kernel
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c512 -o k -- python $GRAFT_REPO_ROOT/tools/prof_case.py topk default 2.0 2024 100 512 8 60 > $OUT/c512.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c1024 -o k -- python $GRAFT_REPO_ROOT/tools/prof_case.py adaptive_topk sparse 1.7 2024 100 1024 16 12 > $OUT/c1024.log 2>&1
# (round 6) HBM-side traffic of the dense regime's and the training path's kernels: FETCH_SIZE / WRITE_SIZE, separate passes
DENSE="python $GRAFT_REPO_ROOT/bench.py --mode adaptive --variant default --steps 10 --warmup 3 --no-cpu-baseline --no-quality --no-extra"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/dense_fetch -o k -- $DENSE > $OUT/dense_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/dense_write -o k -- $DENSE > $OUT/dense_write.log 2>&1
TRAIN="python $GRAFT_REPO_ROOT/bench.py --train --mode topk --steps 6 --warmup 3"
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/train_fetch -o k -- $TRAIN > $OUT/train_fetch.log 2>&1
MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/train_write -o k -- $TRAIN > $OUT/train_write.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_profiles.py $OUT $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_summary $TAG
ls $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_summary
# gpurun merges at most 64 MiB back: the raw traces (kernel traces of ~2000-launch runs) go, the summaries stay
rm -rf $OUT
