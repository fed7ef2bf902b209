#!/bin/bash
# Run ON THE GPU BOX: per-kernel averages of one regime.  tools/prof_case.sh <name> <args of tools/prof_case.py ...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o $name -- python tools/prof_case.py "$@" > gpurun_out/prof_$name.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/prof_$name/*.db")[0]
c = sqlite3.connect(db)
for r in c.execute("select name,total_calls,average,percentage from top_kernels limit 24"):
    print(f"{r[0][:70]:70s} {r[1]:6d} {r[2]:9.2f} us {r[3]:5.1f} %")
PY
tail -1 gpurun_out/prof_$name.log
