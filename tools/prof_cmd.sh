#!/bin/bash
# Run ON THE GPU BOX: per-kernel averages of an arbitrary command.  tools/prof_cmd.sh <name> <command ...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o $name -- "$@" > gpurun_out/prof_$name.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/prof_$name/*.db")[0]
c = sqlite3.connect(db)
for r in c.execute("select name,total_calls,average,percentage from top_kernels limit 22"):
    print(f"{r[0][:70]:70s} {r[1]:6d} {r[2]:9.2f} us {r[3]:5.1f} %")
PY
