cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in base pf3 prio1 pf1; do
    cp ab_libs/$v.so dagl_amd/csrc/libdagl_ce.so
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), {k: round(x*1e3,1) for k,x in d['stage_ms'].items()})"
done; done
