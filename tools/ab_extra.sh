cd $GRAFT_REPO_ROOT
for v in prev cur prev cur; do
cp ab_libs/$v.so dagl_amd/csrc/libdagl_ce.so
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-quality 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {k[8:30]: round(v['ms_per_step'],3) for k,v in d['extra_configs'].items()})"
done
cp ab_libs/cur.so dagl_amd/csrc/libdagl_ce.so
