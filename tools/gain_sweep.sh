cd $GRAFT_REPO_ROOT
for g in 1.8 1.75 1.7 1.65 1.6 1.5 1.4 1.3; do
python bench.py --mode adaptive --variant sparse --sparse-gain $g --wseed 41 --fseed 41 --steps 10 --warmup 3 --no-cpu-baseline --no-quality --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('gain $g', round(d['ms_per_step'],3), 'path', c.get('selection_path'), 'max', c.get('max_degree'), 'mean', c.get('mean_degree'))"
done
