cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/scale_times.py C2 16 2>&1 | tail -6
python bench.py --config C2 --steps 10 --warmup 2 --no-cpu --no-full --no-c2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('C2 ms/step', d['ms_per_step'], 'wino frac', r['frac'], 'share', r['share_of_step'], 'train', d['train'])"
