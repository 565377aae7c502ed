for i in 1 2 3; do
for v in 0 1; do
NRL_NEWS_FORK=$v python bench.py --steps 200 --warmup 30 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fork=$v', d['ms_per_step'], d['median_ms_per_step'], d['value'])"
done; done
