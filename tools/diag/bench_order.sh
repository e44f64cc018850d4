for i in 1 2 3; do
python bench.py --no-cpu-baseline --workload static --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run',$i,'static ms',j['ms_per_step'],'frac',j['roofline']['frac'])"
done
python bench.py --workload static --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with cpu: static ms',j['ms_per_step'],'frac',j['roofline']['frac'])"
python bench.py --no-cpu-baseline --workload static --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20/5: static ms',j['ms_per_step'],'frac',j['roofline']['frac'])"
