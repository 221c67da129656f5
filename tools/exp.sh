cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -30
python bench.py --steps 16 --no-cpu-baseline --host-fed 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['verified_streams'], d['worst_case_value'], d['worst_case'])"
python tools/bench_single.py 3000
