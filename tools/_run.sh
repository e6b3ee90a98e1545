export SPP=200 ITERS=3
short() { grep -o "best kernel ms [0-9.]*\|trace ms [0-9.]*\|primary ms [0-9.]*\|flagged [0-9]*" | tr '\n' ' '; echo; }
echo -n "default 200: "; python3 tools/perf_sweep.py | short
echo -n "default 500: "; SPP=500 ITERS=2 python3 tools/perf_sweep.py | short
echo -n "default 100: "; SPP=100 python3 tools/perf_sweep.py | short
echo "== gpu tests"; timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
