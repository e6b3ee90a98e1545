export SPP=100
echo "== default"; python3 tools/perf_sweep.py
echo "== no primary"; RTP_NO_PRIMARY=1 python3 tools/perf_sweep.py
echo "== general kernel + primary"; RTP_NO_SIMPLE=1 python3 tools/perf_sweep.py
echo "== general kernel no primary"; RTP_NO_SIMPLE=1 RTP_NO_PRIMARY=1 python3 tools/perf_sweep.py
export SPP=500 ITERS=2
echo "== default 500"; python3 tools/perf_sweep.py
