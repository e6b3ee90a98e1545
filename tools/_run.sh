V=$PWD/ray-tracing-practice_amd/variants
export SPP=500 ITERS=3
short() { grep -o "best kernel ms [0-9.]*\|trace ms [0-9.]*" | tr '\n' ' '; echo; }
for i in 1 2; do
echo -n "clamped (default): "; python3 tools/perf_sweep.py | short
echo -n "not clamped: "; RTP_AMD_LIB=$V/librtp_amd_noclamp.so python3 tools/perf_sweep.py | short
done
