V=$PWD/ray-tracing-practice_amd/variants
export SPP=500 ITERS=2
echo "default:"; python3 tools/perf_sweep.py | grep -o "primary ms [0-9.]*"
echo "allsky (wrong image, timing only):"; RTP_AMD_LIB=$V/librtp_amd_allsky.so python3 tools/perf_sweep.py | grep -o "primary ms [0-9.]*\|trace ms [0-9.]*"
