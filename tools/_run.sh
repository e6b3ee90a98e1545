export SPP=500 ITERS=3
V=$PWD/ray-tracing-practice_amd/variants
short() { grep -o "best kernel ms [0-9.]*\|trace ms [0-9.]*\|flagged [0-9]*" | tr '\n' ' '; echo; }
echo -n "default: "; python3 tools/perf_sweep.py | short
for v in r1 u3 u5 u6; do echo -n "$v: "; RTP_AMD_LIB=$V/librtp_amd_$v.so python3 tools/perf_sweep.py | short; done
for ki in 24 28 36 40 44; do echo -n "k_inner $ki: "; RTP_K_INNER=$ki python3 tools/perf_sweep.py | short; done
for ks in 44 48 56 58; do echo -n "k_shade $ks: "; RTP_K_SHADE=$ks python3 tools/perf_sweep.py | short; done
echo -n "k 40/56: "; RTP_K_INNER=40 RTP_K_SHADE=56 python3 tools/perf_sweep.py | short
echo -n "k 36/56: "; RTP_K_INNER=36 RTP_K_SHADE=56 python3 tools/perf_sweep.py | short
