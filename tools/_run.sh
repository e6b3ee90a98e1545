V=$PWD/ray-tracing-practice_amd/variants
echo "== C5 8 spp (row check)"; SPP=8 python3 tools/c5_run.py 2>&1 | tail -3
echo "== C5 125 spp"; SPP=125 python3 tools/c5_run.py 2>&1 | tail -2
echo "== C5 125 spp old step"; RTP_AMD_LIB=$V/librtp_amd_norot.so SPP=125 python3 tools/c5_run.py 2>&1 | tail -2
