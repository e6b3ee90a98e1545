#!/bin/bash
# Dev tool: builds ONE kernel variant with arbitrary defines for A/B timing:
#   tools/build_variant.sh NAME [-DRTP_STATS] [-DRTP_WF_BLOCK=640 -DRTP_WF_MIN_WAVES=5] …
# → ray-tracing-practice_amd/variants/librtp_amd_NAME.so, used through RTP_AMD_LIB.
set -e
cd "$(dirname "$0")/../ray-tracing-practice_amd" && mkdir -p variants
NAME=$1; shift
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -DRTP_PARITY_FLAGS=1 -DRTP_DEV_BUILD -DRTP_TRIPWIRE=0"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o variants/rt_capi_$NAME.o csrc/rt_capi.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/librtp_amd_$NAME.so variants/rt_capi_$NAME.o csrc/rt_accel.o csrc/rt_build.o csrc/rt_multi.o -ldl
echo "built variants/librtp_amd_$NAME.so"
