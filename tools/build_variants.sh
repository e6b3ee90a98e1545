#!/bin/bash
# Dev tool: builds kernel variants (block size / min waves per SIMD / box-step unroll) for A/B timing.
cd "$(dirname "$0")/../ray-tracing-practice_amd" && mkdir -p variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -shared"
for v in "$@"; do
  IFS=_ read -r b w u x <<< "$v"; EXTRA=""; [ "$x" = "stats" ] && EXTRA="-DRTP_STATS"
  ( /opt/rocm/bin/hipcc ${FLAGS% -shared} -DRTP_BLOCK=$b -DRTP_MIN_WAVES=$w -DRTP_UNROLL=$u $EXTRA -c -o variants/rt_capi_$v.o csrc/rt_capi.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/librtp_amd_$v.so variants/rt_capi_$v.o csrc/rt_accel.o csrc/rt_build.o ) &
done
wait
ls variants
