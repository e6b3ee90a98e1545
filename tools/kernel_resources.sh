#!/bin/bash
# Compiler's per-kernel resource usage (VGPR / SGPR / scratch / occupancy) of the shipped build → profiles/<round>/kernel_resource_usage.txt
# (CPU only: hipcc cross-compiles for gfx950).
set -e
ROUND=${1:-r04}
cd "$(dirname "$0")/../ray-tracing-practice_amd"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero"
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c -o /tmp/rt_capi_ru.o csrc/rt_capi.hip 2> /tmp/rt_capi_ru.txt
mkdir -p ../profiles/$ROUND
{
  echo "# hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c csrc/rt_capi.hip   ($(/opt/rocm/bin/hipcc --version | head -1))"
  echo "# render_kernel<kLds, kThreaded, kDyn, kWide, kSimple, kPrim>: <true,false,false,false,true,true> = the headline trace kernel (sphere-only build fed by the primary-visibility pass), <false,false,true,true,false,true> = BASELINE configs[4] (distance-aware margins in parametric form on 4-wide nodes, records through L1/L2), <true,false,false,false,false,*> = the general octant kernel, <*,true,…> = the exact walks"
  grep -E "Function Name|TotalSGPRs|VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size" /tmp/rt_capi_ru.txt | sed 's/.*remark: //;s/ \[-Rpass-analysis=kernel-resource-usage\]//' | sed 's/^    /  /' | c++filt
} > ../profiles/$ROUND/kernel_resource_usage.txt
wc -l ../profiles/$ROUND/kernel_resource_usage.txt
