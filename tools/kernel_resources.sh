#!/bin/bash
# kernel resource usage (VGPRs, SGPRs, spills, LDS, occupancy) of one csrc/*.hip for gfx950:  tools/kernel_resources.sh conv_wino43.hip [extra flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$(dirname "$0")/../multipoint_amd/csrc/$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|Spill|LDS Size|Occupancy|ScratchSize" | \
  sed -e 's/.*remark: [^ ]* *//' | paste - - - - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
