#!/bin/bash
# Register / scratch / LDS use of every kernel of one .hip file (compiler remarks; no GPU needed).
#   scripts/kernel_resources.sh grtcode_amd/csrc/hip/k_gas_optics_mp.hip
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Iinclude \
      -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" |
  sed -e 's/^.*remark: //' | paste - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' -e 's/  */ /g'
