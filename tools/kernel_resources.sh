#!/bin/bash
# Register / LDS / scratch use and occupancy of every kernel of the library, as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage).  Usage: tools/kernel_resources.sh [extra hipcc flags] > profiles/<round>_kernel_resources.txt
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c -Rpass-analysis=kernel-resource-usage "$@" \
  nimblephysics_amd/csrc/nimble_amd.hip -o /dev/null 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' |
  awk '/Function Name:/ {n=$NF; sub(/^_ZN3nbl[0-9]+/, "", n); sub(/(ENS_|ILi|IL[bj]|EPK|EP[a-z]|ENS).*/, "", n)}
       / VGPRs:/ {v=$NF} / AGPRs:/ {a=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} / VGPRs Spill/ {sp=$NF}
       /LDS Size/ {printf "%-34s VGPR %3s AGPR %3s scratch %4s B  LDS %6s B/block  waves/SIMD(regs) %s\n", n, v, a, s, $NF, o}'
