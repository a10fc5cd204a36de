#!/bin/bash
# Registers, scratch, occupancy and LDS of every kernel of one translation unit, one line per kernel (CPU only: hipcc
# cross-compiles).    tools/kernel_resources.sh vvc_amd/csrc/lvgpu_tiles_any.hip [name filter]
src=$1; filt=${2:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -o /dev/null "$src" -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {printf "%s vgpr=%s scratch=%s waves/simd=%s lds=%s\n", name, v, s, o, $(NF-1)}' |
  c++filt | grep -E "$filt"
