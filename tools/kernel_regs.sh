#!/bin/bash
# tools/kernel_regs.sh <name> [filter]: VGPRs / AGPRs / LDS / scratch of every kernel of hopwise_amd/csrc/<name>.o (the gfx950 code object
# embedded in the host object), one line per kernel; occupancy follows from 512 / (vgpr + agpr rounded up to 8) waves per SIMD.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp "$ROOT/hopwise_amd/csrc/$1.o" "$T/x.o"
(cd "$T" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null 2>&1)
CO=$(ls "$T"/x.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" | grep -E "^\s+\.name:|\.vgpr_count|\.agpr_count|\.group_segment_fixed_size|\.private_segment_fixed_size|\.vgpr_spill_count" \
  | paste - - - - - - | sed 's/  */ /g; s/\.group_segment_fixed_size/lds/; s/\.private_segment_fixed_size/scratch/; s/\.vgpr_spill_count/spill/; s/\.vgpr_count/vgpr/; s/\.agpr_count/agpr/' \
  | awk '{name=""; rest=""; for(i=1;i<=NF;i++){ if($i==".name:"){name=$(i+1)} } gsub(/\.name: [^ \t]+/,""); print $0 "\t" name}' | grep "${2:-.}" | while IFS=$'\t' read -r nums name; do echo "$nums  $(echo $name | c++filt | cut -c1-110)"; done
rm -rf "$T"
