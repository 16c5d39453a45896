#!/bin/bash
# Registers, scratch and LDS of every kernel in kernels.o (from the code object's metadata): spills in a classify class show here first.
set -e
cd "$(dirname "$0")/../lmat_amd/csrc"
T=$(mktemp -d)
objcopy --dump-section .hip_fatbin=$T/fat.bin kernels.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | python3 -c '
import sys,re
txt=sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    g=lambda k:(re.search(r"\."+k+r":\s*(\S+)",blk) or [None,"?"])[1]
    name=g("name")
    print("%-28s vgpr %-4s sgpr %-4s spill_v %-4s spill_s %-4s scratch %-6s lds %-6s" % (name[:28],g("vgpr_count"),g("sgpr_count"),g("vgpr_spill_count"),g("sgpr_spill_count"),g("private_segment_fixed_size"),g("group_segment_fixed_size")), name[28:90])
'
rm -rf $T
