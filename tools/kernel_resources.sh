#!/bin/bash
# registers / scratch / LDS of every kernel in one object of csrc/build (no GPU needed):
#   tools/kernel_resources.sh igemm_f16x3.o
set -e
OBJ=${1:?object name under csrc/build}
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$(dirname "$0")/../self-guided-diffusion-models_amd/csrc/build/$OBJ"
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$LLVM/llvm-readelf --notes $T/dev.co | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    m = re.search(r"igemm_kernelILi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELb(\d)", name)
    short = ("igemm<BN=%s,PREC=%s,VEC=%s,TAPS=%s,DEFER=%s>" % m.groups()) if m else name[:60]
    print("%-52s vgpr %3s sgpr %3s vspill %3s sspill %3s scratch %5s lds %6s" % (short, g("vgpr_count"), g("sgpr_count"),
          g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
rm -rf $T
