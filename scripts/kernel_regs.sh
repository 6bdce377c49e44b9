#!/bin/bash
# VGPR / spill / scratch of the kernels in a built library whose (mangled) name contains $2:  scripts/kernel_regs.sh lib.so pattern
L=/opt/rocm/lib/llvm/bin; so=${1:-fisr_amd/libfisr_hip.so}; pat=${2:-conv3x3}
d=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$so" $d/fat.bin
$L/clang-offload-bundler --unbundle --type=o --input=$d/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$d/dev.co
$L/llvm-readelf --notes $d/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+)(.*?)(?=\.name:\s+_Z|\Z)', txt, re.S):
    name=m.group(1)
    if '$pat' not in name: continue
    body=m.group(2)
    g=lambda k:(re.search(k+r':\s+(\d+)',body) or [0,'?'])[1]
    print(name[:110], 'vgpr',g(r'\.vgpr_count'),'spill',g(r'\.vgpr_spill_count'),'scratch',g(r'\.private_segment_fixed_size'))
" | sort
rm -rf $d
