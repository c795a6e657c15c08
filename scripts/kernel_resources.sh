#!/bin/bash
# scripts/kernel_resources.sh <object.o> [name filter]: VGPR / AGPR / SGPR / LDS / scratch of the
# gfx950 kernels in one translation unit's object (code-object metadata), e.g.
#   scripts/kernel_resources.sh climatemachine.jl_amd/csrc/engine_atmos.o DryAtmosILb1ELb1ELb1ELb0ELb0EEELi5ELi5
obj=$(readlink -f "$1")
tmp=$(mktemp -d)
cd $tmp
objcopy -O binary --only-section=.hip_fatbin "$obj" fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=fat.bin --unbundle \
    --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes dev.co 2>/dev/null | python3 -c "
import sys, re
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for k in sys.stdin.read().split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', k).group(1)
    if flt and flt not in name: continue
    g = lambda key: re.search(r'\.' + key + r':\s+(\d+)', k).group(1)
    print('%-100s vgpr %3s agpr %s sgpr %3s lds %6s scratch %s' % (name[:100], g('vgpr_count'), re.match(r':\s+(\d+)', k).group(1), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
" "$2"
rm -rf $tmp
