#!/bin/bash
# Register / spill / LDS metadata of the kernels in one object of the build (code-object notes), e.g.
#   tools/kernel_regs.sh vibo_msplit_a 'msplit_kernelILi2ELb1ELi0ELb0ELb1ELb0E'
# usage: kernel_regs.sh <object stem under csrc/build> [substring of the mangled kernel name]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/variational-item-response-theory-public_amd/csrc/build/$1.o
TMP=$(mktemp -d)
cd $TMP
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin $OBJ
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes dev.co | python3 -c "
import sys, re
txt = sys.stdin.read()
pat = sys.argv[1] if len(sys.argv) > 1 else ''
for blk in txt.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk)
    if not name or pat not in name.group(1): continue
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    print(name.group(1)[:110], 'vgpr', g('vgpr_count'), 'agpr', blk.split()[0], 'sgpr', g('sgpr_count'), 'vspill', g('vgpr_spill_count'), 'sspill', g('sgpr_spill_count'), 'scratch', g('private_segment_fixed_size'), 'lds', g('group_segment_fixed_size'))
" "$2"
rm -rf $TMP
