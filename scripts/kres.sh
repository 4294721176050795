#!/bin/bash
# usage: bash scripts/kres.sh <file.hip> [grep pattern]   -- per-kernel register / scratch / spill figures from the code object
src=$1; pat=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 $KRES_EXTRA --offload-arch=gfx950 -fno-gpu-rdc -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=fast --cuda-device-only -c $src -o $tmp/a.bundle -save-temps=obj 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/a.bundle --output=$tmp/a.elf --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/a.elf | grep -E "^\s+\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|private_segment_fixed|vgpr_spill|sgpr_spill" | paste - - - - - - - | sed 's/\s\+/ /g' | grep -E "$pat" | sed -E "s/\.name: [^ ]+ //"
cp $tmp/*.s /tmp/last_kernel.s 2>/dev/null
rm -rf $tmp
