#!/bin/bash
# Per-kernel registers / LDS / spills of a compiled object: tools/kres_obj.sh ishara_amd/csrc/gemm_as.o [name filter]
set -e
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.o --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.o > $T/notes.txt
python3 "$(dirname "$0")/kres.py" $T/notes.txt | grep -- "${2:-.}"
rm -rf $T
