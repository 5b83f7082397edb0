#!/bin/bash
# VGPR / spill counts of one translation unit's kernels (device-only compile of csrc/<tu>.hip; no GPU needed):
#   scripts/kernel_regs.sh coarse_bf16 [extra hipcc flags]
TU=${1:-coarse_bf16}; shift
D=$(dirname $0)/../codegraph-rust_amd/csrc
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "$@" \
    --cuda-device-only -c $D/$TU.hip -o $T/dev.o || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | grep -E "\.name:|\.vgpr_count|\.vgpr_spill|\.sgpr_spill|private_segment_fixed" | paste - - - - - | sed 's/  */ /g; s/\.private_segment_fixed_size/scratch/; s/_count//g' | cut -c1-220
rm -rf $T
