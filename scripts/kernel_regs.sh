#!/bin/bash
# Register / scratch / LDS use of every kernel of one source: scripts/kernel_regs.sh ransac_kernels.hip [pattern]
# (compiles to assembly with the build's flags and reads the .amdhsa_ directives)
src=$1; pat=${2:-.}
root=$(cd "$(dirname "$0")/.." && pwd)
extra=""; [ "$src" = "ransac_kernels.hip" ] && extra="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1"
out=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt $extra \
    -S --cuda-device-only -o $out/k.s $root/cybervision_amd/csrc/$src 2>/dev/null || exit 1
awk '/\.amdhsa_kernel /{k=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_accum_offset/{a=$2} /\.amdhsa_private_segment_fixed_size/{s=$2} /\.amdhsa_group_segment_fixed_size/{l=$2} /\.end_amdhsa_kernel/{printf "%-110s vgpr+agpr %4s (arch %4s) scratch %5s lds %6s\n", k, v, a, s, l}' $out/k.s | c++filt | grep -E "$pat"
rm -rf $out
