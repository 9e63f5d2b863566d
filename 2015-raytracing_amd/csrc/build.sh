#!/bin/bash
# Builds libmirt.so (HIP kernels + C ABI) for gfx950, in-tree.  Cross-compiles without a GPU.
#   -ffp-contract=off / no fast-math : part of the numerics contract (pt_numerics.hpp)
#   -fhip-fp32-correctly-rounded-divide-sqrt : the hipcc default, spelled out because parity depends on it
#   -fno-slp-vectorize : -O3 packs adjacent scalar fp32 mul/add into v_pk_mul_f32 / v_pk_add_f32, which are slower than the two
#                        scalar ops they replace on gfx950 (MI355X_MICROARCH.md, cycle constants); measured on k_fusedPass: 196.7 -> 178.3 ms
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${MIRT_OUT:-${HERE}/../libmirt.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden
       -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize
       -Wall -Wextra -Wno-unused-parameter)
"${HIPCC}" "${FLAGS[@]}" -shared -o "${OUT}" \
    "${HERE}/mirt_abi.cpp" "${HERE}/pt_kernels_granular.hip" "${HERE}/pt_kernels_fused.hip" "${HERE}/pt_kernels_frame.hip" "${HERE}/pt_grid_build.hip" "$@"
echo "built ${OUT}"
