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
SRC=("${HERE}/mirt_abi.cpp" "${HERE}/pt_kernels_granular.hip" "${HERE}/pt_kernels_fused.hip" "${HERE}/pt_kernels_frame.hip" "${HERE}/pt_grid_build.hip")
# libmirt_default.so -- the SECOND numerics contract: the reference as its own host builds it, program.build() without options (A10 code.js:599),
# i.e. AMD's default 2.5-ulp division and 3-ulp sqrt.  Same sources: -fno-hip-fp32-correctly-rounded-divide-sqrt makes every `/` and sqrt the
# sequence AMD's OpenCL compiler emits without -cl-fp32-correctly-rounded-divide-sqrt (the same LLVM lowering: frexp-scaled v_rcp_f32, ldexp-scaled
# v_sqrt_f32); PT_PLAIN_DIV spells every division of the kernels as a division on the reference's operands -- the refined-reciprocal forms of the
# optimistic kernel are exact-division identities, not this contract's -- and keeps that kernel's STRUCTURE (candidate lists, plane list, shared-test walk,
# guard windows with the exact kernel behind them), which does not depend on how a quotient is rounded.  Same C ABI, a drop-in file;
# pyhost picks it with MIRT_CONTRACT=default.  Built only for the default output path (an A/B build names its own).
if [ -z "${MIRT_OUT:-}" ] && [ "${MIRT_SKIP_DEFAULT_CONTRACT:-0}" != 1 ]; then
    "${HIPCC}" "${FLAGS[@]/-fhip-fp32-correctly-rounded-divide-sqrt/-fno-hip-fp32-correctly-rounded-divide-sqrt}" -DPT_PLAIN_DIV=1 \
        -shared -o "${HERE}/../libmirt_default.so" "${SRC[@]}" "$@" &
fi
"${HIPCC}" "${FLAGS[@]}" -shared -o "${OUT}" "${SRC[@]}" "$@"
wait
echo "built ${OUT}"
