#!/bin/bash
# profiles/round_final.sh <tag> -- the round's closing measurement on the GPU box (through gpurun): full GPU test suite, the default
# bench line, the grid-scene bench line, every fixture scene, the kernel-by-kernel path (with and without command-stream fusion),
# then the rocprofv3 passes of the default bench (run_profile.sh) and the SQ counters of the grid scene (scene_pmc.sh).
set -uo pipefail
TAG="${1:-final}"
OUT="gpurun_out/${TAG}"; mkdir -p "${OUT}"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "${OUT}/gpu_tests.txt" 2>&1 || { tail -20 "${OUT}/gpu_tests.txt"; exit 21; }
tail -1 "${OUT}/gpu_tests.txt"
timeout -k 10 600 python bench.py > "${OUT}/bench.json" 2> "${OUT}/bench.err" || exit 22
timeout -k 10 300 python bench.py --scene cornell_teapot3 --no-cpu > "${OUT}/bench_teapot3.json" 2> "${OUT}/bench_teapot3.err" || exit 23
timeout -k 10 300 python profiles/scene_bench.py > "${OUT}/scene_bench.txt" 2>&1 || exit 24
# the second numerics contract (libmirt_default.so: the reference as its own host builds it), same two bench lines
MIRT_CONTRACT=default timeout -k 10 300 python bench.py --no-cpu --no-extras --no-depth5 > "${OUT}/bench_default_contract.json" 2> "${OUT}/bench_default_contract.err" || exit 32
MIRT_CONTRACT=default timeout -k 10 300 python bench.py --scene cornell_teapot3 --no-cpu --no-extras > "${OUT}/bench_default_contract_teapot3.json" 2>> "${OUT}/bench_default_contract.err" || exit 33
timeout -k 10 300 python profiles/granular_bench.py > "${OUT}/granular.txt" 2>&1 || exit 25
timeout -k 10 300 python profiles/grid_split.py > "${OUT}/grid_split.txt" 2>&1 || exit 26
timeout -k 10 900 bash profiles/run_profile.sh "${TAG}" > "${OUT}/run_profile.log" 2>&1 || exit 27
timeout -k 10 600 bash profiles/scene_pmc.sh cornell_teapot3_32x24_r4 "${TAG}_teapot3" > "${OUT}/teapot3_pmc.txt" 2>&1 || exit 28
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}/trace_teapot3" -- python3 bench.py --scene cornell_teapot3 --steps 2 --warmup 1 --no-cpu --no-extras > "${OUT}/trace_teapot3.log" 2>&1 || exit 29
SCENE=cornell_teapot3_32x24_r4 RPPS=16 timeout -k 10 600 python profiles/ref_gpu_bench.py > "${OUT}/ref_gpu_teapot3.txt" 2>&1 || exit 30
timeout -k 10 600 python profiles/ref_gpu_bench.py > "${OUT}/ref_gpu_bench.txt" 2>&1 || exit 31
echo "all done"
