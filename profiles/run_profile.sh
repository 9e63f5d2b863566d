#!/bin/bash
# profiles/run_profile.sh <tag> -- run on the GPU box (through gpurun) from the repo root.
# Pass 1: kernel trace + stats of the default bench command.  Passes 2..4: PMC counters, each in
# its own run with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
# Outputs land in gpurun_out/prof_<tag>/; copy the summaries into profiles/ afterwards.
set -uo pipefail
TAG="${1:-r1}"
OUT="gpurun_out/prof_${TAG}"
mkdir -p "${OUT}"
export TMPDIR=/tmp
BENCH=(python3 bench.py --steps 2 --warmup 1 --no-cpu --no-depth5 --no-extras)
# the stamp bench.py checks before it quotes these counters: the kernel sources of THIS tree and the workload of THIS command
python3 -c 'import json, bench; print(json.dumps({"csrc_sha256": bench.csrc_sha256(), "workload": "cornell_1920x1080_r256_b8_n1", "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu --no-depth5 --no-extras"}))' > "${OUT}/stamp.json" || exit 10
rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}/trace" -- "${BENCH[@]}" > "${OUT}/trace.log" 2>&1 || exit 11
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "${OUT}/pmc_fetch" -- "${BENCH[@]}" > "${OUT}/pmc_fetch.log" 2>&1 || exit 12
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "${OUT}/pmc_write" -- "${BENCH[@]}" > "${OUT}/pmc_write.log" 2>&1 || exit 13
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "${OUT}/pmc_sq" -- "${BENCH[@]}" > "${OUT}/pmc_sq.log" 2>&1 || exit 14
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "${OUT}/pmc_sq2" -- "${BENCH[@]}" > "${OUT}/pmc_sq2.log" 2>&1 || exit 15
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_VALU --output-format csv -d "${OUT}/pmc_mix" -- "${BENCH[@]}" > "${OUT}/pmc_mix.log" 2>&1 || exit 16
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "${OUT}/pmc_sq3" -- "${BENCH[@]}" > "${OUT}/pmc_sq3.log" 2>&1 || exit 17
rocprofv3 --kernel-trace --pmc SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_DCACHE_REQ --output-format csv -d "${OUT}/pmc_sqc" -- "${BENCH[@]}" > "${OUT}/pmc_sqc.log" 2>&1 || exit 18
find "${OUT}" -name "*.csv" | head -40 > "${OUT}/files.txt"
echo done
