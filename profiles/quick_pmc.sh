#!/bin/bash
# profiles/quick_pmc.sh <lib.so> <tag> -- SQ instruction counters of bench.py's fused kernel for one A/B build
set -uo pipefail
export TMPDIR=/tmp
export MIRT_LIB_PATH="$PWD/$1"
OUT="gpurun_out/quick_$2"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/a" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-depth5 --no-extras > "$OUT/a.log" 2>&1 || exit 11
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d "$OUT/b" -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-depth5 --no-extras > "$OUT/b.log" 2>&1 || exit 12
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_fusedPass' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
w=sum(agg['SQ_WAVES'])/len(agg['SQ_WAVES'])
print(' '.join(f"{k.replace('SQ_','')}={sum(v)/len(v)/w:.0f}" for k,v in sorted(agg.items())))
PY
