#!/bin/bash
# profiles/lds_pmc.sh <fixture-name> <tag> -- LDS counters of the fused kernel on one fixture scene at 1080p x 16 (scene_bench.py)
set -uo pipefail
export TMPDIR=/tmp
export SCENES="$1"
OUT="gpurun_out/lds_$2"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/a" -- python3 profiles/scene_bench.py > "$OUT/a.log" 2>&1 || { tail -5 "$OUT/a.log"; exit 11; }
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d "$OUT/b" -- python3 profiles/scene_bench.py > "$OUT/b.log" 2>&1 || { tail -5 "$OUT/b.log"; echo "pass b failed"; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_fusedPass<true' in r['Kernel_Name'] or 'k_fusedPassILb1' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
w=sum(agg['SQ_WAVES'])/len(agg['SQ_WAVES'])
print(' '.join(f"{k.replace('SQ_','')}={sum(v)/len(v)/w:.0f}" for k,v in sorted(agg.items())))
PY
