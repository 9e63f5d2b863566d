#!/bin/bash
# profiles/frame_pmc.sh <tag> -- SQ counters of the frame kernels (BASELINE configs 2 / 3 at full size: profiles/frame_bench.py), per wave
set -uo pipefail
export TMPDIR=/tmp
OUT="gpurun_out/frames_pmc_$1"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/a" -- python3 profiles/frame_bench.py > "$OUT/a.log" 2>&1 || exit 11
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d "$OUT/b" -- python3 profiles/frame_bench.py > "$OUT/b.log" 2>&1 || exit 12
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'meshTrace' in r['Kernel_Name'] or 'molTrace' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), v in sorted(agg.items()):
    w=sum(v['SQ_WAVES'])/len(v['SQ_WAVES'])
    print(k, 'grid', g, 'dispatches', len(v['SQ_WAVES']), ' '.join(f"{c.replace('SQ_','')}={sum(x)/len(x)/w:.0f}" for c,x in sorted(v.items())))
PY
