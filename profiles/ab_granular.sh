#!/bin/bash
# profiles/ab_granular.sh -- A/B the reference-shaped path: per-kernel average durations (rocprofv3 --kernel-trace --stats) of
# profiles/granular_bench.py at 64 rays per pixel, one run per ab/libmirt_*.so, same box.
set -uo pipefail
export TMPDIR=/tmp RPPS=64
for lib in ab/libmirt_*.so; do
  tag=$(basename $lib .so); out=gpurun_out/abg_$tag; mkdir -p $out
  MIRT_LIB_PATH="$PWD/$lib" rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 profiles/granular_bench.py > $out/log.txt 2>&1 || { echo "$tag FAILED"; continue; }
  echo "== $tag $(grep granular_ms $out/log.txt | tail -1 | cut -c1-150)"
  python3 - $out <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/trace/*/*kernel_stats.csv')[0]
print('   '+'  '.join(f"{r['Name'].split('(')[0].replace('void pt::','').replace('pt::','')}={float(r['AverageNs'])/1e6:.2f}" for r in list(csv.DictReader(open(f)))[1:10]))
PY
done
