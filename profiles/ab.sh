#!/bin/bash
# profiles/ab.sh -- A/B the fused kernel: one bench.py run per ab/libmirt_*.so (built with
# MIRT_OUT=... csrc/build.sh -D...), same process layout, same box; prints launch_ms per variant.
set -uo pipefail
for lib in ab/libmirt_*.so; do
  r=$(MIRT_LIB_PATH="$PWD/$lib" timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-depth5 2>/dev/null)
  echo "$(basename $lib) $(echo "$r" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["launch_ms"], d["value"])' 2>/dev/null || echo FAILED)"
done
if [ "${AB_EXACT_ONLY:-0}" = 1 ]; then for lib in ab/libmirt_*.so; do
  r=$(MIRT_EXACT_ONLY=1 MIRT_LIB_PATH="$PWD/$lib" timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-depth5 2>/dev/null)
  echo "exact_only $(basename $lib) $(echo "$r" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["launch_ms"], d["value"])' 2>/dev/null || echo FAILED)"
done; fi
if [ -n "${AB_SCENES:-}" ]; then for lib in ab/libmirt_*.so; do
  echo "== $(basename $lib)"; SCENES="$AB_SCENES" MIRT_LIB_PATH="$PWD/$lib" timeout -k 10 200 python3 profiles/scene_bench.py 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print("  ", d["scene"], d["ms_per_pass"], d["Msamples_s"])'
done; fi
