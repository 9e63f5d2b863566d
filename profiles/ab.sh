#!/bin/bash
# profiles/ab.sh -- A/B the fused kernel: one bench.py run per ab/libmirt_*.so (built with
# MIRT_OUT=... csrc/build.sh -D..., or profiles/tools/build_at.sh <commit> for an older tree), same process layout, same box; prints
# launch_ms / ms_per_step / value per variant, headline frame first, then (AB_SCENES="a b") profiles/scene_bench.py per variant.
# A library older than the in-pass resolve needs MIRT_INPASS_RESOLVE=0 (bench.py then keeps the per-ray accumulator): set AB_OLD="name1 name2".
set -uo pipefail
for lib in ab/libmirt_*.so; do
  old=0; for o in ${AB_OLD:-}; do [ "$(basename $lib)" = "$o" ] && old=1; done
  r=$(MIRT_INPASS_RESOLVE=$((1-old)) MIRT_LIB_PATH="$PWD/$lib" timeout -k 10 200 python3 bench.py --steps ${AB_STEPS:-5} --warmup 2 --no-cpu --no-depth5 --no-extras 2>/dev/null)
  echo "$(basename $lib) $(echo "$r" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("launch_ms", d["roofline"]["launch_ms"], "ms_per_step", d["ms_per_step"], "value", d["value"])' 2>/dev/null || echo FAILED)"
done
if [ -n "${AB_SCENES:-}" ]; then for lib in ab/libmirt_*.so; do
  old=0; for o in ${AB_OLD:-}; do [ "$(basename $lib)" = "$o" ] && old=1; done
  echo "== $(basename $lib)"; MIRT_INPASS_RESOLVE=$((1-old)) SCENES="$AB_SCENES" MIRT_LIB_PATH="$PWD/$lib" timeout -k 10 200 python3 profiles/scene_bench.py 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print("  ", d["scene"], d["ms_per_pass"], d["Msamples_s"])'
done; fi
