#!/bin/bash
# profiles/ab_scene.sh -- A/B over ab/libmirt_*.so on the grid scenes (scene_bench.py at 1080p x 16), same box.
set -uo pipefail
for lib in ab/libmirt_*.so; do
  echo "== $(basename $lib)"; SCENES="${AB_SCENES:-cornell_teapot3_32x24_r4,own_gems_48x36_r4}" MIRT_LIB_PATH="$PWD/$lib" timeout -k 10 200 python3 profiles/scene_bench.py 2>/dev/null | python3 -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print("  ", d["scene"], d["ms_per_pass"], d["Msamples_s"])'
done
