#!/bin/bash
# profiles/ab_resolve.sh <outdir> -- in-pass resolve A/B on one box: round-3 library (separate copyToPixel), this library with the switch off,
# this library resolving in the pass (no per-ray accumulator); headline frame and the grid scene.  Prints ms_per_step / launch_ms / value.
set -uo pipefail
OUT="${1:-gpurun_out/ab_resolve}"; mkdir -p "$OUT"
run() {  # tag, env..., -- bench args
  local tag="$1"; shift
  local r; r=$(env "$@" timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-depth5 --no-extras ${BENCH_ARGS:-} 2>"$OUT/$tag.err") || { echo "$tag FAILED"; tail -3 "$OUT/$tag.err"; return 1; }
  echo "$r" > "$OUT/$tag.json"
  echo "$tag $(echo "$r" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "launch_ms", d["roofline"]["launch_ms"], "resolve_ms", d["roofline"]["resolve_ms"], "value", d["value"])')"
}
for scene in cornell cornell_teapot3; do
  export BENCH_ARGS="--scene $scene"
  run "${scene}_r3lib" MIRT_INPASS_RESOLVE=0 MIRT_LIB_PATH="$PWD/ab/libmirt_0_r3.so" &&
  run "${scene}_off" MIRT_INPASS_RESOLVE=0 &&
  run "${scene}_keepacu" BENCH_KEEP_ACU=1 &&
  run "${scene}_noacu" X=1 || exit 1
done
