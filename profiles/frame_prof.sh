#!/bin/bash
# profiles/frame_prof.sh <tag> -- rocprofv3 kernel trace of the full-size Assign04 / Assign07 frame tests (BASELINE configs 2 and 3)
set -uo pipefail
export TMPDIR=/tmp
OUT="gpurun_out/frames_$1"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 -m pytest tests/test_frames.py -q -m gpu -k "full_size or variants" > "$OUT/trace.log" 2>&1 || exit 11
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs cat
