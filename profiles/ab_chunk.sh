#!/bin/bash
# profiles/ab_chunk.sh <lib.so> -- k_fusedPassQ's wave chunk (MIRT_Q_CHUNK) swept on the headline frame
set -uo pipefail
for c in 512 1024 2048 4096 8192; do
  r=$(MIRT_Q_CHUNK=$c MIRT_LIB_PATH="$PWD/$1" timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-depth5 2>/dev/null)
  echo "chunk $c $(echo "$r" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["launch_ms"], d["value"])' 2>/dev/null || echo FAILED)"
done
