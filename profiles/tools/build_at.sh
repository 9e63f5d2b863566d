#!/bin/bash
# profiles/tools/build_at.sh <commit> <out.so> [-D...] -- libmirt.so from the csrc/ of another commit (A/B baselines), built out of tree
set -euo pipefail
C="$1"; OUT="$(realpath -m "$2")"; shift 2
T="$(mktemp -d)"
git -C "$(dirname "$0")/../.." archive "$C" 2015-raytracing_amd/csrc include | tar -x -C "$T"
MIRT_OUT="$OUT" bash "$T/2015-raytracing_amd/csrc/build.sh" "$@"
rm -rf "$T"
