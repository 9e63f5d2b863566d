#!/bin/bash
# Builds mirt.node (N-API addon over libmirt.so) next to libmirt.so.  Needs the Node headers
# (/usr/include/node in this image); skipped with a note when they are absent.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="${HERE}/../.."
INC="${NODE_INCLUDE:-/usr/include/node}"
if [ ! -f "${INC}/node_api.h" ]; then echo "node_api.h not found under ${INC}: mirt.node not built"; exit 0; fi
g++ -std=c++17 -O2 -fPIC -shared -Wall -Wextra -Wno-unused-parameter -DNODE_GYP_MODULE_NAME=mirt -I"${INC}" \
    -o "${PKG}/mirt.node" "${HERE}/mirt_napi.cc" -L"${PKG}" -lmirt -Wl,-rpath,'$ORIGIN'
echo "built ${PKG}/mirt.node"
# the same addon over libmirt_default.so (the reference's own build contract, csrc/build.sh): host/webcl.js loads it when MIRT_CONTRACT=default
if [ -f "${PKG}/libmirt_default.so" ]; then
    g++ -std=c++17 -O2 -fPIC -shared -Wall -Wextra -Wno-unused-parameter -DNODE_GYP_MODULE_NAME=mirt_default -I"${INC}" \
        -o "${PKG}/mirt_default.node" "${HERE}/mirt_napi.cc" -L"${PKG}" -lmirt_default -Wl,-rpath,'$ORIGIN'
    echo "built ${PKG}/mirt_default.node"
fi
