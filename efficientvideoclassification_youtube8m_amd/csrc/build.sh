#!/bin/bash
# Builds libevc_hip.so (gfx950 only) next to this script's parent package.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${EVC_OUT:-$HERE/../libevc_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=default -Wno-unused-result"
"$HIPCC" $FLAGS "$HERE/evc_gemm.hip" "$HERE/evc_elementwise.hip" -o "$OUT" "$@"
echo "built $OUT"
