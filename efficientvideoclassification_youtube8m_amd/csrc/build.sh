#!/bin/bash
# Builds libevc_hip.so (gfx950 only) next to this script's parent package.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${EVC_OUT:-$HERE/../libevc_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=default -Wno-unused-result"
"$HIPCC" $FLAGS "$HERE/evc_gemm.hip" "$HERE/evc_elementwise.hip" "$HERE/evc_dbof.hip" "$HERE/evc_netvlad.hip" "$HERE/evc_moe_norms.hip" "$HERE/evc_optim.hip" -o "$OUT" "$@"
echo "built $OUT"
# Host-side input library (TFRecord / SequenceExample parsing); plain C++, no HIP.
IO_OUT="${EVC_IO_OUT:-$HERE/../libevc_io.so}"
g++ -O3 -std=c++17 -fPIC -shared -Wall "$HERE/evc_io.cpp" -o "$IO_OUT"
echo "built $IO_OUT"
