#!/bin/bash
# Builds libevc_hip.so (gfx950 only) next to this script's parent package: one hipcc per source file, in parallel, then one link.
# Extra arguments go to every compile (e.g. -DEVC_STAMPS); EVC_OUT / EVC_OBJ_DIR redirect the outputs (A/B builds).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${EVC_OUT:-$HERE/../libevc_hip.so}"
OBJ="${EVC_OBJ_DIR:-$HERE/build}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -Wno-unused-result"
SRCS="evc_gemm evc_gemm_tn evc_lstm_fwd evc_lstm_bwd evc_elementwise evc_dbof evc_netvlad evc_moe_norms evc_optim"
mkdir -p "$OBJ"
# a stale object is only reused when neither its source nor any header changed (and the extra flags are the same)
STAMP="$(cat "$HERE"/*.h "$HERE/../../include/evc.h" | md5sum | cut -d' ' -f1)-$(echo "$FLAGS $*" | md5sum | cut -d' ' -f1)"
pids=()
for s in $SRCS; do
  [ -f "$HERE/$s.hip" ] || continue
  key="$STAMP-$(md5sum < "$HERE/$s.hip" | cut -d' ' -f1)"
  if [ -f "$OBJ/$s.o" ] && [ "$(cat "$OBJ/$s.key" 2>/dev/null)" = "$key" ]; then continue; fi
  ( "$HIPCC" $FLAGS -c "$HERE/$s.hip" -o "$OBJ/$s.o" "$@" && echo "$key" > "$OBJ/$s.key" ) &
  pids+=($!)
done
for p in "${pids[@]:-}"; do [ -z "$p" ] || wait "$p"; done
OBJS=""
for s in $SRCS; do [ -f "$HERE/$s.hip" ] && OBJS="$OBJS $OBJ/$s.o"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
echo "built $OUT"
# Host-side input library (TFRecord / SequenceExample parsing); plain C++, no HIP.
IO_OUT="${EVC_IO_OUT:-$HERE/../libevc_io.so}"
g++ -O3 -std=c++17 -fPIC -shared -Wall "$HERE/evc_io.cpp" -o "$IO_OUT"
echo "built $IO_OUT"
