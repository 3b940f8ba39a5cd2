#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py command on the GPU box (run from the repo root):
#   bash scripts/prof.sh <out-name> <bench.py args...>     -> gpurun_out/<out-name>/run_kernel_stats.csv + a digest on stdout
set -u
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$NAME
mkdir -p "$OUT"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 bench.py "$@" > "$OUT/stdout.log" 2> "$OUT/stderr.log"
python3 scripts/profile_digest.py "$OUT/run_kernel_trace.csv" ${TOP:-40}
