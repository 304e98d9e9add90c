#!/bin/bash
# GPU box: kernel timeline of the multi-rank step driver (rank 3 of 8, RCCL self-exchange). usage: r04_trace.sh <tag> [ENV=VAL ...]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/halo_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export T8GPU_HALO_ONLY=c
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/scripts/halo_overhead.py" ${HALO_ARGS:-8 3 60} > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
grep -v "amdgpu.ids" "$OUT/trace.log" | tail -8
python3 "$ROOT/scripts/trace_timeline.py" "$OUT/trace" 48 8 > "$OUT/timeline.md" 2>&1
cat "$OUT/timeline.md"
find "$OUT/trace" -type f -name "*.csv" -size +2M -delete
