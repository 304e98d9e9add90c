#!/bin/bash
# GPU box: host-cost profile + kernel timeline of the multi-rank step driver (rank 3 of 8, RCCL self-exchange)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/halo_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
T8GPU_STEPPER_PROFILE=1 date +%T; timeout -k 10 500 python3 "$ROOT/scripts/halo_overhead.py" 8 3 200 > "$OUT/overhead.log" 2>&1 || { tail -20 "$OUT/overhead.log"; exit 1; }
cat "$OUT/overhead.log"; date +%T
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/scripts/halo_overhead.py" 8 3 30 > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
python3 "$ROOT/scripts/trace_timeline.py" "$OUT/trace" 60 140 > "$OUT/timeline.md" 2>&1
cat "$OUT/timeline.md"
find "$OUT/trace" -type f -name "*.csv" -size +2M -delete
