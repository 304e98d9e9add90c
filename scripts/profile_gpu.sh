#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + PMC passes of one bench.py command.
# usage: scripts/profile_gpu.sh <tag> <bench args...>   -> gpurun_out/prof/<tag>/{stats,fetch,write,sq}/
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
T8GPU_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline > "$OUT/stats.log" 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline > "$OUT/fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline > "$OUT/write.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline > "$OUT/sq.log" 2>&1 || exit 1
# instruction mix (optional: a missing counter name must not void the passes above)
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT/mix" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline > "$OUT/mix.log" 2>&1 || echo "instruction-mix pass failed (see mix.log)"
python3 "$ROOT/scripts/summarize_profile.py" "$OUT" > "$OUT/summary.md" 2>&1
cat "$OUT/summary.md"
# the raw traces are tens of MB per pass (gpurun copies at most 64 MiB back): keep the summary, traffic.json, the stats tables
find "$OUT" -type f \( -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv" -o -name "*_marker_api_trace.csv" -o -name "*.db" \) -delete
du -sh "$OUT" | cut -f1
