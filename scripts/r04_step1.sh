#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_step1
mkdir -p "$OUT"
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_halo.py -x -q -k "native_stepper or native_subgrid" > "$OUT/tests.log" 2>&1; rc=$?
tail -15 "$OUT/tests.log"
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-default T8GPU_STEPPER_THREADS=0 T8GPU_STEPPER_CLASSES=3 T8GPU_GHOST_WINDOW=0 T8GPU_COMM_CUS=8}; do
  echo "=== $v"
  if [ "$v" = default ]; then env T8GPU_STEPPER_PROFILE=1 timeout -k 10 300 python3 $ROOT/scripts/halo_overhead.py 8 3 200 2>&1 | grep -v "amdgpu.ids\|version\|Hostname\|Librccl"
  else env T8GPU_STEPPER_PROFILE=1 $v timeout -k 10 300 python3 $ROOT/scripts/halo_overhead.py 8 3 200 2>&1 | grep -v "amdgpu.ids\|version\|Hostname\|Librccl"; fi
done > "$OUT/variants.log" 2>&1
cat "$OUT/variants.log"
