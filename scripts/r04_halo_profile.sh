#!/bin/bash
# GPU box: the numbers of profiles/r04_halo_overhead.md -- the multi-rank step driver on one GPU (RCCL self-exchange), ranks of the
# 8- / 4- / 2-way c4 split, this round's driver and the three-stream pipeline of round 3, plus one kernel timeline.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_halo
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
filt() { grep -v "amdgpu.ids\|version\|Hostname\|Librccl\|^done"; }
for cfg in "8 3 200" "4 1 100" "2 0 100"; do
  tag=${cfg// /_}
  T8GPU_STEPPER_PROFILE=1 timeout -k 10 400 python3 $ROOT/scripts/halo_overhead.py $cfg 2>&1 | filt > "$OUT/lanes_$tag.log"
  T8GPU_STEPPER_PROFILE=1 T8GPU_STEPPER=legacy T8GPU_PLAN_CLASSES=3 timeout -k 10 400 python3 $ROOT/scripts/halo_overhead.py $cfg 2>&1 | filt > "$OUT/legacy_$tag.log"
done
T8GPU_STEPPER_THREADS=0 timeout -k 10 400 python3 $ROOT/scripts/halo_overhead.py 8 3 200 2>&1 | filt > "$OUT/lanes_8_3_200_one_thread.log"
T8GPU_GHOST_WINDOW=0 timeout -k 10 400 python3 $ROOT/scripts/halo_overhead.py 8 3 200 2>&1 | filt > "$OUT/lanes_8_3_200_no_window.log"
T8GPU_PATCH_CHUNK=0 timeout -k 10 400 python3 $ROOT/scripts/halo_overhead.py 8 3 200 2>&1 | filt > "$OUT/lanes_8_3_200_persistent.log"
tail -n 20 "$OUT"/*.log
