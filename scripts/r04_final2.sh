#!/bin/bash
# GPU box, round 4 final batch 2: profiles of c3 fp64, c5 fp64, c5u fp64
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04_final
for spec in "r04_c3_f64_family --workload c3 --dtype f64" "r04_c5_f64 --workload c5 --dtype f64" "r04_c5u_f64 --workload c5u --dtype f64"; do
  set -- $spec; tag=$1; shift
  timeout -k 10 420 scripts/profile_gpu.sh $tag "$@" --steps 20 --reps 2 > gpurun_out/r04_final/profile_$tag.log 2>&1; echo "$tag rc=$?"; tail -2 gpurun_out/r04_final/profile_$tag.log
done
