#!/bin/bash
# GPU box, round 4 closing batches (the kernel sources changed after r04_final1/2: launcher settings became thread-safe).
# usage: scripts/r04_final3.sh tests | profA | profB | profC
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04_final
prof() {
  for spec in "$@"; do
    set -- $spec; tag=$1; shift
    timeout -k 10 380 scripts/profile_gpu.sh $tag "$@" --steps 20 --reps 2 > gpurun_out/r04_final/profile_$tag.log 2>&1; echo "$tag rc=$?"; tail -2 gpurun_out/r04_final/profile_$tag.log
  done
}
case "$1" in
  tests)
    timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_final/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r04_final/gpu_tests.log
    timeout -k 10 250 python3 bench.py > gpurun_out/r04_final/bench_default.json 2> gpurun_out/r04_final/bench_default.err; echo "bench rc=$?"; tail -c 1200 gpurun_out/r04_final/bench_default.json ;;
  profA) prof "r04_c4_f64_stage --workload c4" "r04_c3_f64_family --workload c3 --dtype f64" "r04_c5_f64 --workload c5 --dtype f64" ;;
  profC) prof "r04_c4_f32_stage --workload c4 --dtype f32" "r04_c3_f32_family --workload c3 --dtype f32" "r04_c2_f64_stage --workload c2" ;;
  profB) prof "r04_c5u_f64 --workload c5u --dtype f64" "r04_c5p_f64 --workload c5p --dtype f64" "r04_c5t_f64 --workload c5t --dtype f64" ;;
esac
