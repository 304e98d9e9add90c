#!/bin/bash
# GPU box, round 4 final batch 1: the GPU test-suite, the default bench line, the c4 fp64 profile (stats + PMC passes)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r04_final
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_final/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r04_final/gpu_tests.log
timeout -k 10 300 python3 bench.py > gpurun_out/r04_final/bench_default.json 2> gpurun_out/r04_final/bench_default.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r04_final/bench_default.json
timeout -k 10 600 scripts/profile_gpu.sh r04_c4_f64_stage --workload c4 --steps 20 --reps 2 > gpurun_out/r04_final/profile_c4.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/r04_final/profile_c4.log
