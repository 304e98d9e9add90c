#!/bin/bash
# On the GPU box: patch-kernel tests, then bench lines with and without the patch path on ONE box.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_patch.py -x -q > gpurun_out/r3_patch_tests.log 2>&1 || { tail -30 gpurun_out/r3_patch_tests.log; exit 1; }
tail -3 gpurun_out/r3_patch_tests.log
: > gpurun_out/r3_ab_patch.jsonl
for wl in c4 c2 c1; do
  for p in 1 0; do
    T8GPU_PATCH=$p timeout -k 10 300 python bench.py --workload $wl --steps 50 --reps 3 --no-cpu-baseline 2> gpurun_out/r3_ab_patch.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d.get('roofline', {})
print(json.dumps({'wl': '$wl', 'patch': $p, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel_ms': r.get('kernel_ms'), 'frac': r.get('frac')}))" | tee -a gpurun_out/r3_ab_patch.jsonl || exit 1
  done
done
