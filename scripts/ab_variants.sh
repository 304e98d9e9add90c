#!/bin/bash
# A/B timing of experiment builds of the HIP library (t8gpu_amd/build.py: build_hip(variant=...)) on the GPU box.
# usage: scripts/ab_variants.sh "<bench args>" variant1 variant2 ...   ("default" = the product library)
ARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$ROOT/gpurun_out/ab"
for v in "$@"; do
  if [ "$v" = default ]; then unset T8GPU_HIP_LIB; else export T8GPU_HIP_LIB=$ROOT/t8gpu_amd/lib/variants/libt8gpu_hip_$v.so; fi
  python3 "$ROOT/bench.py" $ARGS --no-cpu-baseline > "$ROOT/gpurun_out/ab/$v.json" 2> "$ROOT/gpurun_out/ab/$v.err" || { echo "$v FAILED"; tail -3 "$ROOT/gpurun_out/ab/$v.err"; continue; }
  python3 - "$v" "$ROOT/gpurun_out/ab/$v.json" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = j["roofline"]
print(f"{sys.argv[1]:>16}: {j['value']:9.1f} M/s  {j['ms_per_step']:.4f} ms/step (min {j['ms_per_step_min']:.4f})  stage kernel {r['avg_launch_ms']:.4f} ms  finite={j['config']['finite']}")
PY
done
