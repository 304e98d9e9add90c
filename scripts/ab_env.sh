#!/bin/bash
# like ab_variants.sh, but varies an environment variable for one library variant
# usage: scripts/ab_env.sh "<bench args>" <variant> <ENVVAR> value1 value2 ...
ARGS=$1; V=$2; VAR=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$ROOT/gpurun_out/ab"
if [ "$V" = default ]; then unset T8GPU_HIP_LIB; else export T8GPU_HIP_LIB=$ROOT/t8gpu_amd/lib/variants/libt8gpu_hip_$V.so; fi
for val in "$@"; do
  export $VAR=$val
  python3 "$ROOT/bench.py" $ARGS --no-cpu-baseline > "$ROOT/gpurun_out/ab/$V.$val.json" 2> "$ROOT/gpurun_out/ab/$V.$val.err" || { echo "$V $VAR=$val FAILED"; tail -3 "$ROOT/gpurun_out/ab/$V.$val.err"; continue; }
  python3 - "$V $VAR=$val" "$ROOT/gpurun_out/ab/$V.$val.json" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = j["roofline"]
print(f"{sys.argv[1]:>40}: {j['value']:9.1f} M/s  {j['ms_per_step']:.4f} ms/step  stage kernel {r['avg_launch_ms']:.4f} ms")
PY
done
