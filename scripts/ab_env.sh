#!/bin/bash
# On the GPU box: one bench line per value of an environment variable, same box.
# usage: scripts/ab_env.sh VAR "v1 v2 ..." <bench args...>
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VAR=$1; VALS=$2; shift 2
mkdir -p gpurun_out
for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python bench.py "$@" --no-cpu-baseline 2> gpurun_out/ab_env.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d.get('roofline', {})
print(json.dumps({'$VAR': '$v', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'avg_launch_ms': r.get('avg_launch_ms'), 'frac': r.get('frac')}))" | tee -a gpurun_out/ab_env.jsonl || { tail -5 gpurun_out/ab_env.err; exit 1; }
done
