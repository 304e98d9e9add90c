#!/bin/bash
# GPU box: T8GPU_STREAM_MB=0 (ordinary accesses everywhere) against the default (non-temporal stage results / previous-state loads
# where the 15 planes of a stage exceed 384 MB: flux_math.hpp, stream_store), same box, alternating.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
one() {
  python3 bench.py --no-cpu-baseline --steps 50 --reps 3 "${@:2}" 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j.get('roofline') or {}; print('$1', '${*:2}', j['value'], j['ms_per_step'], r.get('avg_launch_ms'))"
}
for spec in "--workload c4" "--workload c2" "--workload c3" "--workload c3 --dtype f64" "--workload c3q" "--workload c5" "--workload c5u" "--workload c5t" "--workload c4 --dtype f32"; do
  for rep in 1 2; do
    T8GPU_STREAM_MB=0 one never $spec
    one auto $spec
  done
done
