#!/bin/bash
# Same-box A/B of the working tree against a checkout of an earlier commit in .ab_prev/ (git worktree add .ab_prev <rev>,
# built there with `python -m t8gpu_amd.build`): alternates the two, prints value / stage-kernel time of each run.
# usage: scripts/ab_prev.sh "<bench args>" [rounds]
ARGS=$1; ROUNDS=${2:-2}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$ROOT/gpurun_out/ab"
show() {
python3 - "$1" "$2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(f"{sys.argv[1]:>8}: {j['value']:9.1f} M/s  {j['ms_per_step']:.4f} ms/step (min {j['ms_per_step_min']:.4f})  stage kernel {j['roofline']['avg_launch_ms']:.4f} ms")
PY
}
for i in $(seq $ROUNDS); do
  (cd "$ROOT/.ab_prev" && python3 bench.py $ARGS --no-cpu-baseline > "$ROOT/gpurun_out/ab/prev.json" 2> "$ROOT/gpurun_out/ab/prev.err") && show prev "$ROOT/gpurun_out/ab/prev.json" || { echo "prev FAILED"; tail -3 "$ROOT/gpurun_out/ab/prev.err"; }
  (cd "$ROOT" && python3 bench.py $ARGS --no-cpu-baseline > "$ROOT/gpurun_out/ab/new.json" 2> "$ROOT/gpurun_out/ab/new.err") && show new "$ROOT/gpurun_out/ab/new.json" || { echo "new FAILED"; tail -3 "$ROOT/gpurun_out/ab/new.err"; }
done
