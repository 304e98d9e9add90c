#!/bin/bash
# Same-box A/B of the 3D patch forms: T8GPU_PATCH_IRREGULAR = 1 (regular + irregular patches, two launches), all (every patch in
# the irregular form, one launch), 0 (regular patches only; the other blocks stay generic tiles).
# usage: scripts/ab_irregular.sh [workloads...]
show() { python3 - "$1" "$2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = j.get("roofline") or {}
c = j["config"]
extra = f"step {c['step_ms']} ms, stepping {c['stepping_only_M_cell_updates_per_s']} M/s, cycle {c['cycle_ms']} ms" if "step_ms" in c else f"stage {r.get('avg_launch_ms')} ms"
print(f"{sys.argv[1]:>14}: {j['value']:9.1f} M/s  {j['ms_per_step']:.4f} ms/step  {extra}")
PY
}
mkdir -p gpurun_out/ab
for w in ${@:-c5 c5u}; do
  for v in 1 all 0; do
    args="--workload $w --no-cpu-baseline"; [ $w = c5a ] && args="--workload c5a --steps 80 --warmup 5"
    T8GPU_PATCH_IRREGULAR=$v python3 bench.py $args > gpurun_out/ab/$w.$v.json 2> gpurun_out/ab/$w.$v.err && show "$w irr=$v" gpurun_out/ab/$w.$v.json || { echo "$w $v FAILED"; tail -5 gpurun_out/ab/$w.$v.err; }
  done
done
