#!/bin/bash
# One bench.py line per workload / dtype / flux of the DESIGN.md table (GPU box). usage: scripts/bench_matrix.sh <outfile>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/bench_matrix.jsonl}
: > "$OUT"
run() { python3 "$ROOT/bench.py" --no-cpu-baseline --steps 50 --reps 3 "$@" 2>> "$OUT.err" | grep '^{' >> "$OUT"; echo "done: $*"; }
run --workload c4
run --workload c4 --flux hll
run --workload c4 --flux hllc
run --workload c4 --dtype f32
run --workload c2
run --workload c2 --mode compat
run --workload c1
run --workload c3
run --workload c3 --dtype f64
run --workload c3 --flux hll
run --workload c3q
run --workload c3q --dtype f64
run --workload c5
run --workload c5p
run --workload c5t
run --workload c5u
run --workload c5a --steps 100
python3 - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l); r = j.get("roofline") or {}; c = j["config"]
    print(f"{c['workload'][:28]:28s} {j['dtype']} {c.get('flux','')} {c.get('kernels','')}: {j['value']:9.1f} M/s {j['ms_per_step']:.4f} ms/step "
          f"stage {r.get('avg_launch_ms')} frac {r.get('frac')} min-bytes-frac {r.get('frac_fused_min')} driver {c.get('driver','')} "
          f"{'step_ms %s cycle_ms %s' % (c.get('step_ms'), c.get('cycle_ms')) if 'step_ms' in c else ''}")
PY
