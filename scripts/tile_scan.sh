#!/bin/bash
# c5 (or WORKLOAD) under pairs of tile caps "tmax:fcap" for the generic tiles (T8GPU_TMAX / T8GPU_FCAP).
# usage: [WORKLOAD=c5] scripts/tile_scan.sh 72:480 96:480 256:384
for tf in "$@"; do
  t=${tf%%:*}; f=${tf##*:}
  T8GPU_TMAX=$t T8GPU_FCAP=$f python3 bench.py --workload ${WORKLOAD:-c5} --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('tmax $t fcap $f', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline'].get('kernel_launched'))"
done
