#!/bin/bash
# c5 under a range of face caps for the generic tiles (T8GPU_FCAP): the numbers behind the 384-face rule of fused.PlainPlan.
# usage: [WORKLOAD=c5] [FCAPS="320 384 480"] scripts/fcap_scan.sh
for f in ${FCAPS:-200 256 272 300 360 480}; do
  T8GPU_FCAP=$f python3 bench.py --workload ${WORKLOAD:-c5} --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('fcap', $f, j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline'].get('kernel_launched'))"
done
