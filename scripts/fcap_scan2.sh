for f in 480 380; do
  T8GPU_PATCH=0 T8GPU_FCAP=$f python3 bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('c5 nopatch fcap', $f, j['value'], j['roofline']['avg_launch_ms'], j['roofline'].get('kernel_launched'))"
  T8GPU_FCAP=$f python3 bench.py --workload c5a --steps 80 --warmup 5 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); c=j['config']; print('c5a fcap', $f, j['value'], c['step_ms'], c['cycle_ms'], c['stepping_only_M_cell_updates_per_s'])"
  T8GPU_FCAP=$f python3 bench.py --workload c5u --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('c5u fcap', $f, j['value'], j['roofline']['avg_launch_ms'])"
done
