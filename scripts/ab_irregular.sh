show() { python3 - "$1" "$2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(f"{sys.argv[1]:>14}: {j['value']:9.1f} M/s  {j['ms_per_step']:.4f} ms/step  stage {j['roofline']['avg_launch_ms'] if j.get('roofline') else None}  kernel {j['roofline'].get('kernel_launched') if j.get('roofline') else None}")
PY
}
mkdir -p gpurun_out/ab
for w in c5 c5u; do
  for v in 1 0; do
    T8GPU_PATCH_IRREGULAR=$v python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/ab/$w.$v.json 2> gpurun_out/ab/$w.$v.err && show "$w irr=$v" gpurun_out/ab/$w.$v.json || { echo "$w $v FAILED"; tail -5 gpurun_out/ab/$w.$v.err; }
  done
done
