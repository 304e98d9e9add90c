#!/usr/bin/env python3
"""The markdown table of profiles/rNN_bench_matrix.md from the JSON lines scripts/bench_matrix.sh wrote.
usage: matrix_md.py gpurun_out/bench_matrix.jsonl [round] > profiles/rNN_bench_matrix.md"""
import json
import sys

rows = [json.loads(line) for line in open(sys.argv[1]) if line.startswith("{")]
rnd = sys.argv[2] if len(sys.argv) > 2 else "3"
print(f"# Round-{rnd} bench matrix (one MI355X box, `scripts/bench_matrix.sh` = `bench.py --steps 50 --reps 3 --no-cpu-baseline`, median "
      f"repetition; end of round {rnd})\n")
print("`frac` = algorithmic bytes of the reference's unfused data flow per launch / launch time / 8 TB/s (throughput-equivalent, SURVEY 8d);")
print("`traffic frac` = PMC HBM bytes of the committed profile of the SAME kernel(s) / this run's launch time / 8 TB/s (null where no profile "
      "of the launched kernel is committed);")
print("`fused-min frac` = compulsory bytes of the fused data flow / launch time / 8 TB/s. Between boxes the rates differ by +- 2 %.\n")
print("| workload | dtype | flux | tier | kernel launched (last noted) | M cell-updates/s | ms/step | stage kernel(s) ms | frac | traffic frac | "
      "fused-min frac |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for j in rows:
    r, c = j.get("roofline") or {}, j["config"]
    extra = ""
    if "step_ms" in c:
        extra = (f" step {c['step_ms']} ms, cycle {c['cycle_ms']} ms, split {c['cycle_split_ms']}, stepping only "
                 f"{c['stepping_only_M_cell_updates_per_s']} M/s")
    print(f"| {c['workload'].split(':')[0]} | {j['dtype']} | {c.get('flux', '')} | {c.get('kernels', '')} | `{r.get('kernel_launched')}` | "
          f"{j['value']:.0f} | {j['ms_per_step']} | {r.get('avg_launch_ms')} | {r.get('frac')} | {r.get('frac_traffic')} | "
          f"{r.get('frac_fused_min')} |{extra}")
