#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace CSV: the last `n` kernels of the trace as a markdown table
(start / end / duration in microseconds from the first kernel of the window, queue id, short kernel name with
its grid size), plus per-kernel-name mean durations over the window and the window's period per RK stage.
usage: trace_timeline.py <dir with *_kernel_trace.csv> [n=48] [skip_last=0]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void t8gpu_hip::", "").replace("t8gpu_hip::", "")
    if "nccl" in name.lower() or "rccl" in name.lower():
        return "RCCL " + name.split("(")[0][:40]
    return name.split("(")[0][:56]


def main():
    d = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if skip:
        rows = rows[:-skip]
    win = rows[-n:]
    t0 = int(win[0]["Start_Timestamp"])
    queues = {}
    print("| start | end | duration | queue | kernel | workgroups |\n|---|---|---|---|---|---|")
    per = defaultdict(list)
    for r in win:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        q = queues.setdefault(r["Queue_Id"], len(queues) + 1)
        wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) if r.get("Grid_Size") and r.get("Workgroup_Size") else 0
        print(f"| {s:.1f} | {e:.1f} | {e - s:.1f} | {q} | {short(r['Kernel_Name'])} | {wgs} |")
        per[(short(r["Kernel_Name"]), wgs)].append(e - s)
    print("\n| kernel | workgroups | launches | mean us |\n|---|---|---|---|")
    for (k, w), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print(f"| {k} | {w} | {len(v)} | {sum(v) / len(v):.1f} |")
    span = (int(win[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"\nwindow: {len(win)} kernels in {span:.1f} us")


if __name__ == "__main__":
    main()
