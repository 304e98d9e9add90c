#!/usr/bin/env python3
"""Copies one profile of scripts/profile_gpu.sh from gpurun_out/prof/<tag>/ into profiles/ (tracked) and
registers its PMC traffic in profiles/traffic.json under <workload>|<dtype>|<flux>|<mode>, the key bench.py
looks up for roofline.traffic.
usage: commit_profile.py <tag> <workload> <dtype> <flux> <mode>"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag, workload, dtype, flux, mode = sys.argv[1:6]
    src = os.path.join(ROOT, "gpurun_out", "prof", tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "summary.md"), os.path.join(dst, tag + ".md"))
    stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    tj = os.path.join(src, "traffic.json")
    if os.path.exists(tj):
        t = json.load(open(tj))
        path = os.path.join(dst, "traffic.json")
        allt = json.load(open(path)) if os.path.exists(path) else {}
        from t8gpu_amd.build import kernel_source_hash
        # (the hash of the kernel sources the profiled library was built from: bench.py reports this record's figures only
        #  while the sources are the same -- a body change under an unchanged kernel name would otherwise go unnoticed)
        allt[f"{workload}|{dtype}|{flux}|{mode}"] = {"hbm_bytes_per_launch": int(t["avg_hbm_bytes_per_launch"]), "kernels": t["kernels"],
                                                     "source": f"profiles/{tag}.md", "method": t["method"],
                                                     "valu_busy": t.get("valu_busy"), "lds_conflict_frac": t.get("lds_conflict_frac"),
                                                     "kernel_source_hash": kernel_source_hash()}
        json.dump(allt, open(path, "w"), indent=1)
        print(f"{workload}|{dtype}|{flux}|{mode}: {int(t['avg_hbm_bytes_per_launch'])} B per launch")


if __name__ == "__main__":
    main()
