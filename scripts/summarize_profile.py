#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by scripts/profile_gpu.sh into one markdown summary
(per-kernel average duration, HBM bytes per launch with the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md section HBM, and SQ counter ratios)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void t8gpu_hip::", "").replace("t8gpu_hip::", "")
    return name.split("(")[0][:60]


def kernel_stats(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def counters(d):
    """kernel -> counter -> (sum, n) over dispatches."""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


def main(out):
    print(f"# rocprofv3 summary: {os.path.basename(out)}\n")
    for log in ("stats.log",):
        p = os.path.join(out, log)
        if os.path.exists(p):
            for line in open(p):
                if line.startswith("{"):
                    j = json.loads(line)
                    print(f"bench line (under kernel-trace): value={j['value']} {j['unit']}, ms_per_step={j['ms_per_step']}, "
                          f"roofline={json.dumps(j['roofline'])}\n")
    print("## kernel-trace --stats\n\n| kernel | calls | avg us | total % |\n|---|---|---|---|")
    for r in sorted(kernel_stats(os.path.join(out, "stats")), key=lambda r: -float(r["TotalDurationNs"]))[:12]:
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    fetch, write, sq = (counters(os.path.join(out, k)) for k in ("fetch", "write", "sq"))
    print("\n## HBM traffic per launch (PMC, separate passes)\n\nFETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled "
          "(gfx950 counts 128-B read requests as 64 B for wide coalesced streams, MI355X_MICROARCH.md section HBM), so the "
          "read figure is an upper estimate where accesses are narrow.\n\n| kernel | FETCH_SIZE KiB | x2 read MB | WRITE_SIZE KiB | write MB | total MB |\n|---|---|---|---|---|---|")
    for k in sorted(set(fetch) | set(write)):
        fs = fetch[k]["FETCH_SIZE"]
        ws = write[k]["WRITE_SIZE"]
        f = fs[0] / fs[1] if fs[1] else 0.0
        w = ws[0] / ws[1] if ws[1] else 0.0
        print(f"| {k} | {f:.0f} | {2 * f * 1024 / 1e6:.1f} | {w:.0f} | {w * 1024 / 1e6:.1f} | {(2 * f + w) * 1024 / 1e6:.1f} |")
    # machine-readable traffic per launch of the dominant kernel family (bench.py reports it as roofline.traffic)
    # The stage kernels: every family (name without the STAGE argument) of this library's fused-stage kernels that appears in the trace. One
    # RK stage launches one kernel of each family (e.g. k_plain_patch3 + k_plain_persistent on a 3D mesh), so the HBM
    # bytes of a STAGE -- what bench.py divides by its event-timed stage duration -- are the sum over the families of the
    # family's mean bytes per launch (mean over its three stage instances).
    stage_prefixes = ("k_plain_stage", "k_plain_patch", "k_plain_persistent", "k_plain_fused", "k_subgrid_family", "k_subgrid_fused",
                      "k_subgrid444_fused", "k_flux_faces", "k_subgrid_inner")
    def family(k):   # the name without its STAGE template argument (the third): the three stage instances of one kernel
        if "<" not in k:
            return k
        head, args = k.split("<", 1)
        a = [x.strip() for x in args.rstrip(">").split(",")]
        return head + "<" + ", ".join(a[:2] + a[3:]) + ">"
    fams = defaultdict(list)
    for k in sorted(set(fetch) | set(write)):
        if k.startswith(stage_prefixes):
            fams[family(k)].append(k)
    dom = [k for ks in fams.values() for k in ks]
    if dom:
        tot = 0.0
        for ks in fams.values():
            fam_tot = 0.0
            for k in ks:
                fs, ws = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
                fam_tot += (2 * (fs[0] / fs[1] if fs[1] else 0.0) + (ws[0] / ws[1] if ws[1] else 0.0)) * 1024
            tot += fam_tot / len(ks)
        # utilisation from the SQ pass (MI355X_MICROARCH.md: SQ_* count quad-cycles summed over the chip's 1024 SIMDs;
        # GRBM_GUI_ACTIVE is the sum over the 8 XCDs): VALU busy = ACTIVE_INST_VALU * 4 / 1024 / (GUI_ACTIVE / 8)
        def mean(k, n):
            return sq[k][n][0] / sq[k][n][1] if sq[k][n][1] else 0.0
        busy, confl = [], []
        for k in dom:
            gui = mean(k, "GRBM_GUI_ACTIVE") / 8.0
            if gui > 0:
                # (the counter ticks in quad-cycles: an fp32 instruction that occupies the VALU for 2 cycles still counts
                #  one tick, so fp32 kernels read high -- capped at 1; fp64 instructions take the full 4 cycles or more)
                busy.append(min(1.0, mean(k, "SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / gui))
            if mean(k, "SQ_LDS_IDX_ACTIVE") > 0:
                confl.append(mean(k, "SQ_LDS_BANK_CONFLICT") / mean(k, "SQ_LDS_IDX_ACTIVE"))
        with open(os.path.join(out, "traffic.json"), "w") as fjs:
            json.dump({"kernels": dom, "avg_hbm_bytes_per_launch": tot,
                       "valu_busy": round(sum(busy) / len(busy), 4) if busy else None,
                       "lds_conflict_frac": round(sum(confl) / len(confl), 4) if confl else None,
                       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2 (gfx950), KiB -> bytes"}, fjs)
    print("\n## SQ counters per launch (averages)\n")
    names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
             "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
    print("| kernel | " + " | ".join(n.replace("SQ_", "") for n in names) + " |\n|---|" + "---|" * len(names))
    for k in sorted(sq):
        vals = [sq[k][n][0] / sq[k][n][1] if sq[k][n][1] else 0.0 for n in names]
        print(f"| {k} | " + " | ".join(f"{v:.3g}" for v in vals) + " |")
    mix = counters(os.path.join(out, "mix"))
    if mix:
        print("\n## instruction mix per wavefront (SQ_INSTS_* / SQ_WAVES, averages over launches)\n")
        mn = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
        print("| kernel | waves | " + " | ".join(n.replace("SQ_INSTS_", "") for n in mn[1:]) + " |\n|---|" + "---|" * len(mn))
        for k in sorted(mix):
            v = {n: (mix[k][n][0] / mix[k][n][1] if mix[k][n][1] else 0.0) for n in mn}
            w = v["SQ_WAVES"] or 1.0
            print(f"| {k} | {v['SQ_WAVES']:.0f} | " + " | ".join(f"{v[n] / w:.0f}" for n in mn[1:]) + " |")


if __name__ == "__main__":
    main(sys.argv[1].rstrip("/"))
