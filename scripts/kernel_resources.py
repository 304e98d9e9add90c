#!/usr/bin/env python3
"""usage: scripts/kernel_resources.py <file.hip> [extra hipcc flags]
One line per kernel of the file: VGPRs, spills, SGPRs, LDS and the occupancy the register allocation allows
(hipcc -Rpass-analysis=kernel-resource-usage, cross-compiled for gfx950: runs in the build container)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("VGPRs", "AGPRs", "VGPRs Spill", "TotalSGPRs", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "ScratchSize [bytes/lane]")


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "t8gpu_amd", "csrc", "hip"), "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|" + "|".join(re.escape(k) for k in KEYS) + r"): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k in ("Function Name", "Name"):
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("t8gpu_hip::", "").replace("void ", "")
        print(f"{name:60s} VGPR {r.get('VGPRs', '?'):>4} spill {r.get('VGPRs Spill', '?'):>3} scratch {r.get('ScratchSize [bytes/lane]', '?'):>4} "
              f"SGPR {r.get('TotalSGPRs', '?'):>4} waves/SIMD {r.get('Occupancy [waves/SIMD]', '?'):>2} LDS {r.get('LDS Size [bytes/block]', '?')}")


if __name__ == "__main__":
    main()
