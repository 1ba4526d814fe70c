#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of the HIP library, from hipcc's own
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles for gfx950, no GPU needed).

    python tools/kernel_resources.py [substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "scape_amd", "csrc", "scape_hip.hip")


def main():
    pats = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
               "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(td, "x.so"), SRC]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(\w[\w \[\]/]*): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'LDS':>7s} {'occ':>4s}")
    for r in rows:
        if pats and not any(p in r["name"] for p in pats):
            continue
        print(f"{r['name'][:44]:44s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} "
              f"{r.get('VGPRs Spill', 0):6d} {r.get('LDS Size [bytes/block]', 0):7d} "
              f"{r.get('Occupancy [waves/SIMD]', 0):4d}")


if __name__ == "__main__":
    main()
