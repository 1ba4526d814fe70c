#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small per-kernel summaries kept under profiles/.

    python tools/rocprof_summary.py stats <rocprof_dir> <out.csv> "<command line that was profiled>"
    python tools/rocprof_summary.py pmc   <rocprof_dir> <out.csv> "<command line that was profiled>"
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name).strip()
    return re.sub(r"\s*\[clone.*", "", name)


def find(root, pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))


def stats(root, out, cmd):
    rows = defaultdict(lambda: [0, 0.0])
    files = find(root, "*kernel_trace.csv")
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                rows[k][0] += 1
                rows[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in rows.values()) or 1.0
    with open(out, "w") as fh:
        fh.write(f"# rocprofv3 --kernel-trace --stats -- {cmd} (per-kernel sums over the kernel trace; durations in us)\n")
        fh.write("kernel,calls,total_us,average_us,percent\n")
        for k, (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            fh.write(f"\"{k}\",{n},{us:.3f},{us / n:.3f},{100 * us / tot:.3f}\n")


def pmc(root, out, cmd):
    rows = defaultdict(lambda: [0, 0.0])
    for f in find(root, "*counter_collection.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                key = (r["Counter_Name"], short(r["Kernel_Name"]))
                rows[key][0] += 1
                rows[key][1] += float(r["Counter_Value"])
    with open(out, "w") as fh:
        fh.write(f"# rocprofv3 --kernel-trace --pmc ... -- {cmd} (raw counter sums per kernel; FETCH_SIZE / WRITE_SIZE in KB; "
                 "FETCH_SIZE reads half the bytes of 16-B-per-lane loads on gfx950, MI355X_MICROARCH.md)\n")
        fh.write("counter,kernel,dispatches,sum,per_dispatch\n")
        for (c, k), (n, v) in sorted(rows.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
            fh.write(f"{c},\"{k}\",{n},{v:.6g},{v / n:.6g}\n")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
