#!/usr/bin/env python3
"""Per-dispatch table of rocprofv3 --pmc counters for the kernels whose name contains <pattern>, in dispatch order
(the lock-step EM launches one E-step and one M-step kernel per round, so row i of a sweep is EM round i).

    python tools/pmc_rounds.py <rocprof_dir> <pattern> [max_rows]
Columns: duration (us) from the dispatch timestamps, then every counter of the pass summed over its dimensions.
"""
import csv, glob, os, sys
from collections import OrderedDict, defaultdict

def main():
    root, pat = sys.argv[1], sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rows = OrderedDict()
    names = []
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if pat not in r["Kernel_Name"]:
                    continue
                d = rows.setdefault(int(r["Dispatch_Id"]), dict(us=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                                 vgpr=r["VGPR_Count"], lds=r["LDS_Block_Size"], c=defaultdict(float)))
                d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] not in names:
                    names.append(r["Counter_Name"])
    print("# " + root + " kernel~" + pat)
    print("idx dispatch us vgpr lds " + " ".join(names))
    for i, (did, d) in enumerate(rows.items()):
        if i >= limit:
            break
        print(i, did, f"{d['us']:.1f}", d["vgpr"], d["lds"], " ".join(f"{d['c'][n]:.6g}" for n in names))

if __name__ == "__main__":
    main()
