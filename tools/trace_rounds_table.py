"""Joins two runs of tools/trace_rounds.py (plain: per-launch durations; `bytes`: the M-step's own byte tally per round)
into a per-round table.  python tools/trace_rounds_table.py <plain stderr> <bytes stderr>"""
import re, sys
ms, by = {}, {}
for l in open(sys.argv[1]):
    m = re.match(r"\[trace\] kind 5 launch (\d+) ([\d.]+) ms", l)
    if m: ms[int(m.group(1))] = float(m.group(2))
for l in open(sys.argv[2]):
    m = re.match(r"\[em bytes (\d+)\] (\d+)", l)
    if m: by[int(m.group(1))] = int(m.group(2))
prev = 0
print("round  tensor GB  k2_mstep ms  TB/s")
for r in sorted(by):
    b = by[r] - prev; prev = by[r]
    if r in ms: print(f"{r:5d}  {b / 1e9:9.2f}  {ms[r]:11.3f}  {b / ms[r] / 1e9:5.2f}")
