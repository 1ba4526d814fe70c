#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel, from hipcc's device assembly (no GPU needed):
    python tools/isa_loops.py <mangled-name-prefix> [min_instr]
e.g. python tools/isa_loops.py _Z8k2_estepILi12E 300"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pref, min_instr = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 300
asm = os.environ.get("SCAPE_ASM")
if not asm:
    asm = os.path.join(tempfile.gettempdir(), "scape_dev.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                    "--cuda-device-only", "-o", asm, os.path.join(ROOT, "scape_amd", "csrc", "scape_hip.hip")],
                   stderr=subprocess.DEVNULL, check=True)
lines, on = [], False
for l in open(asm):
    if l.startswith(pref):
        on = True
    if on:
        lines.append(l.rstrip("\n"))
        if l.startswith(".Lfunc_end"):
            break
cur, groups = None, {}
for i, l in enumerate(lines):
    m = re.search(r"Header=(BB\d+_\d+) Depth=(\d)", l)
    if m:
        cur = (m.group(1), m.group(2))
    elif re.search(r"Loop Header: Depth=(\d)", l):
        mm = re.match(r"^\.L(BB\d+_\d+):", lines[i - 1])
        if mm:
            cur = (mm.group(1), re.search(r"Depth=(\d)", l).group(1))
    elif re.match(r"^\.LBB|^; %bb", l) and "Depth=" not in l:
        cur = None
    if cur and l.startswith("\t") and not l.startswith("\t;") and not l.startswith("\t."):
        groups.setdefault(cur, []).append(l.split()[0])
for k, ins in groups.items():
    if len(ins) < min_instr:
        continue
    c = Counter(ins)
    sc_l = sum(v for kk, v in c.items() if kk.startswith("scratch_load"))
    sc_s = sum(v for kk, v in c.items() if kk.startswith("scratch_store"))
    print(k, "instr", len(ins), "valu", sum(v for kk, v in c.items() if kk.startswith("v_")), "exps", c["v_rndne_f64_e32"],
          "scratch ld/st", sc_l, sc_s, "ds", sum(v for kk, v in c.items() if kk.startswith("ds_")),
          "global", sum(v for kk, v in c.items() if kk.startswith("global_")), "mfma", sum(v for kk, v in c.items() if "mfma" in kk),
          "salu", sum(v for kk, v in c.items() if kk.startswith("s_")))
