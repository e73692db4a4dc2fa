#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a `hipcc --save-temps` assembly file (the .s of the gfx950 pass).
usage: tools/vgpr_table.py FILE.s [substring]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for blk in s.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    name = g("name").group(1)
    rows.append((name, int(g("vgpr_count").group(1)), int(g("sgpr_count").group(1)), int(g("private_segment_fixed_size").group(1)), int(blk.split()[0])))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (name, v, sg, scr, ag), dn in zip(rows, names):
    dn = dn.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if want in dn:
        print(f"{dn:70s} vgpr {v:4d} agpr {ag:4d} sgpr {sg:4d} scratch {scr}")
