#!/usr/bin/env python3
"""MFMA pipe utilisation of ONE whole edit from hardware counters (BASELINE: "rocprof MFMA utilisation ... reported against gfx950 peak").
  run     rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d DIR -o runc -- python3 tools/shape_profile.py run DIR/oplog.txt
  report  python3 tools/edit_mfma_utilisation.py DIR OUT.md
The last edit of the run = the last N library dispatches (N = launches in DIR/oplog.txt).  Utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024):
busy cycles of the matrix pipes over (cycles per XCD x the chip's 1 024 SIMDs), the formula of tools/pmc_summary.py (FF1 alone: 0.34 by counters, 0.37-0.41 by time)."""
import collections
import csv
import glob
import os
import re
import sys

d, out = sys.argv[1], sys.argv[2]
n = sum(1 for l in open(os.path.join(d, "oplog.txt")) if l.strip() and not l.startswith("#"))
rows = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
ours = lambda name: not (name.startswith("void at::") or name.startswith("at::") or "at::native" in name or name.startswith("__amd_rocclr"))
per = collections.OrderedDict()
for r in rows:
    if ours(r["Kernel_Name"]):
        e = per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(per)[-n:]
assert len(ids) == n, (len(ids), n)
fam = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
tot = [0.0, 0.0, 0.0]
for i in ids:
    e = per[i]
    short = re.sub(r"\(anonymous namespace\)::|void |fie_gemm::", "", e["name"]).split("(")[0][:70]
    v = (e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), e.get("GRBM_GUI_ACTIVE", 0.0), e.get("SQ_INSTS_MFMA", 0.0))
    for k in range(3):
        fam[short][k] += v[k]
        tot[k] += v[k]
    fam[short][3] += 1
util = lambda busy, gui: busy / (gui / 8 * 1024) if gui else 0.0
with open(out, "w") as f:
    f.write("# MFMA pipe utilisation of one whole edit by hardware counters (SSD-1B-A1 + ControlNet-full, 1024^2, 2 evaluations, CFG batch 2)\n\n"
            f"`tools/edit_mfma_utilisation.py`: {n} launches of one eager single-stream edit under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA`.\n\n"
            f"**Whole edit: {util(tot[0], tot[1]):.3f}** of the matrix pipes' cycles busy ({tot[2] / 1e6:.1f} M MFMA instructions per wave-lane group, "
            f"{tot[1] / 1e6:.1f} M GRBM_GUI_ACTIVE cycles summed over the launches).\n\n| kernel | launches | share of GPU-active cycles | MFMA pipe utilisation |\n|---|---:|---:|---:|\n")
    for name, (busy, gui, insts, cnt) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
        f.write(f"| `{name}` | {cnt} | {100 * gui / tot[1]:.1f} % | {util(busy, gui):.3f} |\n")
print(open(out).read())
