#!/usr/bin/env python3
"""Turn a `rocprofv3 --kernel-trace --stats --output-format csv` directory into the summary kept under profiles/.
usage: tools/rocprof_summary.py gpurun_out/profN profiles/rNN_tag [images_in_run]"""
import csv
import glob
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
images = float(sys.argv[3]) if len(sys.argv) > 3 else None
stats = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)[0]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
shutil.copy(stats, dst + "_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst + "_summary.md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(dst)})\n\n")
    f.write(f"source: `{stats}`; total kernel time {tot / 1e6:.1f} ms")
    if images:
        f.write(f" over {images:g} images = {tot / 1e6 / images:.1f} ms/image")
    f.write("\n\n| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:30]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:80]
        f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
print(open(dst + "_summary.md").read())
