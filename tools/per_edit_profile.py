#!/usr/bin/env python3
"""Kernel time of exactly ONE edit: difference of two `rocprofv3 --kernel-trace --stats --output-format csv` runs of
`bench.py --no-extras --no-cpu-baseline --steps K` with K = k_small and K = k_big, divided by (k_big - k_small): set-up,
weight synthesis and the roofline passes cancel, what is left is exactly the launches one serial edit replays.
usage: tools/per_edit_profile.py <dir_small> <k_small> <dir_big> <k_big> <out.md> [title]"""
import csv
import glob
import os
import sys


def load(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}, f


(a, fa), ka = load(sys.argv[1]), int(sys.argv[2])
(b, fb), kb = load(sys.argv[3]), int(sys.argv[4])
out = sys.argv[5]
title = sys.argv[6] if len(sys.argv) > 6 else ""
n = kb - ka
rows = []
for name in set(a) | set(b):
    ca, ta = a.get(name, (0, 0.0))
    cb, tb = b.get(name, (0, 0.0))
    calls, us = (cb - ca) / n, (tb - ta) / n / 1e3
    if calls > 0.01 or us > 0.5:
        rows.append((us, calls, name))
rows.sort(reverse=True)
tot, launches = sum(r[0] for r in rows), sum(r[1] for r in rows)
with open(out, "w") as f:
    f.write(f"# Kernel time of ONE edit {title}\n\nDifference of two rocprofv3 --kernel-trace --stats runs ({kb} minus {ka} timed edits, divided by {n}):\n"
            f"`{fa}`, `{fb}`.\n\nSum of kernel durations **{tot / 1e3:.2f} ms** in **{launches:.0f} launches**.\n\n"
            "| kernel | calls / edit | us / edit | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for us, calls, name in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:110]
        f.write(f"| `{short}` | {calls:.1f} | {us:.0f} | {us / max(calls, 1e-9):.1f} | {100 * us / tot:.1f} |\n")
print(open(out).read())
