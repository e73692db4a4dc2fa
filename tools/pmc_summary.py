#!/usr/bin/env python3
"""Average the counters of `rocprofv3 --pmc ... --output-format csv` passes over the launches of ONE kernel and write the
record bench.py reads for `roofline.traffic` (profiles/dominant_kernel_pmc.json).
usage: tools/pmc_summary.py <kernel-name substring> M N K out.json <pass dir> [<pass dir> ...]
With FIE_PMC_TILE=<code> in the environment the record is filed under records[<code>] of out.json (one record per tile code
the autotuner may pick for the shape); without it out.json is the single record."""
import csv
import glob
import json
import os
import sys

sub, m, n, k, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
vals, sources = {}, []
for d in sys.argv[6:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        sources.append(f)
        per = {}
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for c, v in per.items():
            vals[c] = (sum(v) / len(v), len(v))
rec = {"kernel_substring": sub, "shape": {"M": m, "N": n, "K": k}, "source": sources,
       "counters_avg_per_launch": {c: round(v, 3) for c, (v, _) in sorted(vals.items())},
       "launches_averaged": {c: cnt for c, (_, cnt) in sorted(vals.items())}}
if "FETCH_SIZE" in vals:
    rec["fetch_size_kib"] = round(vals["FETCH_SIZE"][0], 1)
if "WRITE_SIZE" in vals:
    rec["write_size_kib"] = round(vals["WRITE_SIZE"][0], 1)
if "fetch_size_kib" in rec and "write_size_kib" in rec:
    # gfx950: FETCH_SIZE counts a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM)
    rec["fabric_mb_per_launch"] = round((2 * rec["fetch_size_kib"] + rec["write_size_kib"]) * 1024 / 1e6, 1)
c = rec["counters_avg_per_launch"]
if "SQ_WAVE_CYCLES" in c:
    for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if name in c:
            rec[name.lower() + "_frac"] = round(c[name] / c["SQ_WAVE_CYCLES"], 4)
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    rec["mfma_pipe_utilisation"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)   # busy cycles / (cycles per XCD x 1024 SIMDs)
tile = os.environ.get("FIE_PMC_TILE")
if tile:
    rec["tile_code"] = int(tile)
    doc = {"shape": rec["shape"], "records": {}}
    if os.path.exists(out):
        old = json.load(open(out))
        if old.get("shape") == rec["shape"] and "records" in old:
            doc = old
    doc["records"][tile] = rec
    json.dump(doc, open(out, "w"), indent=1)
else:
    json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k_: v for k_, v in rec.items() if k_ != "source"}, indent=1))
