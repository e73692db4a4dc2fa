#!/bin/bash
# Fabric traffic and L2 hit rate of the thin-conv kernel on the decoder's conv_out (separate rocprofv3 --pmc passes, no trace domains).  Run through gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/r4g_*
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r4g_fetch -o runc -- python3 tools/conv_thin_time.py decoder thin-only > gpurun_out/r4g_fetch.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r4g_hit -o runc -- python3 tools/conv_thin_time.py decoder thin-only > gpurun_out/r4g_hit.log 2>&1 &&
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum --output-format csv -d gpurun_out/r4g_req -o runc -- python3 tools/conv_thin_time.py decoder thin-only > gpurun_out/r4g_req.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r4g_sq -o runc -- python3 tools/conv_thin_time.py decoder thin-only > gpurun_out/r4g_sq.log 2>&1
python3 - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("gpurun_out/r4g_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_thin" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k}: avg {sum(v) / len(v):.1f} over {len(v)} launches")
PY
