#!/bin/bash
# Third session of round 4, second evidence call on ONE box: whole-edit A/B of the thin-conv kernel, the GPU suite on the LIVE tuner,
# BASELINE config 3 (SDXL, 8 images per job) and config 5 (SDXL W8A8, calibrated) beside fp16.  Outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT
FIE_AUTOTUNE=0 timeout -k 10 200 python tools/thin_conv_edit_ab.py > gpurun_out/r4e_thin_edit_ab.log 2>&1; tail -n 8 gpurun_out/r4e_thin_edit_ab.log
FIE_TUNE_LIVE=1 timeout -k 10 480 python -m pytest tests -x -q -m gpu > gpurun_out/r4e_gpu_suite_live.log 2>&1; tail -n 2 gpurun_out/r4e_gpu_suite_live.log
grep -q " passed" gpurun_out/r4e_gpu_suite_live.log && ! grep -q " failed" gpurun_out/r4e_gpu_suite_live.log && \
timeout -k 10 300 python bench.py --model sdxl --batch 8 --no-cpu-baseline > gpurun_out/r4e_bench_sdxl_b8.json 2> gpurun_out/r4e_bench_sdxl_b8.err && \
timeout -k 10 300 python bench.py --model sdxl --weights f8e4m3 --no-cpu-baseline --no-extras > gpurun_out/r4e_bench_sdxl_fp8.json 2> gpurun_out/r4e_bench_sdxl_fp8.err && \
python3 - <<'PY'
import json
for f in ("r4e_bench_sdxl_b8", "r4e_bench_sdxl_fp8"):
    d = json.loads(open("gpurun_out/" + f + ".json").read().strip().split("\n")[-1])
    print(f, d["value"], d["ms_per_step"], d["dtype"], d.get("batched_images_per_sec"), d["roofline"]["unet_forward"]["ms"])
PY
