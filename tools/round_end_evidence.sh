#!/bin/bash
# Round-end evidence on ONE box (run from the repo root through gpurun): rocprofv3 --kernel-trace --stats of a short bench run, the per-shape in-situ table,
# and the fp8 configuration beside fp16.  Outputs under gpurun_out/ (copy what is to be judged into profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/r3s_prof gpurun_out/r3s_shape
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3s_prof -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3s_prof_bench.json 2> gpurun_out/r3s_prof_bench.err
python3 tools/rocprof_summary.py gpurun_out/r3s_prof gpurun_out/r03_final_bench > /dev/null; head -12 gpurun_out/r03_final_bench_summary.md | cut -c1-160
mkdir -p gpurun_out/r3s_shape && timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3s_shape/runc -- python3 tools/shape_profile.py run gpurun_out/r3s_shape/oplog.txt > gpurun_out/r3s_shape_run.log 2>&1
python3 tools/shape_profile.py report gpurun_out/r3s_shape gpurun_out/r3s_shape/report.md > /dev/null 2> gpurun_out/r3s_report.err; head -16 gpurun_out/r3s_shape/report.md | tail -9
timeout -k 10 300 python bench.py --weights f8e4m3 --no-cpu-baseline > gpurun_out/r3s_bench_fp8.json 2> gpurun_out/r3s_bench_fp8.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r3s_bench_f16.json 2> gpurun_out/r3s_bench_f16.err
python3 - <<'PY'
import json
for f in ("r3s_bench_fp8", "r3s_bench_f16"):
    d = json.loads(open("gpurun_out/" + f + ".json").read().strip().split("\n")[-1])
    print(f, d["value"], d["ms_per_step"], d["dtype"], d["roofline"]["unet_forward"]["ms"])
PY
