cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/r4d_prof gpurun_out/r4d_shape
timeout -k 10 480 python -m pytest tests -x -q -m gpu > gpurun_out/r4d_gpu_suite.log 2>&1; tail -n 3 gpurun_out/r4d_gpu_suite.log
grep -q " passed" gpurun_out/r4d_gpu_suite.log && ! grep -q " failed" gpurun_out/r4d_gpu_suite.log && \
timeout -k 10 300 python bench.py > gpurun_out/r4d_bench.json 2> gpurun_out/r4d_bench.err && tail -c 600 gpurun_out/r4d_bench.json && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4d_prof -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r4d_prof_bench.json 2> gpurun_out/r4d_prof_bench.err && \
python3 tools/rocprof_summary.py gpurun_out/r4d_prof gpurun_out/r04_thin_bench > /dev/null && head -12 gpurun_out/r04_thin_bench_summary.md | cut -c1-160 && \
mkdir -p gpurun_out/r4d_shape && timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4d_shape/runc -- python3 tools/shape_profile.py run gpurun_out/r4d_shape/oplog.txt > gpurun_out/r4d_shape_run.log 2>&1 && \
python3 tools/shape_profile.py report gpurun_out/r4d_shape gpurun_out/r4d_shape/report.md > /dev/null 2> gpurun_out/r4d_report.err; head -16 gpurun_out/r4d_shape/report.md | tail -9
