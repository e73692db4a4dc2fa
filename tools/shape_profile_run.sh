cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/r4l_shape
mkdir -p gpurun_out/r4l_shape && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4l_shape/runc -- python3 tools/shape_profile.py run gpurun_out/r4l_shape/oplog.txt > gpurun_out/r4l_shape_run.log 2>&1
python3 tools/shape_profile.py report gpurun_out/r4l_shape gpurun_out/r4l_shape/report.md > /dev/null 2> gpurun_out/r4l_report.err; head -60 gpurun_out/r4l_shape/report.md | cut -c1-230
