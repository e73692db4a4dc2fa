#!/bin/bash
# Third session of round 4, LAST evidence call (final code): GPU suite on the frozen table (verbose, with durations), the default bench line, smoke().
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -v -m gpu --durations=25 > gpurun_out/r4h_gpu_suite.log 2>&1; tail -n 32 gpurun_out/r4h_gpu_suite.log | cut -c1-150
grep -q " passed" gpurun_out/r4h_gpu_suite.log && ! grep -q " failed" gpurun_out/r4h_gpu_suite.log && \
timeout -k 10 300 python bench.py > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err && tail -c 300 gpurun_out/r4h_bench.json && \
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4h_smoke.log 2>&1; tail -n 2 gpurun_out/r4h_smoke.log
