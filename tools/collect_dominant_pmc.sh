#!/bin/bash
# Collects the PMC record bench.py reads for roofline.traffic (profiles/dominant_kernel_pmc.json) and the kernel-trace summaries of the FF1
# GEMM for EVERY tile code the autotuner may pick for it (63 / 64 = 256x320 since round 3, 64 with alternating refill; 96 / 54 / 52 / 62 before): separate rocprofv3 --pmc
# passes per counter group (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950), then a --kernel-trace --stats pass per code.
# Run on the GPU box from the repo root; copy gpurun_out/dominant_kernel_pmc.json and gpurun_out/r03_dominant_kernel_tile*_{summary.md,kernel_stats.csv} to profiles/.
# CODES="64" tools/collect_dominant_pmc.sh adds one tile to the existing record (profiles/dominant_kernel_pmc.json is the starting point).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && if [ -n "$CODES" ]; then cp profiles/dominant_kernel_pmc.json gpurun_out/dominant_kernel_pmc.json; else rm -f gpurun_out/dominant_kernel_pmc.json; fi && for code in ${CODES:-63 64 96 54 52 62}; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3f_fetch_$code -o runc -- python3 tools/one_kernel.py gemm 2048 10240 1280 $code 20 geglu > /dev/null 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3f_write_$code -o runc -- python3 tools/one_kernel.py gemm 2048 10240 1280 $code 20 geglu > /dev/null 2>&1 &&
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r3f_sq_$code -o runc -- python3 tools/one_kernel.py gemm 2048 10240 1280 $code 20 geglu > /dev/null 2>&1 &&
  FIE_PMC_TILE=$code python3 tools/pmc_summary.py gemm3_kernel 2048 10240 1280 gpurun_out/dominant_kernel_pmc.json gpurun_out/r3f_fetch_$code gpurun_out/r3f_write_$code gpurun_out/r3f_sq_$code | grep -E "fabric|frac|utilis" || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3f_kt_$code -o runc -- python3 tools/one_kernel.py gemm 2048 10240 1280 $code 20 geglu > /dev/null 2>&1 &&
  python3 tools/rocprof_summary.py gpurun_out/r3f_kt_$code gpurun_out/r03_dominant_kernel_tile$code > /dev/null && head -7 gpurun_out/r03_dominant_kernel_tile${code}_summary.md | tail -2 || exit 1
done
