#!/bin/bash
# Collects the PMC record bench.py reads for roofline.traffic (profiles/dominant_kernel_pmc.json): separate rocprofv3 --pmc passes
# per counter group and per tile code the autotuner may pick for the FF1 shape, then a kernel-trace pass.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -f gpurun_out/dominant_kernel_pmc.json && for code in 54 52 96 62; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2f_fetch_$code -o runc -- python3 tools/one_kernel.py gemm 2048 10240 1280 $code 20 geglu > /dev/null 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2f_write_$code -o runc -- python3 tools/one_kernel.py gemm 2048 10240 1280 $code 20 geglu > /dev/null 2>&1 &&
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r2f_sq_$code -o runc -- python3 tools/one_kernel.py gemm 2048 10240 1280 $code 20 geglu > /dev/null 2>&1 &&
  FIE_PMC_TILE=$code python3 tools/pmc_summary.py gemm3_kernel 2048 10240 1280 gpurun_out/dominant_kernel_pmc.json gpurun_out/r2f_fetch_$code gpurun_out/r2f_write_$code gpurun_out/r2f_sq_$code | grep -E "fabric|frac|utilis" || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f_kt_54 -o runc -- python3 tools/one_kernel.py gemm 2048 10240 1280 54 20 geglu > /dev/null 2>&1 && python3 tools/rocprof_summary.py gpurun_out/r2f_kt_54 gpurun_out/r02_dominant_kernel && head -12 gpurun_out/r02_dominant_kernel_summary.md
