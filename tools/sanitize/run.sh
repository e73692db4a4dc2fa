#!/bin/bash
# AddressSanitizer + UBSan over the HOST code of the library: the bookkeeping of the C++ graph walks (csrc/graphs.cpp: arena, tensor lifetimes, workspace queries), which runs without a GPU:
# the *_workspace_bytes entries replay every walk's allocation sequence in plan mode.  (GPU sanitizers are not available on this pool.)  Run from the repo root.
set -e
CS=fast-image-editing-with-generative-models_amd/csrc
OUT=${TMPDIR:-/tmp}/fie_asan_drv
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer --offload-arch=gfx950 -Iinclude -I$CS $CS/graphs.cpp tools/sanitize/graph_walks_plan_driver.cpp \
    -L$CS -lfie_hip -Wl,-rpath,$PWD/$CS -o $OUT 2>&1 | grep -v "warning" || true
ASAN_OPTIONS=detect_leaks=0 $OUT
# the host Canny entry (csrc/canny.cpp): odd sizes down to 1x1, flat and noisy images
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer --offload-arch=gfx950 -Iinclude -I$CS $CS/canny.cpp tools/sanitize/canny_driver.cpp \
    -L$CS -lfie_hip -Wl,-rpath,$PWD/$CS -o ${OUT}_canny 2>&1 | grep -v "warning" || true
ASAN_OPTIONS=detect_leaks=0 ${OUT}_canny
