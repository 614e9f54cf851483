#!/bin/bash
# The GPU test tier with the HOST side of libhvs under AddressSanitizer (repo root, on a GPU box; build first, on the CPU:
# make -C hirag-prod_amd/csrc asan).  Device code is not instrumented.  The sanitizer's runtime is gcc's (the one in ROCm's
# clang intercepts the HSA allocator and cannot serve an unsanitized HIP runtime); libstdc++ is preloaded with it so that it
# finds __cxa_throw, and torch's own library directory is named because an intercepted dlopen does not search a caller's RUNPATH.
# One test is left out: it opens RCCL with RTLD_DEEPBIND, which the sanitizer refuses.
#   bash tools/asan_gpu_suite.sh [output directory]
set -o pipefail
OUT=${1:-gpurun_out/asan}
mkdir -p "$OUT"
ROOT=$(pwd)
RT=/usr/lib/x86_64-linux-gnu/libasan.so.6:/usr/lib/x86_64-linux-gnu/libstdc++.so.6
LD_LIBRARY_PATH="/usr/local/lib/python3.10/dist-packages/torch/lib${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}" \
LD_PRELOAD="${LD_PRELOAD:+$LD_PRELOAD:}$RT" \
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:protect_shadow_gap=0:log_path=$ROOT/$OUT/asan \
HVS_LIB=$ROOT/tools/_build/libhvs_asan.so \
timeout -k 10 1000 python3 -X faulthandler -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread \
    --deselect tests/test_gpu_sharded.py::test_rccl_communicator_of_one_rank > "$OUT/pytest.txt" 2>&1
rc=$?
tail -n 3 "$OUT/pytest.txt"
ls "$OUT"/asan.* 2>/dev/null && echo "sanitizer reports above" || echo "no sanitizer report"
exit $rc
