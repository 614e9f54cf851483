#!/bin/bash
# MFMA-pipe occupancy and sustained clock of the 16-bit batched search kernel and of the vendor GEMM on the same shape
# (4096 x 2M x 1536 f16), one MI355X: a kernel trace and a counter pass for each, every pass a run of its own.
# Usage (repo root, GPU box): bash tools/mfma_ceiling.sh   -> gpurun_out/ceiling/
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/ceiling
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d "$OUT/search_trace" -o t --output-format csv -- python3 "$ROOT/tools/mfma_perf.py" --dtype f16 --rows 2000000 --nq 4096 --k 10 --iters 5 > "$OUT/search_trace.txt" 2>&1
echo "search trace done"
timeout -s KILL 150 rocprofv3 --pmc $CTR -d "$OUT/search_pmc" -o c --output-format csv -- python3 "$ROOT/tools/mfma_perf.py" --dtype f16 --rows 2000000 --nq 4096 --k 10 --iters 5 > "$OUT/search_pmc.txt" 2>&1
echo "search counters done"
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d "$OUT/gemm_trace" -o t --output-format csv -- python3 "$ROOT/tools/gemm_pmc_probe.py" > "$OUT/gemm_trace.txt" 2>&1
echo "gemm trace done"
timeout -s KILL 150 rocprofv3 --pmc $CTR -d "$OUT/gemm_pmc" -o c --output-format csv -- python3 "$ROOT/tools/gemm_pmc_probe.py" > "$OUT/gemm_pmc.txt" 2>&1
echo "gemm counters done"
python3 "$ROOT/tools/mfma_ceiling_summary.py" "$OUT" | tee "$OUT/summary.txt"
