#!/bin/bash
# Kernel-trace statistics and HBM/issue counters of the default bench.py run (one MI355X).
# Usage (from the repo root, on the GPU box):  bash tools/profile_bench.sh <tag> [trace|fetch|write|sq ...]
# Writes gpurun_out/prof_<tag>/...; each counter pass is a run of its own.
set -e -o pipefail
TAG=${1:-run}
shift || true
PASSES=${*:-trace}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for pass in $PASSES; do
  case $pass in
    trace)
      timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- \
          python3 "$ROOT/bench.py" > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log" ;;
    fetch)
      timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o c --output-format csv -- \
          python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/fetch.json" 2> "$OUT/fetch.log" ;;
    write)
      timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o c --output-format csv -- \
          python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/write.json" 2> "$OUT/write.log" ;;
    sq)
      timeout -s KILL 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
          -d "$OUT/sq" -o c --output-format csv -- \
          python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/sq.json" 2> "$OUT/sq.log" ;;
  esac
  echo "$pass done"
done
