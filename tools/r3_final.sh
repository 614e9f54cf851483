#!/bin/bash
# Closing measurements on one MI355X (repo root), in parts that each fit one GPU call:
#   bash tools/r3_final.sh probes   the probes whose numbers the design notes quote
#   bash tools/r3_final.sh trace    bench.py plain, then under the kernel tracer
#   bash tools/r3_final.sh pmc      the fabric-byte passes (FETCH_SIZE, WRITE_SIZE) of bench.py
# Everything lands under gpurun_out/r3_final/.  Each step has its own time limit; a failing step ends the script.
set -o pipefail
OUT=gpurun_out/r3_final
mkdir -p $OUT
step() { echo "== $1" | tee -a $OUT/steps.log; }
case ${1:-probes} in
probes)
  step "workspace filter as a row range (cfg5 shard)"
  timeout -k 10 200 python3 tools/partition_probe.py > $OUT/partition_probe.txt 2>&1 || { tail -20 $OUT/partition_probe.txt; exit 1; }
  grep -E "partitioned|plain" $OUT/partition_probe.txt | tail -12
  step "range sizes"
  timeout -k 10 250 python3 tools/range_sweep.py > $OUT/range_sweep.txt 2>&1 || { tail -20 $OUT/range_sweep.txt; exit 1; }
  tail -14 $OUT/range_sweep.txt
  step "whole shard, batches"
  timeout -k 10 250 python3 tools/whole_sweep.py 2000000 1536 f16 10 > $OUT/whole_sweep.txt 2>&1 || { tail -20 $OUT/whole_sweep.txt; exit 1; }
  tail -10 $OUT/whole_sweep.txt
  step "k beyond 256"
  timeout -k 10 200 python3 tools/large_k_probe.py > $OUT/large_k.txt 2>&1 || { tail -20 $OUT/large_k.txt; exit 1; }
  tail -10 $OUT/large_k.txt
  ;;
trace)
  step "bench.py"
  timeout -k 10 500 python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.log || { tail -20 $OUT/bench_plain.log; exit 1; }
  step "bench.py under the kernel tracer"
  bash tools/profile_bench.sh r3final trace > $OUT/profile_trace.log 2>&1 || { tail -20 $OUT/profile_trace.log; exit 1; }
  python3 - <<'PY'
import json
for path in ("gpurun_out/r3_final/bench_plain.json", "gpurun_out/prof_r3final/bench_under_trace.json"):
    r = json.loads(open(path).read().strip().splitlines()[-1])
    print(path, "value", r["value"], "ms/step", r["ms_per_step"], "frac", r["roofline"]["frac"])
    for k, v in r.get("aux", {}).items():
        print("  ", k, json.dumps(v)[:400])
PY
  ;;
pmc)
  step "fetch / write passes"
  bash tools/profile_bench.sh r3final fetch write > $OUT/profile_pmc.log 2>&1 || { tail -20 $OUT/profile_pmc.log; exit 1; }
  python3 tools/pmc_summary.py gpurun_out/prof_r3final/fetch gpurun_out/prof_r3final/write 2>/dev/null | grep -E "mfma_search|mfma_stream|mfma_ring|scan_kernel" | head -24
  ;;
esac
step "done $1"
