#!/bin/bash
# Closing measurements on one MI355X (repo root): GPU tests, the probes whose numbers the design notes quote, and the
# bench.py run under the kernel tracer plus its fabric-byte passes.  Everything lands under gpurun_out/r2_final/.
# Each step has its own time limit; a failing step ends the script.
set -o pipefail
OUT=gpurun_out/r2_final
mkdir -p $OUT
step() { echo "== $1" | tee -a $OUT/steps.log; }
step "pytest -m gpu"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
step "clustered corpus leg (two list lengths in the filter route)"
timeout -k 10 300 python3 bench.py --leg clustered > $OUT/clustered.log 2>&1 || { tail -20 $OUT/clustered.log; exit 1; }
grep LEG-RESULT $OUT/clustered.log | cut -c1-1500
step "sample pass on a 4096-query batch (f16, 2M rows)"
for p in 0 1; do HVS_PREPASS=$p timeout -k 10 120 python3 tools/mfma_perf.py --dtype f16 --rows 2000000 --nq 4096 --iters 5 2>&1 | grep "mfma search" | sed "s/^/HVS_PREPASS=$p /" | tee -a $OUT/prepass_4096.log; done
step "k-means, cfg5"
timeout -k 10 200 python3 tools/km_wall_probe.py 2>&1 | grep -E "filter|three" | tee $OUT/kmeans.log
timeout -k 10 200 python3 tools/cfg_probe.py cfg5 2>&1 | grep cfg5 | tee $OUT/cfg5.log
step "batch sweep"
timeout -k 10 300 python3 tools/batch_sweep.py > $OUT/batch_sweep.txt 2>&1; tail -32 $OUT/batch_sweep.txt | grep -E "nq= *(256|512|1024) mfma"
step "bench.py under the kernel tracer, then the fetch / write passes"
bash tools/profile_bench.sh r2final trace > $OUT/profile_trace.log 2>&1 || { tail -20 $OUT/profile_trace.log; exit 1; }
python3 - <<'PY'
import json
r = json.load(open("gpurun_out/prof_r2final/bench_under_trace.json"))
print("value", r["value"], "ms/step", r["ms_per_step"], "frac", r["roofline"]["frac"])
for k, v in r["aux"].items():
    print(k, json.dumps(v)[:300])
PY
bash tools/profile_bench.sh r2final fetch write > $OUT/profile_pmc.log 2>&1 || { tail -20 $OUT/profile_pmc.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/prof_r2final/fetch gpurun_out/prof_r2final/write 2>/dev/null | grep -E "mfma_search|mfma_stream" | head -12
step "done"
