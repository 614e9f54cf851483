#!/bin/bash
# Issue / wait counters per kernel of one probe script (one MI355X), one counter pass.
# Usage (repo root, GPU box):  bash tools/pmc_kernels.sh <tag> <script.py>
set -e -o pipefail
TAG=$1
SCRIPT=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    -d "$OUT" -o c --output-format csv -- python3 "$ROOT/$SCRIPT" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
cat "$OUT/stdout.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"][:60]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[name] += 1
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    w = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f'{name:60s} launches {n[name]:4d} wave_cycles {w:.3e} wait_any {c.get("SQ_WAIT_ANY", 0) / w:.3f} wait_inst {c.get("SQ_WAIT_INST_ANY", 0) / w:.3f} '
          f'active {c.get("SQ_ACTIVE_INST_ANY", 0) / w:.3f} wait_lds {c.get("SQ_WAIT_INST_LDS", 0) / w:.3f} mfma_busy/busy {c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1):.3f} '
          f'lds_conflict {c.get("SQ_LDS_BANK_CONFLICT", 0):.3e} gui_active {c.get("GRBM_GUI_ACTIVE", 0):.3e}')
PY
