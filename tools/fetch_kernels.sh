#!/bin/bash
# Fabric-side read bytes (FETCH_SIZE, kilobytes; doubled for 16-byte-per-lane streaming reads on gfx950) per kernel of one probe script.
# Usage (repo root, GPU box):  bash tools/fetch_kernels.sh <tag> <script.py>
set -e -o pipefail
TAG=$1
SCRIPT=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/fetch_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT" -o c --output-format csv -- python3 "$ROOT/$SCRIPT" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
cat "$OUT/stdout.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print(f'{name:70s} launches {len(v):4d}  FETCH_SIZE x 2 per launch: max {max(v) * 2 * 1024 / 1e9:8.3f} GB  mean {sum(v) / len(v) * 2 * 1024 / 1e9:8.3f} GB')
PY
