#!/bin/bash
# Kernel-trace statistics of one probe script (one MI355X).
# Usage (repo root, GPU box):  bash tools/trace_kernels.sh <tag> <script.py> ; environment variables reach the script.
# Writes gpurun_out/trace_<tag>/ and prints the ten kernels with the most time.
set -e -o pipefail
TAG=$1
SCRIPT=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT" -o t --output-format csv -- python3 "$ROOT/$SCRIPT" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
cat "$OUT/stdout.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:10.1f} us total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
PY
