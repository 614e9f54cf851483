"""Scan bandwidth by number of folded embeddings (1 - 4) and row type, on 2M rows.
    python tools/fold_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scan_perf import run  # noqa: E402

for dtype, dim in (("f16", 1536), ("bf16", 768), ("f32", 1536)):
    for fold in (1, 2, 3, 4):
        run(dtype, dim, 2_000_000, 1, fold, 10, 10)
