"""Per kernel: mean duration (kernel trace), GRBM cycles per XCD and launch -> sustained clock, MFMA-pipe busy share.
    python tools/mfma_ceiling_summary.py gpurun_out/ceiling"""
import collections
import csv
import glob
import sys

out = sys.argv[1]
FLOP = 2.0 * 4096 * 2_000_000 * 1536
for what in ("search", "gemm"):
    stats = glob.glob(f"{out}/{what}_trace/**/*kernel_stats.csv", recursive=True)
    ctr = glob.glob(f"{out}/{what}_pmc/**/*counter_collection.csv", recursive=True)
    if not stats or not ctr:
        print(what, "missing output")
        continue
    dur = {}
    for r in csv.DictReader(open(stats[0])):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(ctr[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
    # the kernel that holds most of the MFMA cycles
    name = max(acc, key=lambda k: acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0))
    c, n = acc[name], len(launches[name])
    key = next((k for k in dur if k.split("(")[0] == name.split("(")[0]), None) or max(dur, key=lambda k: dur[k][0] * dur[k][1])
    calls, avg_ns = dur[key]
    gui_per_xcd = c["GRBM_GUI_ACTIVE"] / n / 8.0
    clock = gui_per_xcd / avg_ns  # GHz
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (1024.0 * gui_per_xcd)
    print(f"{what}: {name[:90]}")
    print(f"  launches counted {n}, traced {calls}; mean duration {avg_ns / 1e6:.3f} ms = {FLOP / avg_ns / 1e3:.1f} TFLOP/s")
    print(f"  GRBM_GUI_ACTIVE per XCD and launch {gui_per_xcd:.4e} -> sustained clock {clock:.3f} GHz")
    print(f"  SQ_VALU_MFMA_BUSY_CYCLES per launch {c['SQ_VALU_MFMA_BUSY_CYCLES'] / n:.4e} -> MFMA pipe busy {busy * 100:.1f} % of 1024 SIMDs x cycles")
    print(f"  busy x clock / 2.4 GHz = {busy * clock / 2.4:.3f}; wave time waiting at s_waitcnt / s_barrier {c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.3f}, "
          f"waiting to issue {c['SQ_WAIT_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.3f}")
