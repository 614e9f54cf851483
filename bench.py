"""Headline benchmark: exact top-10 QPS on a 10M x 1536 fp32 corpus (batched, 4096 queries per step).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch: 4096 queries against the whole corpus, top-10
per query.  With N > 1 the 10M rows are split into N contiguous shards (strong scaling: the total
work is fixed), every rank searches its shard with the fp32 MFMA kernel and an RCCL all-gather of
the per-rank top-10 lists is merged on every rank.  The corpus is synthetic, generated in HBM by a
counter-based generator before the timed region.

The timed path computes in fp32 end to end (v_mfma_f32_32x32x2_f32, algo "mfma_f32").  The library's
own choices for such batches are faster and are timed after the headline, on the same resident corpus
and the same queries, under "aux": "split_f16x3" contracts a [hi | lo] fp16 split copy of the same
fp32 rows on the 16-bit matrix pipe (three products per element, fp32 accumulation, scores within
~1e-8 of exact arithmetic); "filter_f16_rescore_f32" (what algo "auto" / "mfma" run for k <= 64) uses
the hi slabs alone as a filter with a proven error bound, scores 4 k candidates per query again from
the fp32 rows in fp32 and searches a query again in full when the bound does not prove its list --
the same ids as the fp32 kernel.  Neither is the headline `value`, because their matrix products are
formed from 16-bit operands.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (mfma_search_kernel) against
the dense fp32 MFMA peak: algorithmic flops = 2 * queries * shard rows * dim per launch, divided
by the kernel's mean duration measured with HIP events on the stream it runs on.  `cpu_baseline`
is the C restatement of the pgvector scan (oracle/pgv_scan.c, kind "port") timed on this host's
cores over a bounded sample and scaled linearly to the 10M-row corpus.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ROWS = 10_000_000
DIM = 1536
BATCH = 4096
TOPK = 10
SEED_CORPUS = 1234
SEED_QUERY = 4321
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak of MI355X
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 matrix peak
HBM_PEAK_GBS = 8000.0
VENDOR_GEMM_F16_TFLOPS = 1190.0  # what the vendor GEMM (hipBLASLt behind torch.matmul) sustains on this chip at 4096 x 262144 x 1536 fp16, product written out, nothing selected


def cpu_baseline(n_threads: int, rows: np.ndarray):
    """pgvector-style scan in C on the host over `rows` (a >= 1M-row slice of the benchmark corpus, exported from
    the GPU): queries/s scaled linearly to the 10M-row corpus."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import pgv_scan

    pgv_scan.load()
    sample_rows = rows.shape[0]
    rng = np.random.default_rng(SEED_QUERY)
    # calibrate: one query on one thread
    q1 = rng.standard_normal((1, DIM), dtype=np.float32)
    t0 = time.perf_counter()
    pgv_scan.scan_topk(rows, q1, TOPK)
    t1 = time.perf_counter() - t0
    # aim at ~20 s of CPU work in total, at least two queries per thread
    n_queries = int(max(2 * n_threads, min(4096, 20.0 / max(t1, 1e-4))))
    n_queries = (n_queries + n_threads - 1) // n_threads * n_threads
    queries = rng.standard_normal((n_queries, DIM), dtype=np.float32)

    def work(i):
        pgv_scan.scan_topk(rows, queries[i : i + 1], TOPK)  # ctypes releases the GIL

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=n_threads) as ex:
        list(ex.map(work, range(n_queries)))
    wall = time.perf_counter() - t0
    qps_sample = n_queries / wall
    qps_full = qps_sample * sample_rows / N_ROWS
    # second variant: the numpy statement of the same search (hirag.py:1305-1327: normalised rows, C @ q, argsort),
    # batched -- one sgemm of the sample against a block of queries on the BLAS threads, then argpartition + sort
    nb = 128
    qb = rng.standard_normal((nb, DIM), dtype=np.float32)
    t0 = time.perf_counter()
    inv = 1.0 / np.sqrt(np.einsum("ij,ij->i", rows, rows))
    sims = (qb / np.linalg.norm(qb, axis=1, keepdims=True)) @ rows.T
    sims *= inv[None, :]
    part = np.argpartition(-sims, TOPK, axis=1)[:, :TOPK]
    order = np.take_along_axis(part, np.argsort(np.take_along_axis(-sims, part, axis=1), axis=1), axis=1)
    wall_np = time.perf_counter() - t0
    assert order.shape == (nb, TOPK)
    return {
        "numpy_blas": {
            "value": nb / wall_np * sample_rows / N_ROWS,
            "unit": "queries/s",
            "sample": f"{nb} queries x {sample_rows} rows x {DIM} f32 in {wall_np:.2f} s wall (numpy sgemm on the BLAS threads + "
                      f"argpartition, the formulation of hirag.py:1305-1327 batched), scaled linearly to {N_ROWS} rows",
        },
        "value": qps_full,
        "unit": "queries/s",
        "cores": n_threads,
        "kind": "port",
        "sample": f"{n_queries} queries x {sample_rows} rows x {DIM} f32 (the first rows of the benchmark corpus) in {wall:.2f} s "
                  f"wall on {n_threads} threads (oracle/pgv_scan.c, pgvector-style scan + top-{TOPK} heap), scaled linearly to "
                  f"{N_ROWS} rows",
    }


def stats_ms(samples):
    """median / p95 / mean of per-iteration times in ms"""
    a = np.sort(np.asarray(samples, dtype=np.float64))
    return {"median_ms": float(np.median(a)), "p95_ms": float(a[min(len(a) - 1, int(np.ceil(0.95 * len(a))) - 1)]),
            "mean_ms": float(a.mean()), "iterations": int(len(a))}


def timed_calls(torch, fn, iters, warm=3):
    """per-call wall time (ms) of fn(), each call bracketed by a device synchronisation"""
    for _ in range(warm):
        fn()
    out = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return out


def kernel_ms(idx, which, fn, torch, iters, warm=2, per_call=False):
    """per-launch device time of kernel classes (HIP events on the index's stream), one sample per call of fn();
    `which`: one counter -> list of ms, or a tuple of counters -> one list per counter; per_call: the sum over the
    launches of one call instead of their mean (a search that starts with a pass over a sample launches twice)"""
    many = isinstance(which, tuple)
    counters = which if many else (which,)
    for _ in range(warm):
        fn()
    idx.profile(True)
    out = [[] for _ in counters]
    for _ in range(iters):
        idx.profile_reset()
        fn()
        for o, w in zip(out, counters):
            ms, n = idx.profile_get(w)
            o.append(ms if per_call else ms / max(n, 1))
    idx.profile(False)
    return out if many else out[0]


def leg_cfg2(hvs, _lib, torch, dev):
    """BASELINE config 2: 1M x 1536 fp32, one query per call, top-10 (GEMV, HBM-bound)."""
    n = 1_000_000
    idx = hvs.HvsIndex(DIM, "f32", device=dev.index)
    idx.fill_synthetic(n, seed=SEED_CORPUS)
    gen = torch.Generator(device=dev).manual_seed(SEED_QUERY)
    q = torch.randn((32, DIM), device=dev, dtype=torch.float32, generator=gen)
    qh = q.cpu().numpy()
    state = {"i": 0}

    def one(algo, host):
        def f():
            i = state["i"] = (state["i"] + 1) % 32
            if host:
                idx.search(qh[i : i + 1], k=TOPK, algo=algo)
            else:
                idx.search(q[i : i + 1], k=TOPK, algo=algo)
        return f

    bytes_q = n * DIM * 4
    kern = stats_ms(kernel_ms(idx, _lib.HVS_PROF_SCAN, one("scan", False), torch, 30))
    e2e = stats_ms(timed_calls(torch, one("scan", True), 30))
    out = {
        "workload": f"{n} x {DIM} fp32, 1 query per call, top-{TOPK}, no filter",
        "scan_kernel": kern,
        "scan_kernel_GBps": bytes_q / (kern["median_ms"] * 1e-3) / 1e9,
        "scan_kernel_frac_of_hbm_peak": bytes_q / (kern["median_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "algorithmic_bytes_per_query": bytes_q,
        "end_to_end_host_buffers": e2e,
        "end_to_end_qps": 1e3 / e2e["median_ms"],
    }
    # the library's own choice once the table's split copy exists: hi slabs as a filter, exact fp32 rescoring
    idx.prepare()
    os.environ["HVS_FILTER_MIN_ROWS"] = "1"
    try:
        e2e_auto = stats_ms(timed_calls(torch, one("auto", True), 30))
    finally:
        del os.environ["HVS_FILTER_MIN_ROWS"]
    out["end_to_end_filter_route"] = e2e_auto
    out["end_to_end_filter_route_qps"] = 1e3 / e2e_auto["median_ms"]
    idx.close()
    return out


def leg_cfg4(idx, hvs, _lib, torch, dev, rows):
    """BASELINE config 4: one Lloyd iteration, K = 4096 centroids (the first rows) over the resident 10M corpus."""
    K = 4096
    c = torch.from_numpy(idx.gather(np.arange(K))).to(dev)
    out = {"workload": f"{rows} x {DIM} fp32 rows, K = {K}, one Lloyd iteration (assign + partial sums + centroid division)"}
    flops = 2.0 * K * rows * DIM
    # the library's default: hi slabs as a filter (one product, best two values per centroid tile and row), exact fp32
    # values where the bound leaves several centroids open, all three products for the rows it cannot settle;
    # then all three products for every row; then the fp32 MFMA kernel
    for label, env in (("filter_f16_exact_fp32", None), ("split_f16x3", "HVS_NO_FILTER"), ("fp32_mfma", "HVS_NO_SPLIT")):
        if env:
            os.environ[env] = "1"
        try:
            def it():
                assign, sums, counts = idx.kmeans_assign(c)
                torch.where((counts > 0).unsqueeze(1), sums / counts.clamp(min=1).unsqueeze(1).float(), c)
            iters = 3 if env == "HVS_NO_SPLIT" else 5
            wall = stats_ms(timed_calls(torch, it, iters, warm=1))
            idx.profile_reset()
            a, u = (stats_ms(v) for v in kernel_ms(idx, (_lib.HVS_PROF_KMEANS_ASSIGN, _lib.HVS_PROF_KMEANS_UPDATE), it, torch, iters, warm=0))
            flagged = idx.profile_get(_lib.HVS_PROF_REDO)[1]  # of the last iteration (the counters are reset per call)
        finally:
            if env:
                del os.environ[env]
        peak = FP32_MFMA_PEAK_TFLOPS if env == "HVS_NO_SPLIT" else F16_MFMA_PEAK_TFLOPS
        executed = flops * (1.0 if env == "HVS_NO_SPLIT" else (3.0 if env else 1.0 + 3.0 * flagged / rows))
        out[label] = {
            "iteration": wall,
            "assign_kernel": a,
            "assign_algorithmic_TFLOPs": flops / (a["median_ms"] * 1e-3) / 1e12,
            "assign_executed_TFLOPs": executed / (a["median_ms"] * 1e-3) / 1e12,
            "assign_frac_of_peak_executed": executed / (a["median_ms"] * 1e-3) / 1e12 / peak,
            "peak_TFLOPs": peak,
            "update_pass": u,
            "update_GBps": rows * DIM * 4 / (u["median_ms"] * 1e-3) / 1e9,
        }
        if env is None:
            out[label]["rows_sent_through_all_three_products"] = int(flagged)
            out[label]["rows_sent_through_all_three_products_frac"] = flagged / rows
    return out


def leg_cfg5(hvs, _lib, torch, dev, n=12_500_000, first_global_row=25_000_000, what="shard (rank 2 of 8 of a 100M-row table)",
             arrives_in_order=False):
    """BASELINE config 5: 12.5M x 768 bf16 (one GPU's 1/8 of 100M rows) or the whole 100M-row table on one GPU; top-100,
    64 workspaces.  The table is put in group order first (hvs_index_partition), as the store keeps its tables: the
    workspace filter then selects a row range.  The whole table (154 GB) cannot be reordered on one GPU -- the rows move
    through a second set of arrays -- so it is loaded in group order (a bulk load ordered by workspace), which
    hvs_index_partition recognises and only indexes."""
    d, k = 768, 100
    idx = hvs.HvsIndex(d, "bf16", device=dev.index)
    idx.reserve(n)
    idx.fill_synthetic(n, seed=SEED_CORPUS, first_global_row=first_global_row, n_groups=-64 if arrives_in_order else 64)
    gen = torch.Generator(device=dev).manual_seed(SEED_QUERY)
    q = torch.randn((256, d), device=dev, dtype=torch.float32, generator=gen)
    codes = "64 equal runs of group codes, in order" if arrives_in_order else "group code = row mod 64"
    out = {"workload": f"{n} x {d} bf16 {what}, top-{k}, {codes}"}
    t0 = time.perf_counter()
    try:
        idx.partition()
        out["group_order"] = {"seconds": time.perf_counter() - t0, "rows_in_order": idx.partition_info()[0]}
    except hvs.HvsError as e:
        out["group_order"] = {"skipped": str(e)}
    for label, kw, nq, algo, which in (
        ("single_query", {}, 1, "scan", _lib.HVS_PROF_SCAN),
        ("single_query_workspace_filter", {"group_filter": 3}, 1, "scan", _lib.HVS_PROF_SCAN),
        ("batch_256", {}, 256, "mfma", _lib.HVS_PROF_MFMA_SEARCH),
        ("batch_256_workspace_filter", {"group_filter": 3}, 256, "mfma", _lib.HVS_PROF_MFMA_SEARCH),
    ):
        qq = q[:nq].contiguous()

        def f():
            idx.search(qq, k=k, algo=algo, **kw)

        iters = 20 if n <= 20_000_000 else 8
        wall = stats_ms(timed_calls(torch, f, iters))
        kern = stats_ms(kernel_ms(idx, which, f, torch, iters, per_call=True))  # all launches of one call
        live = n // 64 if kw else n
        hbm_floor = live * d * 2 / (HBM_PEAK_GBS * 1e9) * 1e3
        mfma_floor = 2.0 * nq * live * d / (F16_MFMA_PEAK_TFLOPS * 1e12) * 1e3
        out[label] = {
            "call": wall, "qps": nq * 1e3 / wall["median_ms"], "kernel": kern,
            "live_rows": live, "hbm_floor_ms": hbm_floor, "mfma_floor_ms": mfma_floor,
            "kernel_over_floor": kern["median_ms"] / max(hbm_floor, mfma_floor),
            "mfma_floor_at_vendor_gemm_rate_ms": 2.0 * nq * live * d / (VENDOR_GEMM_F16_TFLOPS * 1e12) * 1e3,
            "kernel_GBps_of_live_bytes": live * d * 2 / (kern["median_ms"] * 1e-3) / 1e9,
            "kernel_TFLOPs": 2.0 * nq * live * d / (kern["median_ms"] * 1e-3) / 1e12,
        }
    if not arrives_in_order and "rows_in_order" in out.get("group_order", {}):
        # rows that arrive after the table was put in order form a tail behind it (the store reorders once the tail is a
        # quarter of the table): 5 % more rows, then the filtered calls again -- one scan launch over range + tail for the
        # single query, range and tail separately plus a merge for the batch
        extra = n // 20
        idx.fill_synthetic(extra, seed=SEED_CORPUS, first_global_row=first_global_row + n, n_groups=64)
        tail = {"rows_behind_the_ordered_part": extra}
        for label, nq, algo, which in (("single_query_workspace_filter", 1, "scan", _lib.HVS_PROF_SCAN),
                                       ("batch_256_workspace_filter", 256, "mfma", _lib.HVS_PROF_MFMA_SEARCH)):
            qq = q[:nq].contiguous()

            def g():
                idx.search(qq, k=k, algo=algo, group_filter=3)

            tail[label] = {"call": stats_ms(timed_calls(torch, g, 20)), "kernel": stats_ms(kernel_ms(idx, which, g, torch, 20, per_call=True))}
        out["with_a_tail_of_5_percent"] = tail
    idx.close()
    return out


def leg_cfg5_full(hvs, _lib, torch, dev):
    """The N = 1 anchor of configuration 5: all 100M x 768 bf16 rows (154 GB) on one GPU."""
    return leg_cfg5(hvs, _lib, torch, dev, n=100_000_000, first_global_row=0, what="table, whole, on one GPU", arrives_in_order=True)


def leg_clustered(hvs, _lib, torch, dev, batch):
    """The library's default route on a corpus of tight clusters (where the filter pass's bound proves little)."""
    n, n_clusters = 2_000_000, 10_000
    idx = hvs.HvsIndex(DIM, "f32", device=dev.index)
    idx.fill_clustered(n, seed=SEED_CORPUS, n_clusters=n_clusters)
    qi = hvs.HvsIndex(DIM, "f32", device=dev.index)
    qi.fill_clustered(batch, seed=SEED_CORPUS, first_global_row=1_000_000_000, n_clusters=n_clusters)
    q = torch.from_numpy(qi.export()[0]).to(dev)
    qi.close()
    idx.prepare()
    out = {"workload": f"{n} x {DIM} fp32 rows in {n_clusters} clusters (cosine spread 1e-3..1e-2 inside a cluster), "
                       f"{batch} queries of the same distribution per call, top-{TOPK}"}
    os.environ["HVS_NO_ADAPT"] = "1"
    try:
        idx.profile_reset()
        d_f, i_f, _ = idx.search(q, k=TOPK, algo="mfma")
        redone = idx.profile_get(_lib.HVS_PROF_REDO)[1]
        fixed = stats_ms(timed_calls(torch, lambda: idx.search(q, k=TOPK, algo="mfma"), 5, warm=1))
    finally:
        del os.environ["HVS_NO_ADAPT"]
    d_s, i_s, _ = idx.search(q, k=TOPK, algo="mfma_split")
    auto = stats_ms(timed_calls(torch, lambda: idx.search(q, k=TOPK), 20, warm=2))
    out.update({
        "queries_searched_again_in_full": int(redone), "queries": batch,
        "filter_route_every_call": fixed, "filter_route_every_call_qps": batch * 1e3 / fixed["median_ms"],
        "default_route_with_back_off": auto, "default_route_qps": batch * 1e3 / auto["median_ms"],
        "ids_equal_to_three_product_route": float((i_f == i_s).float().mean().item()),
        # inside a cluster hundreds of rows lie within a few 1e-7 of each other: the two routes agree on the distances and
        # may order such rows differently (the parity bar allows swaps between distances that tie within 1e-6)
        "max_abs_distance_diff_vs_three_product_route": float((d_f - d_s).abs().max().item()),
        "ids_equal_or_tied_within_1e-6": float(((i_f == i_s) | ((d_f - d_s).abs() <= 1e-6)).float().mean().item()),
    })
    idx.close()
    return out


class Watchdog:
    """N > 1 only.  More than one RCCL rank meets for the first time in the driver's scaling run: if a collective hangs,
    every rank leaves after `seconds` instead of holding the node, and rank 0 first prints the JSON line with what it has
    (the headline, once that exists) and the reason."""

    def __init__(self, rank, seconds):
        import threading

        self.rank, self.seconds, self.result, self.leg = rank, seconds, None, "headline"
        self._timer = threading.Timer(seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()

    def _fire(self):
        if self.rank == 0 and self.result is not None:
            self.result.setdefault("aux", {})["watchdog"] = f"no progress within {self.seconds:.0f} s during '{self.leg}': the legs after it are missing"
            print(json.dumps(self.result), flush=True)
        print(f"[rank {self.rank}] bench.py watchdog: still in '{self.leg}' after {self.seconds:.0f} s, leaving", file=sys.stderr, flush=True)
        os._exit(0 if self.rank == 0 and self.result is not None else 3)

    def cancel(self):
        self._timer.cancel()


def sharded_leg(dist, name, fn, *a):
    """A leg every rank takes part in.  A rank that fails ahead of the leg's collectives must not leave the others waiting
    in them: set-up errors are agreed on first (`ready`), and the collectives themselves carry a status word."""
    try:
        return fn(dist, *a)
    except Exception as e:  # noqa: BLE001 -- reported in the leg's place
        import traceback

        traceback.print_exc()
        return {"error": f"{type(e).__name__}: {e}"}


def all_ready(dist, torch, ok):
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def leg_cfg4_sharded(dist, idx, comm, collective, hvs, _lib, torch, dev, total_rows, max_over_ranks):
    """BASELINE config 4 at N > 1: one Lloyd iteration over the row shards of the resident 10M corpus -- local assignment
    and partial sums, the all-reduce of sums and counts inside libhvs, the division."""
    from hirag_prod_amd.sharded import _ptr
    from hirag_prod_amd._lib import check

    K = 4096
    tmp = hvs.HvsIndex(DIM, "f32", device=dev.index)
    tmp.fill_synthetic(K, seed=SEED_CORPUS, first_global_row=0)  # the centroids: global rows 0 .. K-1, regenerated on every rank
    c = torch.from_numpy(tmp.gather(np.arange(K))).to(dev)
    tmp.close()
    walls, assigns, reduces = [], [], []
    for it in range(4):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        idx.profile(True)
        idx.profile_reset()
        assign, sums, counts = idx.kmeans_assign(c)
        a_ms = idx.profile_get(_lib.HVS_PROF_KMEANS_ASSIGN)[0] + idx.profile_get(_lib.HVS_PROF_KMEANS_UPDATE)[0]
        idx.profile(False)
        idx.timer_start()
        with idx._lock:
            check(idx._lib.hvs_allreduce_centroids(idx._h, comm._h, _ptr(sums), _ptr(counts), K, 1))
            check(idx._lib.hvs_sync(idx._h))
        r_ms = idx.timer_stop()
        new_c = torch.where((counts > 0).unsqueeze(1), sums / counts.clamp(min=1).unsqueeze(1).float(), c)
        torch.cuda.synchronize()
        wall = max_over_ranks((time.perf_counter() - t0) * 1e3)
        if it:  # the first iteration builds the shard's search-side copy
            walls.append(wall)
            assigns.append(a_ms)
            reduces.append(r_ms)
    assert int(counts.sum().item()) == total_rows and bool(torch.isfinite(new_c).all().item())
    return {"workload": f"{total_rows} x {DIM} fp32 rows over {comm.world} GPUs, K = {K}, one Lloyd iteration (assign + partial sums + "
                        f"all-reduce of {K * DIM * 4 + K * 8} bytes + division)",
            "transport": collective, "iteration_max_over_ranks": stats_ms(walls), "assign_and_update_rank0": stats_ms(assigns),
            "all_reduce_rank0": stats_ms(reduces)}


def leg_cfg5_sharded(dist, comm, collective, hvs, _lib, torch, dev, rank, world, max_over_ranks):
    """BASELINE config 5 at N > 1: 100M x 768 bf16 rows split over the GPUs, top-100, 64 workspaces; every shard in group
    order; one packed all-gather and a merge per call (hvs_search_sharded)."""
    from hirag_prod_amd.sharded import ShardedSearch, shard_range

    n_total, d, k = int(os.environ.get("BENCH_CFG5_ROWS", "100000000")), 768, 100
    lo, hi = shard_range(n_total, rank, world)
    idx, err = None, None
    try:
        idx = hvs.HvsIndex(d, "bf16", device=dev.index)
        idx.reserve(hi - lo)
        idx.fill_synthetic(hi - lo, seed=SEED_CORPUS, first_global_row=lo, n_groups=64)
        try:
            idx.partition()
        except hvs.HvsError as e:  # device memory for the second set of arrays: the shard keeps the per-row test
            print(f"[rank {rank}] cfg5 shard stays in arrival order: {e}", file=sys.stderr, flush=True)
    except Exception as e:  # noqa: BLE001
        err = e
    if not all_ready(dist, torch, err is None):
        if idx is not None:
            idx.close()
        raise RuntimeError(f"a rank could not set its shard up ({err})")
    searcher = ShardedSearch.for_index(idx, comm)
    gen = torch.Generator(device=dev).manual_seed(SEED_QUERY)
    q = torch.randn((256, d), device=dev, dtype=torch.float32, generator=gen)
    out = {"workload": f"{n_total} x {d} bf16 rows over {world} GPUs ({hi - lo} per GPU), top-{k}, group code = row mod 64",
           "transport": collective, "rows_in_group_order_rank0": idx.partition_info()[0]}
    for label, kw, nq, algo in (("single_query", {}, 1, "scan"), ("single_query_workspace_filter", {"group_filter": 3}, 1, "scan"),
                                ("batch_256", {}, 256, "mfma"), ("batch_256_workspace_filter", {"group_filter": 3}, 256, "mfma")):
        qq = q[:nq].contiguous()
        for _ in range(3):
            searcher.search(qq, k, algo=algo, **kw)
        walls = []
        idx.profile(True)
        idx.profile_reset()
        for _ in range(12):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            dd, ii, cc = searcher.search(qq, k, algo=algo, **kw)
            torch.cuda.synchronize()
            walls.append(max_over_ranks((time.perf_counter() - t0) * 1e3))
        scan_ms, scan_n = idx.profile_get(_lib.HVS_PROF_SCAN)
        mfma_ms, mfma_n = idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)
        merge_ms, merge_n = idx.profile_get(_lib.HVS_PROF_MERGE)
        idx.profile(False)
        assert int(cc.min().item()) == k and bool((dd[:, 1:] >= dd[:, :-1]).all().item())
        if kw:
            assert bool((ii % 64 == 3).all().item())
        out[label] = {"call_max_over_ranks": stats_ms(walls), "qps": nq * 1e3 / float(np.median(walls)),
                      "search_kernels_ms_per_call_rank0": (scan_ms + mfma_ms) / 12, "merges_ms_per_call_rank0": merge_ms / 12,
                      "all_gather_bytes_per_rank": nq * (k * 12 + 4)}
    dist.barrier()
    idx.close()
    return out


# Per-configuration legs that fill a table of their own: each runs in a child process after the headline table is
# freed, so that whatever happens in one of them (a failed assertion, a device fault that ends its process) costs that
# leg and nothing else -- the parent prints the one JSON line either way.
CHILD_LEGS = {"cfg2": leg_cfg2, "cfg5_shard": leg_cfg5, "cfg5_full": leg_cfg5_full, "clustered": leg_clustered}


def run_child_leg(name, batch):
    import torch

    import hirag_prod_amd as hvs
    from hirag_prod_amd import _lib

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    extra = (batch,) if name == "clustered" else ()
    print("LEG-RESULT " + json.dumps(CHILD_LEGS[name](hvs, _lib, torch, dev, *extra)), flush=True)


def child_leg(name, batch, timeout_s=600):
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", name, "--batch", str(batch)], capture_output=True, text=True,
                           timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"leg {name}: no result within {timeout_s} s"}
    except OSError as e:  # no child process to be had here: run the leg in this one
        print(f"bench.py: could not start a child process for leg {name} ({e}); running it in-process", file=sys.stderr, flush=True)
        try:
            import torch

            import hirag_prod_amd as hvs
            from hirag_prod_amd import _lib

            dev = torch.device("cuda", torch.cuda.current_device())
            return CHILD_LEGS[name](hvs, _lib, torch, dev, *((batch,) if name == "clustered" else ()))
        except Exception as e2:  # noqa: BLE001 -- reported in the leg's place
            return {"error": f"leg {name}: {type(e2).__name__}: {e2}"}
    for line in r.stdout.splitlines():
        if line.startswith("LEG-RESULT "):
            return json.loads(line[len("LEG-RESULT "):])
    return {"error": f"leg {name}: exit status {r.returncode}", "stderr_tail": r.stderr[-800:]}


def measured_traffic(rows_per_gpu, batch):
    """HBM-side bytes per launch of the search kernel from the committed counter passes (rocprofv3
    cannot be driven from inside this process); None when they were taken at another size."""
    for name in ("r3_traffic.json", "r2_traffic.json"):  # the latest pass that covers this size
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                t = json.load(fh)["mfma_search_kernel"]
        except (OSError, KeyError, ValueError):
            continue
        if t["rows_per_gpu"] == rows_per_gpu and t["batch"] == batch and t["dim"] == DIM:
            return t["bytes"], f"profiles/{name}: " + t["how"]
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=N_ROWS, help="total corpus rows (default: the headline 10M)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true")
    ap.add_argument("--leg", choices=sorted(CHILD_LEGS), help="run one per-configuration leg alone and print its JSON object (used by the main run)")
    args = ap.parse_args()
    if args.leg:
        return run_child_leg(args.leg, args.batch)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # torch.distributed carries a 128-byte rendezvous token, the timing barriers and the max over ranks of one float: a gloo
    # group over CPU tensors does all three.  No "nccl" group is opened, so the RCCL communicator inside libhvs is the only
    # one this process ever forms.
    if world > 1:
        from datetime import timedelta

        dist.init_process_group("gloo", timeout=timedelta(seconds=900))
        print(f"[rank {rank}] process group backend: {dist.get_backend()}", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        if world > 1:  # a launcher on a box without a GPU: meet the other ranks, then say what is missing
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit("bench.py: no HIP device visible (this benchmark has no CPU path)")
    if os.environ.get("BENCH_ONE_DEVICE"):  # rehearsal of the N > 1 code path on a one-GPU box: every rank on GPU 0, where no RCCL
        local_rank = 0                      # communicator can form, so the exchange runs host-staged over the gloo group
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    watchdog = Watchdog(rank, float(os.environ.get("BENCH_WATCHDOG_S", "1500"))) if world > 1 else None

    def max_over_ranks(value):
        t = torch.tensor([value], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    import hirag_prod_amd as hvs
    from hirag_prod_amd import _lib
    from hirag_prod_amd.sharded import HvsComm, ShardedSearch, shard_range

    # the exchange between the shards runs inside libhvs: RCCL on the index's stream.  torch.distributed carries the
    # rendezvous token and the timing barriers only.  If the RCCL communicator cannot be formed the same exchange runs
    # host-staged over a gloo group, and the JSON line says so.
    comm, collective = None, None
    if world > 1:
        try:
            comm = HvsComm.from_process_group(device=local_rank, transport="rccl")
            collective = "rccl (ncclAllGather in libhvs)"
        except hvs.HvsError as e:
            print(f"[rank {rank}] RCCL communicator failed ({e}); falling back to the host-staged transport", file=sys.stderr, flush=True)
        ok = torch.tensor([1 if comm is not None else 0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.close()
            comm = HvsComm.from_process_group(device=local_rank, transport="host")
            collective = "host-staged over gloo (RCCL communicator could not be formed)"

    lo, hi = shard_range(args.rows, rank, world)
    idx = hvs.HvsIndex(DIM, "f32", device=local_rank)
    idx.reserve(hi - lo)
    idx.fill_synthetic(hi - lo, seed=SEED_CORPUS, first_global_row=lo)
    gen = torch.Generator(device=dev).manual_seed(SEED_QUERY)
    queries = torch.randn((args.batch, DIM), device=dev, dtype=torch.float32, generator=gen)
    searcher = ShardedSearch.for_index(idx, comm)

    def step():
        return searcher.search(queries, TOPK, algo="mfma_f32")

    for _ in range(args.warmup):
        step()
    idx.profile(True)
    idx.profile_reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)
    idx.profile(False)
    if world > 1:
        elapsed = max_over_ranks(elapsed)

    # median / p95 of single steps, each synchronised on its own (outside the timed region above)
    step_stats = stats_ms(timed_calls(torch, step, min(max(args.steps, 5), 10), warm=0)) if world == 1 else None

    # sanity on the last result: sorted, full, ids in range (not a parity check; tests do that)
    d_last, i_last, c_last = out
    assert int(c_last.min().item()) == TOPK
    assert bool((d_last[:, 1:] >= d_last[:, :-1]).all().item())
    assert int(i_last.min().item()) >= 0 and int(i_last.max().item()) < args.rows

    result = None
    if rank == 0:
        per_launch_ms = kern_ms / max(kern_n, 1)
        flops = 2.0 * args.batch * (hi - lo) * DIM
        achieved = flops / (per_launch_ms * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic(hi - lo, args.batch)
        result = {
            "metric": "exact top-10 QPS, 10M x 1536 corpus",
            "value": args.batch * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "step_ms_stats": step_stats,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.rows} x {DIM} fp32 corpus row-sharded over {world} GPU(s), batches of "
                            f"{args.batch} queries, exact cosine top-{TOPK} (fp32 MFMA GEMM + fused top-k, "
                            f"one packed all-gather + merge per batch)",
                "rows": args.rows,
                "dim": DIM,
                "batch": args.batch,
                "k": TOPK,
                "parallelism": f"row-shard x{world}",
                "collective": collective,
                "process_group": "gloo (rendezvous token, barriers, max over ranks)" if world > 1 else None,
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "mfma_search_kernel<f32, fold 1>",
                "achieved": achieved,
                "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_unit": "bytes per launch",
                "traffic_source": traffic_src,
                "ms_per_launch": per_launch_ms,
                "launches": kern_n,
            },
        }

    if watchdog is not None:
        watchdog.result = result  # from here on a hang costs the legs below, not the headline
    # single-query (GEMV, HBM-bound) figures on the same resident shard, outside the timed region
    if not args.no_aux:
        q1 = queries[:1].contiguous()
        for _ in range(3):
            idx.search(q1, k=TOPK, algo="scan")
        idx.profile(True)
        idx.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            searcher.search(q1, TOPK, algo="scan")
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t0
        sms, sn = idx.profile_get(_lib.HVS_PROF_SCAN)
        idx.profile(False)
        if rank == 0:
            gbs = (hi - lo) * DIM * 4 / (sms / sn * 1e-3) / 1e9
            result["aux"] = {
                "single_query_qps": reps / e1,
                "scan_kernel_ms": sms / sn,
                "scan_hbm_GBps": gbs,
                "scan_hbm_frac": gbs / HBM_PEAK_GBS,
            }

    # the routes the library takes for this batch when left to choose: fp32 rows as [hi | lo] fp16 slabs on the 16-bit
    # matrix pipe -- all three products ("mfma_split"), or one product as a filter and exact fp32 scores for 4 k
    # candidates per query ("mfma", what AUTO picks for k <= 64)
    if not args.no_aux:
        d_ref, i_ref, _ = out

        def timed_route(algo):
            searcher.search(queries, TOPK, algo=algo)  # builds the split copy of the shard (once per table)
            idx.profile(True)
            idx.profile_reset()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                d_s, i_s, _ = searcher.search(queries, TOPK, algo=algo)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            e2 = time.perf_counter() - t0
            kms, kn = idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)
            sms, sn = idx.profile_get(_lib.HVS_PROF_SCAN)
            idx.profile(False)
            if world > 1:
                e2 = max_over_ranks(e2)
            per = kms / max(kn, 1)
            alg = 2.0 * args.batch * (hi - lo) * DIM / (per * 1e-3) / 1e12
            return {
                "qps": args.batch * args.steps / e2,
                "ms_per_step": e2 / args.steps * 1e3,
                "ms_per_launch": per,
                "algorithmic_TFLOPs": alg,
                "max_abs_score_diff_vs_fp32_kernel": float((d_s - d_ref).abs().max().item()),
                "ids_equal_to_fp32_kernel": float((i_s == i_ref).float().mean().item()),
                "extra_device_memory_bytes": int((hi - lo) * ((DIM + 63) // 64 * 64) * 4),
            }, sn

        split, _ = timed_route("mfma_split")
        filt, redone = timed_route("mfma")
        if rank == 0:
            split["kernel"] = "mfma_search_kernel<f16, fold 1, Tile<256,256,2,4>> over [hi | lo] slabs, K = 3 x dim"
            split["executed_TFLOPs"] = 3.0 * split["algorithmic_TFLOPs"]
            split["frac_of_f16_peak_algorithmic"] = split["algorithmic_TFLOPs"] / F16_MFMA_PEAK_TFLOPS
            split["frac_of_f16_peak_executed"] = split["executed_TFLOPs"] / F16_MFMA_PEAK_TFLOPS
            result["aux"]["split_f16x3"] = split
            filt["kernel"] = ("mfma_search_kernel<f16, fold 1, Tile<256,256,2,4>> over the hi slabs (top-40 by approximate score, "
                              "error bound 1.07e-3), then rescore_kernel: fp32 dot products of the candidates from the fp32 rows")
            filt["executed_TFLOPs"] = filt["algorithmic_TFLOPs"]
            filt["frac_of_f16_peak_executed"] = filt["algorithmic_TFLOPs"] / F16_MFMA_PEAK_TFLOPS
            filt["queries_searched_again_in_full"] = int(redone)  # scan launches: one per query whose list was not proven
            result["aux"]["filter_f16_rescore_f32"] = filt

        # one query at a time, left to the library: with the split copy in place it streams the hi slabs (half the bytes
        # of the fp32 rows) as a filter and scores the candidates exactly
        d_scan, i_scan, _ = searcher.search(q1, TOPK, algo="scan")
        for _ in range(3):
            searcher.search(q1, TOPK)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            d_auto, i_auto, _ = searcher.search(q1, TOPK)
        torch.cuda.synchronize()
        e4 = time.perf_counter() - t0
        if rank == 0:
            result["aux"]["single_query_auto"] = {
                "qps": 20 / e4, "ms_per_query": e4 / 20 * 1e3,
                # the route streams the fp16 hi slabs of the split copy: 2 bytes per element, half the fp32 rows
                "bytes_streamed_per_query": (hi - lo) * ((DIM + 63) // 64 * 64) * 2,
                "streamed_bytes_per_s_over_hbm_peak": (hi - lo) * ((DIM + 63) // 64 * 64) * 2 / (e4 / 20) / (HBM_PEAK_GBS * 1e9),
                "ids_equal_to_scan": float((i_auto == i_scan).float().mean().item()),
                "max_abs_score_diff_vs_scan": float((d_auto - d_scan).abs().max().item()),
            }

        # mid-size batch on the same corpus: 64 queries per call (the library's default route: ring kernel on the split copy)
        q64 = queries[:64].contiguous()
        for _ in range(2):
            searcher.search(q64, TOPK)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        reps64 = 10
        for _ in range(reps64):
            searcher.search(q64, TOPK)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e3 = time.perf_counter() - t0
        if rank == 0:
            result["aux"]["batch_of_64"] = {"qps": 64 * reps64 / e3, "ms_per_batch": e3 / reps64 * 1e3,
                                            "hbm_floor_ms": (hi - lo) * DIM * 4 / (HBM_PEAK_GBS * 1e9) * 1e3}

    if rank == 0:
        # timed at N = 1 only, on this host's cores (at most 16: the GPU box's CPU share per GPU)
        if world == 1 and not args.no_cpu_baseline:
            sample = min(hi - lo, 1_048_576)
            result["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1), idx.export(0, sample)[0])
        else:
            result["cpu_baseline"] = None
    # the other BASELINE configurations, each on a freshly filled table of its own (one GPU only: they are per-GPU figures).
    # A leg that fails says so in its place; the headline line is printed regardless.
    def aux_leg(name, fn, *a):
        try:
            result["aux"][name] = fn(*a)
        except Exception as e:  # noqa: BLE001 -- reported, not swallowed
            import traceback

            traceback.print_exc()
            result["aux"][name] = {"error": f"{type(e).__name__}: {e}"}

    if world == 1 and not args.no_aux:
        aux_leg("cfg4_kmeans", leg_cfg4, idx, hvs, _lib, torch, dev, hi - lo)
    if world > 1 and not args.no_aux:
        watchdog.leg = "cfg4_kmeans over the shards"
        out4 = sharded_leg(dist, "cfg4_kmeans", leg_cfg4_sharded, idx, comm, collective, hvs, _lib, torch, dev, args.rows, max_over_ranks)
        if rank == 0:
            result["aux"]["cfg4_kmeans"] = out4
    if world > 1:
        dist.barrier()
    idx.close()
    if world > 1 and not args.no_aux:
        torch.cuda.empty_cache()
        watchdog.leg = "cfg5 over the shards"
        out5 = sharded_leg(dist, "cfg5", leg_cfg5_sharded, comm, collective, hvs, _lib, torch, dev, rank, world, max_over_ranks)
        if rank == 0:
            result["aux"]["cfg5"] = out5
    if world == 1 and not args.no_aux:
        torch.cuda.empty_cache()
        for name in ("cfg2", "cfg5_shard", "clustered", "cfg5_full"):
            result["aux"][name] = child_leg(name, args.batch)
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
