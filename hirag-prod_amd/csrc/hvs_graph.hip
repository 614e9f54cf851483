// Personalised PageRank over the knowledge graph, resident in HBM (gfx950).
//
// The reference rebuilds a networkx graph from the Graph table's rows on every query and runs
// networkx.pagerank on its undirected view (src/hirag_prod/storage/pgvector.py:381-450); the fusion
// step that calls it is src/hirag_prod/storage/query_service.py:246-402.  Here the undirected simple
// graph is built once per (re)load as a CSR adjacency on the device and each query is a handful of
// pull-style sparse matrix-vector products in float64 with the library's stopping rule.
//
// Layout: row_ptr int64 [n+1], col int32 [nnz] (ascending in each row), inv_deg double [n] (0 for an
// isolated node), three row lists by degree (sub-wave groups, one wave per row, several blocks per row), two
// score vectors x and two pre-scaled copies y = x * inv_deg (so the gather reads 8 bytes per edge and
// no degree), a dense reset vector p.
//
// Per iteration and per edge the kernels read 4 bytes of column index and gather 8 bytes of y; per node
// 8 (row_ptr) + 4 (row list) + 8 (p) + 8 (x) + 8 (inv_deg) bytes are read and 16 written.  All sums use
// fixed reduction trees and per-block partials folded in index order: results are reproducible from run
// to run.  No floating-point atomics.
#include "hvs_internal.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <map>
#include <vector>

#define GRAPH_THREADS 256
#define GRAPH_MAX_BLOCKS 2048
#define SMALL_LPR 8
#define SMALL_MAX_DEG 16  // = 2 * SMALL_LPR: ppr_small_rows_kernel reads at most two entries per lane
#define MID_MAX_DEG 2048
#define BIG_SEGMENT 8192
#define TOP_BINS 4096  // sign + exponent of a double

struct hvs_graph {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0, nnz = 0, n_dangling = 0;
    int64_t *row_ptr = nullptr;
    int32_t *col = nullptr;
    double *inv_deg = nullptr;
    uint32_t *chunk_mask = nullptr;
    int32_t *rows[3] = {nullptr, nullptr, nullptr};
    int64_t n_rows[3] = {0, 0, 0};
    int blocks[3] = {0, 0, 0};
    double *x[2] = {nullptr, nullptr}, *y[2] = {nullptr, nullptr}, *p = nullptr;
    double *partials = nullptr;  // [2][sum of blocks]: |x' - x| and dangling mass per block
    double *scal = nullptr;      // [0] err of the last iteration, [1] dangling mass of the current iterate
    int *state = nullptr;        // [0] converged flag, [1] iterations done
    uint32_t *hist = nullptr;    // [TOP_BINS] + [1]: candidate count
    // rows of the third list are cut into segments of at most BIG_SEGMENT adjacency entries, one block each
    int64_t *seg_begin = nullptr;   // [n_segs] first adjacency entry
    int32_t *seg_len = nullptr;     // [n_segs]
    int32_t *row_seg = nullptr;     // [n_rows[2] + 1] first segment of each big row
    double *seg_sum = nullptr;      // [n_segs]
    int64_t n_segs = 0;
    int result = -1;             // which x buffer holds the last result
    hvs_workspace ws;
};

static int graph_ws(hvs_graph *g, size_t bytes, void **out) {
    if (g->ws.bytes < bytes) {
        if (g->ws.ptr) HVS_HIP(hipFree(g->ws.ptr));
        g->ws.ptr = nullptr;
        g->ws.bytes = 0;
        HVS_HIP(hipMalloc(&g->ws.ptr, bytes));
        g->ws.bytes = bytes;
    }
    *out = g->ws.ptr;
    return 0;
}

static inline int blocks_for(int64_t work, int per_block) {
    int64_t b = (work + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > GRAPH_MAX_BLOCKS ? GRAPH_MAX_BLOCKS : b));
}

// ---- build ---------------------------------------------------------------------------------------
__global__ void edge_keys_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ dst, int64_t n_edges, int64_t n_nodes,
                                 uint64_t *__restrict__ keys, int *__restrict__ bad) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_edges; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = src[e], v = dst[e];
        if (u < 0 || v < 0 || u >= n_nodes || v >= n_nodes) {
            *bad = 1;
            keys[2 * e] = keys[2 * e + 1] = 0;
            continue;
        }
        keys[2 * e] = ((uint64_t)u << 32) | (uint64_t)v;
        keys[2 * e + 1] = ((uint64_t)v << 32) | (uint64_t)u;
    }
}

__global__ void csr_cols_kernel(const uint64_t *__restrict__ ukeys, int64_t nnz, int32_t *__restrict__ col) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x)
        col[t] = (int32_t)(ukeys[t] & 0xFFFFFFFFull);
}

// row_ptr[r] = first position whose key has row >= r
__global__ void csr_rows_kernel(const uint64_t *__restrict__ ukeys, int64_t nnz, int64_t n, int64_t *__restrict__ row_ptr) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r <= n; r += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t want = (uint64_t)r << 32;
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (ukeys[mid] < want) lo = mid + 1;
            else hi = mid;
        }
        row_ptr[r] = lo;
    }
}

__global__ void degrees_kernel(const int64_t *__restrict__ row_ptr, int64_t n, double *__restrict__ inv_deg,
                               unsigned long long *__restrict__ n_dangling) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = row_ptr[r + 1] - row_ptr[r];
        inv_deg[r] = d ? 1.0 / (double)d : 0.0;
        if (d == 0) atomicAdd(n_dangling, 1ull);
    }
}

struct DegreeBin {
    const int64_t *row_ptr;
    int bin;
    __device__ bool operator()(const int32_t &r) const {
        const int64_t d = row_ptr[r + 1] - row_ptr[r];
        const int b = d <= SMALL_MAX_DEG ? 0 : (d <= MID_MAX_DEG ? 1 : 2);
        return b == bin;
    }
};

struct KeyEqual {
    __device__ bool operator()(const uint64_t &a, const uint64_t &b) const { return a == b; }
};

// ---- power iteration -----------------------------------------------------------------------------
struct PprParams {
    const int64_t *row_ptr;
    const int32_t *col;
    const double *inv_deg;
    const double *p;
    const double *x_in;
    const double *y_in;
    double *x_out;
    double *y_out;
    const double *scal;  // [1] dangling mass of x_in
    const int *state;    // [0] converged
    double *partials;    // [0 .. total) errors, [total .. 2 total) dangling mass
    int total_blocks;
    double alpha;
};

__device__ __forceinline__ double wave_sum(double v, int width) {
    for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum of the block's (err, dang) pairs in a fixed order -> partials[slot], partials[total + slot]
template <int THREADS>
__device__ __forceinline__ void block_partials(double err, double dang, const PprParams &p, int slot) {
    __shared__ double s_err[THREADS / 64], s_dang[THREADS / 64];
    err = wave_sum(err, 64);
    dang = wave_sum(dang, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_err[wave] = err;
        s_dang[wave] = dang;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double e = 0.0, d = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) {
            e += s_err[w];
            d += s_dang[w];
        }
        p.partials[slot] = e;
        p.partials[p.total_blocks + slot] = d;
    }
}

__device__ __forceinline__ void finish_row(const PprParams &p, int64_t j, double sum, double dang_in, double &err, double &dang) {
    const double pj = p.p[j];
    const double xn = p.alpha * (sum + dang_in * pj) + (1.0 - p.alpha) * pj;
    const double inv = p.inv_deg[j];
    p.x_out[j] = xn;
    p.y_out[j] = xn * inv;
    err += fabs(xn - p.x_in[j]);
    if (inv == 0.0) dang += xn;
}

// LPR lanes per row (used with LPR = 64: one wave per row of 17 .. 2048 neighbours)
template <int LPR>
__global__ void __launch_bounds__(GRAPH_THREADS) ppr_rows_kernel(PprParams p, const int32_t *__restrict__ rows, int64_t n_rows,
                                                                 int slot_base) {
    if (p.state[0]) return;
    const int sub = threadIdx.x & (LPR - 1);
    const int64_t group = (blockIdx.x * (int64_t)GRAPH_THREADS + threadIdx.x) / LPR;
    const int64_t n_groups = (int64_t)gridDim.x * GRAPH_THREADS / LPR;
    const double dang_in = p.scal[1];
    double err = 0.0, dang = 0.0;
    // every lane of a group runs the same number of outer steps, so the shuffles below are convergent
    for (int64_t base = 0; base < n_rows; base += n_groups) {
        const int64_t r = base + group;
        double sum = 0.0;
        int64_t j = -1;
        if (r < n_rows) {
            j = rows[r];
            const int64_t b = p.row_ptr[j], e = p.row_ptr[j + 1];
            for (int64_t t = b + sub; t < e; t += LPR) sum += p.y_in[p.col[t]];
        }
        sum = wave_sum(sum, LPR);
        if (sub == 0 && j >= 0) finish_row(p, j, sum, dang_in, err, dang);
    }
    block_partials<GRAPH_THREADS>(err, dang, p, slot_base + blockIdx.x);
}

// rows of at most SMALL_MAX_DEG = 2 * SMALL_LPR neighbours: a group of SMALL_LPR lanes works on SMALL_ROWS rows at a time
// so that the four dependent loads of a row (row number, row_ptr, column, score) overlap across rows
#define SMALL_ROWS 4
__global__ void __launch_bounds__(GRAPH_THREADS) ppr_small_rows_kernel(PprParams p, const int32_t *__restrict__ rows, int64_t n_rows,
                                                                       int slot_base) {
    if (p.state[0]) return;
    const int sub = threadIdx.x & (SMALL_LPR - 1);
    const int64_t group = (blockIdx.x * (int64_t)GRAPH_THREADS + threadIdx.x) / SMALL_LPR;
    const int64_t n_groups = (int64_t)gridDim.x * GRAPH_THREADS / SMALL_LPR;
    const double dang_in = p.scal[1];
    double err = 0.0, dang = 0.0;
    for (int64_t base = 0; base < n_rows; base += n_groups * SMALL_ROWS) {
        int64_t j[SMALL_ROWS], b[SMALL_ROWS], e[SMALL_ROWS];
        int32_t c0[SMALL_ROWS], c1[SMALL_ROWS];
        double sum[SMALL_ROWS];
#pragma unroll
        for (int q = 0; q < SMALL_ROWS; ++q) {
            const int64_t r = base + group * SMALL_ROWS + q;
            j[q] = r < n_rows ? rows[r] : -1;
        }
#pragma unroll
        for (int q = 0; q < SMALL_ROWS; ++q) {
            b[q] = j[q] >= 0 ? p.row_ptr[j[q]] : 0;
            e[q] = j[q] >= 0 ? p.row_ptr[j[q] + 1] : 0;
        }
#pragma unroll
        for (int q = 0; q < SMALL_ROWS; ++q) {
            c0[q] = b[q] + sub < e[q] ? p.col[b[q] + sub] : -1;
            c1[q] = b[q] + sub + SMALL_LPR < e[q] ? p.col[b[q] + sub + SMALL_LPR] : -1;
        }
#pragma unroll
        for (int q = 0; q < SMALL_ROWS; ++q) {
            const double v0 = c0[q] >= 0 ? p.y_in[c0[q]] : 0.0;
            const double v1 = c1[q] >= 0 ? p.y_in[c1[q]] : 0.0;
            sum[q] = v0 + v1;
        }
#pragma unroll
        for (int q = 0; q < SMALL_ROWS; ++q) {
            const double total = wave_sum(sum[q], SMALL_LPR);
            if (sub == 0 && j[q] >= 0) finish_row(p, j[q], total, dang_in, err, dang);
        }
    }
    block_partials<GRAPH_THREADS>(err, dang, p, slot_base + blockIdx.x);
}

// rows with very many neighbours: one block per segment of BIG_SEGMENT entries, then one thread per row adds the
// row's segment sums in order
__global__ void __launch_bounds__(GRAPH_THREADS) ppr_big_segments_kernel(PprParams p, const int64_t *__restrict__ seg_begin,
                                                                         const int32_t *__restrict__ seg_len,
                                                                         double *__restrict__ seg_sum) {
    if (p.state[0]) return;
    __shared__ double s_sum[GRAPH_THREADS / 64];
    const int64_t b = seg_begin[blockIdx.x];
    const int len = seg_len[blockIdx.x];
    const int32_t *col = p.col + b;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int t = threadIdx.x;
    for (; t + 3 * GRAPH_THREADS < len; t += 4 * GRAPH_THREADS) {
        s0 += p.y_in[col[t]];
        s1 += p.y_in[col[t + GRAPH_THREADS]];
        s2 += p.y_in[col[t + 2 * GRAPH_THREADS]];
        s3 += p.y_in[col[t + 3 * GRAPH_THREADS]];
    }
    for (; t < len; t += GRAPH_THREADS) s0 += p.y_in[col[t]];
    const double sum = wave_sum((s0 + s1) + (s2 + s3), 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
        for (int w = 0; w < GRAPH_THREADS / 64; ++w) v += s_sum[w];
        seg_sum[blockIdx.x] = v;
    }
}

__global__ void __launch_bounds__(GRAPH_THREADS) ppr_big_finish_kernel(PprParams p, const int32_t *__restrict__ rows, int64_t n_rows,
                                                                       const int32_t *__restrict__ row_seg,
                                                                       const double *__restrict__ seg_sum, int slot_base) {
    if (p.state[0]) return;
    const double dang_in = p.scal[1];
    double err = 0.0, dang = 0.0;
    for (int64_t r = blockIdx.x * (int64_t)GRAPH_THREADS + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GRAPH_THREADS) {
        double sum = 0.0;
        for (int sgm = row_seg[r]; sgm < row_seg[r + 1]; ++sgm) sum += seg_sum[sgm];
        finish_row(p, rows[r], sum, dang_in, err, dang);
    }
    block_partials<GRAPH_THREADS>(err, dang, p, slot_base + blockIdx.x);
}

__global__ void row_degrees_kernel(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ rows, int64_t n_rows,
                                   int64_t *__restrict__ begin, int64_t *__restrict__ degree) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        begin[r] = row_ptr[rows[r]];
        degree[r] = row_ptr[rows[r] + 1] - row_ptr[rows[r]];
    }
}

// fold the per-block partials in index order; decide convergence (err < n * tol)
__global__ void __launch_bounds__(GRAPH_THREADS) ppr_fold_kernel(const double *__restrict__ partials, int total_blocks, double *scal,
                                                                 int *state, double bound) {
    if (state[0]) return;
    __shared__ double s_e[GRAPH_THREADS], s_d[GRAPH_THREADS];
    double e = 0.0, d = 0.0;
    for (int i = threadIdx.x; i < total_blocks; i += GRAPH_THREADS) {
        e += partials[i];
        d += partials[total_blocks + i];
    }
    s_e[threadIdx.x] = e;
    s_d[threadIdx.x] = d;
    __syncthreads();
    for (int off = GRAPH_THREADS / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            s_e[threadIdx.x] += s_e[threadIdx.x + off];
            s_d[threadIdx.x] += s_d[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scal[0] = s_e[0];
        scal[1] = s_d[0];
        state[1] += 1;
        if (s_e[0] < bound) state[0] = 1;
    }
}

__global__ void ppr_init_kernel(int64_t n, const double *__restrict__ inv_deg, double *__restrict__ x, double *__restrict__ y,
                                double *__restrict__ p) {
    const double x0 = 1.0 / (double)n;
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        x[j] = x0;
        y[j] = x0 * inv_deg[j];
        p[j] = 0.0;
    }
}

__global__ void ppr_seed_kernel(const int32_t *__restrict__ nodes, const double *__restrict__ weights, int64_t n_seeds, double *__restrict__ p,
                                double *scal, int *state, double dang0) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_seeds) p[nodes[i]] = weights[i];
    if (i == 0) {
        scal[0] = 0.0;
        scal[1] = dang0;
        state[0] = 0;
        state[1] = 0;
    }
}

// ---- top chunk nodes -----------------------------------------------------------------------------
// key of a chunk node: the bits of its (non-negative) score + 1; 0 for every other node
__device__ __forceinline__ uint64_t top_key(const double *x, const uint32_t *mask, int64_t j) {
    if (!((mask[j >> 5] >> (j & 31)) & 1u)) return 0;
    const double v = x[j];
    return v >= 0.0 ? (uint64_t)__double_as_longlong(v) + 1 : 1;
}

__global__ void __launch_bounds__(GRAPH_THREADS) top_hist_kernel(const double *__restrict__ x, const uint32_t *__restrict__ mask, int64_t n,
                                                                 uint32_t *__restrict__ hist) {
    __shared__ uint32_t s_hist[TOP_BINS];
    for (int i = threadIdx.x; i < TOP_BINS; i += GRAPH_THREADS) s_hist[i] = 0;
    __syncthreads();
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = top_key(x, mask, j);
        if (key) atomicAdd(&s_hist[key >> 52], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TOP_BINS; i += GRAPH_THREADS)
        if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

__global__ void top_collect_kernel(const double *__restrict__ x, const uint32_t *__restrict__ mask, int64_t n, uint32_t bin,
                                   uint32_t *counter, uint32_t *__restrict__ out_nodes, uint64_t *__restrict__ out_keys,
                                   uint32_t capacity) {
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = top_key(x, mask, j);
        if (key && (key >> 52) >= bin) {
            const uint32_t at = atomicAdd(counter, 1u);
            if (at < capacity) {
                out_nodes[at] = (uint32_t)j;
                out_keys[at] = key;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------
static void graph_free(hvs_graph *g) {
    void *ptrs[] = {g->row_ptr, g->col, g->inv_deg, g->chunk_mask, g->rows[0], g->rows[1], g->rows[2], g->x[0], g->x[1],
                    g->y[0], g->y[1], g->p, g->partials, g->scal, g->state, g->hist, g->seg_begin, g->seg_len, g->row_seg, g->seg_sum,
                    g->ws.ptr};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (g->own_stream && g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

// device buffer that lives for one host function
struct Scoped {
    void *p = nullptr;
    ~Scoped() {
        if (p) (void)hipFree(p);
    }
    template <class T>
    T *as() const {
        return (T *)p;
    }
};

static int graph_build(hvs_graph *g, const int32_t *src, const int32_t *dst, int64_t n_edges, const uint32_t *chunk_mask, int on_device) {
    const int64_t n = g->n;
    hipStream_t st = g->stream;
    const size_t mask_bytes = (size_t)((n + 31) / 32) * 4;
    HVS_HIP(hipMalloc((void **)&g->chunk_mask, mask_bytes + 4));
    HVS_HIP(hipMemcpyAsync(g->chunk_mask, chunk_mask, mask_bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    HVS_HIP(hipMalloc((void **)&g->row_ptr, (size_t)(n + 1) * 8));
    HVS_HIP(hipMalloc((void **)&g->inv_deg, (size_t)n * 8 + 8));
    Scoped counters;  // [0] size_t count of a rocPRIM call, [1] dangling nodes, [2] bad-edge flag
    HVS_HIP(hipMalloc(&counters.p, 32));
    HVS_HIP(hipMemsetAsync(counters.p, 0, 32, st));
    size_t *d_count = counters.as<size_t>();
    unsigned long long *d_dangling = counters.as<unsigned long long>() + 1;
    int *d_bad = (int *)(counters.as<unsigned long long>() + 2);

    if (n_edges > 0) {
        const size_t n_keys = (size_t)n_edges * 2;
        Scoped keys, sorted, d_src, d_dst, temp;
        HVS_HIP(hipMalloc(&keys.p, n_keys * 8));
        HVS_HIP(hipMalloc(&sorted.p, n_keys * 8));
        const int32_t *s = src, *d = dst;
        if (!on_device) {
            HVS_HIP(hipMalloc(&d_src.p, (size_t)n_edges * 4));
            HVS_HIP(hipMalloc(&d_dst.p, (size_t)n_edges * 4));
            HVS_HIP(hipMemcpyAsync(d_src.p, src, (size_t)n_edges * 4, hipMemcpyHostToDevice, st));
            HVS_HIP(hipMemcpyAsync(d_dst.p, dst, (size_t)n_edges * 4, hipMemcpyHostToDevice, st));
            s = d_src.as<int32_t>();
            d = d_dst.as<int32_t>();
        }
        edge_keys_kernel<<<blocks_for(n_edges, 256), 256, 0, st>>>(s, d, n_edges, n, keys.as<uint64_t>(), d_bad);
        HVS_HIP(hipGetLastError());
        int bits = 33;  // row in the high half: 32 column bits + the bits of the largest row
        while (bits < 64 && ((uint64_t)1 << (bits - 32)) < (uint64_t)n) ++bits;
        size_t sort_bytes = 0, uniq_bytes = 0;
        HVS_HIP(rocprim::radix_sort_keys(nullptr, sort_bytes, keys.as<uint64_t>(), sorted.as<uint64_t>(), n_keys, 0, bits, st));
        HVS_HIP(rocprim::unique(nullptr, uniq_bytes, sorted.as<uint64_t>(), keys.as<uint64_t>(), d_count, n_keys, KeyEqual(), st));
        HVS_HIP(hipMalloc(&temp.p, (sort_bytes > uniq_bytes ? sort_bytes : uniq_bytes) + 256));
        HVS_HIP(rocprim::radix_sort_keys(temp.p, sort_bytes, keys.as<uint64_t>(), sorted.as<uint64_t>(), n_keys, 0, bits, st));
        HVS_HIP(rocprim::unique(temp.p, uniq_bytes, sorted.as<uint64_t>(), keys.as<uint64_t>(), d_count, n_keys, KeyEqual(), st));
        int bad = 0;
        size_t count = 0;
        HVS_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
        HVS_HIP(hipMemcpyAsync(&count, d_count, sizeof(size_t), hipMemcpyDeviceToHost, st));
        HVS_HIP(hipStreamSynchronize(st));
        HVS_REQUIRE(!bad, "hvs_graph_create: an edge names a node outside [0, %lld)", (long long)n);
        g->nnz = (int64_t)count;
        HVS_HIP(hipMalloc((void **)&g->col, (size_t)g->nnz * 4 + 4));
        csr_cols_kernel<<<blocks_for(g->nnz, 256), 256, 0, st>>>(keys.as<uint64_t>(), g->nnz, g->col);
        csr_rows_kernel<<<blocks_for(n + 1, 256), 256, 0, st>>>(keys.as<uint64_t>(), g->nnz, n, g->row_ptr);
        HVS_HIP(hipGetLastError());
        HVS_HIP(hipStreamSynchronize(st));  // the key buffers are released at the end of this scope
    } else {
        HVS_HIP(hipMalloc((void **)&g->col, 4));
        HVS_HIP(hipMemsetAsync(g->row_ptr, 0, (size_t)(n + 1) * 8, st));
    }
    degrees_kernel<<<blocks_for(n, 256), 256, 0, st>>>(g->row_ptr, n, g->inv_deg, d_dangling);
    HVS_HIP(hipGetLastError());

    // row lists by degree, ascending row order in each
    Scoped sel_temp, list;
    size_t sel_bytes = 0;
    rocprim::counting_iterator<int32_t> all_rows(0);
    HVS_HIP(rocprim::select(nullptr, sel_bytes, all_rows, (int32_t *)nullptr, d_count, (size_t)n, DegreeBin{g->row_ptr, 0}, st));
    HVS_HIP(hipMalloc(&sel_temp.p, sel_bytes + 256));
    HVS_HIP(hipMalloc(&list.p, (size_t)n * 4 + 4));
    int total_blocks = 0;
    for (int b = 0; b < 3; ++b) {
        size_t count = 0;
        HVS_HIP(rocprim::select(sel_temp.p, sel_bytes, all_rows, list.as<int32_t>(), d_count, (size_t)n, DegreeBin{g->row_ptr, b}, st));
        HVS_HIP(hipMemcpyAsync(&count, d_count, sizeof(size_t), hipMemcpyDeviceToHost, st));
        HVS_HIP(hipStreamSynchronize(st));
        HVS_HIP(hipMalloc((void **)&g->rows[b], count * 4 + 4));
        HVS_HIP(hipMemcpyAsync(g->rows[b], list.p, count * 4, hipMemcpyDeviceToDevice, st));
        g->n_rows[b] = (int64_t)count;
        const int rows_per_block = b == 0 ? GRAPH_THREADS / SMALL_LPR : (b == 1 ? GRAPH_THREADS / 64 : GRAPH_THREADS);
        g->blocks[b] = count ? blocks_for((int64_t)count, rows_per_block) : 0;
        total_blocks += g->blocks[b];
    }
    if (g->n_rows[2] > 0) {  // segments of the big rows, laid out on the host (there are few such rows)
        const int64_t nb = g->n_rows[2];
        Scoped d_tmp;
        HVS_HIP(hipMalloc(&d_tmp.p, (size_t)nb * 16));
        row_degrees_kernel<<<blocks_for(nb, 256), 256, 0, st>>>(g->row_ptr, g->rows[2], nb, d_tmp.as<int64_t>(), d_tmp.as<int64_t>() + nb);
        HVS_HIP(hipGetLastError());
        std::vector<int64_t> begin_deg((size_t)nb * 2);
        HVS_HIP(hipMemcpyAsync(begin_deg.data(), d_tmp.p, (size_t)nb * 16, hipMemcpyDeviceToHost, st));
        HVS_HIP(hipStreamSynchronize(st));
        std::vector<int64_t> seg_begin;
        std::vector<int32_t> seg_len, row_seg((size_t)nb + 1);
        for (int64_t r = 0; r < nb; ++r) {
            row_seg[(size_t)r] = (int32_t)seg_begin.size();
            for (int64_t off = 0; off < begin_deg[(size_t)(nb + r)]; off += BIG_SEGMENT) {
                const int64_t left = begin_deg[(size_t)(nb + r)] - off;
                seg_begin.push_back(begin_deg[(size_t)r] + off);
                seg_len.push_back((int32_t)(left < BIG_SEGMENT ? left : BIG_SEGMENT));
            }
        }
        row_seg[(size_t)nb] = (int32_t)seg_begin.size();
        g->n_segs = (int64_t)seg_begin.size();
        HVS_HIP(hipMalloc((void **)&g->seg_begin, (size_t)g->n_segs * 8));
        HVS_HIP(hipMalloc((void **)&g->seg_len, (size_t)g->n_segs * 4));
        HVS_HIP(hipMalloc((void **)&g->seg_sum, (size_t)g->n_segs * 8));
        HVS_HIP(hipMalloc((void **)&g->row_seg, (size_t)(nb + 1) * 4));
        HVS_HIP(hipMemcpyAsync(g->seg_begin, seg_begin.data(), (size_t)g->n_segs * 8, hipMemcpyHostToDevice, st));
        HVS_HIP(hipMemcpyAsync(g->seg_len, seg_len.data(), (size_t)g->n_segs * 4, hipMemcpyHostToDevice, st));
        HVS_HIP(hipMemcpyAsync(g->row_seg, row_seg.data(), (size_t)(nb + 1) * 4, hipMemcpyHostToDevice, st));
        HVS_HIP(hipStreamSynchronize(st));  // the host vectors go out of scope
    }
    unsigned long long dangling = 0;
    HVS_HIP(hipMemcpyAsync(&dangling, d_dangling, 8, hipMemcpyDeviceToHost, st));
    HVS_HIP(hipStreamSynchronize(st));
    g->n_dangling = (int64_t)dangling;

    for (int i = 0; i < 2; ++i) {
        HVS_HIP(hipMalloc((void **)&g->x[i], (size_t)n * 8 + 8));
        HVS_HIP(hipMalloc((void **)&g->y[i], (size_t)n * 8 + 8));
    }
    HVS_HIP(hipMalloc((void **)&g->p, (size_t)n * 8 + 8));
    HVS_HIP(hipMalloc((void **)&g->partials, (size_t)(2 * total_blocks + 2) * 8));
    HVS_HIP(hipMalloc((void **)&g->scal, 16));
    HVS_HIP(hipMalloc((void **)&g->state, 8));
    HVS_HIP(hipMalloc((void **)&g->hist, (TOP_BINS + 1) * 4));
    return 0;
}

extern "C" int hvs_graph_create(int device_id, void *stream, int64_t n_nodes, const int32_t *src, const int32_t *dst, int64_t n_edges,
                                const uint32_t *chunk_mask, int on_device, hvs_graph **out) {
    HVS_REQUIRE(out != nullptr, "hvs_graph_create: out is NULL");
    *out = nullptr;
    HVS_REQUIRE(n_nodes >= 1 && n_nodes <= 0x7FFFFFFFll, "hvs_graph_create: n_nodes %lld not in [1, 2^31)", (long long)n_nodes);
    HVS_REQUIRE(n_edges >= 0 && (n_edges == 0 || (src && dst)), "hvs_graph_create: bad edge arrays");
    HVS_REQUIRE(chunk_mask != nullptr, "hvs_graph_create: chunk_mask is NULL");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        hvs_set_error("hvs_graph_create: no HIP device visible (this library has no CPU path)");
        return 1;
    }
    HVS_REQUIRE(device_id >= 0 && device_id < n_dev, "hvs_graph_create: device %d of %d", device_id, n_dev);
    HVS_HIP(hipSetDevice(device_id));
    hvs_graph *g = new hvs_graph();
    g->device = device_id;
    g->n = n_nodes;
    if (stream) {
        g->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) {
            hvs_set_error("hvs_graph_create: hipStreamCreate failed");
            delete g;
            return 1;
        }
        g->own_stream = true;
    }
    const int rc = graph_build(g, src, dst, n_edges, chunk_mask, on_device);
    if (rc) {
        (void)hipStreamSynchronize(g->stream);
        graph_free(g);
        return rc;
    }
    *out = g;
    return 0;
}

extern "C" int hvs_graph_destroy(hvs_graph *g) {
    if (!g) return 0;
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->stream);
    graph_free(g);
    return 0;
}

extern "C" int hvs_graph_info(hvs_graph *g, int64_t *n_nodes, int64_t *nnz, int64_t *n_dangling) {
    HVS_REQUIRE(g != nullptr, "hvs_graph_info: graph is NULL");
    if (n_nodes) *n_nodes = g->n;
    if (nnz) *nnz = g->nnz;
    if (n_dangling) *n_dangling = g->n_dangling;
    return 0;
}

extern "C" int hvs_graph_export(hvs_graph *g, int64_t *row_ptr, int32_t *col) {
    HVS_REQUIRE(g != nullptr, "hvs_graph_export: graph is NULL");
    HVS_HIP(hipSetDevice(g->device));
    if (row_ptr) HVS_HIP(hipMemcpyAsync(row_ptr, g->row_ptr, (size_t)(g->n + 1) * 8, hipMemcpyDeviceToHost, g->stream));
    if (col && g->nnz) HVS_HIP(hipMemcpyAsync(col, g->col, (size_t)g->nnz * 4, hipMemcpyDeviceToHost, g->stream));
    HVS_HIP(hipStreamSynchronize(g->stream));
    return 0;
}

extern "C" int hvs_graph_pagerank(hvs_graph *g, const int32_t *seed_nodes, const double *seed_weights, int64_t n_seeds, double alpha,
                                  int max_iter, double tol, int *out_iters, double *out_err) {
    HVS_REQUIRE(g != nullptr, "hvs_graph_pagerank: graph is NULL");
    HVS_REQUIRE(seed_nodes && seed_weights && n_seeds >= 1, "hvs_graph_pagerank: no seeds");
    HVS_REQUIRE(max_iter >= 1, "hvs_graph_pagerank: max_iter %d", max_iter);
    HVS_REQUIRE(alpha >= 0.0 && alpha <= 1.0 && tol > 0.0, "hvs_graph_pagerank: alpha %g / tol %g out of range", alpha, tol);
    double total = 0.0;
    for (int64_t i = 0; i < n_seeds; ++i) {
        HVS_REQUIRE(seed_nodes[i] >= 0 && seed_nodes[i] < g->n, "hvs_graph_pagerank: seed node %d outside [0, %lld)", seed_nodes[i],
                    (long long)g->n);
        HVS_REQUIRE(std::isfinite(seed_weights[i]) && seed_weights[i] > 0.0, "hvs_graph_pagerank: seed weight %g is not a positive number",
                    seed_weights[i]);
        total += seed_weights[i];
    }
    // a node named twice keeps the sum of its weights (the seed kernel stores, it does not add)
    std::map<int32_t, double> merged;
    for (int64_t i = 0; i < n_seeds; ++i) merged[seed_nodes[i]] += seed_weights[i] / total;
    std::vector<int32_t> nodes;
    std::vector<double> w;
    for (const auto &kv : merged) {
        nodes.push_back(kv.first);
        w.push_back(kv.second);
    }
    const int64_t m = (int64_t)nodes.size();

    HVS_HIP(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    void *stage = nullptr;
    const size_t o_w = ((size_t)m * 4 + 255) & ~(size_t)255;
    HVS_TRY(graph_ws(g, o_w + (size_t)m * 8 + 256, &stage));
    HVS_HIP(hipMemcpyAsync(stage, nodes.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    HVS_HIP(hipMemcpyAsync((char *)stage + o_w, w.data(), (size_t)m * 8, hipMemcpyHostToDevice, st));
    ppr_init_kernel<<<blocks_for(g->n, 256), 256, 0, st>>>(g->n, g->inv_deg, g->x[0], g->y[0], g->p);
    const double dang0 = (double)g->n_dangling * (1.0 / (double)g->n);
    ppr_seed_kernel<<<(int)((m + 255) / 256), 256, 0, st>>>((const int32_t *)stage, (const double *)((char *)stage + o_w), m, g->p, g->scal,
                                                          g->state, dang0);
    HVS_HIP(hipGetLastError());

    const int total_blocks = g->blocks[0] + g->blocks[1] + g->blocks[2];
    PprParams p;
    p.row_ptr = g->row_ptr;
    p.col = g->col;
    p.inv_deg = g->inv_deg;
    p.p = g->p;
    p.scal = g->scal;
    p.state = g->state;
    p.partials = g->partials;
    p.total_blocks = total_blocks;
    p.alpha = alpha;
    int state[2] = {0, 0};
    int it = 0;
    while (it < max_iter && !state[0]) {
        // iterations are queued in batches; once the flag is set the remaining kernels of a batch return at once
        int batch = it < 2 ? 2 : (it < 8 ? it : 8);  // 2, 2, 4, 8, 8, ...: big graphs meet the N-scaled bound after one or two
        if (batch > max_iter - it) batch = max_iter - it;
        for (int b = 0; b < batch; ++b, ++it) {
            p.x_in = g->x[it & 1];
            p.y_in = g->y[it & 1];
            p.x_out = g->x[(it + 1) & 1];
            p.y_out = g->y[(it + 1) & 1];
            if (g->blocks[0]) ppr_small_rows_kernel<<<g->blocks[0], GRAPH_THREADS, 0, st>>>(p, g->rows[0], g->n_rows[0], 0);
            if (g->blocks[1]) ppr_rows_kernel<64><<<g->blocks[1], GRAPH_THREADS, 0, st>>>(p, g->rows[1], g->n_rows[1], g->blocks[0]);
            if (g->blocks[2]) {
                ppr_big_segments_kernel<<<(int)g->n_segs, GRAPH_THREADS, 0, st>>>(p, g->seg_begin, g->seg_len, g->seg_sum);
                ppr_big_finish_kernel<<<g->blocks[2], GRAPH_THREADS, 0, st>>>(p, g->rows[2], g->n_rows[2], g->row_seg, g->seg_sum,
                                                                             g->blocks[0] + g->blocks[1]);
            }
            ppr_fold_kernel<<<1, GRAPH_THREADS, 0, st>>>(g->partials, total_blocks, g->scal, g->state, (double)g->n * tol);
        }
        HVS_HIP(hipGetLastError());
        HVS_HIP(hipMemcpyAsync(state, g->state, 8, hipMemcpyDeviceToHost, st));
        HVS_HIP(hipStreamSynchronize(st));
    }
    double err = 0.0;
    HVS_HIP(hipMemcpyAsync(&err, g->scal, 8, hipMemcpyDeviceToHost, st));
    HVS_HIP(hipStreamSynchronize(st));
    g->result = state[1] & 1;  // iteration i (1-based) writes buffer i & 1
    if (out_iters) *out_iters = state[1];
    if (out_err) *out_err = err;
    if (!state[0]) {
        hvs_set_error("hvs_graph_pagerank: power iteration failed to converge within %d iterations (err %g >= %g)", max_iter, err,
                      (double)g->n * tol);
        return 3;
    }
    return 0;
}

extern "C" int hvs_graph_scores(hvs_graph *g, int64_t first, int64_t n, double *out, int on_device) {
    HVS_REQUIRE(g != nullptr && out != nullptr, "hvs_graph_scores: NULL argument");
    HVS_REQUIRE(g->result >= 0, "hvs_graph_scores: no PageRank has been computed on this graph");
    HVS_REQUIRE(first >= 0 && n >= 0 && first + n <= g->n, "hvs_graph_scores: range [%lld, +%lld) outside %lld nodes", (long long)first,
                (long long)n, (long long)g->n);
    HVS_HIP(hipSetDevice(g->device));
    if (n) HVS_HIP(hipMemcpyAsync(out, g->x[g->result] + first, (size_t)n * 8, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, g->stream));
    HVS_HIP(hipStreamSynchronize(g->stream));
    return 0;
}

extern "C" int hvs_graph_top_chunks(hvs_graph *g, int k, int32_t *out_nodes, double *out_scores, int *out_count) {
    HVS_REQUIRE(g != nullptr && out_nodes && out_scores && out_count, "hvs_graph_top_chunks: NULL argument");
    HVS_REQUIRE(g->result >= 0, "hvs_graph_top_chunks: no PageRank has been computed on this graph");
    HVS_REQUIRE(k >= 1, "hvs_graph_top_chunks: k %d", k);
    HVS_HIP(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    const double *x = g->x[g->result];
    // 1. histogram of sign+exponent over the chunk nodes -> the lowest bin that still has to be looked at
    HVS_HIP(hipMemsetAsync(g->hist, 0, (TOP_BINS + 1) * 4, st));
    top_hist_kernel<<<blocks_for(g->n, GRAPH_THREADS * 8), GRAPH_THREADS, 0, st>>>(x, g->chunk_mask, g->n, g->hist);
    HVS_HIP(hipGetLastError());
    std::vector<uint32_t> hist(TOP_BINS);
    HVS_HIP(hipMemcpyAsync(hist.data(), g->hist, TOP_BINS * 4, hipMemcpyDeviceToHost, st));
    HVS_HIP(hipStreamSynchronize(st));
    // the lowest bin b such that bins b.. hold at least k chunk nodes (bin 0 if there are fewer)
    uint64_t cand = 0;
    uint32_t bin = TOP_BINS - 1;
    for (;; --bin) {
        cand += hist[bin];
        if (cand >= (uint64_t)k || bin == 0) break;
    }
    if (cand == 0) {
        *out_count = 0;
        return 0;
    }
    // 2. collect those nodes (any order), 3. order by node, then stably by descending score: ties keep the lower node first
    size_t sort1 = 0, sort2 = 0;
    HVS_HIP(rocprim::radix_sort_pairs(nullptr, sort1, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                      (size_t)cand, 0, 32, st));
    HVS_HIP(rocprim::radix_sort_pairs_desc(nullptr, sort2, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                           (uint32_t *)nullptr, (size_t)cand, 0, 64, st));
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_n0 = 0, o_n1 = o_n0 + up(cand * 4), o_k0 = o_n1 + up(cand * 4), o_k1 = o_k0 + up(cand * 8),
                 o_t = o_k1 + up(cand * 8), bytes = o_t + up(sort1 > sort2 ? sort1 : sort2);
    void *w = nullptr;
    HVS_TRY(graph_ws(g, bytes, &w));
    char *base = (char *)w;
    uint32_t *n0 = (uint32_t *)(base + o_n0), *n1 = (uint32_t *)(base + o_n1);
    uint64_t *k0 = (uint64_t *)(base + o_k0), *k1 = (uint64_t *)(base + o_k1);
    top_collect_kernel<<<blocks_for(g->n, GRAPH_THREADS * 8), GRAPH_THREADS, 0, st>>>(x, g->chunk_mask, g->n, bin, g->hist + TOP_BINS, n0, k0,
                                                                                      (uint32_t)cand);
    HVS_HIP(hipGetLastError());
    HVS_HIP(rocprim::radix_sort_pairs(base + o_t, sort1, n0, n1, k0, k1, (size_t)cand, 0, 32, st));
    HVS_HIP(rocprim::radix_sort_pairs_desc(base + o_t, sort2, k1, k0, n1, n0, (size_t)cand, 0, 64, st));
    const int m = (uint64_t)k < cand ? k : (int)cand;
    std::vector<uint64_t> keys((size_t)m);
    HVS_HIP(hipMemcpyAsync(out_nodes, n0, (size_t)m * 4, hipMemcpyDeviceToHost, st));
    HVS_HIP(hipMemcpyAsync(keys.data(), k0, (size_t)m * 8, hipMemcpyDeviceToHost, st));
    HVS_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < m; ++i) {
        const uint64_t bits = keys[i] - 1;
        memcpy(&out_scores[i], &bits, 8);
    }
    *out_count = m;
    return 0;
}
