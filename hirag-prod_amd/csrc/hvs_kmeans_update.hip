// M-step partial sums of the k-means step without floating-point atomics: rows are grouped by their
// assignment with the library's stable counting sort (hvs_sort.hip), then every centroid's rows are
// added up in ascending row order by blocks that own a slice of them.  The sums are bit-reproducible
// (fixed order of additions) and the pass is bandwidth-shaped: the rows are read once, whole rows at a
// time, where the atomic version was bound by the chip's atomic rate (~0.8 TB/s of added bytes).
//
// The k-means step has no counterpart in the reference (its only clustering is scikit-learn's ward
// linkage over <= 10 vectors, cluster/hierachical_cluster.py:51-103); it is defined by the float64 oracle.
#include <cstring>

#include "hvs_internal.h"

#define UPD_THREADS 256
#define UPD_SLICE_ROWS 1024  // rows added up by one block

// hist[key] counts the rows per key.  A block counts its rows in LDS first (UPD_LDS_BUCKETS counters, as many as a
// block's LDS holds with room to spare) and adds its non-zero counters to the global ones at the end: one global atomic
// per row, all aimed at a few thousand addresses, took 8.5 ms for 10M rows; this takes a fraction of a millisecond.
#define UPD_LDS_BUCKETS 16384
__global__ __launch_bounds__(1024) void upd_keys_kernel(const int32_t *__restrict__ assign, int64_t n, uint32_t *__restrict__ keys,
                                                        uint32_t *__restrict__ hist, int buckets) {
    __shared__ uint32_t local[UPD_LDS_BUCKETS];
    const bool in_lds = buckets <= UPD_LDS_BUCKETS;
    if (in_lds) {
        for (int b = threadIdx.x; b < buckets; b += blockDim.x) local[b] = 0;
        __syncthreads();
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t key = (uint32_t)(assign[i] + 1);  // 0 = row without an assignment (deleted)
        keys[i] = key;
        if (in_lds) atomicAdd(&local[key], 1u);  // integer: order-independent
        else atomicAdd(&hist[key], 1u);
    }
    if (in_lds) {
        __syncthreads();
        for (int b = threadIdx.x; b < buckets; b += blockDim.x)
            if (local[b]) atomicAdd(&hist[b], local[b]);
    }
}

// One block.  start[b] = number of pairs with key < b (b = 0 .. buckets); counts[k] = hist[k + 1];
// slice_start[k] = number of row slices (UPD_SLICE_ROWS rows, the last one of a centroid shorter) of the
// centroids before k (k = 0 .. K): centroid k is summed by blocks slice_start[k] .. slice_start[k + 1] - 1.
__global__ __launch_bounds__(1024) void upd_scan_kernel(const uint32_t *__restrict__ hist, int buckets, uint32_t *__restrict__ start,
                                                        uint32_t *__restrict__ slice_start, long long *__restrict__ counts) {
    __shared__ uint32_t part[1024], spart[1024];
    const int t = threadIdx.x;
    const int per = (buckets + 1023) / 1024;
    const int lo = t * per, hi = min(buckets, lo + per);
    uint32_t sum = 0, ssum = 0;
    for (int i = lo; i < hi; i++) {
        sum += hist[i];
        if (i >= 1) ssum += (hist[i] + UPD_SLICE_ROWS - 1) / UPD_SLICE_ROWS;
    }
    part[t] = sum;
    spart[t] = ssum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = t >= o ? part[t - o] : 0u, sv = t >= o ? spart[t - o] : 0u;
        __syncthreads();
        part[t] += v;
        spart[t] += sv;
        __syncthreads();
    }
    uint32_t run = part[t] - sum, srun = spart[t] - ssum;
    for (int i = lo; i < hi; i++) {
        start[i] = run;
        run += hist[i];
        if (i >= 1) {
            counts[i - 1] = (long long)hist[i];
            slice_start[i - 1] = srun;
            srun += (hist[i] + UPD_SLICE_ROWS - 1) / UPD_SLICE_ROWS;
        }
    }
    if (t == 1023) {
        start[buckets] = part[1023];
        slice_start[buckets - 1] = spart[1023];
    }
}

// 16 bytes of a row, read once by the whole pass: non-temporal, so that the 61 GB of rows stream past L2 instead of through it
typedef float upd_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_row4(const float4 *p) {
    const upd_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const upd_f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// Block b adds the rows of one slice, in order, into partial[b][ld]; thread t owns the 16-byte column
// groups t, t + 256, ... .  Blocks beyond the last slice do nothing.
template <int COLS4>
__global__ __launch_bounds__(UPD_THREADS) void upd_sum_kernel(const float *__restrict__ X, int ld, const uint32_t *__restrict__ order,
                                                              const uint32_t *__restrict__ start,
                                                              const uint32_t *__restrict__ slice_start, int K,
                                                              float *__restrict__ partial) {
    const uint32_t b = blockIdx.x;
    if (b >= slice_start[K]) return;
    int lo_k = 0, hi_k = K;  // largest k with slice_start[k] <= b (empty centroids share a value with their successor)
    while (hi_k - lo_k > 1) {
        const int mid = (lo_k + hi_k) >> 1;
        if (slice_start[mid] <= b) lo_k = mid;
        else hi_k = mid;
    }
    const int k = lo_k;
    const uint32_t first = start[k + 1] + (b - slice_start[k]) * UPD_SLICE_ROWS;
    const uint32_t end = start[k + 2];
    const uint32_t last = first + UPD_SLICE_ROWS < end ? first + UPD_SLICE_ROWS : end;
    const int ld4 = ld >> 2;
    const float4 *X4 = reinterpret_cast<const float4 *>(X);
    float4 acc[COLS4];
#pragma unroll
    for (int j = 0; j < COLS4; j++) acc[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t i = first;
    for (; i + 4 <= last; i += 4) {  // four rows in flight, added in order
        const uint32_t r0 = order[i], r1 = order[i + 1], r2 = order[i + 2], r3 = order[i + 3];
        float4 v[4][COLS4];
#pragma unroll
        for (int j = 0; j < COLS4; j++) {
            const int c = threadIdx.x + j * UPD_THREADS;
            const bool in = c < ld4;
            v[0][j] = in ? load_row4(X4 + (size_t)r0 * ld4 + c) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            v[1][j] = in ? load_row4(X4 + (size_t)r1 * ld4 + c) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            v[2][j] = in ? load_row4(X4 + (size_t)r2 * ld4 + c) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            v[3][j] = in ? load_row4(X4 + (size_t)r3 * ld4 + c) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int j = 0; j < COLS4; j++) {
                acc[j].x += v[q][j].x;
                acc[j].y += v[q][j].y;
                acc[j].z += v[q][j].z;
                acc[j].w += v[q][j].w;
            }
    }
    for (; i < last; i++) {
        const uint32_t r0 = order[i];
#pragma unroll
        for (int j = 0; j < COLS4; j++) {
            const int c = threadIdx.x + j * UPD_THREADS;
            if (c < ld4) {
                const float4 x = load_row4(X4 + (size_t)r0 * ld4 + c);
                acc[j].x += x.x;
                acc[j].y += x.y;
                acc[j].z += x.z;
                acc[j].w += x.w;
            }
        }
    }
    float4 *dst = reinterpret_cast<float4 *>(partial + (size_t)b * ld);
#pragma unroll
    for (int j = 0; j < COLS4; j++) {
        const int c = threadIdx.x + j * UPD_THREADS;
        if (c < ld4) dst[c] = acc[j];
    }
}

// sums[k][c] = partial[slice_start[k]][c] + partial[slice_start[k] + 1][c] + ... in slice order (0 for an empty centroid)
__global__ void upd_fold_kernel(const float *__restrict__ partial, const uint32_t *__restrict__ slice_start, int K, int dim, int ld,
                                float *__restrict__ sums) {
    const int64_t total = (int64_t)K * dim;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i / dim;
        const int c = (int)(i - k * dim);
        float s = 0.0f;
        for (uint32_t b = slice_start[k]; b < slice_start[k + 1]; b++) s += partial[(size_t)b * ld + c];
        sums[i] = s;
    }
}

template <int COLS4>
static void launch_sum(int grid, hipStream_t st, const float *X, int ld, const uint32_t *order, const uint32_t *start,
                       const uint32_t *slice_start, int K, float *partial) {
    upd_sum_kernel<COLS4><<<grid, UPD_THREADS, 0, st>>>(X, ld, order, start, slice_start, K, partial);
}

// d_assign: device, [n] (-1 = no assignment).  d_sums [K, dim] and d_counts [K]: device outputs.
int hvs_kmeans_update(hvs_index *idx, const int32_t *d_assign, int64_t n, int K, float *d_sums, int64_t *d_counts) {
    hipStream_t st = idx->stream;
    const int dim = idx->dim, ld = idx->ld;
    HVS_REQUIRE(ld <= 4 * 16 * UPD_THREADS, "hvs_kmeans_assign: sums need dim <= %d", 4 * 16 * UPD_THREADS);
    const int buckets = K + 1;
    int bits = 1;
    while ((1 << bits) < buckets) bits++;
    const int64_t max_slices = n / UPD_SLICE_ROWS + K;  // every centroid has at most one short slice

    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_keys = 0, o_order = o_keys + up((size_t)n * 4), o_hist = o_order + up((size_t)n * 4),
                 o_start = o_hist + up((size_t)buckets * 4), o_slices = o_start + up((size_t)(buckets + 1) * 4),
                 o_temp = o_slices + up((size_t)(K + 1) * 4), o_partial = o_temp + up(hvs_sort_temp_bytes(n)),
                 total = o_partial + (size_t)max_slices * ld * 4;
    void *ws = nullptr;
    HVS_TRY(hvs_ws_get(idx, 6, total, &ws));
    char *w = (char *)ws;
    uint32_t *keys = (uint32_t *)(w + o_keys), *order = (uint32_t *)(w + o_order);
    uint32_t *hist = (uint32_t *)(w + o_hist), *start = (uint32_t *)(w + o_start), *slice_start = (uint32_t *)(w + o_slices);
    float *partial = (float *)(w + o_partial);

    HVS_HIP(hipMemsetAsync(hist, 0, (size_t)buckets * 4, st));
    int g = (int)((n + 1023) / 1024);
    if (g > 512) g = 512;  // two blocks per CU: each adds its counters to the global ones once
    upd_keys_kernel<<<g, 1024, 0, st>>>(d_assign, n, keys, hist, buckets);
    upd_scan_kernel<<<1, 1024, 0, st>>>(hist, buckets, start, slice_start, (long long *)d_counts);
    HVS_HIP(hipGetLastError());
    HVS_TRY(hvs_sort_by_key(st, keys, n, bits, order, w + o_temp));  // rows grouped by key, ascending row number inside a key
    const int cols4 = (ld / 4 + UPD_THREADS - 1) / UPD_THREADS;
    const float *X = (const float *)idx->rows;
    if (cols4 <= 1) launch_sum<1>((int)max_slices, st, X, ld, order, start, slice_start, K, partial);
    else if (cols4 <= 2) launch_sum<2>((int)max_slices, st, X, ld, order, start, slice_start, K, partial);
    else if (cols4 <= 4) launch_sum<4>((int)max_slices, st, X, ld, order, start, slice_start, K, partial);
    else if (cols4 <= 8) launch_sum<8>((int)max_slices, st, X, ld, order, start, slice_start, K, partial);
    else launch_sum<16>((int)max_slices, st, X, ld, order, start, slice_start, K, partial);
    HVS_HIP(hipGetLastError());
    int gf = (int)(((int64_t)K * dim + 255) / 256);
    if (gf > 256 * 16) gf = 256 * 16;
    upd_fold_kernel<<<gf, 256, 0, st>>>(partial, slice_start, K, dim, ld, d_sums);
    HVS_HIP(hipGetLastError());
    return 0;
}
