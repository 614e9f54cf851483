// Index life cycle, ingest, synthetic fill, gather (C ABI: include/hvs.h).
#include <algorithm>
#include <cstdarg>
#include <vector>

#include "hvs_internal.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

void hvs_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char *hvs_last_error(void) { return g_last_error.c_str(); }
extern "C" int hvs_version(void) { return 100; }

extern "C" int hvs_device_count(int *out) {
    HVS_REQUIRE(out != nullptr, "hvs_device_count: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *out = n;
    return 0;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <typename SRC, typename DST> __device__ __forceinline__ DST hvs_cvt(SRC v);
template <> __device__ __forceinline__ float hvs_cvt<float, float>(float v) { return v; }
template <> __device__ __forceinline__ __half hvs_cvt<float, __half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ __half hvs_cvt<__half, __half>(__half v) { return v; }
template <> __device__ __forceinline__ uint16_t hvs_cvt<float, uint16_t>(float v) { return hvs_float_to_bf16(v); }
template <> __device__ __forceinline__ uint16_t hvs_cvt<uint16_t, uint16_t>(uint16_t v) { return v; }

// dst[pos(r)][c] = round(src[r][c]) for c < dim, 0 for dim <= c < ld.  positions == nullptr: pos(r) = first + r.
template <typename SRC, typename DST>
__global__ void store_rows_kernel(const SRC *__restrict__ src, DST *__restrict__ dst, int64_t n, int dim, int ld,
                                  int64_t first, const int64_t *__restrict__ positions) {
    const int64_t total = n * (int64_t)ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        const int64_t p = positions ? positions[r] : first + r;
        DST v;
        if (c < dim) v = hvs_cvt<SRC, DST>(src[r * dim + c]);
        else v = hvs_cvt<float, DST>(0.0f);
        dst[p * ld + c] = v;
    }
}

template <typename T> __device__ __forceinline__ float hvs_widen(T v);
template <> __device__ __forceinline__ float hvs_widen<float>(float v) { return v; }
template <> __device__ __forceinline__ float hvs_widen<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float hvs_widen<uint16_t>(uint16_t v) { return hvs_bf16_to_float(v); }

// one wave per row: inv_norm[p] = 1/sqrt(sum x^2) over the stored (rounded) values; zero row -> +inf
template <typename T>
__global__ void inv_norm_kernel(const T *__restrict__ rows, float *__restrict__ inv_norm, int64_t n, int ld,
                                int64_t first, const int64_t *__restrict__ positions) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n; r += nwaves) {
        const int64_t p = positions ? positions[r] : first + r;
        const T *row = rows + p * ld;
        float s = 0.0f;
        for (int c = lane; c < ld; c += 64) {
            const float v = hvs_widen<T>(row[c]);
            s = fmaf(v, v, s);
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) inv_norm[p] = (float)(1.0 / sqrt((double)s));
    }
}

template <typename T>
__global__ void fill_synth_kernel(T *__restrict__ rows, int64_t n, int dim, int ld, int64_t first_pos,
                                  int64_t first_global, uint64_t seed_mix) {
    const int64_t total = n * (int64_t)ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        float v = 0.0f;
        if (c < dim) {
            const uint64_t ctr = (uint64_t)(first_global + r) * (uint64_t)dim + (uint64_t)c;
            const uint64_t h = hvs_splitmix64(ctr ^ seed_mix);
            const int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) +
                          (int)((h >> 48) & 0xFFFF) - 131070;
            v = (float)s * (1.0f / 37837.0f);
        }
        rows[(first_pos + r) * ld + c] = hvs_cvt<float, T>(v);
    }
}

// Clustered corpus: row r = centre[c(r)] + sigma[c(r)] * noise[r], every term a value of the generator above.
//   c(r)      = splitmix64(r ^ splitmix64(seed ^ 0xC1)) mod n_clusters      (members scattered over the table)
//   level(c)  = splitmix64(c ^ splitmix64(seed ^ 0xC2)) & 7, sigma = sqrt of {1, 1.4, 2, 2.8, 4, 5.6, 8, 10}e-3 as float
//   centre[c] = row c of the plain generator under seed + 1;  noise[r] = row r under seed
// Two members of one cluster lie at cosine distance ~ sigma^2 of each other (1e-3 .. 1e-2).  The product and the sum
// are rounded separately (no fused multiply-add), so that float32 arithmetic on any host gives the same bits.
__device__ __forceinline__ float hvs_synth_value(uint64_t ctr, uint64_t seed_mix) {
    const uint64_t h = hvs_splitmix64(ctr ^ seed_mix);
    const int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)((h >> 48) & 0xFFFF) - 131070;
    return (float)s * (1.0f / 37837.0f);
}

struct hvs_cluster_params {
    uint64_t mix_noise, mix_centre, mix_assign, mix_level;
    uint64_t n_clusters;
    float sigma[8];
};

template <typename T>
__global__ void fill_clustered_kernel(T *__restrict__ rows, int64_t n, int dim, int ld, int64_t first_pos, int64_t first_global,
                                      hvs_cluster_params cp) {
    const int64_t total = n * (int64_t)ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        float v = 0.0f;
        if (c < dim) {
            const uint64_t g = (uint64_t)(first_global + r);
            const uint64_t cl = hvs_splitmix64(g ^ cp.mix_assign) % cp.n_clusters;
            const float sigma = cp.sigma[hvs_splitmix64(cl ^ cp.mix_level) & 7];
            const float centre = hvs_synth_value(cl * (uint64_t)dim + (uint64_t)c, cp.mix_centre);
            const float noise = hvs_synth_value(g * (uint64_t)dim + (uint64_t)c, cp.mix_noise);
            {
#pragma clang fp contract(off)  // two roundings, as plain float32 arithmetic on a host does
                const float prod = sigma * noise;
                v = centre + prod;
            }
        }
        rows[(first_pos + r) * ld + c] = hvs_cvt<float, T>(v);
    }
}

__global__ void fill_meta_kernel(int64_t *ids, int32_t *groups, int64_t n, int64_t first_pos, int64_t first_id,
                                 int32_t n_groups, const int64_t *src_ids, const int32_t *src_groups) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = src_ids ? src_ids[i] : first_id + i;
        ids[first_pos + i] = id;
        int32_t g = 0;
        if (src_groups) g = src_groups[i];
        else if (n_groups > 0) g = (int32_t)(id % n_groups);
        else if (n_groups < 0) g = (int32_t)((__int128)i * (-(int64_t)n_groups) / n);  // -n_groups equal runs, in order
        groups[first_pos + i] = g;
    }
}

template <typename T>
__global__ void gather_kernel(const T *__restrict__ rows, const int64_t *__restrict__ positions, int64_t n, int dim,
                              int ld, float *__restrict__ out) {
    const int64_t total = n * (int64_t)dim;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dim;
        const int c = (int)(i - r * dim);
        out[i] = hvs_widen<T>(rows[positions[r] * ld + c]);
    }
}

// flips counts the bits that actually changed, so that the index's tombstone count stays exact when a
// position is deleted twice or restored without having been deleted
__global__ void set_deleted_kernel(uint32_t *bits, const int64_t *positions, int64_t n, int deleted, unsigned long long *flips) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = positions[i];
        const uint32_t bit = 1u << (p & 31);
        const uint32_t old = deleted ? atomicOr(&bits[p >> 5], bit) : atomicAnd(&bits[p >> 5], ~bit);
        if (((old & bit) != 0) != (deleted != 0)) atomicAdd(flips, 1ull);
    }
}

__global__ void unpack_deleted_kernel(const uint32_t *bits, int64_t first, int64_t n, uint8_t *out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = first + i;
        out[i] = bits ? (uint8_t)((bits[p >> 5] >> (p & 31)) & 1u) : (uint8_t)0;
    }
}

__global__ void gather_meta_kernel(const int64_t *ids, const int32_t *groups, const uint32_t *positions, int64_t n,
                                   int64_t *out_ids, int32_t *out_groups) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        out_ids[i] = ids[positions[i]];
        out_groups[i] = groups[positions[i]];
    }
}

static inline int grid_for(int64_t work, int block = 256, int max_blocks = 256 * 8) {
    int64_t b = (work + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

// ---------------------------------------------------------------------------
// workspace / profiling helpers
// ---------------------------------------------------------------------------
int hvs_ws_get(hvs_index *idx, int slot, size_t bytes, void **out) {
    hvs_workspace &w = idx->ws[slot];
    if (w.bytes < bytes) {
        if (w.ptr) {
            HVS_HIP(hipStreamSynchronize(idx->stream));
            HVS_HIP(hipFree(w.ptr));
            w.ptr = nullptr;
            w.bytes = 0;
        }
        size_t want = bytes + bytes / 4;
        if (want < (1u << 20)) want = 1u << 20;
        HVS_HIP(hipMalloc(&w.ptr, want));
        w.bytes = want;
    }
    *out = w.ptr;
    return 0;
}

int hvs_pinned_get(hvs_index *idx, size_t bytes, void **out) {
    if (idx->h_pin_bytes < bytes) {
        if (idx->h_pin) {
            HVS_HIP(hipStreamSynchronize(idx->stream));
            HVS_HIP(hipHostFree(idx->h_pin));
            idx->h_pin = nullptr;
            idx->h_pin_bytes = 0;
        }
        size_t want = bytes + bytes / 4;
        if (want < (64u << 10)) want = 64u << 10;
        HVS_HIP(hipHostMalloc(&idx->h_pin, want, hipHostMallocDefault));
        idx->h_pin_bytes = want;
    }
    *out = idx->h_pin;
    return 0;
}

int hvs_prof_begin(hvs_index *idx) {
    if (!idx->profiling) return 0;
    HVS_HIP(hipEventRecord(idx->ev0, idx->stream));
    return 0;
}

int hvs_prof_end(hvs_index *idx, int which) {
    if (!idx->profiling) return 0;
    HVS_HIP(hipEventRecord(idx->ev1, idx->stream));
    HVS_HIP(hipEventSynchronize(idx->ev1));
    float ms = 0.0f;
    HVS_HIP(hipEventElapsedTime(&ms, idx->ev0, idx->ev1));
    idx->prof_ms[which] += ms;
    idx->prof_launches[which] += 1;
    return 0;
}

// ---------------------------------------------------------------------------
// live-row lists
// ---------------------------------------------------------------------------
// Ordered list of the rows that pass a filter (group code, bitmask, tombstones, row range), used by
// filtered batched searches (which then contract over a dense copy of just those rows) and by
// tombstone compaction.  Three small passes: count live rows per block of
// LIVE_CHUNK rows, exclusive scan of the block counts, ordered write of the live row positions; then
// the rows and inverse norms are gathered.  Order is preserved, so ties still go to the lower position.
#define LIVE_THREADS 256
#define LIVE_ITEMS 16
#define LIVE_CHUNK (LIVE_THREADS * LIVE_ITEMS)

__device__ __forceinline__ bool row_is_live(const hvs_live_params &p, int64_t row) {
    if (row >= p.row_end) return false;
    bool ok = true;
    if (p.group_filter >= 0) ok = p.groups[row] == p.group_filter;
    if (ok && p.mask) ok = (p.mask[row >> 5] >> (row & 31)) & 1u;
    if (ok && p.deleted) ok = !((p.deleted[row >> 5] >> (row & 31)) & 1u);
    if (ok && p.dist) ok = p.dist[row - p.row_begin] < p.max_dist;  // false for NaN
    return ok;
}

__global__ void range_gather_kernel(const uint32_t *__restrict__ positions, int64_t n, const float *__restrict__ dist,
                                    int64_t row_begin, const int64_t *__restrict__ ids, float *__restrict__ out_dist,
                                    int64_t *__restrict__ out_ids) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t r = positions[i];
        out_dist[i] = dist[r - row_begin];
        out_ids[i] = ids[r];
    }
}

int hvs_range_collect(hvs_index *idx, const hvs_live_params &lp, int64_t capacity, float *d_out_dist, int64_t *d_out_ids,
                      int64_t *total) {
    uint32_t n_hit = 0;
    HVS_TRY(hvs_live_count(idx, lp, &n_hit));
    *total = n_hit;
    if (n_hit == 0 || capacity <= 0) return 0;
    void *pos = nullptr;
    HVS_TRY(hvs_ws_get(idx, 6, (size_t)n_hit * 4 + 256, &pos));
    HVS_TRY(hvs_live_write(idx, lp, (uint32_t *)pos));
    const int64_t m = (int64_t)n_hit < capacity ? (int64_t)n_hit : capacity;
    range_gather_kernel<<<grid_for(m), 256, 0, idx->stream>>>((const uint32_t *)pos, m, lp.dist, lp.row_begin, idx->ids, d_out_dist,
                                                             d_out_ids);
    HVS_HIP(hipGetLastError());
    return 0;
}

// Thread t of block b looks at rows b * LIVE_CHUNK + i * LIVE_THREADS + t (coalesced); a block's rows are
// ordered by (i, t), which is ascending row order.
template <bool WRITE>
__global__ __launch_bounds__(LIVE_THREADS) void live_rows_kernel(hvs_live_params p, uint32_t *__restrict__ block_counts,
                                                                 const uint32_t *__restrict__ block_offsets,
                                                                 uint32_t *__restrict__ positions) {
    __shared__ uint32_t wave_tot[LIVE_THREADS / 64];
    __shared__ uint32_t running;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = p.row_begin + (int64_t)blockIdx.x * LIVE_CHUNK;
    if (threadIdx.x == 0) running = WRITE ? block_offsets[blockIdx.x] : 0;
    __syncthreads();
    uint32_t count = 0;
    for (int i = 0; i < LIVE_ITEMS; i++) {
        const int64_t row = base + (int64_t)i * LIVE_THREADS + threadIdx.x;
        const bool live = row_is_live(p, row);
        const uint64_t b = __ballot(live);
        if (!WRITE) {
            count += (lane == 0) ? (uint32_t)__popcll(b) : 0u;
        } else {
            if (lane == 0) wave_tot[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t off = running;
            for (int w = 0; w < wave; w++) off += wave_tot[w];
            if (live) positions[off + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)row;
            __syncthreads();
            if (threadIdx.x == 0) running += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            __syncthreads();
        }
    }
    if (!WRITE) {
        if (lane == 0) wave_tot[wave] = count;
        __syncthreads();
        if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    }
}

// Exclusive scan of n block counts by one block; offsets[n] = total.
// host_total: a pinned host word that receives the total as well (read after the stream's next synchronisation).
__global__ __launch_bounds__(1024) void live_scan_kernel(const uint32_t *__restrict__ counts, uint32_t *__restrict__ offsets, int n,
                                                         uint32_t *__restrict__ host_total) {
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    uint32_t sum = 0;
    for (int i = lo; i < hi; i++) sum += counts[i];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
        const uint32_t v = t >= o ? part[t - o] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int i = lo; i < hi; i++) {
        offsets[i] = run;
        run += counts[i];
    }
    if (t == 1023) {
        offsets[n] = part[1023];
        *host_total = part[1023];
    }
}

// dst row i = src row positions[i] (16-byte chunks, one wave per row), with its inverse norm.
__global__ __launch_bounds__(256) void gather_live_kernel(const char *__restrict__ src, const float *__restrict__ src_inv,
                                                          const uint32_t *__restrict__ positions, int64_t n_live, int chunks,
                                                          char *__restrict__ dst, float *__restrict__ dst_inv) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_live) return;
    const int64_t r = positions[i];
    const uint4 *s = reinterpret_cast<const uint4 *>(src) + r * chunks;
    uint4 *d = reinterpret_cast<uint4 *>(dst) + i * chunks;
    for (int c = lane; c < chunks; c += 64) d[c] = s[c];
    if (lane == 0) dst_inv[i] = src_inv[r];
}


int hvs_live_count(hvs_index *idx, const hvs_live_params &lp, uint32_t *n_live) {
    *n_live = 0;
    const int64_t n_range = lp.row_end - lp.row_begin;
    if (n_range <= 0) return 0;
    HVS_REQUIRE(lp.row_end <= (int64_t)0xFFFFFFFFll, "row positions beyond 2^32-1");
    const int nb = (int)((n_range + LIVE_CHUNK - 1) / LIVE_CHUNK);
    const size_t o_off = ((size_t)nb * 4 + 255) & ~(size_t)255;
    void *cw = nullptr;
    HVS_TRY(hvs_ws_get(idx, 5, o_off + (size_t)(nb + 1) * 4, &cw));
    uint32_t *counts = (uint32_t *)cw, *offsets = (uint32_t *)((char *)cw + o_off);
    live_rows_kernel<false><<<nb, LIVE_THREADS, 0, idx->stream>>>(lp, counts, nullptr, nullptr);
    volatile uint32_t *host_total = reinterpret_cast<volatile uint32_t *>(idx->h_flag) + 8;  // second half of the pinned line
    live_scan_kernel<<<1, 1024, 0, idx->stream>>>(counts, offsets, nb, const_cast<uint32_t *>(host_total));
    HVS_HIP(hipGetLastError());
    HVS_HIP(hipStreamSynchronize(idx->stream));
    *n_live = *host_total;
    return 0;
}

// Must follow hvs_live_count with the same parameters (it uses the block offsets that call left in
// workspace 5); positions needs room for the count it returned.
int hvs_live_write(hvs_index *idx, const hvs_live_params &lp, uint32_t *positions) {
    const int64_t n_range = lp.row_end - lp.row_begin;
    if (n_range <= 0) return 0;
    const int nb = (int)((n_range + LIVE_CHUNK - 1) / LIVE_CHUNK);
    const size_t o_off = ((size_t)nb * 4 + 255) & ~(size_t)255;
    const uint32_t *offsets = (const uint32_t *)((char *)idx->ws[5].ptr + o_off);
    live_rows_kernel<true><<<nb, LIVE_THREADS, 0, idx->stream>>>(lp, nullptr, offsets, positions);
    HVS_HIP(hipGetLastError());
    return 0;
}

int hvs_live_gather(hvs_index *idx, const void *src_rows, const float *src_inv, int row_bytes, const uint32_t *positions,
                    int64_t n_live, char *dst_rows, float *dst_inv) {
    if (n_live <= 0) return 0;
    gather_live_kernel<<<(unsigned)((n_live + 3) / 4), 256, 0, idx->stream>>>((const char *)src_rows, src_inv, positions, n_live,
                                                                             row_bytes / 16, dst_rows, dst_inv);
    HVS_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// row attributes and the filter's row bitmask
// ---------------------------------------------------------------------------
__global__ void set_attrs_kernel(int4 *__restrict__ attrs, int64_t first, int64_t n, const int32_t *__restrict__ ws,
                                 const int32_t *__restrict__ kb, const int32_t *__restrict__ uri, const uint8_t *__restrict__ priv) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int4 a = attrs[first + i];  // a field no call has set yet reads -1 and matches no clause
        if (ws) a.x = ws[i];
        if (kb) a.y = kb[i];
        if (uri) a.z = uri[i];
        if (priv) a.w = priv[i] ? 1 : 0;
        attrs[first + i] = a;
    }
}

__global__ void gather_attrs_kernel(const int4 *__restrict__ src, const uint32_t *__restrict__ positions, int64_t n, int4 *__restrict__ dst) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[positions[i]];
}

// One wave per 64 consecutive rows and step: lane i tests row base + i (one coalesced 16-byte load per lane), a ballot
// makes the two mask words.  The uri codes sit sorted in LDS; membership is a binary search.
#define FILTER_URIS_LDS 8192
__global__ __launch_bounds__(256) void filter_mask_kernel(const int4 *__restrict__ attrs, int64_t n, int64_t first_step, int64_t last_step, int32_t ws,
                                                           int32_t kb, int32_t access, const int32_t *__restrict__ uris, int n_uris,
                                                           uint32_t *__restrict__ mask) {
    __shared__ int32_t s_uris[FILTER_URIS_LDS];
    for (int i = threadIdx.x; i < n_uris; i += blockDim.x) s_uris[i] = uris[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t step = first_step + wave; step < last_step; step += n_waves) {  // 64 rows per step
        const int64_t row = step * 64 + lane;
        bool ok = row < n;
        if (ok) {
            const int4 a = attrs[row];
            if (ws >= 0) ok = a.x == ws;
            if (ok && kb >= 0) ok = a.y == kb;
            if (ok && access >= 0) ok = a.w >= 0 && (a.w != 0) == (access != 0);
            if (ok && uris) {
                int lo = 0, hi = n_uris;  // first entry >= a.z
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_uris[mid] < a.z) lo = mid + 1;
                    else hi = mid;
                }
                ok = lo < n_uris && s_uris[lo] == a.z;
            }
        }
        const uint64_t bits = __ballot(ok);
        if (lane == 0) {
            mask[step * 2] = (uint32_t)bits;
            if (step * 64 + 32 < ((n + 31) / 32) * 32) mask[step * 2 + 1] = (uint32_t)(bits >> 32);
        }
    }
}

int hvs_attrs_ensure(hvs_index *idx) {
    if (idx->attrs || idx->cap == 0) return 0;
    HVS_HIP(hipMalloc((void **)&idx->attrs, (size_t)idx->cap * sizeof(int4)));
    HVS_HIP(hipMemsetAsync(idx->attrs, 0xFF, (size_t)idx->cap * sizeof(int4), idx->stream));  // -1: no value yet, matches no clause
    return 0;
}

int hvs_filter_mask(hvs_index *idx, const hvs_filter *f, const uint32_t **d_mask, int64_t row_begin, int64_t row_end) {
    *d_mask = nullptr;
    HVS_REQUIRE(f->n_uris >= 0 && f->n_uris <= HVS_MAX_FILTER_URIS, "hvs_filter: %d uri codes, at most %d", f->n_uris, HVS_MAX_FILTER_URIS);
    HVS_REQUIRE(f->n_uris == 0 || f->uris != nullptr, "hvs_filter: n_uris = %d with a NULL list", f->n_uris);
    static_assert(HVS_MAX_FILTER_URIS <= FILTER_URIS_LDS, "uri list must fit the kernel's LDS array");
    const bool has_uris = f->uris != nullptr;
    if (f->workspace < 0 && f->knowledge_base < 0 && f->access < 0 && !has_uris) return 0;
    if (idx->n == 0) return 0;
    HVS_TRY(hvs_attrs_ensure(idx));
    const size_t words = (size_t)((idx->n + 31) / 32);
    void *m = nullptr, *u = nullptr;
    const int32_t *d_uris = nullptr;
    int n_uris = 0;
    HVS_TRY(hvs_ws_get(idx, 8, words * 4 + 8, &m));
    if (has_uris) {
        // WHERE uri IN (): an empty list matches nothing -- one code no row can carry keeps the kernel's path uniform
        std::vector<int32_t> sorted(f->uris, f->uris + f->n_uris);
        std::sort(sorted.begin(), sorted.end());
        if (sorted.empty()) sorted.push_back(INT32_MIN);
        HVS_TRY(hvs_ws_get(idx, 9, sorted.size() * 4, &u));
        HVS_HIP(hipMemcpyAsync(u, sorted.data(), sorted.size() * 4, hipMemcpyHostToDevice, idx->stream));
        HVS_HIP(hipStreamSynchronize(idx->stream));  // `sorted` is pageable host memory about to go out of scope
        d_uris = (const int32_t *)u;
        n_uris = (int)sorted.size();
    }
    // the rows a search can reach: [row_begin, row_end), and on a table in group order with a group clause only that
    // group's range and the rows added since -- the words of the mask that cover other rows are left as they are
    if (row_end < 0 || row_end > idx->n) row_end = idx->n;
    if (row_begin < 0) row_begin = 0;
    int64_t spans[2][2] = {{row_begin, row_end}, {0, 0}};
    int64_t gb = 0, ge = 0;
    if (f->group >= 0 && !getenv("HVS_NO_PARTITION") && hvs_part_range(idx, f->group, &gb, &ge)) {
        spans[0][0] = row_begin > gb ? row_begin : gb;
        spans[0][1] = row_end < ge ? row_end : ge;
        spans[1][0] = row_begin > idx->part_n ? row_begin : idx->part_n;
        spans[1][1] = row_end;
    }
    for (int i = 0; i < 2; i++) {
        if (spans[i][0] >= spans[i][1]) continue;
        const int64_t first_step = spans[i][0] / 64, last_step = (spans[i][1] + 63) / 64;
        filter_mask_kernel<<<grid_for((last_step - first_step) * 64, 256, 2048), 256, 0, idx->stream>>>(
            idx->attrs, idx->n, first_step, last_step, f->workspace, f->knowledge_base, f->access, d_uris, n_uris, (uint32_t *)m);
    }
    HVS_HIP(hipGetLastError());
    *d_mask = (const uint32_t *)m;
    return 0;
}

extern "C" int hvs_index_set_attributes(hvs_index *idx, int64_t first_pos, int64_t n, const int32_t *workspace,
                                        const int32_t *knowledge_base, const int32_t *uri, const uint8_t *is_private) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_set_attributes: idx is NULL");
    HVS_REQUIRE(first_pos >= 0 && n >= 0 && first_pos + n <= idx->n, "hvs_index_set_attributes: rows [%lld, %lld) outside [0, %lld)",
                (long long)first_pos, (long long)(first_pos + n), (long long)idx->n);
    if (n == 0 || (!workspace && !knowledge_base && !uri && !is_private)) return 0;
    HVS_HIP(hipSetDevice(idx->device));
    HVS_TRY(hvs_attrs_ensure(idx));
    // staging (workspace 3): the four host arrays, 256-byte aligned
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t col = up((size_t)n * 4);
    void *stage = nullptr;
    HVS_TRY(hvs_ws_get(idx, 3, 3 * col + up((size_t)n), &stage));
    char *b = (char *)stage;
    const int32_t *d_ws = nullptr, *d_kb = nullptr, *d_uri = nullptr;
    const uint8_t *d_priv = nullptr;
    if (workspace) { HVS_HIP(hipMemcpyAsync(b, workspace, (size_t)n * 4, hipMemcpyHostToDevice, idx->stream)); d_ws = (const int32_t *)b; }
    if (knowledge_base) { HVS_HIP(hipMemcpyAsync(b + col, knowledge_base, (size_t)n * 4, hipMemcpyHostToDevice, idx->stream)); d_kb = (const int32_t *)(b + col); }
    if (uri) { HVS_HIP(hipMemcpyAsync(b + 2 * col, uri, (size_t)n * 4, hipMemcpyHostToDevice, idx->stream)); d_uri = (const int32_t *)(b + 2 * col); }
    if (is_private) { HVS_HIP(hipMemcpyAsync(b + 3 * col, is_private, (size_t)n, hipMemcpyHostToDevice, idx->stream)); d_priv = (const uint8_t *)(b + 3 * col); }
    set_attrs_kernel<<<grid_for(n), 256, 0, idx->stream>>>(idx->attrs, first_pos, n, d_ws, d_kb, d_uri, d_priv);
    HVS_HIP(hipGetLastError());
    HVS_HIP(hipStreamSynchronize(idx->stream));  // the host arrays may be released on return
    return 0;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int hvs_index_create(hvs_index **out, int device_id, int dim, int dtype, void *stream) {
    HVS_REQUIRE(out != nullptr, "hvs_index_create: out is NULL");
    *out = nullptr;
    HVS_REQUIRE(dim > 0 && dim <= 65536, "hvs_index_create: dim %d out of range [1, 65536]", dim);
    HVS_REQUIRE(dtype == HVS_F32 || dtype == HVS_F16 || dtype == HVS_BF16, "hvs_index_create: unknown dtype %d", dtype);
    int ndev = 0;
    HVS_TRY(hvs_device_count(&ndev));
    HVS_REQUIRE(ndev > 0, "hvs_index_create: no HIP device visible (this library has no CPU path)");
    HVS_REQUIRE(device_id >= 0 && device_id < ndev, "hvs_index_create: device %d not in [0, %d)", device_id, ndev);
    HVS_HIP(hipSetDevice(device_id));
    HVS_TRY(hvs_noalloc_setup(device_id));  // once per device, so that hvs_search_noalloc never has to
    hvs_index *idx = new hvs_index();
    idx->device = device_id;
    idx->dim = dim;
    idx->dtype = dtype;
    idx->elem_size = hvs_elem_size(dtype);
    const int per_chunk = 16 / idx->elem_size;
    idx->ld = (dim + per_chunk - 1) / per_chunk * per_chunk;
    if (stream) {
        idx->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            hvs_set_error("hvs_index_create: hipStreamCreate failed: %s", hipGetErrorString(e));
            delete idx;
            return 1;
        }
        idx->own_stream = true;
    }
    hipEventCreate(&idx->ev0);
    hipEventCreate(&idx->ev1);
    hipEventCreate(&idx->tm0);
    hipEventCreate(&idx->tm1);
    if (hipHostMalloc((void **)&idx->h_flag, 64, hipHostMallocDefault) != hipSuccess) {
        hvs_set_error("hvs_index_create: no pinned host memory for 64 bytes");
        hvs_index_destroy(idx);
        return 1;
    }
    idx->h_flag[0] = 0;
    *out = idx;
    return 0;
}

extern "C" int hvs_index_destroy(hvs_index *idx) {
    if (!idx) return 0;
    (void)hipSetDevice(idx->device);
    (void)hipStreamSynchronize(idx->stream);
    hipFree(idx->rows);
    hipFree(idx->inv_norm);
    hipFree(idx->ids);
    hipFree(idx->groups);
    hipFree(idx->deleted);
    hipFree(idx->attrs);
    hipFree(idx->split);
    hipFree(idx->split_inv);
    hipFree(idx->split_rel);
    for (auto &w : idx->ws) hipFree(w.ptr);
    if (idx->h_pin) hipHostFree(idx->h_pin);
    if (idx->h_flag) hipHostFree(idx->h_flag);
    hipEventDestroy(idx->ev0);
    hipEventDestroy(idx->ev1);
    hipEventDestroy(idx->tm0);
    hipEventDestroy(idx->tm1);
    if (idx->own_stream) hipStreamDestroy(idx->stream);
    delete idx;
    return 0;
}

template <typename P>
static int grow(hvs_index *idx, P **ptr, size_t old_count, size_t new_count, bool zero_new = false, int fill_byte = 0) {
    P *np = nullptr;
    HVS_HIP(hipMalloc((void **)&np, new_count * sizeof(P)));
    if (*ptr && old_count) HVS_HIP(hipMemcpyAsync(np, *ptr, old_count * sizeof(P), hipMemcpyDeviceToDevice, idx->stream));
    if (zero_new && new_count > old_count)
        HVS_HIP(hipMemsetAsync(np + old_count, fill_byte, (new_count - old_count) * sizeof(P), idx->stream));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    if (*ptr) HVS_HIP(hipFree(*ptr));
    *ptr = np;
    return 0;
}

extern "C" int hvs_index_reserve(hvs_index *idx, int64_t capacity_rows) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_reserve: idx is NULL");
    HVS_REQUIRE(capacity_rows < (int64_t)0xFFFFFFFFll, "hvs_index_reserve: at most 2^32-1 rows per index");
    if (capacity_rows <= idx->cap) return 0;
    HVS_HIP(hipSetDevice(idx->device));
    const size_t row_bytes = (size_t)idx->ld * idx->elem_size;
    {
        // Rows beyond n and HVS_ROWS_SLACK bytes after the last one are kept zero: the MFMA kernels fetch whole
        // 128-byte k-steps, so the last step of a row whose pitch is not a multiple of 128 bytes reads into the
        // row after it (times a zero-padded query: harmless as long as what it reads is finite).
        char *np = nullptr;
        HVS_HIP(hipMalloc((void **)&np, (size_t)capacity_rows * row_bytes + HVS_ROWS_SLACK));
        if (idx->rows && idx->n)
            HVS_HIP(hipMemcpyAsync(np, idx->rows, (size_t)idx->n * row_bytes, hipMemcpyDeviceToDevice, idx->stream));
        HVS_HIP(hipMemsetAsync(np + (size_t)idx->n * row_bytes, 0, (size_t)(capacity_rows - idx->n) * row_bytes + HVS_ROWS_SLACK, idx->stream));
        HVS_HIP(hipStreamSynchronize(idx->stream));
        if (idx->rows) HVS_HIP(hipFree(idx->rows));
        idx->rows = np;
    }
    HVS_TRY(grow(idx, &idx->inv_norm, (size_t)idx->n, (size_t)capacity_rows));
    HVS_TRY(grow(idx, &idx->ids, (size_t)idx->n, (size_t)capacity_rows));
    HVS_TRY(grow(idx, &idx->groups, (size_t)idx->n, (size_t)capacity_rows));
    if (idx->deleted)
        HVS_TRY(grow(idx, &idx->deleted, (size_t)(idx->cap + 31) / 32, (size_t)(capacity_rows + 31) / 32, true));
    if (idx->attrs) HVS_TRY(grow(idx, &idx->attrs, (size_t)idx->cap, (size_t)capacity_rows, true, 0xFF));
    idx->cap = capacity_rows;
    return 0;
}

extern "C" int hvs_index_size(hvs_index *idx, int64_t *out_rows) {
    HVS_REQUIRE(idx && out_rows, "hvs_index_size: NULL argument");
    *out_rows = idx->n;
    return 0;
}

extern "C" int hvs_index_info(hvs_index *idx, int *dim, int *dtype, int *device_id, int64_t *capacity_rows) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_info: idx is NULL");
    if (dim) *dim = idx->dim;
    if (dtype) *dtype = idx->dtype;
    if (device_id) *device_id = idx->device;
    if (capacity_rows) *capacity_rows = idx->cap;
    return 0;
}

extern "C" int hvs_sync(hvs_index *idx) {
    HVS_REQUIRE(idx != nullptr, "hvs_sync: idx is NULL");
    HVS_HIP(hipSetDevice(idx->device));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    return 0;
}

// store n rows from a DEVICE source buffer at [first, first+n) or at positions[]
static int store_rows_device(hvs_index *idx, const void *src, int src_dtype, int64_t n, int64_t first,
                             const int64_t *positions) {
    const int g = grid_for(n * (int64_t)idx->ld);
    hipStream_t s = idx->stream;
    hvs_split_invalidate(idx, positions ? 0 : first);
    if (idx->dtype == HVS_F32) {
        HVS_REQUIRE(src_dtype == HVS_F32, "rows for an f32 index must be f32");
        store_rows_kernel<float, float><<<g, 256, 0, s>>>((const float *)src, (float *)idx->rows, n, idx->dim, idx->ld, first, positions);
        inv_norm_kernel<float><<<grid_for(n * 64), 256, 0, s>>>((const float *)idx->rows, idx->inv_norm, n, idx->ld, first, positions);
    } else if (idx->dtype == HVS_F16) {
        if (src_dtype == HVS_F32)
            store_rows_kernel<float, __half><<<g, 256, 0, s>>>((const float *)src, (__half *)idx->rows, n, idx->dim, idx->ld, first, positions);
        else if (src_dtype == HVS_F16)
            store_rows_kernel<__half, __half><<<g, 256, 0, s>>>((const __half *)src, (__half *)idx->rows, n, idx->dim, idx->ld, first, positions);
        else HVS_REQUIRE(false, "rows for an f16 index must be f32 or f16");
        inv_norm_kernel<__half><<<grid_for(n * 64), 256, 0, s>>>((const __half *)idx->rows, idx->inv_norm, n, idx->ld, first, positions);
    } else {
        if (src_dtype == HVS_F32)
            store_rows_kernel<float, uint16_t><<<g, 256, 0, s>>>((const float *)src, (uint16_t *)idx->rows, n, idx->dim, idx->ld, first, positions);
        else if (src_dtype == HVS_BF16)
            store_rows_kernel<uint16_t, uint16_t><<<g, 256, 0, s>>>((const uint16_t *)src, (uint16_t *)idx->rows, n, idx->dim, idx->ld, first, positions);
        else HVS_REQUIRE(false, "rows for a bf16 index must be f32 or bf16");
        inv_norm_kernel<uint16_t><<<grid_for(n * 64), 256, 0, s>>>((const uint16_t *)idx->rows, idx->inv_norm, n, idx->ld, first, positions);
    }
    HVS_HIP(hipGetLastError());
    return 0;
}

extern "C" int hvs_index_add(hvs_index *idx, const void *rows, int src_dtype, int64_t n, const int64_t *ids,
                             const int32_t *group_codes, int on_device) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_add: idx is NULL");
    HVS_REQUIRE(n >= 0, "hvs_index_add: negative row count");
    if (n == 0) return 0;
    HVS_REQUIRE(rows != nullptr, "hvs_index_add: rows is NULL");
    HVS_REQUIRE(src_dtype == HVS_F32 || src_dtype == idx->dtype, "hvs_index_add: source dtype %d not accepted by index dtype %d", src_dtype, idx->dtype);
    HVS_HIP(hipSetDevice(idx->device));
    if (idx->n + n > idx->cap) {
        int64_t want = idx->cap * 2;
        if (want < idx->n + n) want = idx->n + n;
        if (want < 1024) want = 1024;
        HVS_TRY(hvs_index_reserve(idx, want));
    }
    const int64_t first = idx->n;
    const size_t src_row_bytes = (size_t)idx->dim * hvs_elem_size(src_dtype);
    if (on_device) {
        HVS_TRY(store_rows_device(idx, rows, src_dtype, n, first, nullptr));
        fill_meta_kernel<<<grid_for(n), 256, 0, idx->stream>>>(idx->ids, idx->groups, n, first, first, 0, ids, group_codes);
        HVS_HIP(hipGetLastError());
    } else {
        // stage through device scratch in slices of <= 128 MiB
        int64_t slice = (int64_t)((128u << 20) / src_row_bytes);
        if (slice < 1) slice = 1;
        for (int64_t off = 0; off < n; off += slice) {
            const int64_t m = (n - off < slice) ? n - off : slice;
            void *stage = nullptr;
            const size_t meta_off = ((size_t)m * src_row_bytes + 255) & ~(size_t)255;
            HVS_TRY(hvs_ws_get(idx, 3, meta_off + (size_t)m * 12 + 256, &stage));
            HVS_HIP(hipMemcpyAsync(stage, (const char *)rows + (size_t)off * src_row_bytes, (size_t)m * src_row_bytes, hipMemcpyHostToDevice, idx->stream));
            int64_t *d_ids = nullptr;
            int32_t *d_groups = nullptr;
            if (ids) {
                d_ids = (int64_t *)((char *)stage + meta_off);
                HVS_HIP(hipMemcpyAsync(d_ids, ids + off, (size_t)m * 8, hipMemcpyHostToDevice, idx->stream));
            }
            if (group_codes) {
                d_groups = (int32_t *)((char *)stage + meta_off + (size_t)m * 8);
                HVS_HIP(hipMemcpyAsync(d_groups, group_codes + off, (size_t)m * 4, hipMemcpyHostToDevice, idx->stream));
            }
            HVS_TRY(store_rows_device(idx, stage, src_dtype, m, first + off, nullptr));
            fill_meta_kernel<<<grid_for(m), 256, 0, idx->stream>>>(idx->ids, idx->groups, m, first + off, first + off, 0, d_ids, d_groups);
            HVS_HIP(hipGetLastError());
            HVS_HIP(hipStreamSynchronize(idx->stream));  // the stage buffer is reused by the next slice
        }
    }
    idx->n += n;
    if (!on_device) HVS_HIP(hipStreamSynchronize(idx->stream));
    return 0;
}

// positions that arrive in device memory cannot be read by the host: a kernel checks them before anything uses them
__global__ void check_positions_kernel(const int64_t *__restrict__ positions, int64_t n, int64_t limit, int *__restrict__ bad) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (positions[i] < 0 || positions[i] >= limit) *bad = 1;  // every writer stores the same value
}

static int validate_device_positions(hvs_index *idx, const int64_t *d_pos, int64_t n, const char *who) {
    void *flag = nullptr;
    HVS_TRY(hvs_ws_get(idx, 5, 256, &flag));
    HVS_HIP(hipMemsetAsync(flag, 0, 4, idx->stream));
    check_positions_kernel<<<grid_for(n), 256, 0, idx->stream>>>(d_pos, n, idx->n, (int *)flag);
    HVS_HIP(hipGetLastError());
    int bad = 0;
    HVS_HIP(hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, idx->stream));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    HVS_REQUIRE(bad == 0, "%s: a position lies outside [0, %lld)", who, (long long)idx->n);
    return 0;
}

extern "C" int hvs_index_update(hvs_index *idx, const int64_t *positions, const void *rows, int src_dtype, int64_t n,
                                int on_device) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_update: idx is NULL");
    if (n <= 0) return 0;
    HVS_REQUIRE(positions && rows, "hvs_index_update: NULL argument");
    HVS_REQUIRE(src_dtype == HVS_F32 || src_dtype == idx->dtype, "hvs_index_update: source dtype %d not accepted", src_dtype);
    HVS_HIP(hipSetDevice(idx->device));
    const size_t src_row_bytes = (size_t)idx->dim * hvs_elem_size(src_dtype);
    if (on_device) {
        HVS_TRY(validate_device_positions(idx, positions, n, "hvs_index_update"));
        HVS_TRY(store_rows_device(idx, rows, src_dtype, n, 0, positions));
        return 0;
    }
    for (int64_t i = 0; i < n; i++)
        HVS_REQUIRE(positions[i] >= 0 && positions[i] < idx->n, "hvs_index_update: position %lld out of range", (long long)positions[i]);
    void *stage = nullptr;
    const size_t pos_off = ((size_t)n * src_row_bytes + 255) & ~(size_t)255;
    HVS_TRY(hvs_ws_get(idx, 3, pos_off + (size_t)n * 8, &stage));
    HVS_HIP(hipMemcpyAsync(stage, rows, (size_t)n * src_row_bytes, hipMemcpyHostToDevice, idx->stream));
    HVS_HIP(hipMemcpyAsync((char *)stage + pos_off, positions, (size_t)n * 8, hipMemcpyHostToDevice, idx->stream));
    HVS_TRY(store_rows_device(idx, stage, src_dtype, n, 0, (const int64_t *)((char *)stage + pos_off)));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    return 0;
}

extern "C" int hvs_index_set_deleted(hvs_index *idx, const int64_t *positions, int64_t n, int deleted) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_set_deleted: idx is NULL");
    if (n <= 0) return 0;
    HVS_REQUIRE(positions != nullptr, "hvs_index_set_deleted: positions is NULL");
    for (int64_t i = 0; i < n; i++)
        HVS_REQUIRE(positions[i] >= 0 && positions[i] < idx->n, "hvs_index_set_deleted: position %lld out of range", (long long)positions[i]);
    HVS_HIP(hipSetDevice(idx->device));
    if (!idx->deleted) {
        const size_t words = (size_t)(idx->cap + 31) / 32;
        HVS_HIP(hipMalloc((void **)&idx->deleted, words * 4));
        HVS_HIP(hipMemsetAsync(idx->deleted, 0, words * 4, idx->stream));
    }
    void *stage = nullptr;
    const size_t cnt_off = ((size_t)n * 8 + 255) & ~(size_t)255;
    HVS_TRY(hvs_ws_get(idx, 3, cnt_off + 8, &stage));
    unsigned long long *d_flips = (unsigned long long *)((char *)stage + cnt_off);
    HVS_HIP(hipMemcpyAsync(stage, positions, (size_t)n * 8, hipMemcpyHostToDevice, idx->stream));
    HVS_HIP(hipMemsetAsync(d_flips, 0, 8, idx->stream));
    set_deleted_kernel<<<grid_for(n), 256, 0, idx->stream>>>(idx->deleted, (const int64_t *)stage, n, deleted, d_flips);
    HVS_HIP(hipGetLastError());
    unsigned long long flips = 0;
    HVS_HIP(hipMemcpyAsync(&flips, d_flips, 8, hipMemcpyDeviceToHost, idx->stream));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    idx->n_deleted += deleted ? (int64_t)flips : -(int64_t)flips;
    return 0;
}

extern "C" int hvs_index_deleted_count(hvs_index *idx, int64_t *out_rows) {
    HVS_REQUIRE(idx && out_rows, "hvs_index_deleted_count: NULL argument");
    *out_rows = idx->n_deleted;
    return 0;
}

extern "C" int hvs_index_prepare(hvs_index *idx, int *out_ready) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_prepare: idx is NULL");
    HVS_HIP(hipSetDevice(idx->device));
    bool ok = false;
    if (idx->dtype == HVS_F32 && idx->n > 0 && !getenv("HVS_NO_SPLIT")) {
        idx->split_failed = false;  // an explicit request tries again: memory may have been freed since the copy was refused
        HVS_TRY(hvs_split_ensure(idx, &ok));
        if (!ok) hvs_set_error("hvs_index_prepare: %s", idx->split_why.c_str());  // readable through hvs_last_error; the call still succeeds
        HVS_HIP(hipStreamSynchronize(idx->stream));
    }
    if (out_ready) *out_ready = ok ? 1 : 0;
    return 0;
}

extern "C" int hvs_index_search_copy(hvs_index *idx, int allow, int64_t *out_bytes) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_search_copy: idx is NULL");
    if (allow >= 0) {
        idx->split_allowed = allow != 0;
        if (!idx->split_allowed && idx->split) {  // give the memory back now
            HVS_HIP(hipSetDevice(idx->device));
            HVS_HIP(hipStreamSynchronize(idx->stream));
            (void)hipFree(idx->split);
            (void)hipFree(idx->split_inv);
            idx->split = nullptr;
            idx->split_inv = nullptr;
            idx->split_cap = idx->split_n = 0;
        }
        if (idx->split_allowed) idx->split_failed = false;
    }
    if (out_bytes) {
        const int ld16 = (idx->dim + 63) / 64 * 64;
        *out_bytes = idx->split ? (int64_t)idx->split_cap * ((int64_t)ld16 * 4 + 4) + HVS_ROWS_SLACK : 0;
    }
    return 0;
}

extern "C" int hvs_index_export(hvs_index *idx, int64_t first, int64_t n, void *rows, int64_t *ids, int32_t *group_codes,
                                uint8_t *deleted) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_export: idx is NULL");
    HVS_REQUIRE(first >= 0 && n >= 0 && first + n <= idx->n, "hvs_index_export: rows [%lld, %lld) outside [0, %lld)",
                (long long)first, (long long)(first + n), (long long)idx->n);
    if (n == 0) return 0;
    HVS_HIP(hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    if (rows) {
        const size_t row_bytes = (size_t)idx->dim * idx->elem_size, pitch = (size_t)idx->ld * idx->elem_size;
        if (pitch == row_bytes)  // no pad columns: one linear copy (a 2-D copy was seen to deliver nothing under rocprofv3's tracer)
            HVS_HIP(hipMemcpyAsync(rows, (const char *)idx->rows + (size_t)first * pitch, row_bytes * (size_t)n, hipMemcpyDeviceToHost, s));
        else
            HVS_HIP(hipMemcpy2DAsync(rows, row_bytes, (const char *)idx->rows + (size_t)first * pitch, pitch, row_bytes, (size_t)n,
                                     hipMemcpyDeviceToHost, s));
    }
    if (ids) HVS_HIP(hipMemcpyAsync(ids, idx->ids + first, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    if (group_codes) HVS_HIP(hipMemcpyAsync(group_codes, idx->groups + first, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (deleted) {
        void *stage = nullptr;
        HVS_TRY(hvs_ws_get(idx, 3, (size_t)n, &stage));
        unpack_deleted_kernel<<<grid_for(n), 256, 0, s>>>(idx->n_deleted > 0 ? idx->deleted : nullptr, first, n, (uint8_t *)stage);
        HVS_HIP(hipGetLastError());
        HVS_HIP(hipMemcpyAsync(deleted, stage, (size_t)n, hipMemcpyDeviceToHost, s));
    }
    HVS_HIP(hipStreamSynchronize(s));
    return 0;
}

// The table's arrays replaced by the rows at `positions` (device memory, n_live entries, any order), in that order:
// new arrays sized for them, the old ones released only when every copy has succeeded.  Tombstones are gone afterwards
// (the caller lists live rows only).  renumber_ids: ids become the new positions.
static int rebuild_rows(hvs_index *idx, const uint32_t *positions, int64_t n_live, int renumber_ids, const char *who) {
    const size_t pitch = (size_t)idx->ld * idx->elem_size;
    const int64_t cap = n_live > 1024 ? n_live : 1024;
    char *rows = nullptr;
    float *inv = nullptr;
    int64_t *ids = nullptr;
    int32_t *groups = nullptr;
    int4 *attrs = nullptr;
    int rc = 0;
    do {
        hipError_t e;
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)cap * (pitch + 16 + (idx->attrs ? sizeof(int4) : 0)) + HVS_ROWS_SLACK;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < need + ((size_t)64 << 20)) {
            hvs_set_error("%s: the rows move into new arrays of %.2f GB, %.2f GB of device memory are free", who, need / 1e9, free_b / 1e9);
            rc = 1;
            break;
        }
        if (idx->attrs && (e = hipMalloc((void **)&attrs, (size_t)cap * sizeof(int4))) != hipSuccess) {
            hvs_set_error("%s: hipMalloc failed: %s", who, hipGetErrorString(e));
            rc = 1;
            break;
        }
        if (attrs && (e = hipMemsetAsync(attrs, 0xFF, (size_t)cap * sizeof(int4), idx->stream)) != hipSuccess) {
            hvs_set_error("%s: %s", who, hipGetErrorString(e));
            rc = 1;
            break;
        }
        if ((e = hipMalloc((void **)&rows, (size_t)cap * pitch + HVS_ROWS_SLACK)) != hipSuccess ||
            (e = hipMalloc((void **)&inv, (size_t)cap * 4)) != hipSuccess || (e = hipMalloc((void **)&ids, (size_t)cap * 8)) != hipSuccess ||
            (e = hipMalloc((void **)&groups, (size_t)cap * 4)) != hipSuccess) {
            hvs_set_error("%s: hipMalloc failed: %s", who, hipGetErrorString(e));
            rc = 1;
            break;
        }
        if ((e = hipMemsetAsync(rows + (size_t)n_live * pitch, 0, (size_t)(cap - n_live) * pitch + HVS_ROWS_SLACK, idx->stream)) != hipSuccess) {
            hvs_set_error("%s: %s", who, hipGetErrorString(e));
            rc = 1;
            break;
        }
        if (n_live > 0) {
            if ((rc = hvs_live_gather(idx, idx->rows, idx->inv_norm, idx->ld * idx->elem_size, positions, n_live, rows, inv)) != 0) break;
            gather_meta_kernel<<<grid_for(n_live), 256, 0, idx->stream>>>(idx->ids, idx->groups, positions, n_live, ids, groups);
            if (attrs) gather_attrs_kernel<<<grid_for(n_live), 256, 0, idx->stream>>>(idx->attrs, positions, n_live, attrs);
            if (renumber_ids) fill_meta_kernel<<<grid_for(n_live), 256, 0, idx->stream>>>(ids, groups, n_live, 0, 0, 0, nullptr, groups);
        }
        if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(idx->stream)) != hipSuccess) {
            hvs_set_error("%s: %s", who, hipGetErrorString(e));
            rc = 1;
        }
    } while (0);
    if (rc != 0) {
        (void)hipFree(rows);
        (void)hipFree(inv);
        (void)hipFree(ids);
        (void)hipFree(groups);
        (void)hipFree(attrs);
        return rc;
    }
    (void)hipFree(idx->rows);
    (void)hipFree(idx->inv_norm);
    (void)hipFree(idx->ids);
    (void)hipFree(idx->groups);
    (void)hipFree(idx->deleted);
    idx->rows = rows;
    idx->inv_norm = inv;
    idx->ids = ids;
    idx->groups = groups;
    if (attrs) {
        (void)hipFree(idx->attrs);
        idx->attrs = attrs;
    }
    idx->deleted = nullptr;
    idx->n_deleted = 0;
    idx->n = n_live;
    hvs_split_invalidate(idx, 0);
    idx->split_failed = false;  // the table changed size or order: the copy may fit now
    idx->cap = cap;
    return 0;
}

// ---------------------------------------------------------------------------
// group-partitioned order
// ---------------------------------------------------------------------------
// stats[0] = smallest group code of a live row, [1] = largest, [2] = rows whose code is below their predecessor's
__global__ void part_stats_kernel(const int32_t *__restrict__ groups, const uint32_t *__restrict__ deleted, int64_t n, int *__restrict__ stats) {
    int lo = INT32_MAX, hi = INT32_MIN, disorder = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = groups[i];
        if (i > 0 && g < groups[i - 1]) disorder = 1;
        if (deleted && ((deleted[i >> 5] >> (i & 31)) & 1u)) continue;
        lo = g < lo ? g : lo;
        hi = g > hi ? g : hi;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const int l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
        disorder |= __shfl_xor(disorder, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&stats[0], lo);
        atomicMax(&stats[1], hi);
        if (disorder) atomicOr(&stats[2], 1);
    }
}

// sort key of a row: its group code, or dead_key (above every code) for a deleted row
__global__ void part_keys_kernel(const int32_t *__restrict__ groups, const uint32_t *__restrict__ deleted, int64_t n, uint32_t dead_key,
                                 uint32_t *__restrict__ keys) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool dead = deleted && ((deleted[i >> 5] >> (i & 31)) & 1u);
        keys[i] = dead ? dead_key : (uint32_t)groups[i];
    }
}

// (code, first row) of every run of equal codes in groups[0, n); the runs arrive in any order, *count may pass capacity
__global__ void part_runs_kernel(const int32_t *__restrict__ groups, int64_t n, unsigned int capacity, unsigned int *__restrict__ count,
                                 int64_t *__restrict__ runs) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = groups[i];
        if (i == 0 || g != groups[i - 1]) {
            const unsigned int slot = atomicAdd(count, 1u);
            if (slot < capacity) {
                runs[2 * (size_t)slot] = g;
                runs[2 * (size_t)slot + 1] = i;
            }
        }
    }
}

__global__ void widen_positions_kernel(const uint32_t *__restrict__ src, int64_t n, int64_t *__restrict__ dst) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

#define PART_MAX_GROUPS (1u << 22)  // runs of equal codes the range table holds (32 MB of staging); beyond it the table has no partition

bool hvs_part_range(const hvs_index *idx, int32_t code, int64_t *begin, int64_t *end) {
    if (idx->part_n <= 0) return false;
    const auto it = std::lower_bound(idx->part_codes.begin(), idx->part_codes.end(), code);
    if (it == idx->part_codes.end() || *it != code) {
        *begin = *end = 0;  // no row of the ordered part carries this code
        return true;
    }
    const size_t i = (size_t)(it - idx->part_codes.begin());
    *begin = idx->part_begin[i];
    *end = idx->part_begin[i + 1];
    return true;
}

// Range table of rows [0, part_n), which must be ordered by group code.
static int build_part_table(hvs_index *idx, int64_t part_n) {
    idx->part_n = 0;
    idx->part_codes.clear();
    idx->part_begin.clear();
    if (part_n <= 0) return 0;
    void *ws = nullptr;
    const unsigned int capacity = part_n < (int64_t)PART_MAX_GROUPS ? (unsigned int)part_n : PART_MAX_GROUPS;  // a run has at least one row
    HVS_TRY(hvs_ws_get(idx, 5, 256 + (size_t)capacity * 16, &ws));
    unsigned int *d_count = (unsigned int *)ws;
    int64_t *d_runs = (int64_t *)((char *)ws + 256);
    HVS_HIP(hipMemsetAsync(d_count, 0, 4, idx->stream));
    part_runs_kernel<<<grid_for(part_n), 256, 0, idx->stream>>>(idx->groups, part_n, capacity, d_count, d_runs);
    HVS_HIP(hipGetLastError());
    unsigned int n_runs = 0;
    HVS_HIP(hipMemcpyAsync(&n_runs, d_count, 4, hipMemcpyDeviceToHost, idx->stream));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    HVS_REQUIRE(n_runs <= capacity, "hvs_index_partition: %u runs of group codes, the range table holds %u", n_runs, capacity);
    std::vector<int64_t> runs((size_t)n_runs * 2);
    HVS_HIP(hipMemcpy(runs.data(), d_runs, runs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<std::pair<int64_t, int32_t>> by_begin(n_runs);
    for (unsigned int i = 0; i < n_runs; i++) by_begin[i] = {runs[2 * (size_t)i + 1], (int32_t)runs[2 * (size_t)i]};
    std::sort(by_begin.begin(), by_begin.end());
    idx->part_codes.reserve(n_runs);
    idx->part_begin.reserve((size_t)n_runs + 1);
    for (unsigned int i = 0; i < n_runs; i++) {
        HVS_REQUIRE(i == 0 || by_begin[i].second > by_begin[i - 1].second, "hvs_index_partition: rows [0, %lld) are not ordered by group code",
                    (long long)part_n);
        idx->part_codes.push_back(by_begin[i].second);
        idx->part_begin.push_back(by_begin[i].first);
    }
    idx->part_begin.push_back(part_n);
    idx->part_n = part_n;
    return 0;
}

extern "C" int hvs_index_partition(hvs_index *idx, int renumber_ids, int64_t *out_rows, int64_t *out_old_positions) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_partition: idx is NULL");
    HVS_HIP(hipSetDevice(idx->device));
    const int64_t n = idx->n;
    if (n == 0) {
        idx->part_n = 0;
        idx->part_codes.clear();
        idx->part_begin.clear();
        if (out_rows) *out_rows = 0;
        return 0;
    }
    HVS_REQUIRE(n < (int64_t)0xFFFFFFFFll, "hvs_index_partition: row positions beyond 2^32 - 2");
    const uint32_t *dead = idx->n_deleted > 0 ? idx->deleted : nullptr;
    // staging (workspace 6): [stats][keys n][order n][sort scratch]
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_keys = 256, o_order = o_keys + up((size_t)n * 4), o_tmp = o_order + up((size_t)n * 4);
    void *ws = nullptr;
    HVS_TRY(hvs_ws_get(idx, 6, o_tmp + hvs_sort_temp_bytes(n), &ws));
    int *d_stats = (int *)ws;
    uint32_t *d_keys = (uint32_t *)((char *)ws + o_keys), *d_order = (uint32_t *)((char *)ws + o_order);
    const int init[3] = {INT32_MAX, INT32_MIN, 0};
    HVS_HIP(hipMemcpyAsync(d_stats, init, sizeof(init), hipMemcpyHostToDevice, idx->stream));
    part_stats_kernel<<<grid_for(n), 256, 0, idx->stream>>>(idx->groups, dead, n, d_stats);
    HVS_HIP(hipGetLastError());
    int stats[3] = {0, 0, 0};
    HVS_HIP(hipMemcpyAsync(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, idx->stream));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    const bool any_live = idx->n_deleted < n;
    HVS_REQUIRE(!any_live || stats[0] >= 0, "hvs_index_partition: group code %d; the partition takes codes >= 0 (a filter can name no other)", stats[0]);
    const bool in_order = stats[2] == 0 && idx->n_deleted == 0;
    if (!in_order) {
        const uint32_t dead_key = any_live ? (uint32_t)stats[1] + 1u : 0u;
        int key_bits = 1;
        while (key_bits < 32 && (dead_key >> key_bits) != 0) key_bits++;
        part_keys_kernel<<<grid_for(n), 256, 0, idx->stream>>>(idx->groups, dead, n, dead_key, d_keys);
        HVS_HIP(hipGetLastError());
        HVS_TRY(hvs_sort_by_key(idx->stream, d_keys, n, key_bits, d_order, (char *)ws + o_tmp));
        const int64_t n_live = n - idx->n_deleted;
        if (out_old_positions && n_live > 0) {  // before the rows move: the order lives in a workspace the range table reuses
            widen_positions_kernel<<<grid_for(n_live), 256, 0, idx->stream>>>(d_order, n_live, (int64_t *)((char *)ws + o_tmp));
            HVS_HIP(hipGetLastError());
            HVS_HIP(hipMemcpyAsync(out_old_positions, (char *)ws + o_tmp, (size_t)n_live * 8, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipStreamSynchronize(idx->stream));
        }
        HVS_TRY(rebuild_rows(idx, d_order, n_live, renumber_ids, "hvs_index_partition"));
    } else {
        if (out_old_positions)
            for (int64_t i = 0; i < n; i++) out_old_positions[i] = i;
        if (renumber_ids) {
            fill_meta_kernel<<<grid_for(n), 256, 0, idx->stream>>>(idx->ids, idx->groups, n, 0, 0, 0, nullptr, idx->groups);
            HVS_HIP(hipGetLastError());
        }
    }
    HVS_TRY(build_part_table(idx, idx->n));
    if (out_rows) *out_rows = idx->n;
    return 0;
}

extern "C" int hvs_index_partition_info(hvs_index *idx, int64_t *ordered_rows, int64_t *n_groups) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_partition_info: idx is NULL");
    if (ordered_rows) *ordered_rows = idx->part_n;
    if (n_groups) *n_groups = (int64_t)idx->part_codes.size();
    return 0;
}

extern "C" int hvs_index_set_ids(hvs_index *idx, int64_t first_pos, int64_t n, const int64_t *ids) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_set_ids: idx is NULL");
    HVS_REQUIRE(first_pos >= 0 && n >= 0 && first_pos + n <= idx->n, "hvs_index_set_ids: rows [%lld, %lld) outside [0, %lld)",
                (long long)first_pos, (long long)(first_pos + n), (long long)idx->n);
    if (n == 0) return 0;
    HVS_REQUIRE(ids != nullptr, "hvs_index_set_ids: ids is NULL");
    HVS_HIP(hipSetDevice(idx->device));
    HVS_HIP(hipMemcpyAsync(idx->ids + first_pos, ids, (size_t)n * 8, hipMemcpyHostToDevice, idx->stream));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    return 0;
}

extern "C" int hvs_index_compact(hvs_index *idx, int renumber_ids, int64_t *out_rows) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_compact: idx is NULL");
    HVS_HIP(hipSetDevice(idx->device));
    if (idx->n_deleted > 0 && idx->n > 0) {
        // compaction keeps the order of the survivors, so the ordered part of a partitioned table stays ordered: it
        // shrinks to its live rows and the range table is rebuilt below
        uint32_t part_live = 0;
        if (idx->part_n > 0) {
            hvs_live_params lpp{idx->groups, nullptr, idx->deleted, -1, 0, idx->part_n};
            HVS_TRY(hvs_live_count(idx, lpp, &part_live));
        }
        hvs_live_params lp{idx->groups, nullptr, idx->deleted, -1, 0, idx->n};
        uint32_t n_live = 0;
        HVS_TRY(hvs_live_count(idx, lp, &n_live));
        uint32_t *positions = nullptr;
        hipError_t e = hipMalloc((void **)&positions, (size_t)n_live * 4 + 4);
        HVS_REQUIRE(e == hipSuccess, "hvs_index_compact: hipMalloc failed: %s", hipGetErrorString(e));
        int rc = 0;
        if (n_live > 0) rc = hvs_live_write(idx, lp, positions);
        if (rc == 0) rc = rebuild_rows(idx, positions, n_live, renumber_ids, "hvs_index_compact");
        (void)hipFree(positions);
        if (rc != 0) return rc;
        if (idx->part_n > 0) HVS_TRY(build_part_table(idx, part_live));
    } else if (renumber_ids && idx->n > 0) {
        fill_meta_kernel<<<grid_for(idx->n), 256, 0, idx->stream>>>(idx->ids, idx->groups, idx->n, 0, 0, 0, nullptr, idx->groups);
        HVS_HIP(hipGetLastError());
        HVS_HIP(hipStreamSynchronize(idx->stream));
    }
    if (out_rows) *out_rows = idx->n;
    return 0;
}

extern "C" int hvs_index_fill_synthetic(hvs_index *idx, int64_t n, uint64_t seed, int64_t first_global_row,
                                        int32_t n_groups) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_fill_synthetic: idx is NULL");
    HVS_REQUIRE(n >= 0 && first_global_row >= 0, "hvs_index_fill_synthetic: negative argument");
    if (n == 0) return 0;
    HVS_HIP(hipSetDevice(idx->device));
    if (idx->n + n > idx->cap) HVS_TRY(hvs_index_reserve(idx, idx->n + n));
    const int64_t first = idx->n;
    // splitmix64(seed) on the host, same function as the device's
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    const uint64_t seed_mix = z ^ (z >> 31);
    const int g = grid_for(n * (int64_t)idx->ld, 256, 256 * 16);
    hipStream_t s = idx->stream;
    if (idx->dtype == HVS_F32) {
        fill_synth_kernel<float><<<g, 256, 0, s>>>((float *)idx->rows, n, idx->dim, idx->ld, first, first_global_row, seed_mix);
        inv_norm_kernel<float><<<grid_for(n * 64), 256, 0, s>>>((const float *)idx->rows, idx->inv_norm, n, idx->ld, first, nullptr);
    } else if (idx->dtype == HVS_F16) {
        fill_synth_kernel<__half><<<g, 256, 0, s>>>((__half *)idx->rows, n, idx->dim, idx->ld, first, first_global_row, seed_mix);
        inv_norm_kernel<__half><<<grid_for(n * 64), 256, 0, s>>>((const __half *)idx->rows, idx->inv_norm, n, idx->ld, first, nullptr);
    } else {
        fill_synth_kernel<uint16_t><<<g, 256, 0, s>>>((uint16_t *)idx->rows, n, idx->dim, idx->ld, first, first_global_row, seed_mix);
        inv_norm_kernel<uint16_t><<<grid_for(n * 64), 256, 0, s>>>((const uint16_t *)idx->rows, idx->inv_norm, n, idx->ld, first, nullptr);
    }
    HVS_HIP(hipGetLastError());
    fill_meta_kernel<<<grid_for(n), 256, 0, s>>>(idx->ids, idx->groups, n, first, first_global_row, n_groups, nullptr, nullptr);
    HVS_HIP(hipGetLastError());
    HVS_HIP(hipStreamSynchronize(s));
    idx->n += n;
    return 0;
}

static uint64_t host_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

extern "C" int hvs_index_fill_clustered(hvs_index *idx, int64_t n, uint64_t seed, int64_t first_global_row, int32_t n_groups,
                                        int64_t n_clusters) {
    HVS_REQUIRE(idx != nullptr, "hvs_index_fill_clustered: idx is NULL");
    HVS_REQUIRE(n >= 0 && first_global_row >= 0, "hvs_index_fill_clustered: negative argument");
    HVS_REQUIRE(n_clusters >= 1, "hvs_index_fill_clustered: n_clusters %lld < 1", (long long)n_clusters);
    if (n == 0) return 0;
    HVS_HIP(hipSetDevice(idx->device));
    if (idx->n + n > idx->cap) HVS_TRY(hvs_index_reserve(idx, idx->n + n));
    const int64_t first = idx->n;
    hvs_cluster_params cp;
    cp.mix_noise = host_splitmix64(seed);
    cp.mix_centre = host_splitmix64(seed + 1);
    cp.mix_assign = host_splitmix64(seed ^ 0xC1);
    cp.mix_level = host_splitmix64(seed ^ 0xC2);
    cp.n_clusters = (uint64_t)n_clusters;
    static const double spread[8] = {1.0e-3, 1.4e-3, 2.0e-3, 2.8e-3, 4.0e-3, 5.6e-3, 8.0e-3, 1.0e-2};
    for (int i = 0; i < 8; i++) cp.sigma[i] = (float)std::sqrt(spread[i]);
    const int g = grid_for(n * (int64_t)idx->ld, 256, 256 * 16);
    hipStream_t s = idx->stream;
    hvs_split_invalidate(idx, first);
    if (idx->dtype == HVS_F32) {
        fill_clustered_kernel<float><<<g, 256, 0, s>>>((float *)idx->rows, n, idx->dim, idx->ld, first, first_global_row, cp);
        inv_norm_kernel<float><<<grid_for(n * 64), 256, 0, s>>>((const float *)idx->rows, idx->inv_norm, n, idx->ld, first, nullptr);
    } else if (idx->dtype == HVS_F16) {
        fill_clustered_kernel<__half><<<g, 256, 0, s>>>((__half *)idx->rows, n, idx->dim, idx->ld, first, first_global_row, cp);
        inv_norm_kernel<__half><<<grid_for(n * 64), 256, 0, s>>>((const __half *)idx->rows, idx->inv_norm, n, idx->ld, first, nullptr);
    } else {
        fill_clustered_kernel<uint16_t><<<g, 256, 0, s>>>((uint16_t *)idx->rows, n, idx->dim, idx->ld, first, first_global_row, cp);
        inv_norm_kernel<uint16_t><<<grid_for(n * 64), 256, 0, s>>>((const uint16_t *)idx->rows, idx->inv_norm, n, idx->ld, first, nullptr);
    }
    HVS_HIP(hipGetLastError());
    fill_meta_kernel<<<grid_for(n), 256, 0, s>>>(idx->ids, idx->groups, n, first, first_global_row, n_groups, nullptr, nullptr);
    HVS_HIP(hipGetLastError());
    HVS_HIP(hipStreamSynchronize(s));
    idx->n += n;
    return 0;
}

extern "C" int hvs_gather(hvs_index *idx, const int64_t *positions, int64_t n, float *out, int on_device) {
    HVS_REQUIRE(idx != nullptr, "hvs_gather: idx is NULL");
    if (n <= 0) return 0;
    HVS_REQUIRE(positions && out, "hvs_gather: NULL argument");
    HVS_HIP(hipSetDevice(idx->device));
    const int64_t *d_pos = positions;
    float *d_out = out;
    const size_t out_bytes = (size_t)n * idx->dim * 4;
    if (!on_device) {
        for (int64_t i = 0; i < n; i++)
            HVS_REQUIRE(positions[i] >= 0 && positions[i] < idx->n, "hvs_gather: position %lld out of range", (long long)positions[i]);
        void *stage = nullptr;
        const size_t pos_off = (out_bytes + 255) & ~(size_t)255;
        HVS_TRY(hvs_ws_get(idx, 3, pos_off + (size_t)n * 8, &stage));
        d_out = (float *)stage;
        d_pos = (const int64_t *)((char *)stage + pos_off);
        HVS_HIP(hipMemcpyAsync((void *)d_pos, positions, (size_t)n * 8, hipMemcpyHostToDevice, idx->stream));
    } else
        HVS_TRY(validate_device_positions(idx, positions, n, "hvs_gather"));
    const int g = grid_for(n * (int64_t)idx->dim);
    if (idx->dtype == HVS_F32) gather_kernel<float><<<g, 256, 0, idx->stream>>>((const float *)idx->rows, d_pos, n, idx->dim, idx->ld, d_out);
    else if (idx->dtype == HVS_F16) gather_kernel<__half><<<g, 256, 0, idx->stream>>>((const __half *)idx->rows, d_pos, n, idx->dim, idx->ld, d_out);
    else gather_kernel<uint16_t><<<g, 256, 0, idx->stream>>>((const uint16_t *)idx->rows, d_pos, n, idx->dim, idx->ld, d_out);
    HVS_HIP(hipGetLastError());
    if (!on_device) {
        HVS_HIP(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, idx->stream));
        HVS_HIP(hipStreamSynchronize(idx->stream));
    }
    return 0;
}

extern "C" int hvs_profile_enable(hvs_index *idx, int on) {
    HVS_REQUIRE(idx != nullptr, "hvs_profile_enable: idx is NULL");
    idx->profiling = on != 0;
    return 0;
}

extern "C" int hvs_profile_get(hvs_index *idx, int which, double *total_ms, int64_t *launches) {
    HVS_REQUIRE(idx != nullptr, "hvs_profile_get: idx is NULL");
    HVS_REQUIRE(which >= 0 && which < HVS_PROF_COUNT, "hvs_profile_get: unknown counter %d", which);
    if (total_ms) *total_ms = idx->prof_ms[which];
    if (launches) *launches = idx->prof_launches[which];
    return 0;
}

extern "C" int hvs_profile_reset(hvs_index *idx) {
    HVS_REQUIRE(idx != nullptr, "hvs_profile_reset: idx is NULL");
    for (int i = 0; i < HVS_PROF_COUNT; i++) {
        idx->prof_ms[i] = 0.0;
        idx->prof_launches[i] = 0;
    }
    return 0;
}

extern "C" int hvs_timer_start(hvs_index *idx) {
    HVS_REQUIRE(idx != nullptr, "hvs_timer_start: idx is NULL");
    HVS_HIP(hipSetDevice(idx->device));
    HVS_HIP(hipEventRecord(idx->tm0, idx->stream));
    return 0;
}

extern "C" int hvs_timer_stop(hvs_index *idx, double *elapsed_ms) {
    HVS_REQUIRE(idx && elapsed_ms, "hvs_timer_stop: NULL argument");
    HVS_HIP(hipSetDevice(idx->device));
    HVS_HIP(hipEventRecord(idx->tm1, idx->stream));
    HVS_HIP(hipEventSynchronize(idx->tm1));
    float ms = 0.0f;
    HVS_HIP(hipEventElapsedTime(&ms, idx->tm0, idx->tm1));
    *elapsed_ms = ms;
    return 0;
}
