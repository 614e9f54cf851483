// Internal definitions shared by the translation units of libhvs.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hvs.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
void hvs_set_error(const char *fmt, ...);

#define HVS_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            hvs_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

#define HVS_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            hvs_set_error(__VA_ARGS__);   \
            return 2;                     \
        }                                 \
    } while (0)

#define HVS_TRY(expr)              \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != 0) return rc_;  \
    } while (0)

// ---------------------------------------------------------------------------
// index object
// ---------------------------------------------------------------------------
struct hvs_workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
};

struct hvs_index {
    int device = 0;
    int dim = 0;        // logical dimension
    int ld = 0;         // row pitch in elements (dim rounded up so a row is a whole number of 16-B chunks)
    int dtype = HVS_F32;
    int elem_size = 4;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    void *rows = nullptr;         // [cap, ld] elements, pad columns zero
    float *inv_norm = nullptr;    // [cap] 1/|row| (+inf for a zero row, so distance becomes NaN)
    int64_t *ids = nullptr;       // [cap] external ids
    int32_t *groups = nullptr;    // [cap] group codes
    int4 *attrs = nullptr;        // [cap] {workspace code, knowledge-base code, uri code, private flag}; allocated on first use, zero = unset
    uint32_t *deleted = nullptr;  // [ceil(cap/32)] bit set = row deleted; allocated on first delete
    int64_t n_deleted = 0;
    int64_t n = 0;
    int64_t cap = 0;

    // group-partitioned order (hvs_index_partition): rows [0, part_n) are ordered by group code, part_codes (ascending)
    // and part_begin (one more entry than codes; the last is part_n) give every code its row range, so that a search
    // with a group filter scans a range without testing a code; rows [part_n, n) arrived afterwards, in any order
    int64_t part_n = 0;
    std::vector<int32_t> part_codes;
    std::vector<int64_t> part_begin;
    bool emit_positions = false;  // set around the two sub-searches of a (range, tail) pair: their ids are row positions

    // fp32 indexes only: the rows as pairs of fp16 slabs for the 16-bit MFMA pipe (hvs_split.hip), built on the first
    // batched search and extended after appends; rows [0, split_n) are current
    void *split = nullptr;
    float *split_inv = nullptr;
    int64_t split_cap = 0, split_n = 0;
    uint32_t *split_rel = nullptr;  // [1] float bits: largest |row - hi slab| / |row| over the rows of the copy (split_rows_kernel)
    const float *q_rel = nullptr;   // the same per query row of the last search that split its queries (lives in workspace 4 until the next search)
    bool split_allowed = true;  // hvs_index_search_copy(idx, 0): never build the copy (the table keeps its own size in HBM)
    bool split_failed = false;  // device memory did not allow the copy (cleared by hvs_index_compact and by an explicit hvs_index_prepare)
    std::string split_why;      // ... and why, for hvs_index_prepare's caller
    // the filter route learns from its own outcome: after a call in which more than half of the candidate lists proved
    // nothing (a table of tight clusters), the next filter_skip calls go straight to the route that scores every row in
    // full; the run of skipped calls doubles with every failure in a row (4, 8, ... 64) and ends with the first success
    int filter_skip = 0, filter_backoff = 0;
    bool filter_long = false;  // the short candidate lists (4 k) failed for most of a batch: start with the long ones (MFMA_MAX_K)
    int km_filter_skip = 0, km_filter_backoff = 0;  // the same for the k-means assignment's filter pass (more than a quarter of the rows flagged)

    hvs_workspace ws[11];  // scratch buffers reused across calls (contents are lost when one grows): 0 queries, 1/2 candidate
                           // lists, 3 staging, 4 MFMA A operand, 5 live-row counts, 6 compacted rows of a filtered search,
                           // 7 candidates of the filter pass (hvs_search_rescore), 8 row bitmask of an hvs_filter, 9 its uri codes,
                           // 10 the two lists of a (group range, tail) search

    void *h_pin = nullptr;  // pinned host buffer: the results of a search cross to the host in one copy
    int *h_flag = nullptr;  // pinned host word the batched kernels raise when a candidate array overflows (never, by construction):
                            // read after the stream has been synchronised, in place of a copy from the device per call
    size_t h_pin_bytes = 0;

    bool profiling = false;
    double prof_ms[HVS_PROF_COUNT] = {0};
    int64_t prof_launches[HVS_PROF_COUNT] = {0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // profiling pair
    hipEvent_t tm0 = nullptr, tm1 = nullptr;  // user timer pair
};

int hvs_ws_get(hvs_index *idx, int slot, size_t bytes, void **out);
int hvs_pinned_get(hvs_index *idx, size_t bytes, void **out);  // pinned host memory, grown like a workspace

// RAII-less profiling bracket: HVS_PROF_BEGIN(idx) ... launch ... HVS_PROF_END(idx, which)
int hvs_prof_begin(hvs_index *idx);
int hvs_prof_end(hvs_index *idx, int which);

// Rows of [row_begin, row_end) that pass a filter, as an ordered list of positions (hvs_index.hip).
struct hvs_live_params {
    const int32_t *groups;
    const uint32_t *mask;     // optional: bit set = row may be returned
    const uint32_t *deleted;  // optional: bit set = row deleted
    int32_t group_filter;     // -1 = any group
    int64_t row_begin, row_end;
    const float *dist = nullptr;  // optional: distance of row r at dist[r - row_begin]; the row passes only if it is < max_dist
    float max_dist = 0.0f;
};
int hvs_live_count(hvs_index *idx, const hvs_live_params &lp, uint32_t *n_live);          // synchronises the stream
int hvs_live_write(hvs_index *idx, const hvs_live_params &lp, uint32_t *positions);        // after hvs_live_count
// dst row i = row positions[i] of a row matrix with rows of row_bytes (a multiple of 16), with its inverse norm
int hvs_live_gather(hvs_index *idx, const void *src_rows, const float *src_inv, int row_bytes, const uint32_t *positions,
                    int64_t n_live, char *dst_rows, float *dst_inv);
// (distance, id) of the rows that pass lp (lp.dist set), ascending row position, first `capacity` of them; *total = all
int hvs_range_collect(hvs_index *idx, const hvs_live_params &lp, int64_t capacity, float *d_out_dist, int64_t *d_out_ids,
                      int64_t *total);

int hvs_merge_launch(hipStream_t st, const float *in_dist, const int64_t *in_ids, const int32_t *in_count, int nq, int P, int k,
                     int64_t dist_q, int64_t dist_p, int64_t ids_q, int64_t ids_p, int64_t cnt_q, int64_t cnt_p, float *out_dist,
                     int64_t *out_ids, int32_t *out_count);  // hvs_scan.hip
// Row bitmask of a filter's attribute clauses into workspace 8, for the rows a search of [row_begin, row_end) with that
// filter can reach (*d_mask = nullptr when the filter has no such clause; the group clause stays with the kernels' own
// group test, or with the group's row range on a table in group order).  hvs_index.hip
int hvs_filter_mask(hvs_index *idx, const hvs_filter *f, const uint32_t **d_mask, int64_t row_begin, int64_t row_end);
int hvs_attrs_ensure(hvs_index *idx);
// hvs_search / hvs_range_search with a row mask that already lies on the device whatever on_device says (hvs_scan.hip)
int hvs_search_core(hvs_index *idx, const float *queries, int nq, int fold, int k, int32_t group_filter, const uint32_t *row_mask,
                    bool mask_on_device, double max_distance, int64_t row_begin, int64_t row_end, float *out_dist, int64_t *out_ids,
                    int32_t *out_count, int on_device, int algo);
int hvs_range_search_core(hvs_index *idx, const float *queries, int fold, int32_t group_filter, const uint32_t *row_mask,
                          bool mask_on_device, double max_distance, int64_t row_begin, int64_t row_end, int64_t capacity,
                          float *out_dist, int64_t *out_ids, int64_t *out_total, int on_device);
// the row range of a group code in the partitioned part of the table; false when the table has no partition
bool hvs_part_range(const hvs_index *idx, int32_t code, int64_t *begin, int64_t *end);
size_t hvs_sort_temp_bytes(int64_t n);  // hvs_sort.hip
int hvs_sort_by_key(hipStream_t st, const uint32_t *keys, int64_t n, int key_bits, uint32_t *order_out, void *tmp);
int hvs_noalloc_setup(int device);  // hvs_scan.hip: address of the scratch array of hvs_search_noalloc on `device` (current)

int hvs_split_rows(hipStream_t st, const float *src, const float *src_inv, int64_t first, int64_t n, int cols, int ld_src, int ld16,
                   void *dst, float *dst_inv, float *dst_rel, uint32_t *dst_rel_max);
int hvs_split_ensure(hvs_index *idx, bool *ok);
static inline void hvs_split_invalidate(hvs_index *idx, int64_t from_row) {
    if (from_row < idx->split_n) idx->split_n = from_row < 0 ? 0 : from_row;
}

#ifndef MERGE_FAN
#define MERGE_FAN 8  // sorted lists merged by one wave of the merge kernel, one after the other (hvs_scan.hip): the chain of
                     // merges is what a level costs, so a low fan-in and more levels win (32: 36 us, 16: 25-28 us, 8: 19-21 us
                     // for the ~2 000 - 4 000 wave lists of a scan)
#endif

#define HVS_ROWS_SLACK 256  // zeroed bytes kept after the last row of a row matrix (see hvs_index_reserve)

static inline int hvs_elem_size(int dtype) { return dtype == HVS_F32 ? 4 : 2; }

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
#define HVS_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull

// float -> uint32 whose unsigned order is the float order (NaN handled by the caller)
__device__ __forceinline__ uint32_t hvs_ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float hvs_unord(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}
__host__ __device__ __forceinline__ uint64_t hvs_make_key(uint32_t ord, uint32_t row) {
    return ((uint64_t)ord << 32) | row;
}

// cosine distance from a dot product and the two inverse norms, pgvector's way:
// clamp the similarity to [-1, 1] without destroying NaN, return 1 - sim.
__device__ __forceinline__ float hvs_cos_dist(float dot, float inv_a, float inv_b) {
    float sim = dot * inv_a * inv_b;
    sim = sim > 1.0f ? 1.0f : (sim < -1.0f ? -1.0f : sim);
    return 1.0f - sim;
}

__device__ __forceinline__ float hvs_bf16_to_float(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// round-to-nearest-even float -> bf16 bits; NaN stays NaN
__device__ __forceinline__ uint16_t hvs_float_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

__device__ __forceinline__ uint64_t hvs_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// smallest float >= d (so that "x < d" over floats x equals "x < result")
static inline float hvs_ceil_to_float(double d) {
    if (std::isnan(d)) return INFINITY;
    float f = (float)d;
    if ((double)f < d) f = std::nextafterf(f, INFINITY);
    return f;
}
