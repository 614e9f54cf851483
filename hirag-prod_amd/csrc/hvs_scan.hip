// Bandwidth-bound exact scan (GEMV) with a fused wavefront-level top-k, the list-merge kernels and
// the hvs_search / hvs_distances / hvs_merge_topk entry points.
//
// Data flow of one search:  rows stream HBM -> VGPRs exactly once (16 B per lane per load, 16 to 64
// lanes per row, several rows in flight per wave); the query embeddings sit in LDS in the rows' own
// element type; every wave keeps a sorted top-k list per query in LDS and only touches it when a row
// beats the list's current worst (rare after the first few hundred rows), so the score vector is never
// written anywhere.  Every block then merges its waves' lists in LDS and writes k keys; log_32(P) tiny merge
// launches reduce the P block lists to the final answer.
#include "hvs_internal.h"

#include <mutex>
#include <unordered_map>

#define SCAN_THREADS 256
#define SCAN_WAVES (SCAN_THREADS / 64)
#define SCAN_R 4          // rows a wave has in flight (one or two embeddings)
// With three or more embeddings the accumulators and the per-row tail grow; two rows in flight keep the
// kernel at three waves per SIMD instead of two.
// (Eight rows in flight for single-embedding scans of 16-bit rows were measured: 57 % instead of 82 % of the HBM
// peak on 1M x 1536 f16 -- the extra registers cost a wave per SIMD.)
constexpr int scan_rows_in_flight(int nq, int elem_size = 4) { return nq >= 3 ? 2 : SCAN_R; }

struct ScanParams {
    const void *rows;
    const float *inv_norm;
    const int32_t *groups;
    const uint32_t *mask;     // optional: bit set = row may be returned
    const uint32_t *deleted;  // optional: bit set = row deleted
    const float *queries;     // [nq_total, ld] fp32, pad columns zero
    const float *q_inv;       // [nq_total]
    const float *raw;         // optional: the embeddings as the caller gave them, [nq_total, raw_dim] fp32 -- the kernel then rounds
    int raw_dim;              // them to the rows' type and forms the inverse norms itself (what prep_queries_kernel does) and
                              // reads neither queries nor q_inv
    int64_t row_begin, row_end;
    int64_t row_begin2, row_end2;  // optional second range, scanned behind the first in the same launch with its own group test
                                   // (group_filter2): a group's rows in the ordered part of a table, then the tail of rows added since
    int ld;       // row pitch in elements
    int chunks;   // 16-byte chunks per row
    int fold;     // embeddings folded by LEAST per output query
    int nq_out;   // output queries in this launch (nq_out * fold <= NQ)
    int k;
    uint32_t thr_ord;  // ordered-uint image of the strict distance bound
    int32_t group_filter;
    int32_t group_filter2;
    uint64_t *partial;  // [nq_out, gridDim.x, k] keys (topk mode)
    float *dist_out;    // [row_end - row_begin] (distance mode)
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- 16 bytes of a row against 16 bytes of a query (same element type, same k range) ----------
// fp32: two packed fmas (even / odd partial sums).  fp16 / bf16: four v_dot2c_f32_{f16,bf16}: two exact products added to the fp32
// accumulator each; the query is held in the row's own 16-bit type, which loses nothing because it
// has already been rounded to that type.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int DT> struct Chunk;
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <> struct Chunk<HVS_F32> {
    static constexpr int ESZ = 4;
    // two partial sums per dot product (even and odd elements), so that 16 bytes cost two v_pk_fma_f32
    // instead of four v_fma_f32: with several embeddings per row the scan is VALU-issue-limited
    typedef f32x2 Acc;
    __device__ static __forceinline__ Acc zero() { return (f32x2){0.0f, 0.0f}; }
    __device__ static __forceinline__ Acc dot(const u32x4 &v, const u32x4 &q, Acc acc) {
        const f32x2 v0 = {__uint_as_float(v.x), __uint_as_float(v.y)}, v1 = {__uint_as_float(v.z), __uint_as_float(v.w)};
        const f32x2 q0 = {__uint_as_float(q.x), __uint_as_float(q.y)}, q1 = {__uint_as_float(q.z), __uint_as_float(q.w)};
        acc = __builtin_elementwise_fma(v0, q0, acc);
        return __builtin_elementwise_fma(v1, q1, acc);
    }
    __device__ static __forceinline__ float finish(Acc acc) { return acc.x + acc.y; }
    __device__ static __forceinline__ void put_query(char *dst, int c, float v) { reinterpret_cast<float *>(dst)[c] = v; }
};
template <> struct Chunk<HVS_F16> {
    static constexpr int ESZ = 2;
    typedef float Acc;
    __device__ static __forceinline__ Acc zero() { return 0.0f; }
    __device__ static __forceinline__ float finish(Acc acc) { return acc; }
    __device__ static __forceinline__ float dot(const u32x4 &v, const u32x4 &q, float acc) {
        h16x2 a[4], b[4];
        __builtin_memcpy(a, &v, 16);
        __builtin_memcpy(b, &q, 16);
#pragma unroll
        for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_fdot2(a[i], b[i], acc, false);
        return acc;
    }
    __device__ static __forceinline__ void put_query(char *dst, int c, float v) {
        reinterpret_cast<__half *>(dst)[c] = __float2half_rn(v);
    }
};
template <> struct Chunk<HVS_BF16> {
    static constexpr int ESZ = 2;
    typedef float Acc;
    __device__ static __forceinline__ Acc zero() { return 0.0f; }
    __device__ static __forceinline__ float finish(Acc acc) { return acc; }
    __device__ static __forceinline__ float dot(const u32x4 &v, const u32x4 &q, float acc) {
        bf16x2 a[4], b[4];
        __builtin_memcpy(a, &v, 16);
        __builtin_memcpy(b, &q, 16);
#pragma unroll
        for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_fdot2_f32_bf16(a[i], b[i], acc, false);
        return acc;
    }
    __device__ static __forceinline__ void put_query(char *dst, int c, float v) {
        reinterpret_cast<uint16_t *>(dst)[c] = hvs_float_to_bf16(v);
    }
};

// ---- 16-bit rows, two to four embeddings: v_mfma_f32_4x4x4_16B_{f16,bf16} -------------------------
// The instruction multiplies 16 independent blocks: D_b[i][j] += sum_k A_b[i][k] B_b[k][j], i, j, k < 4.
// Lane 4 b + i supplies A_b[i][0..3], lane 4 b + j supplies B_b[0..3][j] and receives D_b[0..3][j].  With
// A = four embeddings and B = four rows, both at the same four elements, one instruction does the work
// of sixteen dot2 instructions and the scan stays bandwidth-bound with several embeddings.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int DT> __device__ __forceinline__ f32x4v mfma4(const u32x4 &q, const u32x4 &v, f32x4v acc) {
    uint32_t qa[4] = {q.x, q.y, q.z, q.w}, va[4] = {v.x, v.y, v.z, v.w};
    if (DT == HVS_F16) {
        h16x4 a[2], b[2];
        __builtin_memcpy(a, qa, 16);
        __builtin_memcpy(b, va, 16);
        acc = __builtin_amdgcn_mfma_f32_4x4x4f16(a[0], b[0], acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_4x4x4f16(a[1], b[1], acc, 0, 0, 0);
    } else {
        s16x4 a[2], b[2];
        __builtin_memcpy(a, qa, 16);
        __builtin_memcpy(b, va, 16);
        acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[0], b[0], acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[1], b[1], acc, 0, 0, 0);
    }
}

// ---- wave-private list in LDS ------------------------------------------------------------
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A wave's list holds up to k keys in no particular order (keys are distinct: the low word is the row).  While it has room,
// the keys below the bound are appended where they arrive, one LDS write for all the lanes' keys.  Once it is full, tau is its
// largest key: a key below tau takes that key's slot and the largest is found again -- each lane the largest of its slots,
// then six DPP steps over the lanes for the distance word and six for the row word.  The lists are put in order once, when the
// wave has seen all its rows (wave_sort).  (A list kept in order costs several dependent LDS round trips per key, and until a
// list is full every key below the bound is one: the first k rows of every wave, and on a short range, where a wave sees
// fewer than k rows, all of them -- a 300 MB range of 768-column bf16 rows, k = 100: 48 rows per wave.)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_umax_step(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);  // no source: the lane's own
    return o > v ? o : v;
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
    v = dpp_umax_step<0x111, 0xF>(v);  // row_shr:1
    v = dpp_umax_step<0x112, 0xF>(v);  // row_shr:2
    v = dpp_umax_step<0x114, 0xF>(v);  // row_shr:4
    v = dpp_umax_step<0x118, 0xF>(v);  // row_shr:8: lane 15 of every row of 16 lanes has the row's
    v = dpp_umax_step<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_umax_step<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
// The largest key of the full list L[0..k), after the slot that holds `out` (a key of the list, or HVS_KEY_EMPTY for none)
// has taken `in`.
__device__ __forceinline__ uint64_t wave_swap_largest(volatile uint64_t *L, int k, uint64_t out, uint64_t in, int lane) {
    wave_fence();  // earlier writes to the list are in it
    uint64_t m = 0;
    for (int i = lane; i < k; i += 64) {
        uint64_t v = L[i];
        if (v == out) {
            v = in;
            L[i] = in;
        }
        m = v > m ? v : m;
    }
    const uint32_t hi = wave_umax((uint32_t)(m >> 32));
    const uint32_t lo = wave_umax((uint32_t)(m >> 32) == hi ? (uint32_t)m : 0u);
    return ((uint64_t)hi << 32) | lo;
}

// Offer one candidate key per lane (HVS_KEY_EMPTY = none) to the list; tau = current admission bound (the caller's bound
// until the list is full).
__device__ __forceinline__ void wave_offer(volatile uint64_t *L, int &cnt, int k, uint64_t &tau, uint64_t thr_key,
                                           uint64_t key, int lane) {
    uint64_t pending = __ballot(key < tau);
    if (!pending) return;
    if (cnt < k) {
        const int room = k - cnt, n = __popcll(pending);
        const int at = __popcll(pending & ((1ull << lane) - 1ull));
        if (key < tau && at < room) L[cnt + at] = key;
        if (n < room) {
            cnt += n;
            return;
        }
        cnt = k;
        tau = wave_swap_largest(L, k, HVS_KEY_EMPTY, HVS_KEY_EMPTY, lane);
        for (int i = 0; i < room; i++) pending &= pending - 1;  // those went in
        pending &= __ballot(key < tau);
    }
    while (pending) {
        const int src = __ffsll((unsigned long long)pending) - 1;
        pending &= pending - 1;
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)key, src);
        const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(key >> 32), src);
        const uint64_t c = ((uint64_t)hi << 32) | lo;
        if (c < tau) tau = wave_swap_largest(L, k, tau, c, lane);
    }
    (void)thr_key;
}

// Put the cnt <= 64 KL keys of a list in ascending order, in place: a key's place is the number of keys below it, counted
// against every key of the list (one LDS broadcast read each).
template <int KL> __device__ __forceinline__ void wave_sort_r(volatile uint64_t *L, int cnt, int lane) {
    wave_fence();
    const uint64_t *Lc = const_cast<const uint64_t *>(L);
    uint64_t mine[KL];
    int rank[KL];
#pragma unroll
    for (int e = 0; e < KL; e++) {
        mine[e] = e * 64 + lane < cnt ? Lc[e * 64 + lane] : HVS_KEY_EMPTY;
        rank[e] = 0;
    }
#pragma unroll 4
    for (int j = 0; j < cnt; j++) {
        const uint64_t o = Lc[j];
#pragma unroll
        for (int e = 0; e < KL; e++) rank[e] += o < mine[e] ? 1 : 0;
    }
    wave_fence();  // every lane has read what it needs
#pragma unroll
    for (int e = 0; e < KL; e++)
        if (e * 64 + lane < cnt) L[rank[e]] = mine[e];
    wave_fence();
}
__device__ __forceinline__ void wave_sort(volatile uint64_t *L, int cnt, int lane) {
    if (cnt <= 1) return;
    if (cnt <= 64) wave_sort_r<1>(L, cnt, lane);
    else if (cnt <= 128) wave_sort_r<2>(L, cnt, lane);
    else if (cnt <= 256) wave_sort_r<4>(L, cnt, lane);
    else if (cnt <= 512) wave_sort_r<8>(L, cnt, lane);
    else wave_sort_r<16>(L, cnt, lane);
}

// ---- dot-product reduction across the lanes of a row group ----------------------------------------
// Each lane holds V partial sums (V a power of two, V <= W) that must be added up over the W lanes of its
// group.  V independent butterflies would cost V log2(W) shuffles; instead the first log2(V) steps halve
// the number of sums a lane carries (it keeps one half and hands the other to its partner), the remaining
// steps are plain butterflies on the one sum left: V - 1 + log2(W / V) shuffles.  Afterwards the total of
// sum j sits in a[0] of every lane whose index inside the group, shifted right by log2(W / V), equals j.
// STRIDE: distance in lanes between neighbours of a group (l = lane / STRIDE % W).
template <int V, int W, int STRIDE = 1> __device__ __forceinline__ void reduce_scatter(float (&a)[V], int l) {
    int n = V;
#pragma unroll
    for (int o = W / 2; o >= 1; o >>= 1) {
        if (n > 1) {
            const int half = n / 2;
            const bool upper = (l & o) != 0;
#pragma unroll
            for (int i = 0; i < V / 2; i++) {
                if (i < half) {
                    const float send = upper ? a[i] : a[i + half];
                    const float keep = upper ? a[i + half] : a[i];
                    a[i] = keep + __shfl_xor(send, o * STRIDE);
                }
            }
            n = half;
        } else {
            a[0] += __shfl_xor(a[0], o * STRIDE);
        }
    }
}
constexpr int hvs_pow2_ceil(int v) { return v <= 1 ? 1 : 2 * hvs_pow2_ceil((v + 1) / 2); }
constexpr int hvs_log2(int v) { return v <= 1 ? 0 : 1 + hvs_log2(v / 2); }

// ---- list merging -------------------------------------------------------------------------------
// A wave holds one ascending list of N = 64 * KL keys in registers, slot i = e * 64 + lane in a[e].
// Merging another ascending list B keeps the N smallest of the union: c[i] = min(a[i], B[N-1-i]) is a
// bitonic sequence holding exactly those keys, and log2(N) compare-exchange rounds sort it (rounds with
// stride >= 64 pair registers of one lane, rounds with stride < 64 pair lanes through a shuffle).
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
    const uint32_t lo = __shfl_xor((uint32_t)v, mask);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), mask);
    return ((uint64_t)hi << 32) | lo;
}

template <int KL>
__device__ __forceinline__ void wl_load(uint64_t (&a)[KL], const uint64_t *src, int n, int lane) {
#pragma unroll
    for (int e = 0; e < KL; e++) {
        const int i = e * 64 + lane;
        a[e] = i < n ? src[i] : HVS_KEY_EMPTY;
    }
}

// the other list, reversed, as the merge below consumes it
template <int KL>
__device__ __forceinline__ void wl_load_reversed(uint64_t (&b)[KL], const uint64_t *other, int n, int lane) {
    constexpr int N = 64 * KL;
#pragma unroll
    for (int e = 0; e < KL; e++) {
        const int i = N - 1 - (e * 64 + lane);
        b[e] = i < n ? other[i] : HVS_KEY_EMPTY;
    }
}

// a = the 64 KL smallest of a (ascending) and b (a list loaded by wl_load_reversed), ascending
template <int KL>
__device__ __forceinline__ void wl_merge(uint64_t (&a)[KL], const uint64_t (&b)[KL], int lane) {
#pragma unroll
    for (int e = 0; e < KL; e++) a[e] = b[e] < a[e] ? b[e] : a[e];
#pragma unroll
    for (int se = KL / 2; se >= 1; se >>= 1) {  // strides 64 * se: register pairs inside a lane
#pragma unroll
        for (int e = 0; e < KL; e++) {
            if ((e & se) == 0) {
                const uint64_t lo = a[e] < a[e + se] ? a[e] : a[e + se];
                const uint64_t hi = a[e] < a[e + se] ? a[e + se] : a[e];
                a[e] = lo;
                a[e + se] = hi;
            }
        }
    }
#pragma unroll
    for (int stride = 32; stride >= 1; stride >>= 1) {
        const bool keep_min = (lane & stride) == 0;
#pragma unroll
        for (int e = 0; e < KL; e++) {
            const uint64_t o = shfl_xor_u64(a[e], stride);
            const uint64_t lo = a[e] < o ? a[e] : o;
            const uint64_t hi = a[e] < o ? o : a[e];
            a[e] = keep_min ? lo : hi;
        }
    }
}

#define MERGE_WAVES 4
#define MERGE_BLOCK_FAN (MERGE_WAVES * MERGE_FAN)

// A block of MERGE_WAVES waves merges nl <= MERGE_BLOCK_FAN consecutive ascending lists of k keys: wave w lists
// [w MERGE_FAN, (w + 1) MERGE_FAN) one after the other, then the first wave the four results, handed over in LDS
// (handed: (MERGE_WAVES - 1) x 64 KL keys).  The result is in the first wave's a[].  All threads of the block call it.
template <int KL>
__device__ __forceinline__ void block_merge(uint64_t (&a)[KL], const uint64_t *lists, int nl_block, int k, uint64_t *handed, int wave, int lane) {
    constexpr int N = 64 * KL;
    const int first = wave * MERGE_FAN;
    const int nl = nl_block - first < MERGE_FAN ? (nl_block - first > 0 ? nl_block - first : 0) : MERGE_FAN;
    const uint64_t *src = lists + (size_t)first * k;
#pragma unroll
    for (int e = 0; e < KL; e++) {
        const int i = e * 64 + lane;
        a[e] = (nl > 0 && i < k) ? src[i] : HVS_KEY_EMPTY;
    }
    // the lists' addresses do not depend on the merges: fetch them a group ahead, so that a wave waits for global
    // memory once per group instead of once per list
    constexpr int AHEAD = KL <= 2 ? 8 : (KL <= 4 ? 4 : 2);
    for (int j0 = 1; j0 < nl; j0 += AHEAD) {
        uint64_t b[AHEAD][KL];
#pragma unroll
        for (int u = 0; u < AHEAD; u++) {
            const uint64_t *other = src + (size_t)(j0 + u < nl ? j0 + u : j0) * k;
#pragma unroll
            for (int e = 0; e < KL; e++) {
                const int i = N - 1 - (e * 64 + lane);  // reversed, as wl_merge consumes it
                b[u][e] = (j0 + u < nl && i < k) ? other[i] : HVS_KEY_EMPTY;
            }
        }
#pragma unroll
        for (int u = 0; u < AHEAD; u++)
            if (j0 + u < nl) wl_merge<KL>(a, b[u], lane);
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < KL; e++) handed[(wave - 1) * N + e * 64 + lane] = a[e];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 1; w < MERGE_WAVES; w++) {
        if (w * MERGE_FAN >= nl_block) break;  // that wave had no list
        uint64_t b[KL];
        wl_load_reversed<KL>(b, handed + (w - 1) * N, N, lane);
        wl_merge<KL>(a, b, lane);
    }
}

template <int KL> __device__ __forceinline__ void merged_to_keys(const uint64_t (&a)[KL], uint64_t *o, int k, int lane) {
#pragma unroll
    for (int e = 0; e < KL; e++) {
        const int i = e * 64 + lane;
        if (i < k) o[i] = a[e];
    }
}

// the final list as distances, ids (ids == nullptr: row positions) and the count of one query
template <int KL>
__device__ __forceinline__ void merged_to_results(const uint64_t (&a)[KL], int k, const int64_t *ids, float *out_dist, int64_t *out_ids,
                                                  int32_t *out_count, int lane) {
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < KL; e++) {
        const int i = e * 64 + lane;
        const bool have = i < k && a[e] != HVS_KEY_EMPTY;
        cnt += __popcll(__ballot(have));
        if (i < k) {
            float d = __builtin_inff();
            int64_t id = -1;
            if (have) {
                d = hvs_unord((uint32_t)(a[e] >> 32));
                const uint32_t row = (uint32_t)a[e];
                id = ids ? ids[row] : (int64_t)row;
            }
            out_dist[i] = d;
            out_ids[i] = id;
        }
    }
    if (lane == 0) *out_count = cnt;
}

// ---- the scan -----------------------------------------------------------------------------------
// LPR = lanes per row: 64 (one row per wave pass) or 32 (two rows per pass, for rows of <= 2 KB, which
// would otherwise leave half the lanes idle and pay the reduction and selection once per 1.5 KB), or 16
// for the 4x4x4 MFMA mode of 16-bit rows with up to four embeddings (NQ = 4): four rows per pass, lane
// 4 b + j works on row j and on the b-th 16-byte chunk of every 256 bytes.
template <int DT, int NQ, int LPR, bool DISTONLY>
__global__ __launch_bounds__(SCAN_THREADS, NQ <= 4 ? 4 : 2) void scan_kernel(ScanParams p) {  // <= 128 VGPRs up to four embeddings: 4 waves per SIMD
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [NQ][ld] query elements in the row's element type, then per-wave lists [SCAN_WAVES][nq_out][k] keys
    constexpr int ESZ = Chunk<DT>::ESZ;
    char *q_lds = smem;
    const int q_bytes = NQ * p.ld * ESZ;
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem + (((size_t)q_bytes + 15) & ~(size_t)15));
    uint32_t *row_queue = reinterpret_cast<uint32_t *>(lists + (size_t)SCAN_WAVES * p.nq_out * p.k);  // [SCAN_WAVES][64 * RG], then NQ inverse norms

    constexpr bool M4 = LPR == 16;
    static_assert(!M4 || (NQ == 4 && DT != HVS_F32 && !DISTONLY), "the 4x4x4 MFMA mode takes four 16-bit embeddings");
    constexpr int R = M4 ? 4 : scan_rows_in_flight(NQ, Chunk<DT>::ESZ);
    constexpr int SUB = 64 / LPR;     // rows per wave pass
    constexpr int RG = R * SUB;  // rows per wave iteration
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int sub = M4 ? (lane & 3) : lane / LPR;   // row of the pass this lane works on
    const int l = M4 ? (lane >> 2) : lane % LPR;    // its position among the LPR lanes of that row
    const int nq_total = p.nq_out * p.fold;

    float *qinv_lds = reinterpret_cast<float *>(row_queue + SCAN_WAVES * 64 * RG);  // [NQ], behind the row queues
    if (p.raw) {
        // the embeddings as they came: a wave per embedding rounds them to the rows' type and sums the squares of the rounded
        // values -- the lanes' partial sums in the order of prep_queries_kernel, so that both give the same bits
        for (int q = wave; q < NQ; q += SCAN_WAVES) {
            float s = 0.0f;
            for (int c = lane; c < p.ld; c += 64) {
                float v = 0.0f;
                if (q < nq_total && c < p.raw_dim) {
                    v = p.raw[(size_t)q * p.raw_dim + c];
                    if (DT == HVS_F16) v = __half2float(__float2half_rn(v));
                    else if (DT == HVS_BF16) v = hvs_bf16_to_float(hvs_float_to_bf16(v));
                }
                Chunk<DT>::put_query(q_lds, q * p.ld + c, v);
                s = fmaf(v, v, s);
            }
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) qinv_lds[q] = q < nq_total ? (float)(1.0 / sqrt((double)s)) : 0.0f;
        }
    } else {
        for (int i = tid; i < NQ * p.ld; i += SCAN_THREADS) {
            const int q = i / p.ld;
            Chunk<DT>::put_query(q_lds, i, q < nq_total ? p.queries[i] : 0.0f);
        }
        if (tid < NQ) qinv_lds[tid] = tid < nq_total ? p.q_inv[tid] : 0.0f;
    }
    __syncthreads();

    float qinv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) qinv[q] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qinv_lds[q])));  // wave-uniform: scalar registers

    volatile uint64_t *myL = lists + (size_t)wave * p.nq_out * p.k;
    const uint64_t thr_key = (uint64_t)p.thr_ord << 32;
    uint64_t tau[NQ];
    int cnt[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { tau[q] = thr_key; cnt[q] = 0; }

    const u32x4 *rows16 = reinterpret_cast<const u32x4 *>(p.rows);
    const int64_t gwave = (int64_t)blockIdx.x * SCAN_WAVES + wave;
    const int64_t nwaves = (int64_t)gridDim.x * SCAN_WAVES;
    volatile uint32_t *rowq = row_queue + wave * (64 * RG);

    // Up to two row ranges per launch, each walked in the way that suits it; the wave's lists carry over.
    // Without a filter this wave owns row groups gwave, gwave + nwaves, ... (RG rows each) of a range.  With a filter
    // (group code, row mask, tombstones) it owns whole spans of 64 * RG consecutive rows instead: lane i
    // evaluates the filter of rows i, i + 64, ... of the span (consecutive lanes read consecutive filter
    // words: one cache line per load instruction, not one per lane), a wave prefix sum packs the surviving
    // row numbers into a small LDS queue, and only those rows are streamed, RG at a time.  (Spans suit filters that
    // thin the rows out; a range whose rows all pass -- a group's range of a table in group order -- is better off
    // without its test: a span of 64 * RG live rows is one wave's work.)
#pragma unroll 1
    for (int seg = 0; seg < 2; seg++) {
    const int64_t seg_begin = seg ? p.row_begin2 : p.row_begin, seg_end = seg ? p.row_end2 : p.row_end;
    if (seg_end <= seg_begin) continue;
    const int32_t seg_group = seg ? p.group_filter2 : p.group_filter;
    const int64_t n_groups = (seg_end - seg_begin + RG - 1) / RG;
    const bool filtered = !DISTONLY && (seg_group >= 0 || p.mask != nullptr || p.deleted != nullptr);
    const int64_t my_groups = n_groups > gwave ? (n_groups - gwave + nwaves - 1) / nwaves : 0;
    // rows of a span per lane: RG on a range long enough to give every wave a span of 64 RG rows, fewer on a shorter one
    // (down to spans of 64 rows), so that the waves share a short range instead of a few of them walking it alone
    int per_lane = (int)((seg_end - seg_begin + 64 * nwaves - 1) / (64 * nwaves));
    per_lane = per_lane < 1 ? 1 : (per_lane > RG ? RG : per_lane);
    const int64_t span_rows = 64 * (int64_t)per_lane;
    const int64_t n_spans = (seg_end - seg_begin + span_rows - 1) / span_rows;
    const int64_t n_outer = filtered ? (n_spans > gwave ? (n_spans - gwave + nwaves - 1) / nwaves : 0) : (my_groups + 63) / 64;
    for (int64_t ob = 0; ob < n_outer; ob++) {
        const int64_t base = ob * 64;
        int n_items;
        int total = 0;
        if (filtered) {
            uint32_t mybits = 0;
            const int64_t row0 = seg_begin + (gwave + ob * nwaves) * span_rows + lane;
#pragma unroll
            for (int i = 0; i < RG; i++) {
                const int64_t row = row0 + i * 64;
                bool ok = i < per_lane && row < seg_end;
                if (ok && seg_group >= 0) ok = p.groups[row] == seg_group;
                if (ok && p.mask) ok = (p.mask[row >> 5] >> (row & 31)) & 1u;
                if (ok && p.deleted) ok = !((p.deleted[row >> 5] >> (row & 31)) & 1u);
                mybits |= (ok ? 1u : 0u) << i;
            }
            // exclusive prefix sum of the live counts over the lanes
            const int mine = __popc(mybits);
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            total = __shfl(incl, 63);
            int pos = incl - mine;
            wave_fence();  // earlier reads of the queue are done
#pragma unroll
            for (int i = 0; i < RG; i++)
                if ((mybits >> i) & 1u) rowq[pos++] = (uint32_t)(row0 + i * 64);
            wave_fence();
            n_items = (total + RG - 1) / RG;
        } else {
            const int64_t left = my_groups - base;
            n_items = left >= 64 ? 64 : (int)left;
        }
      for (int it = 0; it < n_items; it++) {
        int64_t rowv[R];
        bool live[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (filtered) {
                const int qi = it * RG + r * SUB + sub;
                live[r] = qi < total;
                rowv[r] = live[r] ? (int64_t)rowq[qi] : 0;
            } else {
                rowv[r] = seg_begin + (gwave + (base + it) * nwaves) * RG + r * SUB + sub;
                live[r] = rowv[r] < seg_end;
            }
        }
        float acc[R][NQ];
        if (M4) {
            if constexpr (M4) {
                f32x4v part4[R];  // [embedding] partial sums of this lane's row over its chunks
#pragma unroll
                for (int r = 0; r < R; r++) part4[r] = (f32x4v){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 2
                for (int c = l; c < p.chunks; c += LPR) {
                    u32x4 v[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        v[r] = (u32x4){0u, 0u, 0u, 0u};
                        if (live[r]) v[r] = __builtin_nontemporal_load(rows16 + rowv[r] * p.chunks + c);
                    }
                    // this lane supplies embedding `sub` (rows of A) and row `sub` (columns of B)
                    const u32x4 qv = *reinterpret_cast<const u32x4 *>(q_lds + ((size_t)sub * p.chunks + c) * 16);
#pragma unroll
                    for (int r = 0; r < R; r++) part4[r] = mfma4<DT>(qv, v[r], part4[r]);
                }
                // sum over the 16 lanes of the row (lane stride 4), then hand every lane of the row all four totals
#pragma unroll
                for (int r = 0; r < R; r++) {
                    float a[4] = {part4[r][0], part4[r][1], part4[r][2], part4[r][3]};
                    reduce_scatter<4, 16, 4>(a, l);
#pragma unroll
                    for (int q = 0; q < NQ; q++) acc[r][q] = __shfl(a[0], (q << 4) | sub);
                }
            }
        } else {
            typename Chunk<DT>::Acc part[R][NQ];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int q = 0; q < NQ; q++) part[r][q] = Chunk<DT>::zero();

            // 16-bit rows, one embedding: the loads of the next chunk step go out before the products of this one are
            // formed, so that a row pass keeps loads in flight throughout (a 300 MB range: 0.073 -> 0.067 ms).  The other
            // kernels have no registers left for it under the four-waves-per-SIMD bound (fp32 rows with three embeddings
            // lose 5 points of bandwidth to the spills) and fetch two chunk steps, then use them.
            constexpr bool PIPELINED = DT != HVS_F32 && NQ == 1;
            if constexpr (PIPELINED) {
                auto fetch = [&](u32x4 (&v)[R], int c) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        v[r] = (u32x4){0u, 0u, 0u, 0u};
                        if (live[r] && c < p.chunks) v[r] = __builtin_nontemporal_load(rows16 + rowv[r] * p.chunks + c);
                    }
                };
                u32x4 v[R];
                fetch(v, l);
#pragma unroll 2
                for (int c = l; c < p.chunks; c += LPR) {
                    u32x4 w[R];
                    fetch(w, c + LPR);
                    const u32x4 qv = *reinterpret_cast<const u32x4 *>(q_lds + (size_t)c * 16);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        part[r][0] = Chunk<DT>::dot(v[r], qv, part[r][0]);
                        v[r] = w[r];
                    }
                }
            } else {
#pragma unroll 2
                for (int c = l; c < p.chunks; c += LPR) {
                    u32x4 v[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        v[r] = (u32x4){0u, 0u, 0u, 0u};
                        if (live[r]) v[r] = __builtin_nontemporal_load(rows16 + rowv[r] * p.chunks + c);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const u32x4 qv = *reinterpret_cast<const u32x4 *>(q_lds + ((size_t)q * p.chunks + c) * 16);
#pragma unroll
                        for (int r = 0; r < R; r++) part[r][q] = Chunk<DT>::dot(v[r], qv, part[r][q]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int q = 0; q < NQ; q++) acc[r][q] = Chunk<DT>::finish(part[r][q]);
        }
        if (NQ == 1) {
            // One query: the tail below costs as much as the row stream hides, so it is kept short.
            // add up the R * NQ partial sums over the LPR lanes of the row group, then hand every lane of
            // the group all the totals
            constexpr int V = hvs_pow2_ceil(R * NQ);
            static_assert(V <= LPR, "more sums per lane than lanes per row group");
            {
                float a[V];
    #pragma unroll
                for (int i = 0; i < V; i++) a[i] = i < R * NQ ? acc[i / NQ][i % NQ] : 0.0f;
                reduce_scatter<V, LPR>(a, l);
                constexpr int SH = hvs_log2(LPR / V);
    #pragma unroll
                for (int r = 0; r < R; r++)
    #pragma unroll
                    for (int q = 0; q < NQ; q++) acc[r][q] = __shfl(a[0], sub * LPR + ((r * NQ + q) << SH));
            }
            // distances (in place of the dot products): every lane computes them for all R rows of its group
    #pragma unroll
            for (int r = 0; r < R; r++)
    #pragma unroll
                for (int t = 0; t < NQ; t++) acc[r][t] = hvs_cos_dist(acc[r][t], live[r] ? p.inv_norm[rowv[r]] : 0.0f, qinv[t]);
            // lane r of the group offers row r, so that one pass over the wave's list per query takes all RG rows
            // of this iteration
    #pragma unroll
            for (int qo = 0; qo < NQ; qo++) {
                if (qo >= p.nq_out) break;
                float d = __builtin_nanf("");  // this lane's row (r = l), LEAST over the query's embeddings
                int64_t row = 0;
    #pragma unroll
                for (int r = 0; r < R; r++) {
                    float dr = __builtin_nanf("");
    #pragma unroll
                    for (int t = 0; t < NQ; t++) {
                        const int j = t - qo * p.fold;  // wave-uniform
                        if (j >= 0 && j < p.fold) dr = fminf(dr, acc[r][t]);  // fminf drops a NaN operand: LEAST semantics
                    }
                    if (DISTONLY) {
                        if (l == 0 && live[r]) p.dist_out[rowv[r] - p.row_begin] = dr;
                    } else if (l == r && live[r]) {
                        d = dr;
                        row = rowv[r];
                    }
                }
                if (!DISTONLY) {
                    // NaN distances and filtered rows offer nothing
                    const uint64_t key = d == d ? hvs_make_key(hvs_ord(d), (uint32_t)row) : HVS_KEY_EMPTY;
                    wave_offer(myL + (size_t)qo * p.k, cnt[qo], p.k, tau[qo], thr_key, key, lane);
                }
            }
        } else {
            // Several embeddings: rows are finished one at a time, which keeps few values alive (registers
            // decide the occupancy here).
            // butterfly inside each group of LPR lanes: every lane ends with its row's full dot products
            if (!M4) {
    #pragma unroll
                for (int r = 0; r < R; r++)
    #pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        float s = acc[r][q];
    #pragma unroll
                        for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
                        acc[r][q] = s;
                    }
            }
    #pragma unroll
            for (int r = 0; r < R; r++) {
                const int64_t row = rowv[r];
                const float inv = live[r] ? p.inv_norm[row] : 0.0f;
                float dd[NQ];
    #pragma unroll
                for (int t = 0; t < NQ; t++) dd[t] = hvs_cos_dist(acc[r][t], inv, qinv[t]);
    #pragma unroll
                for (int qo = 0; qo < NQ; qo++) {
                    if (qo >= p.nq_out) break;
                    float d = __builtin_nanf("");
    #pragma unroll
                    for (int t = 0; t < NQ; t++) {
                        const int j = t - qo * p.fold;  // wave-uniform
                        if (j >= 0 && j < p.fold) d = fminf(d, dd[t]);  // fminf drops a NaN operand: LEAST semantics
                    }
                    if (DISTONLY) {
                        if (l == 0 && live[r]) p.dist_out[row - p.row_begin] = d;
                    } else {
                        // one lane per row offers the row's key; NaN distances and filtered rows offer nothing
                        const uint64_t key = (l == 0 && live[r] && d == d) ? hvs_make_key(hvs_ord(d), (uint32_t)row) : HVS_KEY_EMPTY;
                        wave_offer(myL + (size_t)qo * p.k, cnt[qo], p.k, tau[qo], thr_key, key, lane);
                    }
                }
            }
        }
      }
    }
    }
    if (DISTONLY) return;

    // publish: one ascending list of k keys per block and query (padded); the merge kernels do the rest.  The block's
    // four wave lists are merged where they lie, in LDS: a key's place in the union is its index in its own list plus
    // the number of keys before it in each of the others (keys carry the row number, so no two are equal).
#pragma unroll
    for (int qo = 0; qo < NQ; qo++)
        if (qo < p.nq_out) wave_sort(myL + (size_t)qo * p.k, cnt[qo], lane);
    int *counts = reinterpret_cast<int *>(row_queue + wave * (64 * RG));  // the wave's own queue is free now
    if (lane < NQ) {
        int c = 0;
#pragma unroll
        for (int qo = 0; qo < NQ; qo++) c = lane == qo ? cnt[qo] : c;
        counts[lane] = c;
    }
    __syncthreads();
    for (int qo = 0; qo < p.nq_out; qo++) {
        uint64_t *out = p.partial + ((size_t)qo * gridDim.x + blockIdx.x) * p.k;
        int cw[SCAN_WAVES], total = 0;
#pragma unroll
        for (int w = 0; w < SCAN_WAVES; w++) {
            cw[w] = reinterpret_cast<const int *>(row_queue + w * (64 * RG))[qo];
            total += cw[w];
        }
#pragma unroll
        for (int w = 0; w < SCAN_WAVES; w++) {
            const uint64_t *mine = lists + ((size_t)w * p.nq_out + qo) * p.k;
            for (int i = tid; i < cw[w]; i += SCAN_THREADS) {
                const uint64_t key = mine[i];
                int rank = i;
#pragma unroll
                for (int o = 0; o < SCAN_WAVES; o++) {
                    if (o == w) continue;
                    const uint64_t *other = lists + ((size_t)o * p.nq_out + qo) * p.k;
                    int lo = 0, hi = cw[o];
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (other[mid] < key) lo = mid + 1;
                        else hi = mid;
                    }
                    rank += lo;
                }
                if (rank < p.k) out[rank] = key;
            }
        }
        for (int i = total + tid; i < p.k; i += SCAN_THREADS) out[i] = HVS_KEY_EMPTY;
    }
}

// ---- list merging: the kernel ------------------------------------------------------------------
// in: [nq, P, k] ascending keys padded with HVS_KEY_EMPTY.  One block merges MERGE_BLOCK_FAN consecutive lists: each of
// its four waves MERGE_FAN of them, one after the other, then the first wave the four results, handed over in LDS.
// A level costs a launch and a round trip to memory whatever it merges, so the levels are few and wide: the 1024
// block lists of a scan take two launches.
// FINAL = false: out keys [nq, P2, k].  FINAL = true (P2 == 1): distances / ids / counts.
template <int KL, bool FINAL>
__global__ __launch_bounds__(64 * MERGE_WAVES) void merge_lists_kernel(const uint64_t *__restrict__ in, int P, int k, uint64_t *out_keys,
                                                                        const int64_t *__restrict__ ids, float *out_dist,
                                                                        int64_t *out_ids, int32_t *out_count) {
    __shared__ uint64_t handed[(MERGE_WAVES - 1) * 64 * KL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = blockIdx.x;
    const int q = blockIdx.y;
    const int P2 = gridDim.x;
    const int first = grp * MERGE_BLOCK_FAN;
    uint64_t a[KL];
    block_merge<KL>(a, in + ((size_t)q * P + first) * k, P - first < MERGE_BLOCK_FAN ? P - first : MERGE_BLOCK_FAN, k, handed, wave, lane);
    if (wave > 0) return;
    if (!FINAL) merged_to_keys<KL>(a, out_keys + ((size_t)q * P2 + grp) * k, k, lane);
    else merged_to_results<KL>(a, k, ids, out_dist + (size_t)q * k, out_ids + (size_t)q * k, out_count + q, lane);
}

template <int KL>
static int reduce_lists_kl(hvs_index *idx, uint64_t *bufA, uint64_t *bufB, int nq, int P, int k, float *d_dist,
                           int64_t *d_ids, int32_t *d_count, bool positions) {
    uint64_t *in = bufA, *out = bufB;
    while (P > MERGE_BLOCK_FAN) {
        const int P2 = (P + MERGE_BLOCK_FAN - 1) / MERGE_BLOCK_FAN;
        merge_lists_kernel<KL, false><<<dim3(P2, nq), 64 * MERGE_WAVES, 0, idx->stream>>>(in, P, k, out, nullptr, nullptr, nullptr, nullptr);
        HVS_HIP(hipGetLastError());
        uint64_t *t = in; in = out; out = t;
        P = P2;
    }
    merge_lists_kernel<KL, true><<<dim3(1, nq), 64 * MERGE_WAVES, 0, idx->stream>>>(in, P, k, nullptr, (positions || idx->emit_positions) ? nullptr : idx->ids,
                                                                                   d_dist, d_ids, d_count);
    HVS_HIP(hipGetLastError());
    return 0;
}

// reduce [nq, P, k] key lists (in bufA) to final outputs; bufB is scratch of at least nq * ceil(P / MERGE_FAN) * k keys.
// positions: d_ids receives row positions instead of external ids
int hvs_reduce_lists(hvs_index *idx, uint64_t *bufA, uint64_t *bufB, int nq, int P, int k, float *d_dist,
                     int64_t *d_ids, int32_t *d_count, bool positions) {
    if (k <= 64) return reduce_lists_kl<1>(idx, bufA, bufB, nq, P, k, d_dist, d_ids, d_count, positions);
    if (k <= 128) return reduce_lists_kl<2>(idx, bufA, bufB, nq, P, k, d_dist, d_ids, d_count, positions);
    if (k <= 256) return reduce_lists_kl<4>(idx, bufA, bufB, nq, P, k, d_dist, d_ids, d_count, positions);
    if (k <= 512) return reduce_lists_kl<8>(idx, bufA, bufB, nq, P, k, d_dist, d_ids, d_count, positions);
    return reduce_lists_kl<16>(idx, bufA, bufB, nq, P, k, d_dist, d_ids, d_count, positions);
}

// ---- query preparation --------------------------------------------------------------------------
// dst[q][c] = query value rounded to the column type (binding a parameter with the column's type), pad 0;
// q_inv[q] = 1/|dst[q]|.  One wave per query.
__global__ void prep_queries_kernel(const float *__restrict__ src, float *__restrict__ dst, float *__restrict__ q_inv,
                                    int nq, int dim, int ld, int dtype) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    float s = 0.0f;
    for (int c = lane; c < ld; c += 64) {
        float v = 0.0f;
        if (c < dim) {
            v = src[(size_t)q * dim + c];
            if (dtype == HVS_F16) v = __half2float(__float2half_rn(v));
            else if (dtype == HVS_BF16) v = hvs_bf16_to_float(hvs_float_to_bf16(v));
        }
        dst[(size_t)q * ld + c] = v;
        s = fmaf(v, v, s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) q_inv[q] = (float)(1.0 / sqrt((double)s));
}

// The embeddings of a call on the device, and whether they have been prepared yet (rounded to the rows' type with their
// inverse norms: prep_queries_kernel).  A search that ends on the scan never launches the preparation: scan_kernel does
// the same arithmetic in its prologue (ScanParams::raw), one launch less per call -- about 5 us of the 50 us that a
// search of a small table takes.  Every other consumer calls queries_prepare first.
struct PendingQueries {
    const float *raw;  // [nq_total, dim] fp32 on the device
    float *d_q;        // [nq_total, ld] prepared, valid once done
    float *d_qinv;     // [nq_total]
    int nq_total;
    bool done;
};

static int queries_stage(hvs_index *idx, const float *queries, int nq_total, int on_device, PendingQueries *pq) {
    const size_t qbytes = (size_t)nq_total * idx->ld * 4;
    const size_t inv_off = (qbytes + 255) & ~(size_t)255;
    const size_t raw_off = (inv_off + (size_t)nq_total * 4 + 255) & ~(size_t)255;
    const size_t raw_bytes = on_device ? 0 : (size_t)nq_total * idx->dim * 4;
    void *ws = nullptr;
    HVS_TRY(hvs_ws_get(idx, 0, raw_off + raw_bytes, &ws));
    const float *src = queries;
    if (!on_device) {
        HVS_HIP(hipMemcpyAsync((char *)ws + raw_off, queries, raw_bytes, hipMemcpyHostToDevice, idx->stream));
        src = (const float *)((char *)ws + raw_off);
    }
    pq->raw = src;
    pq->d_q = (float *)ws;
    pq->d_qinv = (float *)((char *)ws + inv_off);
    pq->nq_total = nq_total;
    pq->done = false;
    return 0;
}

static int queries_prepare(hvs_index *idx, PendingQueries *pq) {
    if (!pq || pq->done) return 0;
    prep_queries_kernel<<<(pq->nq_total + 3) / 4, 256, 0, idx->stream>>>(pq->raw, pq->d_q, pq->d_qinv, pq->nq_total, idx->dim, idx->ld, idx->dtype);
    HVS_HIP(hipGetLastError());
    pq->done = true;
    return 0;
}

int hvs_prepare_queries(hvs_index *idx, const float *queries, int nq_total, int on_device, float **d_q, float **d_qinv) {
    PendingQueries pq;
    HVS_TRY(queries_stage(idx, queries, nq_total, on_device, &pq));
    HVS_TRY(queries_prepare(idx, &pq));
    *d_q = pq.d_q;
    *d_qinv = pq.d_qinv;
    return 0;
}

// ---- launch helpers -----------------------------------------------------------------------------
// *grid = 0 on entry: size the grid so that every block is resident at once (occupancy x CUs); waves
// stride over the rows, so a second, partial round of blocks would run at a fraction of the bandwidth.
template <int DT, int NQ, int LPR, bool DISTONLY>
static int launch_scan_one(int64_t want_blocks, size_t lds, hipStream_t s, const ScanParams &p, int *grid_out, bool dry) {
    const void *fn = reinterpret_cast<const void *>(&scan_kernel<DT, NQ, LPR, DISTONLY>);
    // resident blocks of this kernel at this LDS size on this device: asked of the runtime once (four runtime calls per
    // launch cost a single query on a small table more than its merges)
    int dev = 0;
    HVS_HIP(hipGetDevice(&dev));
    int64_t grid = 0;
    {
        static std::mutex mu;
        static std::unordered_map<uint64_t, int> known;
        static std::unordered_map<const void *, size_t> lds_allowed;  // a kernel's dynamic-LDS limit is only ever raised
        const uint64_t key = ((uint64_t)(uintptr_t)fn * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)lds << 8) ^ (uint64_t)dev;
        std::lock_guard<std::mutex> guard(mu);
        auto it = known.find(key);
        if (it == known.end()) {
            size_t &allowed = lds_allowed[fn];
            if (lds > 48 * 1024 && lds > allowed) {
                HVS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                allowed = lds;
            }
            int per_cu = 0, cus = 256;
            HVS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, SCAN_THREADS, lds));
            if (per_cu < 1) per_cu = 1;
            if (per_cu > 4) per_cu = 4;
            HVS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            it = known.emplace(key, per_cu * cus).first;
        }
        grid = it->second;
    }
    if (const char *e = getenv("HVS_SCAN_MAX_BLOCKS")) grid = atoll(e) > 0 && atoll(e) < grid ? atoll(e) : grid;  // experiments
    if (want_blocks < grid) grid = want_blocks < 1 ? 1 : want_blocks;
    *grid_out = (int)grid;
    if (dry) return 0;
    scan_kernel<DT, NQ, LPR, DISTONLY><<<(int)grid, SCAN_THREADS, lds, s>>>(p);
    HVS_HIP(hipGetLastError());
    return 0;
}

template <int DT, int LPR, bool DISTONLY>
static int launch_scan_nq(int NQ, int64_t want, size_t lds, hipStream_t s, const ScanParams &p, int *grid, bool dry) {
    switch (NQ) {
    case 1: return launch_scan_one<DT, 1, LPR, DISTONLY>(want, lds, s, p, grid, dry);
    case 2: return launch_scan_one<DT, 2, LPR, DISTONLY>(want, lds, s, p, grid, dry);
    case 4: return launch_scan_one<DT, 4, LPR, DISTONLY>(want, lds, s, p, grid, dry);
    case 8: return launch_scan_one<DT, 8, LPR, DISTONLY>(want, lds, s, p, grid, dry);
    default: hvs_set_error("scan: unsupported NQ %d", NQ); return 2;
    }
}

// lanes per row: two rows per wave pass when a row is at most 2 KB
static inline int scan_lpr(int chunks) { return chunks <= 128 ? 32 : 64; }

template <bool DISTONLY>
static int launch_scan(int dtype, int NQ, int lpr, int64_t want, size_t lds, hipStream_t s, const ScanParams &p, int *grid,
                       bool dry) {
    if (DISTONLY || lpr == 64) {
        if (dtype == HVS_F32) return launch_scan_nq<HVS_F32, 64, DISTONLY>(NQ, want, lds, s, p, grid, dry);
        if (dtype == HVS_F16) return launch_scan_nq<HVS_F16, 64, DISTONLY>(NQ, want, lds, s, p, grid, dry);
        return launch_scan_nq<HVS_BF16, 64, DISTONLY>(NQ, want, lds, s, p, grid, dry);
    }
    if (lpr == 16) {
        if (dtype == HVS_F16) return launch_scan_one<HVS_F16, 4, 16, false>(want, lds, s, p, grid, dry);
        return launch_scan_one<HVS_BF16, 4, 16, false>(want, lds, s, p, grid, dry);
    }
    if (dtype == HVS_F32) return launch_scan_nq<HVS_F32, 32, false>(NQ, want, lds, s, p, grid, dry);
    if (dtype == HVS_F16) return launch_scan_nq<HVS_F16, 32, false>(NQ, want, lds, s, p, grid, dry);
    return launch_scan_nq<HVS_BF16, 32, false>(NQ, want, lds, s, p, grid, dry);
}

static inline int round_nq(int n) { return n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : 8)); }

// rg = rows a wave takes per iteration (sizes the live-row queue of filtered scans)
static size_t scan_lds_bytes(int NQ, int ld, int esz, int nq_out, int k, int rg) {
    size_t b = ((size_t)NQ * ld * esz + 15) & ~(size_t)15;
    b += (size_t)SCAN_WAVES * nq_out * k * 8;
    b += (size_t)SCAN_WAVES * 64 * rg * 4;  // live-row queues
    b += 64;                                // the embeddings' inverse norms (up to 8)
    return b;
}

static const size_t LDS_LIMIT = 160 * 1024;

// Run the scan path for out-queries [0, nq) (already prepared on device).  Results to device buffers.
// lists / lists_bytes: where the wave lists and the merge levels go; nullptr = the index's growable workspaces 1 and 2.
// With a caller's buffer nothing is allocated: the out-queries per pass shrink until the lists fit, and a search whose
// lists do not fit even one query at a time is refused.
static int search_scan_ranges(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int32_t group_filter,
                              const uint32_t *d_mask, float thr, int64_t row_begin, int64_t row_end, float *d_dist,
                              int64_t *d_ids, int32_t *d_count, void *lists, size_t lists_bytes, int64_t row_begin2, int64_t row_end2,
                              int32_t group_filter2, const PendingQueries *pq = nullptr) {
    // out-queries per pass: as many as fit 8 dot products and the LDS budget -- or 4, for 16-bit rows with
    // two to four embeddings per query: those passes run the 4x4x4 MFMA mode, which stays bandwidth-bound
    const bool m4 = idx->dtype != HVS_F32 && fold <= 4 && nq * fold >= 2;
    int per_pass = (m4 ? 4 : 8) / fold;
    if (per_pass < 1) per_pass = 1;
    if (per_pass > nq) per_pass = nq;
    const int chunks = idx->ld * idx->elem_size / 16;
    const int lpr = m4 ? 16 : scan_lpr(chunks);
    auto rows_per_iter_of = [&](int nq_rounded) { return m4 ? 16 : scan_rows_in_flight(nq_rounded, idx->elem_size) * (64 / lpr); };
    while (per_pass > 1 && scan_lds_bytes(m4 ? 4 : round_nq(per_pass * fold), idx->ld, idx->elem_size, per_pass, k,
                                          rows_per_iter_of(round_nq(per_pass * fold))) > LDS_LIMIT / 2)
        per_pass--;
    const int NQ = m4 ? 4 : round_nq(per_pass * fold);
    const int rows_per_iter = rows_per_iter_of(NQ);
    const size_t lds = scan_lds_bytes(NQ, idx->ld, idx->elem_size, per_pass, k, rows_per_iter);
    HVS_REQUIRE(lds <= LDS_LIMIT, "hvs_search: dim %d with fold %d and k %d needs %zu bytes of LDS (limit %zu)", idx->dim, fold, k, lds, LDS_LIMIT);

    // [row_begin2, row_end2): a second range for the same launch, under the group test group_filter2
    const bool two_ranges = row_end2 > row_begin2;
    const int64_t n_rows = (row_end - row_begin) + (two_ranges ? row_end2 - row_begin2 : 0);
    const int64_t row_groups = (n_rows + rows_per_iter - 1) / rows_per_iter;
    const int64_t want_blocks = (row_groups + SCAN_WAVES - 1) / SCAN_WAVES;
    int grid = 0;
    {
        ScanParams dummy;
        memset(&dummy, 0, sizeof(dummy));
        HVS_TRY(launch_scan<false>(idx->dtype, NQ, lpr, want_blocks, lds, idx->stream, dummy, &grid, true));
    }
    const int P = grid;  // one list per block
    const int P2 = (P + MERGE_BLOCK_FAN - 1) / MERGE_BLOCK_FAN;
    void *bufA = nullptr, *bufB = nullptr;
    if (lists) {
        // per_pass only bounds how many of the kernel's NQ dot products a launch uses: the kernel and its LDS stay as sized
        auto need = [&](int pp) { return (((size_t)pp * P * k * 8 + 255) & ~(size_t)255) + (size_t)pp * P2 * k * 8; };
        while (per_pass > 1 && need(per_pass) > lists_bytes) per_pass--;
        HVS_REQUIRE(need(per_pass) <= lists_bytes,
                    "hvs_search_noalloc: k = %d needs %zu bytes for the %d block lists of one query and their first merge level, the "
                    "scratch array leaves %zu",
                    k, need(1), P, lists_bytes);
        bufA = lists;
        bufB = (char *)lists + (((size_t)per_pass * P * k * 8 + 255) & ~(size_t)255);
    } else {
        HVS_TRY(hvs_ws_get(idx, 1, (size_t)per_pass * P * k * 8, &bufA));
        HVS_TRY(hvs_ws_get(idx, 2, (size_t)per_pass * P2 * k * 8, &bufB));
    }

    ScanParams p;
    memset(&p, 0, sizeof(p));
    p.rows = idx->rows;
    p.inv_norm = idx->inv_norm;
    p.groups = idx->groups;
    p.mask = d_mask;
    p.deleted = idx->n_deleted > 0 ? idx->deleted : nullptr;
    p.row_begin = row_begin;
    p.row_end = row_end;
    p.row_begin2 = two_ranges ? row_begin2 : 0;
    p.row_end2 = two_ranges ? row_end2 : 0;
    p.ld = idx->ld;
    p.chunks = chunks;
    p.fold = fold;
    p.k = k;
    p.thr_ord = 0;
    {
        // ordered image of thr, computed like the device's hvs_ord
        uint32_t u;
        memcpy(&u, &thr, 4);
        p.thr_ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    p.group_filter = group_filter;
    p.group_filter2 = group_filter2;
    p.partial = (uint64_t *)bufA;
    p.dist_out = nullptr;

    for (int q0 = 0; q0 < nq; q0 += per_pass) {
        const int nqo = (nq - q0 < per_pass) ? nq - q0 : per_pass;
        p.queries = d_q + (size_t)q0 * fold * idx->ld;
        p.q_inv = d_qinv + (size_t)q0 * fold;
        p.raw = (pq && !pq->done) ? pq->raw + (size_t)q0 * fold * idx->dim : nullptr;  // not prepared: the kernel does it
        p.raw_dim = idx->dim;
        p.nq_out = nqo;
        HVS_TRY(hvs_prof_begin(idx));
        int g2 = 0;
        HVS_TRY(launch_scan<false>(idx->dtype, NQ, lpr, grid, lds, idx->stream, p, &g2, false));
        HVS_REQUIRE(g2 == grid, "scan: grid changed between sizing and launch");
        HVS_TRY(hvs_prof_end(idx, HVS_PROF_SCAN));
        HVS_TRY(hvs_prof_begin(idx));
        HVS_TRY(hvs_reduce_lists(idx, (uint64_t *)bufA, (uint64_t *)bufB, nqo, P, k, d_dist + (size_t)q0 * k,
                                 d_ids + (size_t)q0 * k, d_count + q0, false));
        HVS_TRY(hvs_prof_end(idx, HVS_PROF_MERGE));
    }
    return 0;
}

int hvs_search_scan(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int32_t group_filter,
                    const uint32_t *d_mask, float thr, int64_t row_begin, int64_t row_end, float *d_dist,
                    int64_t *d_ids, int32_t *d_count, void *lists, size_t lists_bytes) {
    return search_scan_ranges(idx, d_q, d_qinv, nq, fold, k, group_filter, d_mask, thr, row_begin, row_end, d_dist, d_ids, d_count, lists,
                              lists_bytes, 0, 0, -1);
}

int hvs_search_mfma(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int32_t group_filter,
                    const uint32_t *d_mask, float thr, int64_t row_begin, int64_t row_end, float *d_dist,
                    int64_t *d_ids, int32_t *d_count, int split_mode);
int hvs_search_rescore(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int32_t group_filter,
                       const uint32_t *d_mask, float thr, int64_t row_begin, int64_t row_end, float *d_dist, int64_t *d_ids,
                       int32_t *d_count, bool *used);
bool hvs_mfma_supported(hvs_index *idx, int nq, int fold, int k);

// ---- C ABI --------------------------------------------------------------------------------------
extern "C" int hvs_search(hvs_index *idx, const float *queries, int nq, int fold, int k, int32_t group_filter,
                          const uint32_t *row_mask, double max_distance, int64_t row_begin, int64_t row_end,
                          float *out_dist, int64_t *out_ids, int32_t *out_count, int on_device, int algo) {
    return hvs_search_core(idx, queries, nq, fold, k, group_filter, row_mask, on_device != 0, max_distance, row_begin, row_end, out_dist,
                           out_ids, out_count, on_device, algo);
}

extern "C" int hvs_search_filtered(hvs_index *idx, const float *queries, int nq, int fold, int k, const hvs_filter *filter,
                                   double max_distance, int64_t row_begin, int64_t row_end, float *out_dist, int64_t *out_ids,
                                   int32_t *out_count, int on_device, int algo) {
    HVS_REQUIRE(idx != nullptr, "hvs_search_filtered: idx is NULL");
    const uint32_t *d_mask = nullptr;
    if (filter && nq > 0) {
        HVS_HIP(hipSetDevice(idx->device));
        HVS_TRY(hvs_filter_mask(idx, filter, &d_mask, row_begin, row_end));
    }
    return hvs_search_core(idx, queries, nq, fold, k, filter ? filter->group : -1, d_mask, true, max_distance, row_begin, row_end, out_dist,
                           out_ids, out_count, on_device, algo);
}

// One search over rows [row_begin, row_end) with prepared queries and device outputs: picks the route (scan, matrix
// pipe, filter pass + exact rescoring) as the header describes.
static int search_range(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int32_t group_filter,
                        const uint32_t *d_mask, float thr, int64_t row_begin, int64_t row_end, float *d_dist, int64_t *d_ids, int32_t *d_count,
                        int algo, PendingQueries *pq) {
    const int nq_total = nq * fold;
    bool use_mfma = false;
    if (algo == HVS_ALGO_MFMA || algo == HVS_ALGO_MFMA_F32 || algo == HVS_ALGO_MFMA_SPLIT) {
        HVS_REQUIRE(hvs_mfma_supported(idx, nq, fold, k), "hvs_search: the MFMA path does not support this shape (dtype %d, dim %d, fold %d, k %d)", idx->dtype, idx->dim, fold, k);
        use_mfma = true;
    } else if (algo == HVS_ALGO_AUTO) {
        // Crossover measured on 1M x 1536: the scan takes 8 dot products per pass over the rows, the MFMA path
        // costs a whole 128-query tile whatever the count -- 3.3 ms for fp32 rows (two scan passes), but
        // only 0.65 ms for 16-bit rows, which is what the scan needs for four embeddings.
        // With the split copy of an fp32 table in place its batches run on the 16-bit pipe too (6.5 ms against 8.9 ms
        // for 8 queries on 4M x 1536).
        const bool f32_pipe = idx->dtype == HVS_F32 && !(idx->split && idx->split_n == idx->n && !getenv("HVS_NO_SPLIT"));
        int crossover = f32_pipe ? 16 : 5;  // 16-bit pipe: four embeddings are one scan pass (2.0 ms on 4M x 1536 f16), five are two (3.9 ms vs 2.2 ms)
        // A large fp32 table with a current split copy, small k, no WHERE: even one query is better off on the filter
        // route -- the ring kernel streams the hi slabs, half the bytes the scan has to read, and the candidates are
        // scored exactly afterwards (10M x 1536: 5.8 ms for up to 64 queries against 8.9 ms for one scan pass).  End to
        // end the routes cross at about 1.5 GB of fp32 rows (1536 columns: 0.35 / 0.37 ms at 262 144 rows, 0.48 / 0.61 ms
        // at 524 288, 0.74 / 1.05 ms at 1M); from 2 GB on the filter route is taken.
        int64_t filter_min_rows = ((int64_t)500 * 1000 * 1000 + idx->ld - 1) / idx->ld;
        if (const char *e = getenv("HVS_FILTER_MIN_ROWS")) filter_min_rows = atoll(e);
        if (idx->dtype == HVS_F32 && !f32_pipe && k <= 64 && group_filter < 0 && !d_mask && row_end - row_begin >= filter_min_rows &&
            !getenv("HVS_NO_FILTER"))
            crossover = 1;
        use_mfma = nq_total >= crossover && hvs_mfma_supported(idx, nq, fold, k);
    }
    if (row_end == row_begin) {
        // empty range: counts 0, +inf / -1 entries
        use_mfma = false;
    }
    if (use_mfma) {
        // fp32 rows: 0 = fp32 MFMA kernel, 1 = split copy (built if need be), 2 = split copy if it exists or from 16 embeddings on
        const int split_mode = algo == HVS_ALGO_MFMA_F32 ? 0 : (algo == HVS_ALGO_AUTO ? 2 : 1);
        // ... and where the split copy is in use, small k first tries one product instead of three: a filter pass over
        // the hi slabs alone, then exact fp32 scores of its candidates (MFMA_SPLIT keeps the three products)
        bool done = false;
        HVS_TRY(queries_prepare(idx, pq));
        if (idx->dtype == HVS_F32 && (algo == HVS_ALGO_MFMA || algo == HVS_ALGO_AUTO) &&
            (algo == HVS_ALGO_MFMA || nq_total >= 16 || (idx->split && idx->split_n == idx->n)))
            HVS_TRY(hvs_search_rescore(idx, d_q, d_qinv, nq, fold, k, group_filter, d_mask, thr, row_begin, row_end, d_dist, d_ids,
                                       d_count, &done));
        if (!done)
            HVS_TRY(hvs_search_mfma(idx, d_q, d_qinv, nq, fold, k, group_filter, d_mask, thr, row_begin, row_end, d_dist, d_ids, d_count,
                                    split_mode));
    } else
        HVS_TRY(search_scan_ranges(idx, d_q, d_qinv, nq, fold, k, group_filter, d_mask, thr, row_begin, row_end, d_dist, d_ids, d_count,
                                   nullptr, 0, 0, 0, -1, pq));

    return 0;
}

// Would search_range send this search to the scan?  (The cases in which a group filter is set: the filter route for
// single queries on large fp32 tables takes no WHERE.)
static bool filtered_search_takes_scan(hvs_index *idx, int nq, int fold, int k, int algo) {
    if (algo == HVS_ALGO_SCAN) return true;
    if (algo != HVS_ALGO_AUTO) return false;
    const bool f32_pipe = idx->dtype == HVS_F32 && !(idx->split && idx->split_n == idx->n && !getenv("HVS_NO_SPLIT"));
    return nq * fold < (f32_pipe ? 16 : 5) || !hvs_mfma_supported(idx, nq, fold, k);
}

// out = the k smallest of two ascending (distance, row position) lists per query, positions mapped to ids.  One block
// per query; an entry's place in the union is its own index plus the entries of the other list ordered before it
// (positions are distinct, so no two entries compare equal).
__global__ __launch_bounds__(256) void merge_two_kernel(const float *__restrict__ da, const int64_t *__restrict__ pa, const int32_t *__restrict__ ca,
                                                         const float *__restrict__ db, const int64_t *__restrict__ pb, const int32_t *__restrict__ cb,
                                                         int k, const int64_t *__restrict__ ids, float *__restrict__ od, int64_t *__restrict__ oi,
                                                         int32_t *__restrict__ oc) {
    const int q = blockIdx.x;
    const int na = ca[q], nb = cb[q];
    const float *A = da + (size_t)q * k, *B = db + (size_t)q * k;
    const int64_t *PA = pa + (size_t)q * k, *PB = pb + (size_t)q * k;
    for (int t = threadIdx.x; t < na + nb; t += blockDim.x) {
        const bool from_a = t < na;
        const int i = from_a ? t : t - na;
        const float d = from_a ? A[i] : B[i];
        const int64_t p = from_a ? PA[i] : PB[i];
        const float *O = from_a ? B : A;
        const int64_t *PO = from_a ? PB : PA;
        int lo = 0, hi = from_a ? nb : na;  // entries of the other list ordered before (d, p)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (O[mid] < d || (O[mid] == d && PO[mid] < p)) lo = mid + 1;
            else hi = mid;
        }
        const int rank = i + lo;
        if (rank < k) {
            od[(size_t)q * k + rank] = d;
            oi[(size_t)q * k + rank] = ids[p];
        }
    }
    const int total = na + nb < k ? na + nb : k;
    for (int t = total + threadIdx.x; t < k; t += blockDim.x) {
        od[(size_t)q * k + t] = __builtin_inff();
        oi[(size_t)q * k + t] = -1;
    }
    if (threadIdx.x == 0) oc[q] = total;
}

int hvs_search_core(hvs_index *idx, const float *queries, int nq, int fold, int k, int32_t group_filter, const uint32_t *row_mask,
                    bool mask_on_device, double max_distance, int64_t row_begin, int64_t row_end, float *out_dist, int64_t *out_ids,
                    int32_t *out_count, int on_device, int algo) {
    HVS_REQUIRE(idx != nullptr, "hvs_search: idx is NULL");
    HVS_REQUIRE(nq >= 0, "hvs_search: negative query count");
    if (nq == 0) return 0;
    HVS_REQUIRE(queries && out_dist && out_ids && out_count, "hvs_search: NULL argument");
    HVS_REQUIRE(fold >= 1 && fold <= HVS_MAX_FOLD, "hvs_search: fold %d not in [1, %d]", fold, HVS_MAX_FOLD);
    HVS_REQUIRE(k >= 1 && k <= HVS_MAX_K, "hvs_search: k %d not in [1, %d]", k, HVS_MAX_K);
    HVS_REQUIRE(algo >= HVS_ALGO_AUTO && algo <= HVS_ALGO_MFMA_SPLIT, "hvs_search: unknown algo %d", algo);
    if (row_end < 0 || row_end > idx->n) row_end = idx->n;
    if (row_begin < 0) row_begin = 0;
    if (row_begin > row_end) row_begin = row_end;
    HVS_HIP(hipSetDevice(idx->device));
    const float thr = hvs_ceil_to_float(max_distance);
    const int nq_total = nq * fold;

    PendingQueries pq;
    HVS_TRY(queries_stage(idx, queries, nq_total, on_device, &pq));
    if (getenv("HVS_NO_FUSED_PREP")) HVS_TRY(queries_prepare(idx, &pq));  // experiments: always the separate launch
    float *d_q = pq.d_q, *d_qinv = pq.d_qinv;

    float *d_dist = out_dist;
    int64_t *d_ids = out_ids;
    int32_t *d_count = out_count;
    const uint32_t *d_mask = row_mask;
    const size_t dist_bytes = (size_t)nq * k * 4, ids_bytes = (size_t)nq * k * 8, cnt_bytes = (size_t)nq * 4;
    const size_t mask_bytes = (row_mask && !mask_on_device) ? (size_t)((idx->n + 31) / 32) * 4 : 0;
    if (!on_device) {
        void *stage = nullptr;
        const size_t o_ids = (dist_bytes + 255) & ~(size_t)255;
        const size_t o_cnt = (o_ids + ids_bytes + 255) & ~(size_t)255;
        const size_t o_mask = (o_cnt + cnt_bytes + 255) & ~(size_t)255;
        HVS_TRY(hvs_ws_get(idx, 3, o_mask + mask_bytes, &stage));
        d_dist = (float *)stage;
        d_ids = (int64_t *)((char *)stage + o_ids);
        d_count = (int32_t *)((char *)stage + o_cnt);
        if (mask_bytes) {
            HVS_HIP(hipMemcpyAsync((char *)stage + o_mask, row_mask, mask_bytes, hipMemcpyHostToDevice, idx->stream));
            d_mask = (const uint32_t *)((char *)stage + o_mask);
        }
    }

    // A table in group order (hvs_index_partition) answers a group filter with a row range: the ordered part holds the
    // group's rows in [gb, ge), searched with no code test; rows added since form a tail that keeps the per-row test.
    int64_t gb = 0, ge = 0;
    if (group_filter >= 0 && !getenv("HVS_NO_PARTITION") && hvs_part_range(idx, group_filter, &gb, &ge)) {
        const int64_t mb = row_begin > gb ? row_begin : gb, me = row_end < ge ? row_end : ge;
        const int64_t tb = row_begin > idx->part_n ? row_begin : idx->part_n, te = row_end;
        const bool have_main = mb < me, have_tail = tb < te;
        if (have_main && have_tail && filtered_search_takes_scan(idx, nq, fold, k, algo)) {
            // both, on the scan: one launch walks the group's range (no code test) and then the tail (with it).  Keys order
            // by row position, and the ordered part lies below the tail.
            HVS_TRY(search_scan_ranges(idx, d_q, d_qinv, nq, fold, k, -1, d_mask, thr, mb, me, d_dist, d_ids, d_count, nullptr, 0, tb, te, group_filter, &pq));
        } else if (have_main && have_tail) {
            // both, on the matrix pipe: each into lists of row positions, then one merge by (distance, position) -- the
            // ordered part lies below the tail, so ties still go to the lower position
            auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t o_i = up(dist_bytes), o_c = o_i + up(ids_bytes), one = o_c + up(cnt_bytes);
            void *pair = nullptr;
            HVS_TRY(hvs_ws_get(idx, 10, 2 * one, &pair));
            char *a = (char *)pair, *b = a + one;
            idx->emit_positions = true;
            int rc = search_range(idx, d_q, d_qinv, nq, fold, k, -1, d_mask, thr, mb, me, (float *)a, (int64_t *)(a + o_i), (int32_t *)(a + o_c), algo, &pq);
            if (rc == 0)
                rc = search_range(idx, d_q, d_qinv, nq, fold, k, group_filter, d_mask, thr, tb, te, (float *)b, (int64_t *)(b + o_i),
                                  (int32_t *)(b + o_c), algo, &pq);
            idx->emit_positions = false;
            HVS_TRY(rc);
            HVS_TRY(hvs_prof_begin(idx));
            merge_two_kernel<<<nq, 256, 0, idx->stream>>>((const float *)a, (const int64_t *)(a + o_i), (const int32_t *)(a + o_c), (const float *)b,
                                                         (const int64_t *)(b + o_i), (const int32_t *)(b + o_c), k, idx->ids, d_dist, d_ids, d_count);
            HVS_HIP(hipGetLastError());
            HVS_TRY(hvs_prof_end(idx, HVS_PROF_MERGE));
        } else if (have_tail)
            HVS_TRY(search_range(idx, d_q, d_qinv, nq, fold, k, group_filter, d_mask, thr, tb, te, d_dist, d_ids, d_count, algo, &pq));
        else  // the group's range alone (an empty one gives counts of 0)
            HVS_TRY(search_range(idx, d_q, d_qinv, nq, fold, k, -1, d_mask, thr, mb, have_main ? me : mb, d_dist, d_ids, d_count, algo, &pq));
    } else
        HVS_TRY(search_range(idx, d_q, d_qinv, nq, fold, k, group_filter, d_mask, thr, row_begin, row_end, d_dist, d_ids, d_count, algo, &pq));

    if (!on_device) {
        const size_t o_ids = (dist_bytes + 255) & ~(size_t)255;
        const size_t o_cnt = (o_ids + ids_bytes + 255) & ~(size_t)255;
        const size_t span = o_cnt + cnt_bytes;  // the three outputs lie in one staging buffer (see above)
        if (span <= ((size_t)8 << 20)) {
            // one copy into pinned memory instead of three into pageable buffers: each of those costs about as much
            // as the scan of a small table
            void *h = nullptr;
            HVS_TRY(hvs_pinned_get(idx, span, &h));
            HVS_HIP(hipMemcpyAsync(h, d_dist, span, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipStreamSynchronize(idx->stream));
            memcpy(out_dist, h, dist_bytes);
            memcpy(out_ids, (const char *)h + o_ids, ids_bytes);
            memcpy(out_count, (const char *)h + o_cnt, cnt_bytes);
        } else {
            HVS_HIP(hipMemcpyAsync(out_dist, d_dist, dist_bytes, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipMemcpyAsync(out_ids, d_ids, ids_bytes, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipMemcpyAsync(out_count, d_count, cnt_bytes, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipStreamSynchronize(idx->stream));
        }
    }
    return 0;
}

// ---- the search that allocates nothing ----------------------------------------------------------
// Everything hvs_search takes from growable workspaces (prepared queries, inverse query norms, the wave lists of the
// scan and the levels of their merge) lies in one array of the code object instead.  Host code never uses the array's
// name as an address: each GPU holds its own copy, whose address hipGetSymbolAddress gives, once per device.
#define HVS_NOALLOC_QUERY_BYTES ((size_t)1 << 20)
static __device__ __align__(256) unsigned char hvs_noalloc_scratch[HVS_NOALLOC_SCRATCH_BYTES];

#define HVS_MAX_DEVICES 64
static void *g_noalloc_ptr[HVS_MAX_DEVICES];  // per device: address of that device's copy of the array
static std::mutex g_noalloc_mutex;

// with `device` current: look the array's address on that device up (loads the code object there on first use)
int hvs_noalloc_setup(int device) {
    HVS_REQUIRE(device >= 0 && device < HVS_MAX_DEVICES, "hvs_search_noalloc: device %d not in [0, %d)", device, HVS_MAX_DEVICES);
    std::lock_guard<std::mutex> guard(g_noalloc_mutex);
    if (g_noalloc_ptr[device]) return 0;
    void *ptr = nullptr;
    HVS_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(hvs_noalloc_scratch)));
    HVS_REQUIRE(ptr != nullptr, "hvs_search_noalloc: hipGetSymbolAddress returned no address on device %d", device);
    g_noalloc_ptr[device] = ptr;
    return 0;
}

extern "C" int hvs_search_noalloc(hvs_index *idx, const float *queries, int nq, int fold, int k, int32_t group_filter,
                                  const uint32_t *row_mask, double max_distance, int64_t row_begin, int64_t row_end,
                                  float *out_dist, int64_t *out_ids, int32_t *out_count) {
    HVS_REQUIRE(idx != nullptr, "hvs_search_noalloc: idx is NULL");
    HVS_REQUIRE(nq >= 0, "hvs_search_noalloc: negative query count");
    if (nq == 0) return 0;
    HVS_REQUIRE(queries && out_dist && out_ids && out_count, "hvs_search_noalloc: NULL argument");
    HVS_REQUIRE(fold >= 1 && fold <= HVS_MAX_FOLD, "hvs_search_noalloc: fold %d not in [1, %d]", fold, HVS_MAX_FOLD);
    HVS_REQUIRE(k >= 1 && k <= HVS_MAX_K, "hvs_search_noalloc: k %d not in [1, %d]", k, HVS_MAX_K);
    if (row_end < 0 || row_end > idx->n) row_end = idx->n;
    if (row_begin < 0) row_begin = 0;
    if (row_begin > row_end) row_begin = row_end;
    const int nq_total = nq * fold;
    const size_t q_bytes = (size_t)nq_total * idx->ld * 4;
    const size_t inv_off = (q_bytes + 255) & ~(size_t)255;
    const size_t q_end = (inv_off + (size_t)nq_total * 4 + 255) & ~(size_t)255;
    HVS_REQUIRE(q_end <= HVS_NOALLOC_QUERY_BYTES,
                "hvs_search_noalloc: %d query embeddings of %d columns take %zu bytes, the scratch array keeps %zu for them", nq_total,
                idx->dim, q_end, HVS_NOALLOC_QUERY_BYTES);
    HVS_HIP(hipSetDevice(idx->device));
    HVS_TRY(hvs_noalloc_setup(idx->device));  // a table look-up after hvs_index_create has run on this device
    char *scratch = (char *)g_noalloc_ptr[idx->device];

    float *d_q = (float *)scratch, *d_qinv = (float *)(scratch + inv_off);
    prep_queries_kernel<<<(nq_total + 3) / 4, 256, 0, idx->stream>>>(queries, d_q, d_qinv, nq_total, idx->dim, idx->ld, idx->dtype);
    HVS_HIP(hipGetLastError());
    // a table in group order: the group's row range, and behind it in the same launch the tail of rows added since (those
    // keep the per-row test); without a tail the test is dropped
    int64_t gb = 0, ge = 0, tail_begin = 0, tail_end = 0;
    int32_t tail_group = -1;
    if (group_filter >= 0 && !getenv("HVS_NO_PARTITION") && hvs_part_range(idx, group_filter, &gb, &ge)) {
        tail_begin = row_begin > idx->part_n ? row_begin : idx->part_n;
        tail_end = row_end;
        row_begin = row_begin > gb ? row_begin : gb;
        row_end = row_end < ge ? row_end : ge;
        if (row_begin > row_end) row_begin = row_end;
        if (tail_begin >= tail_end) tail_begin = tail_end = 0;
        tail_group = group_filter;
        group_filter = -1;
    }
    // every wave of the scan writes all k keys of its list and every merge level writes all of its outputs, so nothing
    // in the array has to start at zero: what the previous call left is overwritten before it is read
    return search_scan_ranges(idx, d_q, d_qinv, nq, fold, k, group_filter, row_mask, hvs_ceil_to_float(max_distance), row_begin, row_end,
                              out_dist, out_ids, out_count, scratch + HVS_NOALLOC_QUERY_BYTES,
                              HVS_NOALLOC_SCRATCH_BYTES - HVS_NOALLOC_QUERY_BYTES, tail_begin, tail_end, tail_group);
}

int hvs_distances_mfma(hvs_index *idx, const float *d_q, const float *d_qinv, int m, int64_t row_begin, int64_t row_end, float *d_out,
                       bool *used);  // hvs_gemm.hip

__global__ void fmin_columns_kernel(float *__restrict__ acc, const float *__restrict__ other, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc[i] = fminf(acc[i], other[i]);  // NaN only where both are NaN
}

static int distance_column_scan(hvs_index *idx, const float *d_q, const float *d_qinv, int fold, int64_t row_begin, int64_t row_end,
                                float *d_out, const char *who);

// distance column of rows [row_begin, row_end) into the device buffer d_out (queries as hvs_distances takes them):
// from five embeddings on in one pass on the matrix pipe where that route applies, else one streaming pass per 8
static int distance_column(hvs_index *idx, const float *queries, int fold, int on_device, int64_t row_begin, int64_t row_end,
                           float *d_out, const char *who) {
    float *d_q = nullptr, *d_qinv = nullptr;
    HVS_TRY(hvs_prepare_queries(idx, queries, fold, on_device, &d_q, &d_qinv));
    if (fold >= 5) {
        bool used = false;
        HVS_TRY(hvs_distances_mfma(idx, d_q, d_qinv, fold, row_begin, row_end, d_out, &used));
        if (used) return 0;
    }
    if (fold <= HVS_MAX_FOLD) return distance_column_scan(idx, d_q, d_qinv, fold, row_begin, row_end, d_out, who);
    const int64_t n_rows = row_end - row_begin;
    void *tmp = nullptr;
    HVS_TRY(hvs_ws_get(idx, 6, (size_t)n_rows * 4 + 256, &tmp));
    for (int first = 0; first < fold; first += HVS_MAX_FOLD) {
        const int part = fold - first < HVS_MAX_FOLD ? fold - first : HVS_MAX_FOLD;
        HVS_TRY(distance_column_scan(idx, d_q + (size_t)first * idx->ld, d_qinv + first, part, row_begin, row_end,
                                     first == 0 ? d_out : (float *)tmp, who));
        if (first) {
            int g = (int)((n_rows + 255) / 256);
            fmin_columns_kernel<<<g > 2048 ? 2048 : g, 256, 0, idx->stream>>>(d_out, (const float *)tmp, n_rows);
            HVS_HIP(hipGetLastError());
        }
    }
    return 0;
}

static int distance_column_scan(hvs_index *idx, const float *d_q, const float *d_qinv, int fold, int64_t row_begin, int64_t row_end,
                                float *d_out, const char *who) {
    const int64_t n_rows = row_end - row_begin;
    const int NQ = round_nq(fold);
    const size_t lds = scan_lds_bytes(NQ, idx->ld, idx->elem_size, 1, 1, scan_rows_in_flight(NQ, idx->elem_size));
    HVS_REQUIRE(lds <= LDS_LIMIT, "%s: dim %d with fold %d needs %zu bytes of LDS", who, idx->dim, fold, lds);
    ScanParams p;
    memset(&p, 0, sizeof(p));
    p.rows = idx->rows;
    p.inv_norm = idx->inv_norm;
    p.queries = d_q;
    p.q_inv = d_qinv;
    p.row_begin = row_begin;
    p.row_end = row_end;
    p.ld = idx->ld;
    p.chunks = idx->ld * idx->elem_size / 16;
    p.fold = fold;
    p.nq_out = 1;
    p.k = 1;
    p.group_filter = -1;
    p.dist_out = d_out;
    const int rif = scan_rows_in_flight(NQ, idx->elem_size);
    const int64_t row_groups = (n_rows + rif - 1) / rif;
    const int64_t want_blocks = (row_groups + SCAN_WAVES - 1) / SCAN_WAVES;
    int grid = 0;
    HVS_TRY(hvs_prof_begin(idx));
    HVS_TRY(launch_scan<true>(idx->dtype, NQ, 64, want_blocks, lds, idx->stream, p, &grid, false));
    HVS_TRY(hvs_prof_end(idx, HVS_PROF_SCAN));
    return 0;
}

extern "C" int hvs_distances(hvs_index *idx, const float *queries, int fold, int64_t row_begin, int64_t row_end,
                             float *out, int on_device) {
    HVS_REQUIRE(idx != nullptr, "hvs_distances: idx is NULL");
    HVS_REQUIRE(queries && out, "hvs_distances: NULL argument");
    HVS_REQUIRE(fold >= 1 && fold <= HVS_MAX_DIST_FOLD, "hvs_distances: fold %d not in [1, %d]", fold, HVS_MAX_DIST_FOLD);
    if (row_end < 0 || row_end > idx->n) row_end = idx->n;
    if (row_begin < 0) row_begin = 0;
    if (row_begin >= row_end) return 0;
    HVS_HIP(hipSetDevice(idx->device));
    const int64_t n_rows = row_end - row_begin;
    float *d_out = out;
    if (!on_device) {
        void *stage = nullptr;
        HVS_TRY(hvs_ws_get(idx, 3, (size_t)n_rows * 4, &stage));
        d_out = (float *)stage;
    }
    HVS_TRY(distance_column(idx, queries, fold, on_device, row_begin, row_end, d_out, "hvs_distances"));
    if (!on_device) {
        HVS_HIP(hipMemcpyAsync(out, d_out, (size_t)n_rows * 4, hipMemcpyDeviceToHost, idx->stream));
        HVS_HIP(hipStreamSynchronize(idx->stream));
    }
    return 0;
}

extern "C" int hvs_range_search(hvs_index *idx, const float *queries, int fold, int32_t group_filter, const uint32_t *row_mask,
                                double max_distance, int64_t row_begin, int64_t row_end, int64_t capacity, float *out_dist,
                                int64_t *out_ids, int64_t *out_total, int on_device) {
    return hvs_range_search_core(idx, queries, fold, group_filter, row_mask, on_device != 0, max_distance, row_begin, row_end, capacity,
                                 out_dist, out_ids, out_total, on_device);
}

extern "C" int hvs_range_search_filtered(hvs_index *idx, const float *queries, int fold, const hvs_filter *filter, double max_distance,
                                         int64_t row_begin, int64_t row_end, int64_t capacity, float *out_dist, int64_t *out_ids,
                                         int64_t *out_total, int on_device) {
    HVS_REQUIRE(idx != nullptr, "hvs_range_search_filtered: idx is NULL");
    const uint32_t *d_mask = nullptr;
    if (filter) {
        HVS_HIP(hipSetDevice(idx->device));
        HVS_TRY(hvs_filter_mask(idx, filter, &d_mask, row_begin, row_end));
    }
    return hvs_range_search_core(idx, queries, fold, filter ? filter->group : -1, d_mask, true, max_distance, row_begin, row_end, capacity,
                                 out_dist, out_ids, out_total, on_device);
}

int hvs_range_search_core(hvs_index *idx, const float *queries, int fold, int32_t group_filter, const uint32_t *row_mask,
                          bool mask_on_device, double max_distance, int64_t row_begin, int64_t row_end, int64_t capacity,
                          float *out_dist, int64_t *out_ids, int64_t *out_total, int on_device) {
    HVS_REQUIRE(idx != nullptr, "hvs_range_search: idx is NULL");
    HVS_REQUIRE(queries && out_total, "hvs_range_search: NULL argument");
    HVS_REQUIRE(capacity >= 0 && (capacity == 0 || (out_dist && out_ids)), "hvs_range_search: outputs missing for capacity %lld", (long long)capacity);
    HVS_REQUIRE(fold >= 1 && fold <= HVS_MAX_DIST_FOLD, "hvs_range_search: fold %d not in [1, %d]", fold, HVS_MAX_DIST_FOLD);
    if (row_end < 0 || row_end > idx->n) row_end = idx->n;
    if (row_begin < 0) row_begin = 0;
    int64_t total = 0;
    // a table in group order: the group's row range (no code test), then the rows added since (with it); both in
    // ascending row position, the range below the tail
    struct Seg { int64_t b, e; int32_t gf; } segs[2];
    int n_segs = 0;
    int64_t gb = 0, ge = 0;
    if (group_filter >= 0 && !getenv("HVS_NO_PARTITION") && hvs_part_range(idx, group_filter, &gb, &ge)) {
        const int64_t mb = row_begin > gb ? row_begin : gb, me = row_end < ge ? row_end : ge;
        const int64_t tb = row_begin > idx->part_n ? row_begin : idx->part_n;
        if (mb < me) segs[n_segs++] = {mb, me, -1};
        if (tb < row_end) segs[n_segs++] = {tb, row_end, group_filter};
    } else if (row_begin < row_end)
        segs[n_segs++] = {row_begin, row_end, group_filter};
    if (n_segs > 0) {
        HVS_HIP(hipSetDevice(idx->device));
        int64_t widest = 0;
        for (int i = 0; i < n_segs; i++) widest = segs[i].e - segs[i].b > widest ? segs[i].e - segs[i].b : widest;
        // staging (workspace 3): [distance column][mask words, if they come from the host][outputs, if they go to the host]
        const size_t mask_bytes = (row_mask && !mask_on_device) ? (size_t)((idx->n + 31) / 32) * 4 : 0;
        const size_t o_mask = ((size_t)widest * 4 + 255) & ~(size_t)255;
        const size_t o_dist = (o_mask + mask_bytes + 255) & ~(size_t)255;
        const size_t o_ids = (o_dist + (on_device ? 0 : (size_t)capacity * 4) + 255) & ~(size_t)255;
        void *stage = nullptr;
        HVS_TRY(hvs_ws_get(idx, 3, o_ids + (on_device ? 0 : (size_t)capacity * 8) + 256, &stage));
        float *d_col = (float *)stage;
        const uint32_t *d_mask = row_mask;
        if (mask_bytes) {
            HVS_HIP(hipMemcpyAsync((char *)stage + o_mask, row_mask, mask_bytes, hipMemcpyHostToDevice, idx->stream));
            d_mask = (const uint32_t *)((char *)stage + o_mask);
        }
        float *d_od = on_device ? out_dist : (float *)((char *)stage + o_dist);
        int64_t *d_oi = on_device ? out_ids : (int64_t *)((char *)stage + o_ids);
        int64_t written = 0;
        for (int i = 0; i < n_segs; i++) {
            HVS_TRY(distance_column(idx, queries, fold, on_device, segs[i].b, segs[i].e, d_col, "hvs_range_search"));
            hvs_live_params lp{idx->groups, d_mask, idx->n_deleted > 0 ? idx->deleted : nullptr, segs[i].gf, segs[i].b, segs[i].e};
            lp.dist = d_col;
            lp.max_dist = hvs_ceil_to_float(max_distance);
            int64_t seg_total = 0;
            HVS_TRY(hvs_range_collect(idx, lp, capacity - written, d_od + written, d_oi + written, &seg_total));
            total += seg_total;
            written += seg_total < capacity - written ? seg_total : capacity - written;
        }
        if (!on_device && written > 0) {
            HVS_HIP(hipMemcpyAsync(out_dist, d_od, (size_t)written * 4, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipMemcpyAsync(out_ids, d_oi, (size_t)written * 8, hipMemcpyDeviceToHost, idx->stream));
        }
        HVS_HIP(hipStreamSynchronize(idx->stream));
    }
    *out_total = total;
    return 0;
}

// ---- cross-GPU merge ----------------------------------------------------------------------------
// (dist, id) lists -> keys are not usable here (ids are 64-bit), so this kernel orders pairs directly.
// List pl of query q starts at in_dist[q * dist_q + pl * dist_p] (likewise ids and counts): [nq, P, k] arrays and the
// per-rank records of a packed all-gather are both such layouts.
// One block per query.  Every list is ascending and ids are unique across lists, so an entry's place in the union is
// its own index plus, for every other list, the number of entries ordered before it -- a binary search per list; the
// entries whose place is below k are written there.  (A wave inserting the P k candidates one by one into a sorted list
// took 1.6 ms for 256 queries, k = 100, P = 2: more than the search it followed.)
struct MergeStrides {
    int64_t dist_q, dist_p, ids_q, ids_p, cnt_q, cnt_p;
};
__global__ __launch_bounds__(256) void merge_pairs_kernel(const float *__restrict__ in_dist, const int64_t *__restrict__ in_ids,
                                                           const int32_t *__restrict__ in_count, int P, int k, MergeStrides ms,
                                                           float *out_dist, int64_t *out_ids, int32_t *out_count) {
    __shared__ int s_cnt[64];  // P <= 64 lists
    const int q = blockIdx.x;
    for (int pl = threadIdx.x; pl < P; pl += blockDim.x) {
        int c = in_count[(int64_t)q * ms.cnt_q + (int64_t)pl * ms.cnt_p];
        s_cnt[pl] = c < 0 ? 0 : (c > k ? k : c);  // a count is data from another rank: never index past a list
    }
    __syncthreads();
    int total = 0;
    for (int pl = 0; pl < P; pl++) total += s_cnt[pl];
    for (int t = threadIdx.x; t < P * k; t += blockDim.x) {
        const int pl = t / k, i = t - pl * k;
        if (i >= s_cnt[pl]) continue;
        const float d = in_dist[(int64_t)q * ms.dist_q + (int64_t)pl * ms.dist_p + i];
        const int64_t id = in_ids[(int64_t)q * ms.ids_q + (int64_t)pl * ms.ids_p + i];
        int rank = i;
        for (int o = 0; o < P && rank < k; o++) {
            if (o == pl) continue;
            const float *od = in_dist + (int64_t)q * ms.dist_q + (int64_t)o * ms.dist_p;
            const int64_t *oi = in_ids + (int64_t)q * ms.ids_q + (int64_t)o * ms.ids_p;
            int lo = 0, hi = s_cnt[o];  // entries of list o ordered before (d, id)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float md = od[mid];
                const int64_t mi = oi[mid];
                if (md < d || (md == d && (mi < id || (mi == id && o < pl)))) lo = mid + 1;  // (an id present in two lists: the lower list first)
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_dist[(size_t)q * k + rank] = d;
            out_ids[(size_t)q * k + rank] = id;
        }
    }
    const int kept = total < k ? total : k;
    for (int i = kept + threadIdx.x; i < k; i += blockDim.x) {
        out_dist[(size_t)q * k + i] = __builtin_inff();
        out_ids[(size_t)q * k + i] = -1;
    }
    if (threadIdx.x == 0) out_count[q] = kept;
}

int hvs_merge_launch(hipStream_t st, const float *in_dist, const int64_t *in_ids, const int32_t *in_count, int nq, int P, int k,
                     int64_t dist_q, int64_t dist_p, int64_t ids_q, int64_t ids_p, int64_t cnt_q, int64_t cnt_p, float *out_dist,
                     int64_t *out_ids, int32_t *out_count) {
    HVS_REQUIRE(P >= 1 && P <= 64, "merge: %d lists per query, at most 64", P);
    const MergeStrides ms{dist_q, dist_p, ids_q, ids_p, cnt_q, cnt_p};
    merge_pairs_kernel<<<nq, 256, 0, st>>>(in_dist, in_ids, in_count, P, k, ms, out_dist, out_ids, out_count);
    HVS_HIP(hipGetLastError());
    return 0;
}

extern "C" int hvs_merge_topk(int device_id, void *stream, const float *in_dist, const int64_t *in_ids,
                              const int32_t *in_count, int nq, int P, int k, float *out_dist, int64_t *out_ids,
                              int32_t *out_count) {
    HVS_REQUIRE(nq >= 0 && P >= 1, "hvs_merge_topk: bad nq/P");
    if (nq == 0) return 0;
    HVS_REQUIRE(k >= 1 && k <= HVS_MAX_K, "hvs_merge_topk: k %d not in [1, %d]", k, HVS_MAX_K);
    HVS_REQUIRE(in_dist && in_ids && in_count && out_dist && out_ids && out_count, "hvs_merge_topk: NULL argument");
    HVS_HIP(hipSetDevice(device_id));
    return hvs_merge_launch((hipStream_t)stream, in_dist, in_ids, in_count, nq, P, k, (int64_t)P * k, k, (int64_t)P * k, k, P, 1, out_dist,
                            out_ids, out_count);
}
