// Batched exact search as an LDS-staged MFMA contraction (gfx950).
//
// S[a, n] = sum_k A[a, k] * X[n, k]      A = query embeddings, X = the stored rows (fp32, fp16 or bf16;
//                                          A is held in the same type, accumulation is fp32).
// Block tile 128 (A rows) x 128 (corpus rows), 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain) or v_mfma_f32_32x32x16_{f16,bf16}.  Corpus rows sit on the MFMA column
// (= lane) axis, A rows on the accumulator-register axis, so per-corpus-row quantities (inverse norm,
// filter bit, running argmin) are per-lane scalars and folding consecutive query embeddings (LEAST)
// is a min over adjacent accumulator registers.
//
// The K loop streams 128-byte slices of both operands through a double-buffered LDS image filled by
// LDS-DMA (buffer_load_dwordx4 ... lds, XOR swizzle applied on the source address so that the
// ds_read_b128 fragment reads are conflict-free); the DMA of step g+1 is in flight while step g's
// 64 MFMAs per wave run.
//
// Search never writes the score matrix: the epilogue of every 128 x 128 tile compares each score
// with the query's current admission bound tau (the k-th best distance seen so far by this block)
// and appends the rare survivors to a small per-(block, query) candidate array; when an array nears
// capacity one wave ranks it, keeps the best k and lowers tau.
#include "hvs_gemm_tile.h"

// ---------------------------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------------------------
struct SearchParams {
    const void *A;       // prepared queries [nA] rows of lda_bytes (a multiple of KB, pad zero), element type = the index's
    const float *q_inv;  // [nA]
    int nA;              // nq_out * FOLD
    int lda_bytes;
    const void *X;       // corpus rows of ld_bytes
    const float *inv_norm;
    const int32_t *groups;
    const uint32_t *mask;
    const uint32_t *deleted;
    int32_t group_filter;
    int64_t row_begin, row_end;
    int ld_bytes, nks;
    int nq_out;  // output queries
    int k;
    float thr;
    int cap;            // capacity of one candidate array
    uint64_t *cand;     // [gridDim.x, BM / FOLD, cap]
    uint64_t *partial;  // [nq_out, S, k]
    int QT, S;
    int64_t T;      // corpus tiles in [row_begin, row_end)
    int *overflow;  // set to 1 if a candidate array ever overflowed (must stay 0)
    const uint32_t *remap;  // optional: X holds a compacted copy; remap[r] = row position of its row r in the index
    int dbg;        // timing experiments only (HVS_DBG): 4 no barrier, 8 no epilogue, 16 older block->XCD map
};

// One wave ranks the candidate array of one query, keeps the best `keep`, writes them in ascending
// order to dst (dst == the array itself for an in-place compaction) and returns the new count.
__device__ __forceinline__ int rank_select(const uint64_t *src, int cnt, int keep, uint64_t *dst, bool pad_dst,
                                           volatile uint64_t *scratch, int lane, uint64_t *kth_key) {
    uint64_t own[CAND_MAX / 64];
    int rank[CAND_MAX / 64];
#pragma unroll
    for (int e = 0; e < CAND_MAX / 64; e++) {
        const int i = lane + 64 * e;
        own[e] = i < cnt ? src[i] : HVS_KEY_EMPTY;
        rank[e] = 0;
        if (i < cnt) scratch[i] = own[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < cnt; j++) {
        const uint64_t kj = scratch[j];
#pragma unroll
        for (int e = 0; e < CAND_MAX / 64; e++) rank[e] += (kj < own[e]) ? 1 : 0;
    }
    const int newcnt = cnt < keep ? cnt : keep;
    uint64_t kth = HVS_KEY_EMPTY;
#pragma unroll
    for (int e = 0; e < CAND_MAX / 64; e++) {
        const int i = lane + 64 * e;
        if (i < cnt && rank[e] < keep) {
            dst[rank[e]] = own[e];
            if (rank[e] == keep - 1) kth = own[e];
        }
    }
    if (pad_dst)
        for (int i = newcnt + lane; i < keep; i += 64) dst[i] = HVS_KEY_EMPTY;
    // broadcast the k-th key (held by exactly one lane, if any)
    const uint64_t has = __ballot(kth != HVS_KEY_EMPTY);
    if (has) {
        const int src_lane = __ffsll((unsigned long long)has) - 1;
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)kth, src_lane);
        const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(kth >> 32), src_lane);
        *kth_key = ((uint64_t)hi << 32) | lo;
    } else {
        *kth_key = HVS_KEY_EMPTY;
    }
    return newcnt;
}

// Lower bound on dot * inv_row that a row needs for its distance to be below tau, loosened so that
// it never rejects a row the exact fp32 test (hvs_cos_dist(...) < tau) would accept.
__device__ __forceinline__ float prefilter_bound(float tau, float q_inv) {
    const float t = ((1.0f - tau) - 4e-7f) / q_inv;
    return t - fabsf(t) * 4e-7f;
}

template <int DT, int FOLD, class T>
__global__ __launch_bounds__(T::GT, T::GT == 256 ? 2 : 1) void mfma_search_kernel(SearchParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // device-only types (buffer descriptors) inside
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int GT = T::GT, BM = T::BM, BN = T::BN, STAGE_BYTES = T::STAGE_BYTES, MT = T::MT, NT = T::NT, NW = T::NW;
    char *stages = smem;                                                      // 2 * STAGE_BYTES
    float *tau_lds = reinterpret_cast<float *>(smem + 2 * STAGE_BYTES);       // [BM] per folded query
    int *cnt_lds = reinterpret_cast<int *>(tau_lds + BM);                     // [BM] per folded query
    float *qinv_lds = reinterpret_cast<float *>(cnt_lds + BM);                // [BM] per A row
    float *bound_lds = qinv_lds + BM;                                         // [BM] per A row
    uint64_t *scratch_all = reinterpret_cast<uint64_t *>(bound_lds + BM);     // [NW][CAND_MAX]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int h = lane >> 5;

    // block -> (query tile, corpus split).  Blocks are dealt to the 8 XCDs round-robin (b % 8); each XCD gets
    // a rectangle of (QT / gq) query tiles x (S / gs) splits, gq = gcd(QT, 8), gs = 8 / gq, with as few
    // query tiles as possible: their A slabs (re-read for every corpus tile) then stay in that XCD's L2,
    // and only the corpus stream comes from HBM.
    int qt, s;
    if ((p.S & 7) == 0 && !(p.dbg & 16)) {
        const int gq = (p.QT & 7) == 0 ? 8 : ((p.QT & 3) == 0 ? 4 : ((p.QT & 1) == 0 ? 2 : 1));
        const int gs = 8 / gq;
        const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
        const int qn = p.QT / gq, sn = p.S / gs;
        qt = (xcd % gq) * qn + t % qn;
        s = (xcd / gq) * sn + t / qn;
    } else if ((p.S & 7) == 0) {
        const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
        qt = t % p.QT;
        s = xcd + 8 * (t / p.QT);
    } else {
        qt = blockIdx.x % p.QT;
        s = blockIdx.x / p.QT;
    }
    const int64_t tile_lo = p.T * s / p.S, tile_hi = p.T * (s + 1) / p.S;
    const int64_t a_row0 = (int64_t)qt * BM;
    constexpr int groups_per_tile = BM / FOLD;           // folded queries per tile
    const int q_out0 = qt * groups_per_tile;             // first folded query of this tile
    uint64_t *my_cand = p.cand + (size_t)blockIdx.x * groups_per_tile * p.cap;
    volatile uint64_t *scratch = scratch_all + (size_t)wave * CAND_MAX;

    for (int i = tid; i < BM; i += GT) {
        tau_lds[i] = p.thr;
        cnt_lds[i] = 0;
        const int64_t ar = a_row0 + i;
        const float qi = ar < p.nA ? p.q_inv[ar] : 0.0f;
        qinv_lds[i] = qi;
        bound_lds[i] = prefilter_bound(p.thr, qi);
    }

    f32x16 acc[MT][NT];
    zero_acc(acc);
    const int64_t n_tiles = tile_hi - tile_lo;
    const int64_t total_steps = n_tiles * p.nks;
    TileSrcT<T::DMA_N> srcA, srcB;
    {
        const int64_t a_valid = p.nA - a_row0;
        tile_src_init<T>(srcA, p.A, a_row0, a_valid < BM ? (int)a_valid : BM, p.lda_bytes, tid);
    }
    auto init_b = [&](int64_t tile) {
        const int64_t r0 = p.row_begin + tile * BN;
        const int64_t valid = p.row_end - r0;
        tile_src_init<T>(srcB, p.X, r0, valid < BN ? (int)valid : BN, p.ld_bytes, tid);
    };
    if (total_steps > 0) {
        init_b(tile_lo);
        dma_tiles<T>(srcA, srcB, stages, 0, wave);
    }
    __syncthreads();

    int64_t tile = tile_lo;
    int ks = 0;
    for (int64_t g = 0; g < total_steps; g++) {
        const int cur = (int)(g & 1);
        // next step's coordinates; after the last step the current tile's step 0 is fetched again (harmless,
        // keeps the loop free of branches around the LDS-DMA instructions)
        int nks_next = ks + 1;
        int64_t ntile = tile;
        if (nks_next == p.nks) { nks_next = 0; ntile = tile + 1; }
        if (nks_next == 0 && g + 1 < total_steps) init_b((p.dbg & 32) ? tile_lo : ntile);
        compute_step<DT, T>(acc, stages + cur * STAGE_BYTES, wm, wn, lane, srcA, srcB, stages + (cur ^ 1) * STAGE_BYTES,
                            nks_next, wave);

        if (ks == p.nks - 1) {
          if (!(p.dbg & 8)) {
            // ---- epilogue of tile `tile`: prefilter against the bound, exact test for survivors ----
            const int64_t n_base = p.row_begin + tile * BN + wn * (NT * 32) + (lane & 31);
            int64_t rows2[NT];
            bool ok2[NT];
            float invn2[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int64_t row = n_base + nt * 32;
                bool ok = row < p.row_end;
                float invn = 0.0f;
                if (ok) {
                    invn = p.inv_norm[row];
                    if (p.group_filter >= 0) ok = p.groups[row] == p.group_filter;
                    if (ok && p.mask) ok = (p.mask[row >> 5] >> (row & 31)) & 1u;
                    if (ok && p.deleted) ok = !((p.deleted[row >> 5] >> (row & 31)) & 1u);
                }
                rows2[nt] = row;
                ok2[nt] = ok;
                invn2[nt] = invn;
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int a_loc = wm * (MT * 32) + mt * 32 + 8 * gq + 4 * h;  // A row of register 4*gq (+0..3)
                    const f32x4 bnd = *reinterpret_cast<const f32x4 *>(bound_lds + a_loc);
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        bool any = false;
#pragma unroll
                        for (int e = 0; e < 4; e++) any |= acc[mt][nt][4 * gq + e] * invn2[nt] > bnd[e];
                        any &= ok2[nt];
                        if (__ballot(any)) {  // wave-uniform and rare once tau has tightened
                            int av = a_loc;
                            asm volatile("" : "+v"(av));  // keep this branch's address arithmetic inside it
                            const f32x4 qi = *reinterpret_cast<const f32x4 *>(qinv_lds + av);
                            float d[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) d[e] = hvs_cos_dist(acc[mt][nt][4 * gq + e], invn2[nt], qi[e]);
                            if (FOLD == 2) {
                                d[0] = fminf(d[0], d[1]);
                                d[2] = fminf(d[2], d[3]);
                            } else if (FOLD == 4) {
                                d[0] = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
                            }
#pragma unroll
                            for (int e = 0; e < 4; e += FOLD) {
                                const int qv = (av + e) / FOLD;  // folded query, local to the tile
                                if (any && d[e] < tau_lds[qv]) {
                                    const int pos = atomicAdd(&cnt_lds[qv], 1);
                                    if (pos < p.cap) {
                                        const uint32_t rp = p.remap ? p.remap[rows2[nt]] : (uint32_t)rows2[nt];
                                        my_cand[(size_t)qv * p.cap + pos] = hvs_make_key(hvs_ord(d[e]), rp);
                                    }
                                    else *p.overflow = 1;
                                }
                            }
                        }
                    }
                }
            }
          }
            zero_acc(acc);
        }

        if (!(p.dbg & 4)) __syncthreads();  // the DMA of the next stage has landed (vmcnt) for every wave

        if (ks == p.nks - 1) {
            // ---- compaction: arrays that could overflow during the next tile, and arrays that have grown
            // enough for a fresher tau to pay (a stale tau admits ~count/k times too many survivors) ----
            const int eager = p.k * 2 > 32 ? p.k * 2 : 32;
            const int trigger = eager < p.cap - BN ? eager : p.cap - BN;
            for (int base = 0; base < groups_per_tile; base += 64) {
                const int ql = base + lane;
                // wave w owns the queries with index % NW == w: a static split, so no wave's decision
                // depends on counters another wave is resetting at the same time
                const bool need = ql < groups_per_tile && (ql & (NW - 1)) == wave && cnt_lds[ql] > trigger;
                uint64_t pending = __ballot(need);
                while (pending) {
                    const int b = __ffsll((unsigned long long)pending) - 1;
                    pending &= pending - 1;
                    const int q = base + b;
                    int c = cnt_lds[q];
                    c = c > p.cap ? p.cap : c;
                    uint64_t kth;
                    uint64_t *arr = my_cand + (size_t)q * p.cap;
                    const int nc = rank_select(arr, c, p.k, arr, false, scratch, lane, &kth);
                    if (lane == 0) cnt_lds[q] = nc;
                    if (nc == p.k) {
                        const float tau = hvs_unord((uint32_t)(kth >> 32));
                        if (lane == 0) tau_lds[q] = tau;
                        if (lane < FOLD) bound_lds[q * FOLD + lane] = prefilter_bound(tau, qinv_lds[q * FOLD + lane]);
                    }
                }
            }
            // the next epilogue is a whole K loop (>= 1 barrier) away, so no extra barrier is needed
            // unless the K loop has a single step
            if (p.nks == 1) __syncthreads();
        }
        ks = nks_next;
        tile = ntile;
    }

    // ---- publish: every folded query's best k, ascending, padded ----
    __syncthreads();
    for (int ql = wave; ql < groups_per_tile; ql += NW) {
        const int qo = q_out0 + ql;
        if (qo >= p.nq_out) break;
        int c = cnt_lds[ql];
        c = c > p.cap ? p.cap : c;
        uint64_t kth;
        rank_select(my_cand + (size_t)ql * p.cap, c, p.k, p.partial + ((size_t)qo * p.S + s) * p.k, true, scratch, lane, &kth);
    }
#endif
}

#define HVS_COMPACT_MAX_BYTES ((size_t)16 << 30)  // largest dense copy a filtered search may make

// ---------------------------------------------------------------------------------------------
// host side of the search
// ---------------------------------------------------------------------------------------------
int hvs_reduce_lists(hvs_index *idx, uint64_t *bufA, uint64_t *bufB, int nq, int P, int k, float *d_dist,
                     int64_t *d_ids, int32_t *d_count);

template <class T> static size_t search_lds_bytes() { return (size_t)2 * T::STAGE_BYTES + T::BM * 16 + (size_t)T::NW * CAND_MAX * 8; }

static int padded_fold(int fold) { return fold <= 1 ? 1 : (fold == 2 ? 2 : 4); }

bool hvs_mfma_supported(hvs_index *idx, int nq, int fold, int k) {
    (void)nq;
    (void)idx;
    return fold >= 1 && fold <= 4 && k >= 1 && k <= MFMA_MAX_K;
}

// A operand: groups of `pf` rows per output query (the last embedding repeated when fold < pf), in the
// index's element type (the prepared queries are already rounded to it, so the conversion is exact),
// row pitch lda_bytes (a multiple of KB, pad zero), with the matching inverse norms.
template <typename T> __device__ __forceinline__ T a_cvt(float v);
template <> __device__ __forceinline__ float a_cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ __half a_cvt<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ uint16_t a_cvt<uint16_t>(float v) { return hvs_float_to_bf16(v); }

template <typename T>
__global__ void build_a_kernel(const float *__restrict__ src, const float *__restrict__ src_inv, T *__restrict__ dst,
                               float *__restrict__ dst_inv, int fold, int pf, int ld, int lda) {
    const int row = blockIdx.x;  // destination row
    const int q = row / pf, j = row % pf;
    const int sj = j < fold ? j : fold - 1;
    const float *s = src + ((size_t)q * fold + sj) * ld;
    T *d = dst + (size_t)row * lda;
    for (int c = threadIdx.x; c < lda; c += blockDim.x) d[c] = a_cvt<T>(c < ld ? s[c] : 0.0f);
    if (threadIdx.x == 0) dst_inv[row] = src_inv[q * fold + sj];
}

template <int DT, class T>
static int launch_search(int pf, int grid, hipStream_t st, const SearchParams &p) {
    const size_t lds = search_lds_bytes<T>();
#define HVS_LAUNCH_SEARCH(F)                                                                                         \
    do {                                                                                                             \
        HVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mfma_search_kernel<DT, F, T>),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
        mfma_search_kernel<DT, F, T><<<grid, T::GT, lds, st>>>(p);                                                   \
    } while (0)
    if (pf == 1) HVS_LAUNCH_SEARCH(1);
    else if (pf == 2) HVS_LAUNCH_SEARCH(2);
    else HVS_LAUNCH_SEARCH(4);
#undef HVS_LAUNCH_SEARCH
    HVS_HIP(hipGetLastError());
    return 0;
}

int hvs_search_mfma(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int32_t group_filter,
                    const uint32_t *d_mask, float thr, int64_t row_begin, int64_t row_end, float *d_dist,
                    int64_t *d_ids, int32_t *d_count) {
    const int pf = padded_fold(fold);
    const int nA = nq * pf;
    const int esz = idx->elem_size;
    const int ld_bytes = idx->ld * esz;
    const int lda_bytes = (ld_bytes + KB - 1) / KB * KB;
    const int lda = lda_bytes / esz;
    void *a_ws = nullptr;
    const size_t a_bytes = ((size_t)nA * lda_bytes + 255) & ~(size_t)255;
    HVS_TRY(hvs_ws_get(idx, 4, a_bytes + (size_t)nA * 4, &a_ws));
    float *A_inv = (float *)((char *)a_ws + a_bytes);
    if (idx->dtype == HVS_F32)
        build_a_kernel<float><<<nA, 256, 0, idx->stream>>>(d_q, d_qinv, (float *)a_ws, A_inv, fold, pf, idx->ld, lda);
    else if (idx->dtype == HVS_F16)
        build_a_kernel<__half><<<nA, 256, 0, idx->stream>>>(d_q, d_qinv, (__half *)a_ws, A_inv, fold, pf, idx->ld, lda);
    else
        build_a_kernel<uint16_t><<<nA, 256, 0, idx->stream>>>(d_q, d_qinv, (uint16_t *)a_ws, A_inv, fold, pf, idx->ld, lda);
    HVS_HIP(hipGetLastError());

    SearchParams p;
    memset(&p, 0, sizeof(p));
    p.A = a_ws;
    p.q_inv = A_inv;
    p.nA = nA;
    p.lda_bytes = lda_bytes;
    p.X = idx->rows;
    p.inv_norm = idx->inv_norm;
    p.groups = idx->groups;
    p.mask = d_mask;
    p.deleted = idx->n_deleted > 0 ? idx->deleted : nullptr;
    p.group_filter = group_filter;
    p.row_begin = row_begin;
    p.row_end = row_end;

    // A WHERE on the group code or a row bitmask: count the live rows and, when they are a minority,
    // contract over a dense copy of them (tombstones alone are sparse and stay in the epilogue): the
    // contraction then costs n_live instead of n rows, and the copy is 2 * n_live * row bytes of traffic.
    const int64_t n_range = row_end - row_begin;
    if ((group_filter >= 0 || d_mask) && n_range >= 512 && row_end <= (int64_t)0xFFFFFFFFll && !getenv("HVS_NO_COMPACT")) {
        hvs_live_params lp{idx->groups, d_mask, p.deleted, group_filter, row_begin, row_end};
        uint32_t n_live = 0;
        HVS_TRY(hvs_live_count(idx, lp, &n_live));
        const size_t pos_bytes = ((size_t)n_live * 4 + 255) & ~(size_t)255;
        const size_t copy_bytes = (size_t)n_live * ld_bytes;
        if ((int64_t)n_live * 2 <= n_range && copy_bytes <= HVS_COMPACT_MAX_BYTES) {
            void *cr = nullptr;
            HVS_TRY(hvs_ws_get(idx, 6, 2 * pos_bytes + copy_bytes + 256, &cr));
            uint32_t *positions = (uint32_t *)cr;
            float *c_inv = (float *)((char *)cr + pos_bytes);
            char *c_rows = (char *)cr + 2 * pos_bytes;
            if (n_live > 0) {
                HVS_TRY(hvs_live_write(idx, lp, positions));
                HVS_TRY(hvs_live_gather(idx, positions, n_live, c_rows, c_inv));
            }
            p.X = c_rows;
            p.inv_norm = c_inv;
            p.remap = positions;
            p.groups = nullptr;
            p.mask = nullptr;
            p.deleted = nullptr;
            p.group_filter = -1;
            p.row_begin = row_begin = 0;
            p.row_end = row_end = n_live;
        }
    }
    p.ld_bytes = ld_bytes;
    p.nks = lda_bytes / KB;
    p.nq_out = nq;
    p.k = k;
    p.thr = thr;
    // tile shape: 16-bit rows with more than one 128-row tile of queries take the 256 x 256 block tile
    bool big = idx->dtype != HVS_F32 && nA > Tile128::BM;
    if (const char *e = getenv("HVS_TILE")) big = idx->dtype != HVS_F32 && atoi(e) == 256;
    const int bm = big ? Tile256::BM : Tile128::BM, bn = big ? Tile256::BN : Tile128::BN;
    int cap = k + 2 * bn;
    cap = (cap + 63) / 64 * 64;
    if (cap > CAND_MAX) cap = CAND_MAX;  // >= k + bn for every k <= MFMA_MAX_K: one tile's survivors always fit
    p.cap = cap;
    p.QT = (nA + bm - 1) / bm;
    p.T = (row_end - row_begin + bn - 1) / bn;
    // splits: fill every block slot of the chip (2 per CU for the small tile, 1 for the large), keep S a
    // multiple of 8 (XCD affinity) and every split non-empty
    const int slots = big ? 256 : 512;
    int S = (slots + p.QT - 1) / p.QT;
    S = (S + 7) / 8 * 8;
    if ((int64_t)S > p.T) S = (int)p.T;
    if (S >= 8) S = S / 8 * 8;
    if (S < 1) S = 1;
    p.S = S;
    p.dbg = 0;
    if (const char *e = getenv("HVS_DBG")) p.dbg = atoi(e);
    const int grid = p.QT * S;
    const int groups_per_tile = bm / pf;

    void *cand = nullptr, *bufA = nullptr, *bufB = nullptr;
    const size_t cand_bytes = (size_t)grid * groups_per_tile * cap * 8;
    const size_t part_bytes = (size_t)nq * S * k * 8;
    const int P2 = (S + 31) / 32;
    HVS_TRY(hvs_ws_get(idx, 1, cand_bytes + part_bytes + 512, &cand));
    HVS_TRY(hvs_ws_get(idx, 2, (size_t)nq * P2 * k * 8 + 256, &bufB));
    bufA = (char *)cand + ((cand_bytes + 255) & ~(size_t)255);
    int *overflow = (int *)((char *)bufB + (((size_t)nq * P2 * k * 8 + 15) & ~(size_t)15));
    HVS_HIP(hipMemsetAsync(overflow, 0, 4, idx->stream));
    p.cand = (uint64_t *)cand;
    p.partial = (uint64_t *)bufA;
    p.overflow = overflow;

    HVS_TRY(hvs_prof_begin(idx));
    if (idx->dtype == HVS_F32) HVS_TRY((launch_search<HVS_F32, Tile128>(pf, grid, idx->stream, p)));
    else if (idx->dtype == HVS_F16 && !big) HVS_TRY((launch_search<HVS_F16, Tile128>(pf, grid, idx->stream, p)));
    else if (idx->dtype == HVS_F16) HVS_TRY((launch_search<HVS_F16, Tile256>(pf, grid, idx->stream, p)));
    else if (!big) HVS_TRY((launch_search<HVS_BF16, Tile128>(pf, grid, idx->stream, p)));
    else HVS_TRY((launch_search<HVS_BF16, Tile256>(pf, grid, idx->stream, p)));
    HVS_TRY(hvs_prof_end(idx, HVS_PROF_MFMA_SEARCH));
    HVS_TRY(hvs_prof_begin(idx));
    HVS_TRY(hvs_reduce_lists(idx, (uint64_t *)bufA, (uint64_t *)bufB, nq, S, k, d_dist, d_ids, d_count));
    HVS_TRY(hvs_prof_end(idx, HVS_PROF_MERGE));
    int h_overflow = 0;
    HVS_HIP(hipMemcpyAsync(&h_overflow, overflow, 4, hipMemcpyDeviceToHost, idx->stream));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    HVS_REQUIRE(h_overflow == 0, "hvs_search: internal candidate array overflow (results would be incomplete)");
    return 0;
}
