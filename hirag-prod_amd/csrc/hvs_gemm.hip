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

#include <vector>

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
    unsigned long long *prof;  // timing experiments only: per-phase cycle sums of wave 0 lanes (HVS_DBG & 512)
    const uint32_t *remap;  // optional: X holds a compacted copy; remap[r] = row position of its row r in the index
    float *dist_out;  // distance-column kernels: LEAST over all A rows of every corpus row, [row_end - row_begin]
    const float *tau0;  // optional, [nq_out]: an upper bound on every query's k-th distance known before the launch (the k-th
                        // distance over a sample of the rows); a block then starts from min(thr, tau0) instead of thr
    int split_steps;  // 0: plain operands.  > 0: both operands hold rows [hi slab | lo slab] of split_steps k-steps each
                      // and the K loop runs 3 split_steps steps: hi.hi, hi.lo, lo.hi (see hvs_split.hip)
    unsigned int *lockstep;  // optional: one arrival counter per (XCD, corpus split) -- see lockstep_arrive; zeroed before the launch
    // optional (kernels built with XCHG): the blocks of a query tile pool what their first corpus tile shows -- see exchange_bound
    uint32_t *xchg_min;       // [S * xchg_w][nq_out] ordered images of the smallest distance per (block, wave column); all ones before the launch
    int xchg_w;               // values a block publishes per query: one per wave column of its tile (T::WN), or 1 (their minimum)
                              // where S * WN would pass the 1024 values one wave ranks
    uint32_t *xchg_tau;       // [nq_out] float bits of the bound derived from them; all ones (a NaN) before the launch
    // later rounds of the exchange (exchange_candidates): at tile XCHG_TILE_2 and XCHG_TILE_3 of a block's stream every block
    // publishes its xchg_r smallest candidate distances per query; round j = 1, 2 uses xchg_min2 / xchg_tau2 + (j - 1) * stride
    uint32_t *xchg_min2;      // [2][1024][nq_out], all ones before the launch
    uint32_t *xchg_tau2;      // [2][nq_out]
    int xchg_r;               // values per block, query and later round (a power of two, S * xchg_r <= 1024)
    int xchg_rounds;          // 1: the first tile only; 2, 3: the later rounds too
    int xchg_t2, xchg_t3;     // the tiles (counted from 1 in a block's stream) after which the later rounds run
    unsigned int *pooled_count;  // optional (kernels built with XCHG): [nq_out] keys in a query's pooled list, zeroed before the launch;
                                 // `partial` then holds one list of S * k slots per query that every block appends its survivors to
    int dbg;        // timing experiments only (env HVS_DBG; results are wrong with 4 or 8): 4 no k-step barrier, 8 no epilogue,
                    // 512 per-phase cycle counts printed to stderr
};

// ---- candidate arrays: selection without sorting ------------------------------------------------
// A wave holds a query's candidate array (up to 64 R 64-bit keys, all distinct: the low word is the row
// position) in registers, slot i = e * 64 + lane in own[e], unused slots = HVS_KEY_EMPTY.
template <int R> __device__ __forceinline__ void cand_load(uint64_t (&own)[R], const uint64_t *arr, int cnt, int lane) {
#pragma unroll
    for (int e = 0; e < R; e++) {
        const int i = e * 64 + lane;
        own[e] = i < cnt ? arr[i] : HVS_KEY_EMPTY;
    }
}

// The k-th smallest key (k >= 1, at least k real keys present), found bit by bit from the top: the answer
// is the smallest T with #{key <= T} >= k.  First the 32 distance bits; if exactly one key carries the
// k-th distance that key is the answer, otherwise (ties at the k-th distance) the 32 row bits follow.
// Rounds of R compares + ballots, no data movement.
template <int R> __device__ __forceinline__ uint64_t cand_kth(const uint64_t (&own)[R], int k) {
    uint32_t hi[R];
#pragma unroll
    for (int e = 0; e < R; e++) hi[e] = (uint32_t)(own[e] >> 32);
    uint32_t t = 0;
    for (int bit = 31; bit >= 0; bit--) {
        const uint32_t trial = t | ((1u << bit) - 1u);  // this bit 0, everything below it 1
        int c = 0;
#pragma unroll
        for (int e = 0; e < R; e++) c += __popcll(__ballot(hi[e] <= trial));
        if (c < k) t |= 1u << bit;
    }
    // t = the k-th smallest distance image
    int ties = 0;
    uint64_t at = 0;  // the largest key carrying distance t
#pragma unroll
    for (int e = 0; e < R; e++) {
        const uint64_t m = __ballot(hi[e] == t);
        ties += __popcll(m);
        if (m) {
            const int src = 63 - __builtin_clzll((unsigned long long)m);  // within a register, higher lanes hold later slots -- any tie will do when it is alone
            const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)own[e], src);
            at = ((uint64_t)t << 32) | lo;
        }
    }
    if (ties == 1) return at;
    uint64_t full = (uint64_t)t << 32;
    for (int bit = 31; bit >= 0; bit--) {
        const uint64_t trial = full | ((1ull << bit) - 1ull);
        int c = 0;
#pragma unroll
        for (int e = 0; e < R; e++) c += __popcll(__ballot(own[e] <= trial));
        if (c < k) full |= 1ull << bit;
    }
    return full;
}

// Write the keys <= kth back to the front of arr (any order); returns how many there are.
template <int R> __device__ __forceinline__ int cand_keep(const uint64_t (&own)[R], uint64_t kth, uint64_t *arr, int lane) {
    int base = 0;
#pragma unroll
    for (int e = 0; e < R; e++) {
        const bool keep = own[e] <= kth;
        const uint64_t m = __ballot(keep);
        if (keep) arr[base + __popcll(m & ((1ull << lane) - 1ull))] = own[e];
        base += __popcll(m);
    }
    return base;
}

// Reduce arr[0, cnt) (cnt > k) to its k smallest keys, in place and unordered; returns the k-th smallest.
template <int R> __device__ __forceinline__ uint64_t cand_reduce_r(uint64_t *arr, int cnt, int k, int lane) {
    uint64_t own[R];
    cand_load<R>(own, arr, cnt, lane);
    const uint64_t kth = cand_kth<R>(own, k);
    cand_keep<R>(own, kth, arr, lane);
    return kth;
}
__device__ __forceinline__ uint64_t cand_reduce(uint64_t *arr, int cnt, int k, int lane) {
    // registers for the occupied part of the array only: most compactions see one or two hundred keys
    if (cnt <= 64) return cand_reduce_r<1>(arr, cnt, k, lane);
    if (cnt <= 128) return cand_reduce_r<2>(arr, cnt, k, lane);
    if (cnt <= 256) return cand_reduce_r<4>(arr, cnt, k, lane);
    if (cnt <= CAND_SMALL) return cand_reduce_r<CAND_SMALL / 64>(arr, cnt, k, lane);
    return cand_reduce_r<CAND_MAX / 64>(arr, cnt, k, lane);
}

// Ascending bitonic sort of 64 * KL keys held one slot per (register, lane), slot i = e * 64 + lane.
__device__ __forceinline__ uint64_t sort_shfl_xor(uint64_t v, int mask) {
    const uint32_t lo = __shfl_xor((uint32_t)v, mask);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), mask);
    return ((uint64_t)hi << 32) | lo;
}
template <int KL> __device__ __forceinline__ void bitonic_sort(uint64_t (&a)[KL], int lane) {
    constexpr int N = 64 * KL;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j >= 64) {  // partner slot is another register of the same lane
                const int se = j >> 6;
#pragma unroll
                for (int e = 0; e < KL; e++) {
                    if ((e & se) == 0) {
                        const bool up = k >= N || ((e & (k >> 6)) == 0);
                        const uint64_t x = a[e], y = a[e + se];
                        const bool sw = up ? (y < x) : (x < y);
                        a[e] = sw ? y : x;
                        a[e + se] = sw ? x : y;
                    }
                }
            } else {  // partner slot is the same register of lane ^ j
#pragma unroll
                for (int e = 0; e < KL; e++) {
                    const bool up = k >= N ? true : (k >= 64 ? ((e & (k >> 6)) == 0) : ((lane & k) == 0));
                    const bool lower = (lane & j) == 0;
                    const uint64_t o = sort_shfl_xor(a[e], j);
                    const uint64_t mn = a[e] < o ? a[e] : o, mx = a[e] < o ? o : a[e];
                    a[e] = (lower == up) ? mn : mx;
                }
            }
        }
    }
}

// Lower bound on dot * inv_row that a row needs for its distance to be below tau, loosened so that
// it never rejects a row the exact fp32 test (hvs_cos_dist(...) < tau) would accept.
__device__ __forceinline__ float prefilter_bound(float tau, float q_inv) {
    const float t = ((1.0f - tau) - 4e-7f) / q_inv;
    return t - fabsf(t) * 4e-7f;
}

// Per-block LDS state of a search kernel.
struct BlockState {
    float *tau;                  // [BM / FOLD] admission bound per folded query (strict <)
    int *cnt;                    // [BM / FOLD] entries in the query's candidate array
    float *qinv;                 // [BM] 1/|q| per A row
    float *bound;                // [BM] prefilter bound per A row
    uint64_t *my_cand;           // this block's candidate arrays [BM / FOLD][cap] (global memory)
};

// block -> (query tile, corpus split).  Blocks are dealt to the 8 XCDs round-robin (b % 8); each XCD gets
// a rectangle of (QT / gq) query tiles x (S / gs) splits, gq = gcd(QT, 8), gs = 8 / gq, with as few
// query tiles as possible: their A slabs (re-read for every corpus tile) then stay in that XCD's L2,
// and only the corpus stream comes from HBM.
// Returns the number of blocks that share this block's XCD and corpus split (0: no XCD-aware mapping).
__device__ __forceinline__ int block_coords(const SearchParams &p, int &qt, int &s) {
    if ((p.S & 7) == 0) {
        const int gq = (p.QT & 7) == 0 ? 8 : ((p.QT & 3) == 0 ? 4 : ((p.QT & 1) == 0 ? 2 : 1));
        const int gs = 8 / gq;
        const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
        const int qn = p.QT / gq, sn = p.S / gs;
        qt = (xcd % gq) * qn + t % qn;
        s = (xcd / gq) * sn + t / qn;
        return qn;
    }
    qt = blockIdx.x % p.QT;
    s = blockIdx.x / p.QT;
    return 0;
}

template <class T> __device__ __forceinline__ void block_state_init(const SearchParams &p, const BlockState &b, int64_t a_row0, int tid, int fold) {
    for (int i = tid; i < T::BM; i += T::GT) {
        // tau by folded query (the first BM / fold entries are used), bound and 1/|q| by A row
        const int64_t qo = a_row0 / fold + i;
        float t = p.thr;
        if (p.tau0 && qo < p.nq_out) t = fminf(t, p.tau0[qo]);
        b.tau[i] = t;
        b.cnt[i] = 0;
        const int64_t ar = a_row0 + i;
        const float qi = ar < p.nA ? p.q_inv[ar] : 0.0f;
        b.qinv[i] = qi;
        float tr = p.thr;
        if (p.tau0 && ar < p.nA) tr = fminf(tr, p.tau0[ar / fold]);
        b.bound[i] = ar < p.nA ? prefilter_bound(tr, qi) : INFINITY;  // padding rows of the A tile never pass
    }
}

// Per-lane row data of a block tile, fetched at the START of the tile's K loop so that the latency is long
// gone when the epilogue needs it: 1/|row| of the lane's NT rows, or NaN for a row that is out of range or
// filtered out (group code, bitmask, tombstone) -- a NaN fails every comparison of the epilogue.
template <class T>
__device__ __forceinline__ void load_row_meta(const SearchParams &p, int64_t tile, int wn, int lane, float (&invn2)[T::NT]) {
    const int64_t n_base = p.row_begin + tile * T::BN + wn * (T::NT * 32) + (lane & 31);
#pragma unroll
    for (int nt = 0; nt < T::NT; nt++) {
        const int64_t row = n_base + nt * 32;
        float invn = __builtin_nanf("");
        if (row < p.row_end) {
            bool ok = true;
            if (p.group_filter >= 0) ok = p.groups[row] == p.group_filter;
            if (ok && p.mask) ok = (p.mask[row >> 5] >> (row & 31)) & 1u;
            if (ok && p.deleted) ok = !((p.deleted[row >> 5] >> (row & 31)) & 1u);
            if (ok) invn = p.inv_norm[row];
        }
        invn2[nt] = invn;
    }
}

// counter[0]++ in LDS, the old value returned (16-bit kernels; the fp32 kernel, whose k-steps are four times as long, measured
// 1.3 % slower with it and keeps atomicAdd).  Written as the instruction itself: for an atomicAdd the compiler first
// waits for every outstanding LDS-DMA fetch (vmcnt(0): it cannot tell the stage a fetch fills from the counters, both
// lie in the block's one dynamic LDS array), i.e. the first survivor of a tile stalled its wave for the whole round trip
// of the next tile's first k-step, which the rest of the epilogue is there to cover.
__device__ __forceinline__ int lds_count_up(int *counter) {
    const uint32_t at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)counter;
    int old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(at), "v"(1) : "memory");
    return old;
}

// Epilogue of one block tile: prefilter every score against its query's bound, exact test for the
// survivors, append to the query's candidate array.
template <int FOLD, class T, bool BARE_COUNT = false>
__device__ __forceinline__ void tile_epilogue(const SearchParams &p, const BlockState &b, f32x16 (&acc)[T::MT][T::NT],
                                              const float (&invn2)[T::NT], int64_t tile, int64_t tile_first, int wm, int wn,
                                              int lane) {
    constexpr int MT = T::MT, NT = T::NT;
    const int h = lane >> 5;
    const int64_t n_base = p.row_begin + tile * T::BN + wn * (NT * 32) + (lane & 31);
    // opaque to the optimiser: otherwise every per-group address below is hoisted out of the caller's tile loop
    // and kept in registers (or scratch) across the whole MFMA loop
    int a_base = wm * (MT * 32) + 4 * h;
    asm volatile("" : "+v"(a_base));
    // Two levels.  Level 1, per 32 x 32 MFMA tile: the largest of the lane's 16 scores against the smallest
    // of the 16 bounds -- one v_max3 chain, one multiply, one compare per tile; it is used once the block has
    // seen enough rows for most tiles to hold no survivor (about k / rows_seen of the scores survive).
    // Level 2, per group of four scores (four adjacent queries): the test proper.
    const bool level1 = p.tau0 != nullptr || p.xchg_min != nullptr || (int64_t)p.k * 512 < (tile - tile_first) * T::BN;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        f32x4 bnd[4];
#pragma unroll
        for (int gq = 0; gq < 4; gq++) bnd[gq] = *reinterpret_cast<const f32x4 *>(b.bound + a_base + mt * 32 + 8 * gq);
        float bmin = INFINITY;
        if (level1) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) bmin = fminf(bmin, fminf(fminf(bnd[gq][0], bnd[gq][1]), fminf(bnd[gq][2], bnd[gq][3])));
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            if (level1) {
                float m = acc[mt][nt][0];
#pragma unroll
                for (int e = 1; e < 16; e++) m = fmaxf(m, acc[mt][nt][e]);
                if (!__ballot(m * invn2[nt] > bmin)) continue;
            }
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const int a_loc = a_base + mt * 32 + 8 * gq;  // A row of register 4*gq (+0..3)
                bool any = false;
#pragma unroll
                for (int e = 0; e < 4; e++) any |= acc[mt][nt][4 * gq + e] * invn2[nt] > bnd[gq][e];
                if (__ballot(any)) {  // wave-uniform and rare once tau has tightened
                    int av = a_loc;
                    asm volatile("" : "+v"(av));  // keep this branch's address arithmetic inside it
                    const f32x4 qi = *reinterpret_cast<const f32x4 *>(b.qinv + av);
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
                        if (any && d[e] < b.tau[qv]) {
                            const int pos = BARE_COUNT ? lds_count_up(&b.cnt[qv]) : atomicAdd(&b.cnt[qv], 1);
                            if (pos < p.cap) {
                                // the row as this launch numbers it; a dense copy's numbers are mapped back when the lists
                                // are published (the map is monotone, so ties keep their order) -- a load here would make
                                // the store wait for every LDS-DMA fetch in flight as well
                                b.my_cand[(size_t)qv * p.cap + pos] = hvs_make_key(hvs_ord(d[e]), (uint32_t)(n_base + nt * 32));
                            }
                            else *p.overflow = 1;
                        }
                    }
                }
            }
        }
    }
}

// Compaction after a tile's epilogue (and a block barrier): arrays that could overflow during the next
// tile, and arrays that have grown enough for a fresher tau to pay (a stale tau admits ~count/k times too
// many survivors).  Wave w owns the queries with index % NW == w: a static split, so no wave's decision
// depends on counters another wave is resetting at the same time.
template <int FOLD, class T>
__device__ __forceinline__ void compact_lists(const SearchParams &p, const BlockState &b, int wave, int lane) {
    constexpr int groups_per_tile = T::BM / FOLD;
    const int eager = p.k * 2 > 32 ? p.k * 2 : 32;
    const int trigger = eager < p.cap - T::BN ? eager : p.cap - T::BN;
    for (int base = 0; base < groups_per_tile; base += 64) {
        const int ql = base + lane;
        const bool need = ql < groups_per_tile && (ql & (T::NW - 1)) == wave && b.cnt[ql] > trigger;
        uint64_t pending = __ballot(need);
        while (pending) {
            const int bit = __ffsll((unsigned long long)pending) - 1;
            pending &= pending - 1;
            const int q = base + bit;
            int c = b.cnt[q];
            c = c > p.cap ? p.cap : c;
            if (c <= p.k) continue;  // (cannot happen: trigger >= k)
            uint64_t *arr = b.my_cand + (size_t)q * p.cap;
            const uint64_t kth = cand_reduce(arr, c, p.k, lane);
            const int nc = p.k;  // keys are distinct, so exactly k are <= the k-th
            const float tau = hvs_unord((uint32_t)(kth >> 32));
            if (lane == 0) {
                b.cnt[q] = nc;
                b.tau[q] = tau;
            }
            if (lane < FOLD) b.bound[q * FOLD + lane] = prefilter_bound(tau, b.qinv[q * FOLD + lane]);
        }
    }
}

// Every folded query's best k, ascending, padded, to partial[query][split][k].  KL = registers per lane of the sort (64 KL >= k).
template <int FOLD, class T, int KL>
__device__ __forceinline__ void publish_lists_kl(const SearchParams &p, const BlockState &b, int q_out0, int s, int wave, int lane) {
    constexpr int groups_per_tile = T::BM / FOLD;
    for (int ql = wave; ql < groups_per_tile; ql += T::NW) {
        const int qo = q_out0 + ql;
        if (qo >= p.nq_out) break;
        int c = b.cnt[ql];
        c = c > p.cap ? p.cap : c;
        uint64_t *arr = b.my_cand + (size_t)ql * p.cap;
        if (c > p.k) {  // keep the k smallest (unordered), through the array itself
            cand_reduce(arr, c, p.k, lane);
            c = p.k;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // this wave's stores before its reloads below
        }
        uint64_t a[KL];
#pragma unroll
        for (int e = 0; e < KL; e++) {
            const int i = e * 64 + lane;
            a[e] = i < c ? arr[i] : HVS_KEY_EMPTY;
        }
        // with a good starting bound most lists of a short stream hold a handful of keys: one key per lane sorts in a
        // sixth of the steps of the full network, an empty or single-key list needs none
        if (c > 64) bitonic_sort<KL>(a, lane);
        else if (c > 1) {
            uint64_t a1[1] = {a[0]};
            bitonic_sort<1>(a1, lane);
            a[0] = a1[0];
        }
        uint64_t *dst = p.partial + ((size_t)qo * p.S + s) * p.k;
#pragma unroll
        for (int e = 0; e < KL; e++) {
            const int i = e * 64 + lane;
            uint64_t key = a[e];
            if (p.remap && key != HVS_KEY_EMPTY) key = (key & 0xFFFFFFFF00000000ull) | p.remap[(uint32_t)key];  // rows of a dense copy -> table positions
            if (i < p.k) dst[i] = key;  // ascending, HVS_KEY_EMPTY padding sorts last
        }
    }
}
template <int FOLD, class T>
__device__ __forceinline__ void publish_lists(const SearchParams &p, const BlockState &b, int q_out0, int s, int wave, int lane) {
    if (p.k <= 128) publish_lists_kl<FOLD, T, 2>(p, b, q_out0, s, wave, lane);
    else publish_lists_kl<FOLD, T, MFMA_MAX_K / 64>(p, b, q_out0, s, wave, lane);
}

// ---- a bound every block starts from, pooled from the first corpus tile of all blocks ----------------------------
// A block tightens tau from the rows it has seen itself.  When a table (or the row range a group filter selects) is cut
// into so many splits that a block sees only a few tiles, tau never gets anywhere: every score is a survivor, every
// array is ranked after every tile, and the selection costs several times the contraction (256 queries x 195 k rows of
// 768 bf16, k = 100: 0.62 ms, of which the K loops are 18 %).  All blocks of a launch are resident at once (the grid is
// sized to the chip), so they can pool what their FIRST tile shows before any of them runs its epilogue:
//   1. every wave forms, per query, the smallest distance over its rows of the tile (a min over the accumulator
//      registers of a lane and over the 32 lanes of a half: no candidate array, no ranking) and publishes it;
//   2. the S x WN published values of a query stand for disjoint sets of rows, so their k-th smallest bounds the
//      k-th distance of the whole table from above -- one wave per query finds it (bitwise search, as cand_kth does)
//      and publishes the smallest float above it; the queries of a tile are dealt round-robin to the tile's S blocks;
//   3. every block starts its first epilogue from min(thr, that bound): with 256 blocks of 256 rows the bound sits
//      near the k / 65 536 quantile instead of +inf.
// The accumulators of the first tile stay in registers across the two waits.  Both waits are bounded (blocks of
// another kernel may hold CUs): a block that gives up goes on with what is published so far -- slots nobody has
// filled read as +inf / NaN, so whatever a block derives is still an upper bound, and with nothing it keeps thr.
// What is exchanged travels in agent-scope atomics, which are performed where all XCDs see them, and there are no
// arrival counters: a slot nobody has filled yet reads all ones, which no published value is, so a reader simply looks
// again (a bounded number of times -- blocks of another kernel may hold CUs -- after which it goes on with what is there).
#define XCHG_SPINS 400
// vals[i * stride] for i < m: the k-th smallest, all ones when fewer than k are there
template <int R> __device__ __forceinline__ uint32_t xchg_kth(const uint32_t *vals, int64_t stride, int m, int k, int lane) {
    uint32_t own[R];
    for (int spin = 0; spin < XCHG_SPINS; spin++) {
        bool missing = false;
#pragma unroll
        for (int e = 0; e < R; e++) {
            const int i = e * 64 + lane;
            own[e] = i < m ? __hip_atomic_load(vals + (size_t)i * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            missing |= own[e] == 0xFFFFFFFFu;
        }
        if (!__ballot(missing)) break;
        __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int e = 0; e < R; e++)
        if (e * 64 + lane >= m) own[e] = 0xFFFFFFFFu;
    uint32_t t = 0;  // the smallest T with #{value <= T} >= k
    for (int bit = 31; bit >= 0; bit--) {
        const uint32_t trial = t | ((1u << bit) - 1u);
        int c = 0;
#pragma unroll
        for (int e = 0; e < R; e++) c += __popcll(__ballot(own[e] <= trial));
        if (c < k) t |= 1u << bit;
    }
    return t;
}

// max over the 32 lanes of a wave half, left in the half's last lane (31 / 63): four row_shr steps inside each row of 16
// lanes, then row_bcast:15 hands row 0's result to row 1 (and row 2's to row 3).  A lane without a source keeps its own.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_max_step(float v) {
    const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return fmaxf(v, __int_as_float(o));
}
__device__ __forceinline__ float half_max(float v) {
    v = dpp_max_step<0x111, 0xF>(v);  // row_shr:1
    v = dpp_max_step<0x112, 0xF>(v);  // row_shr:2
    v = dpp_max_step<0x114, 0xF>(v);  // row_shr:4
    v = dpp_max_step<0x118, 0xF>(v);  // row_shr:8
    v = dpp_max_step<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    return v;
}

// scratch: T::WN * T::BM floats of LDS nobody else uses until the next block barrier after the call
template <int FOLD, class T>
__device__ __forceinline__ void exchange_bound(const SearchParams &p, const BlockState &b, const f32x16 (&acc)[T::MT][T::NT],
                                               const float (&invn2)[T::NT], float *scratch, int qt, int s, int64_t a_row0, int wm, int wn,
                                               int lane, int wave, int tid) {
    constexpr int MT = T::MT, NT = T::NT, BMQ = T::BM / FOLD;
    const int h = lane >> 5;
    const int W = p.xchg_w;  // T::WN or 1
    const int M = p.S * W;
    const int64_t q0 = a_row0 / FOLD;
    __syncthreads();  // every wave is done with the fragments of the stage that now serves as scratch
    // 1. smallest distance per query over this wave's rows of the tile -> scratch[wn][query of the tile].  The distance
    //    falls as dot / |row| grows and 1/|q| is a positive factor, so the largest dot / |row| over the rows is taken first
    //    (one multiply per score, a max over the lane's rows, five DPP steps over the 32 lanes of a half: lanes 31 and 63
    //    end up with it) and turned into a distance once; a NaN (filtered row, zero norm) drops out of every max.
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const int a_loc = wm * (MT * 32) + mt * 32 + 8 * gq + 4 * h;
            const f32x4 qi = *reinterpret_cast<const f32x4 *>(b.qinv + a_loc);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float m = -INFINITY;
#pragma unroll
                for (int nt = 0; nt < NT; nt++) m = fmaxf(m, acc[mt][nt][4 * gq + e] * invn2[nt]);
                m = half_max(m);
                v[e] = hvs_cos_dist(m, 1.0f, qi[e]);  // (m * 1) * 1/|q|: the value the epilogue forms for that row
            }
            if (FOLD == 2) {
                v[0] = fminf(v[0], v[1]);
                v[2] = fminf(v[2], v[3]);
            } else if (FOLD == 4) {
                v[0] = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
            }
            if ((lane & 31) == 31) {
#pragma unroll
                for (int e = 0; e < 4; e += FOLD) scratch[wn * BMQ + (a_loc + e) / FOLD] = v[e];
            }
        }
    }
    __syncthreads();
    // the block's WN x BMQ values, consecutive queries in consecutive lanes: xchg_min[(s * WN + wn)][query]
    for (int i = tid; i < W * BMQ; i += T::GT) {
        const int w = i / BMQ, ql = i - w * BMQ;
        float m = scratch[i];
        if (W == 1)
            for (int o = 1; o < T::WN; o++) m = fminf(m, scratch[o * BMQ + ql]);
        if (q0 + ql < p.nq_out)
            __hip_atomic_store(p.xchg_min + (size_t)(s * W + w) * p.nq_out + q0 + ql, hvs_ord(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // 2. this block's share of the tile's queries: the k-th smallest of the M published values, as soon as they are there
    for (int ql = s + wave * p.S; ql < BMQ; ql += T::NW * p.S) {
        const int64_t qo = q0 + ql;
        if (qo >= p.nq_out) break;
        const uint32_t *vals = p.xchg_min + qo;
        uint32_t t;
        if (M <= 64) t = xchg_kth<1>(vals, p.nq_out, M, p.k, lane);
        else if (M <= 128) t = xchg_kth<2>(vals, p.nq_out, M, p.k, lane);
        else if (M <= 256) t = xchg_kth<4>(vals, p.nq_out, M, p.k, lane);
        else if (M <= 512) t = xchg_kth<8>(vals, p.nq_out, M, p.k, lane);
        else t = xchg_kth<16>(vals, p.nq_out, M, p.k, lane);
        if (lane == 0) {
            float bound = __builtin_nanf("");  // (0x7FC00000: not the all-ones of an unfilled slot)
            if (t < hvs_ord(INFINITY)) {
                const float d = hvs_unord(t);  // admission is strict: the smallest float above the k-th value
                bound = d == 0.0f ? 1.4e-45f : (d > 0.0f ? __uint_as_float(__float_as_uint(d) + 1u) : __uint_as_float(__float_as_uint(d) - 1u));
            }
            __hip_atomic_store(p.xchg_tau + qo, __float_as_uint(bound), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // 3. every query of the tile starts from its bound, as soon as that is there
    for (int i = tid; i < BMQ; i += T::GT) {
        if (q0 + i >= p.nq_out) continue;
        uint32_t bits = 0xFFFFFFFFu;
        for (int spin = 0; spin < XCHG_SPINS; spin++) {
            bits = __hip_atomic_load(p.xchg_tau + q0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bits != 0xFFFFFFFFu) break;
            __builtin_amdgcn_s_sleep(8);
        }
        const float t = __uint_as_float(bits);
        if (t < b.tau[i]) b.tau[i] = t;  // false for a NaN (fewer than k values, or a slot nobody filled in time)
    }
    __syncthreads();
    for (int i = tid; i < T::BM; i += T::GT)
        if (a_row0 + i < p.nA && b.tau[i / FOLD] < p.thr) b.bound[i] = prefilter_bound(b.tau[i / FOLD], b.qinv[i]);
    __syncthreads();
}

// Later rounds.  The bound of the first round sits near the k / (S x BN) quantile, and a block that sees a few hundred
// tiles never collects the 2 k candidates a ranking of its own needs, so tau stays there for the whole stream (cfg5 shard,
// 256 queries, k = 100: 0.15 % of the scores survive the tile tests from the first tile to the last, 12 survivor groups
// per wave and tile, 25 % of the launch in epilogues).  But between them the blocks hold, in their candidate arrays, every
// row below that bound seen so far: after t tiles the global k-th smallest of those is the k-th distance over S x BN x t
// rows.  So at tile XCHG_TILE_2 (and XCHG_TILE_3) every block publishes its r smallest candidate distances per query -- a
// lane per query walks its array, which holds a handful of keys -- and the k-th smallest of the S x r published values
// (distinct rows, so an upper bound whatever r is; equal to the exact value unless one block holds more than r of the k
// best) becomes the new bound, found and handed out as in the first round.
#define XCHG_TILE_2 8
#define XCHG_TILE_3 64
template <int FOLD, class T, int R>
__device__ __forceinline__ void exchange_candidates_r(const SearchParams &p, const BlockState &b, int round, int s, int64_t a_row0, int lane,
                                                      int wave, int tid) {
    constexpr int BMQ = T::BM / FOLD;
    const int M = p.S * R;
    const int64_t q0 = a_row0 / FOLD;
    uint32_t *mins = p.xchg_min2 + (size_t)(round - 1) * 1024 * p.nq_out;
    uint32_t *taus = p.xchg_tau2 + (size_t)(round - 1) * p.nq_out;
    __syncthreads();  // the arrays as the tile's epilogue and rankings left them
    for (int base = wave * 64; base < BMQ; base += T::NW * 64) {
        const int ql = base + lane;
        const int64_t qo = q0 + ql;
        if (ql >= BMQ || qo >= p.nq_out) continue;
        int c = b.cnt[ql];
        c = c > p.cap ? p.cap : c;
        const uint64_t *arr = b.my_cand + (size_t)ql * p.cap;
        uint32_t best[R];
#pragma unroll
        for (int i = 0; i < R; i++) best[i] = 0xFF800000u;  // the image of +inf: published, but no row
        for (int j = 0; j < c; j++) {
            uint32_t v = (uint32_t)(arr[j] >> 32);
#pragma unroll
            for (int i = 0; i < R; i++) {  // keeps best[] ascending
                const uint32_t lo = v < best[i] ? v : best[i];
                v = v < best[i] ? best[i] : v;
                best[i] = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < R; i++)
            __hip_atomic_store(mins + (size_t)(s * R + i) * p.nq_out + qo, best[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int ql = s + wave * p.S; ql < BMQ; ql += T::NW * p.S) {
        const int64_t qo = q0 + ql;
        if (qo >= p.nq_out) break;
        const uint32_t *vals = mins + qo;
        uint32_t t;
        if (M <= 64) t = xchg_kth<1>(vals, p.nq_out, M, p.k, lane);
        else if (M <= 128) t = xchg_kth<2>(vals, p.nq_out, M, p.k, lane);
        else if (M <= 256) t = xchg_kth<4>(vals, p.nq_out, M, p.k, lane);
        else if (M <= 512) t = xchg_kth<8>(vals, p.nq_out, M, p.k, lane);
        else t = xchg_kth<16>(vals, p.nq_out, M, p.k, lane);
        if (lane == 0) {
            float bound = __builtin_nanf("");
            if (t < hvs_ord(INFINITY)) {
                const float d = hvs_unord(t);
                bound = d == 0.0f ? 1.4e-45f : (d > 0.0f ? __uint_as_float(__float_as_uint(d) + 1u) : __uint_as_float(__float_as_uint(d) - 1u));
            }
            __hip_atomic_store(taus + qo, __float_as_uint(bound), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (int i = tid; i < BMQ; i += T::GT) {
        if (q0 + i >= p.nq_out) continue;
        uint32_t bits = 0xFFFFFFFFu;
        for (int spin = 0; spin < XCHG_SPINS; spin++) {
            bits = __hip_atomic_load(taus + q0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bits != 0xFFFFFFFFu) break;
            __builtin_amdgcn_s_sleep(8);
        }
        const float t = __uint_as_float(bits);
        if (t < b.tau[i]) b.tau[i] = t;
    }
    __syncthreads();
    for (int i = tid; i < T::BM; i += T::GT)
        if (a_row0 + i < p.nA && b.tau[i / FOLD] < p.thr) b.bound[i] = prefilter_bound(b.tau[i / FOLD], b.qinv[i]);
    __syncthreads();
}
template <int FOLD, class T>
__device__ __forceinline__ void exchange_candidates(const SearchParams &p, const BlockState &b, int round, int s, int64_t a_row0, int lane,
                                                    int wave, int tid) {
    if (p.xchg_r >= 16) exchange_candidates_r<FOLD, T, 16>(p, b, round, s, a_row0, lane, wave, tid);
    else if (p.xchg_r >= 8) exchange_candidates_r<FOLD, T, 8>(p, b, round, s, a_row0, lane, wave, tid);
    else if (p.xchg_r >= 4) exchange_candidates_r<FOLD, T, 4>(p, b, round, s, a_row0, lane, wave, tid);
    else if (p.xchg_r >= 2) exchange_candidates_r<FOLD, T, 2>(p, b, round, s, a_row0, lane, wave, tid);
    else exchange_candidates_r<FOLD, T, 1>(p, b, round, s, a_row0, lane, wave, tid);
}

// The other half of a short-stream launch: with a bound from the pooled first tiles a block ends up with a handful of
// survivors per query, and sorting and merging S lists of k slots each per query costs more than the contraction (256
// queries x 195 k rows, k = 100: 65 us of sorting inside the kernel and three merge launches, against 80 us of K loops).
// Every block instead appends what it holds to ONE list per query (an atomic counter hands out the slots; a lane per
// query does the copying, so a wave publishes 64 queries at a time), and select_pooled_kernel ranks each list once.
// The order of a pooled list depends on timing, its contents do not, and the answer is its k smallest keys in order.
template <int FOLD, class T>
__device__ __forceinline__ void publish_pooled(const SearchParams &p, const BlockState &b, int q_out0, int wave, int lane) {
    constexpr int BMQ = T::BM / FOLD;
    const size_t slots = (size_t)p.S * p.k;
    for (int base = wave * 64; base < BMQ; base += T::NW * 64) {
        const int ql = base + lane;
        const int64_t qo = (int64_t)q_out0 + ql;
        const bool valid = ql < BMQ && qo < p.nq_out;
        int c = valid ? b.cnt[ql] : 0;
        c = c > p.cap ? p.cap : c;
        uint64_t longer = __ballot(c > p.k);  // rare: an array that grew past k since its last ranking
        while (longer) {
            const int src = __ffsll((unsigned long long)longer) - 1;
            longer &= longer - 1;
            cand_reduce(b.my_cand + (size_t)(base + src) * p.cap, __shfl(c, src), p.k, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the rankings' stores before the reloads below
        c = c > p.k ? p.k : c;
        unsigned int at = 0;
        if (c > 0) at = atomicAdd(p.pooled_count + qo, (unsigned int)c);
        const uint64_t *arr = b.my_cand + (size_t)ql * p.cap;
        uint64_t *dst = p.partial + (size_t)qo * slots + at;
        for (int j = 0; __ballot(j < c); j++) {
            if (j < c) {
                uint64_t key = arr[j];
                if (p.remap) key = (key & 0xFFFFFFFF00000000ull) | p.remap[(uint32_t)key];  // rows of a dense copy -> table positions
                dst[j] = key;
            }
        }
    }
}

// One wave per query: the k smallest keys of its pooled list, ascending, as distances / ids / count.  Lists of up to
// 1024 keys are ranked in registers in one go; longer ones in rounds that carry the best k so far at the list's front.
template <int KL, int ROUND>
__global__ __launch_bounds__(64) void select_pooled_kernel(uint64_t *__restrict__ lists, const unsigned int *__restrict__ counts, size_t slots,
                                                            int k, const int64_t *__restrict__ ids, float *__restrict__ out_dist,
                                                            int64_t *__restrict__ out_ids, int32_t *__restrict__ out_count) {
    const int q = blockIdx.x, lane = threadIdx.x;
    uint64_t *list = lists + (size_t)q * slots;
    size_t n = counts[q];
    n = n > slots ? slots : n;
    int have = 0;      // keys at the list's front that are the best of everything read so far
    size_t pos = 0;    // next unread key
    while (pos < n) {
        static_assert(ROUND > 64 * KL, "a round must bring new keys in next to the k it carries");
        const int take = (int)((n - pos) < (size_t)(ROUND - have) ? (n - pos) : (size_t)(ROUND - have));
        const int cnt = have + take;
        if (cnt > k) {
            uint64_t own[ROUND / 64];
#pragma unroll
            for (int e = 0; e < ROUND / 64; e++) {
                const int i = e * 64 + lane;
                own[e] = i < have ? list[i] : (i < cnt ? list[pos + (i - have)] : HVS_KEY_EMPTY);
            }
            const uint64_t kth = cand_kth<ROUND / 64>(own, k);
            cand_keep<ROUND / 64>(own, kth, list, lane);
            have = k;
        } else {
            if (pos != (size_t)have)  // (only the first round can end here: have = 0, pos = 0)
                for (int i = lane; i < take; i += 64) list[have + i] = list[pos + i];
            have = cnt;
        }
        pos += take;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    uint64_t a[KL];
#pragma unroll
    for (int e = 0; e < KL; e++) {
        const int i = e * 64 + lane;
        a[e] = i < have ? list[i] : HVS_KEY_EMPTY;
    }
    if (have > 64 || KL == 1) bitonic_sort<KL>(a, lane);
    else {
        uint64_t a1[1] = {a[0]};
        bitonic_sort<1>(a1, lane);
        a[0] = a1[0];
    }
#pragma unroll
    for (int e = 0; e < KL; e++) {
        const int i = e * 64 + lane;
        if (i < k) {
            const bool real = a[e] != HVS_KEY_EMPTY;
            out_dist[(size_t)q * k + i] = real ? hvs_unord((uint32_t)(a[e] >> 32)) : __builtin_inff();
            out_ids[(size_t)q * k + i] = real ? (ids ? ids[(uint32_t)a[e]] : (int64_t)(uint32_t)a[e]) : -1;
        }
    }
    if (lane == 0) out_count[q] = have;
}

// launch state of a search pass: 16-byte words [0, zero16) = 0, [zero16, all16) = all ones
__global__ __launch_bounds__(256) void init_state_kernel(uint4 *__restrict__ p, size_t zero16, size_t all16) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < all16; i += (size_t)gridDim.x * 256) {
        const uint32_t v = i < zero16 ? 0u : 0xFFFFFFFFu;
        p[i] = make_uint4(v, v, v, v);
    }
}

// LDS layout: [2 stages][tau][cnt][qinv][bound]
template <class T> __device__ __forceinline__ BlockState block_state(char *smem, const SearchParams &p, int fold) {
    BlockState b;
    b.tau = reinterpret_cast<float *>(smem + 2 * T::STAGE_BYTES);
    b.cnt = reinterpret_cast<int *>(b.tau + T::BM);
    b.qinv = reinterpret_cast<float *>(b.cnt + T::BM);
    b.bound = b.qinv + T::BM;
    b.my_cand = p.cand + (size_t)blockIdx.x * (T::BM / fold) * p.cap;
    return b;
}

// The timing hooks of this kernel (HVS_DBG bits 4, 8, 512: branches in every k-step, five live 64-bit counters) cost
// 2 % of every launch even when they are off (2M x 1536 x 4096 queries: f16 995 -> 1 015 TFLOP/s, fp32 176.8 -> 173.3 ms),
// so they are compiled in only with -DHVS_KERNEL_HOOKS (make HOOKS=1); the host refuses HVS_DBG without them.
#ifdef HVS_KERNEL_HOOKS
#define HVS_HOOK(p) ((p).dbg)
#else
#define HVS_HOOK(p) 0
#endif

template <int DT, int FOLD, class T, bool XCHG = false>
__global__ __launch_bounds__(T::GT, T::GT <= 256 ? 2 : 1) void mfma_search_kernel(SearchParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // device-only types (buffer descriptors) inside
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = T::BM, BN = T::BN, STAGE_BYTES = T::STAGE_BYTES, MT = T::MT, NT = T::NT;
    char *stages = smem;  // 2 * STAGE_BYTES
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WN, wn = wave % T::WN;
    int qt, s;
    const int members = block_coords(p, qt, s);
    bool in_step = p.lockstep != nullptr && members > 1;
    unsigned int *step_counter = p.lockstep + ((blockIdx.x & 7) * p.S + s);
    const int64_t tile_lo = p.T * s / p.S, tile_hi = p.T * (s + 1) / p.S;
    const int64_t a_row0 = (int64_t)qt * BM;
    const BlockState b = block_state<T>(smem, p, FOLD);
    block_state_init<T>(p, b, a_row0, tid, FOLD);

    f32x16 acc[MT][NT];
    float invn2[NT];
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
        dma_tiles<T>(srcA, srcB, stages, koff_plain(0), wave);
    }
    __syncthreads();

    // (A, X) byte offsets of k-step k inside a row.  Split operands run three slab pairs, (hi, hi), (hi, lo), (lo, hi), of
    // split_steps steps each; plain operands are the first pair of a split that never ends
    const int slab_steps = p.split_steps ? p.split_steps : 0x3fffffff;
    auto step_off = [&](int k) {
        const int slab = (k >= 2 * slab_steps ? 1 : 0) + (k >= slab_steps ? 1 : 0), w = k - slab * slab_steps;
        return KOff{((slab == 2 ? slab_steps : 0) + w) * KB, ((slab == 1 ? slab_steps : 0) + w) * KB};
    };
    // A tile loop around a k loop without tile bookkeeping in it: the steps whose successor lies in the same tile run
    // in the inner loop, the tile's last step (whose LDS-DMA fetches step 0 of the next tile) is written out after it.
    int cur = 0;
#ifdef HVS_KERNEL_HOOKS
    unsigned long long pr_loop = 0, pr_epi = 0, pr_bar = 0, pr_cmp = 0, pr_x = 0, pr_last = (p.dbg & 512) ? clock64() : 0;
#endif
    for (int64_t tile = tile_lo; tile < tile_hi; tile++) {
        load_row_meta<T>(p, tile, wn, lane, invn2);
        if (tid == 0 && p.lockstep != nullptr && members > 1)
            lockstep_arrive(step_counter, (unsigned int)(tile - tile_lo + 1), members, in_step);
        for (int ks = 0; ks + 1 < p.nks; ks++) {
            compute_step<DT, T>(acc, stages + cur * STAGE_BYTES, wm, wn, lane, srcA, srcB, stages + (cur ^ 1) * STAGE_BYTES,
                                step_off(ks + 1), wave);
            if (!(HVS_HOOK(p) & 4)) __syncthreads();  // the DMA of the next stage has landed (vmcnt) for every wave
            cur ^= 1;
        }
        // after the last tile its own step 0 is fetched again (harmless, keeps the step free of branches around the LDS-DMA)
        if (tile + 1 < tile_hi) init_b(tile + 1);
        compute_step<DT, T>(acc, stages + cur * STAGE_BYTES, wm, wn, lane, srcA, srcB, stages + (cur ^ 1) * STAGE_BYTES, step_off(0),
                            wave);
#ifdef HVS_KERNEL_HOOKS
        const bool prof = p.dbg & 512;
        unsigned long long c1 = 0, c2 = 0, c3 = 0;
        if (prof) c1 = clock64();
#endif
        if (XCHG && tile == tile_lo) {
            exchange_bound<FOLD, T>(p, b, acc, invn2, reinterpret_cast<float *>(stages + cur * STAGE_BYTES), qt, s, a_row0, wm, wn, lane, wave, tid);
#ifdef HVS_KERNEL_HOOKS
            if (prof) {
                const unsigned long long cx = clock64();
                pr_x += cx - c1;
                c1 = cx;
            }
#endif
        }
        if (!(HVS_HOOK(p) & 8)) tile_epilogue<FOLD, T, DT != HVS_F32>(p, b, acc, invn2, tile, tile_lo, wm, wn, lane);
        zero_acc(acc);
#ifdef HVS_KERNEL_HOOKS
        if (prof) c2 = clock64();
#endif
        if (!(HVS_HOOK(p) & 4)) __syncthreads();
        cur ^= 1;
#ifdef HVS_KERNEL_HOOKS
        if (prof) c3 = clock64();
#endif
        compact_lists<FOLD, T>(p, b, wave, lane);
        if (XCHG && p.xchg_rounds > 1) {
            const int64_t done = tile - tile_lo + 1;
            if (done == p.xchg_t2 || (done == p.xchg_t3 && p.xchg_rounds > 2))
                exchange_candidates<FOLD, T>(p, b, done == p.xchg_t2 ? 1 : 2, s, a_row0, lane, wave, tid);
        }
#ifdef HVS_KERNEL_HOOKS
        if (prof) {
            const unsigned long long c4 = clock64();
            pr_loop += c1 - pr_last;
            pr_epi += c2 - c1;
            pr_bar += c3 - c2;
            pr_cmp += c4 - c3;
            pr_last = c4;
        }
#endif
        // the next epilogue is a whole K loop (>= 1 barrier) away, so no extra barrier is needed
        // unless the K loop has a single step
        if (p.nks == 1) __syncthreads();
    }
#ifdef HVS_KERNEL_HOOKS
    if ((p.dbg & 512) && lane == 0) {
        atomicAdd(&p.prof[0], pr_loop);
        atomicAdd(&p.prof[1], pr_epi);
        atomicAdd(&p.prof[2], pr_bar);
        atomicAdd(&p.prof[3], pr_cmp);
        atomicAdd(&p.prof[4], pr_x);
    }
    const unsigned long long c_pub = (p.dbg & 512) ? clock64() : 0;
#endif
    __syncthreads();
    if (p.pooled_count) publish_pooled<FOLD, T>(p, b, qt * (BM / FOLD), wave, lane);
    else publish_lists<FOLD, T>(p, b, qt * (BM / FOLD), s, wave, lane);
#ifdef HVS_KERNEL_HOOKS
    if ((p.dbg & 512) && lane == 0) atomicAdd(&p.prof[5], clock64() - c_pub);
#endif
#endif
}

// ---------------------------------------------------------------------------------------------
// single query tile: the corpus streams from HBM
// ---------------------------------------------------------------------------------------------
// With few query embeddings there is one query tile, every block streams its own slab of the corpus from HBM, and
// the two-stage kernel above keeps one k-step in flight: a step then lasts one loaded-HBM latency.  This variant is
// laid out for that regime: a narrow query tile (BM = 32 or 64 rows) against 256 corpus rows, so that most of a
// stage is corpus, and a ring of NS stages whose fetches run NS - 1 k-steps ahead (explicit vmcnt waits instead of a
// full fence at the step barrier).  4 waves of BM x 64, one block per CU.  16-bit operands only: 16-bit tables, and
// the [hi | lo] split copy of fp32 tables (DUAL), for which a stage carries BOTH slabs of the query tile: the hi
// slab of the corpus meets hi and lo of the queries in one step, the lo slab meets hi in another, so every corpus
// byte is fetched once (the three-product order of the large-batch kernel would fetch the hi slab twice, and here
// the corpus stream is the bound).
template <int BM_, bool DUAL_, int WM_, int WN_, int BN_ = 256> struct StreamTile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr bool DUAL = DUAL_;
    static constexpr int NW = WM * WN, GT = 64 * NW;
    static constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
    static constexpr int A_IMAGES = DUAL ? 2 : 1;
    static constexpr int STAGE_BYTES = (A_IMAGES * BM + BN) * KB;
    // LDS-DMA instructions per thread, image and k-step.  A query image smaller than one instruction per thread is
    // fetched by every group of A_THREADS threads alike (the same bytes to the same place): every wave then issues
    // the same number of instructions, which the vmcnt arithmetic needs.
    static constexpr int A_THREADS = BM * KB / 16 < GT ? BM * KB / 16 : GT;
    static constexpr int NA = BM * KB / 16 / A_THREADS, NB = BN * KB / 16 / GT;
    static_assert(BM * KB / 16 % A_THREADS == 0 && BN * KB / 16 % GT == 0, "whole pieces per thread");
};

// ring depth: as many stages as the 160 KB of LDS hold next to the per-query state, at most 4
template <class T> __host__ __device__ constexpr int stream_stages() {
    return T::STAGE_BYTES * 4 + T::BM * 16 + 1024 <= 160 * 1024 - 1024 ? 4 : 3;
}

template <int N>
__device__ __forceinline__ void stream_src_init(TileSrcT<N> &t, const void *base, int64_t row0, int rows_valid, int pitch_bytes, int tid,
                                                int threads) {
    const char *ptr = reinterpret_cast<const char *>(base) + row0 * pitch_bytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ptr >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane(rows_valid * pitch_bytes);
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < N; i++) t.voff[i] = dma_src_offset(tid, i, pitch_bytes, rows_valid - 1, threads);
}

// DISTCOL: no selection at all -- every corpus row's LEAST over the A rows goes to p.dist_out (the projected
// distance column of cross-language search with many sentence embeddings in one pass over the rows).
template <int DT, int FOLD, class T, bool DISTCOL = false>
__global__ __launch_bounds__(T::GT, T::GT == 512 ? 2 : 1) void mfma_stream_kernel(SearchParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = T::BM, BN = T::BN, STAGE_BYTES = T::STAGE_BYTES, MT = T::MT, NT = T::NT, NS = stream_stages<T>();
    constexpr int PIECES = T::A_IMAGES * T::NA + T::NB;
    constexpr int X_IMAGE = T::A_IMAGES * BM * KB;  // the corpus image follows the query image(s)
    static_assert(PIECES * (NS - 1) <= 63, "vmcnt is a 6-bit counter");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int s = blockIdx.x;  // one query tile: the grid is the corpus splits
    const int64_t tile_lo = p.T * s / p.S, tile_hi = p.T * (s + 1) / p.S;
    BlockState b;
    b.tau = reinterpret_cast<float *>(smem + NS * STAGE_BYTES);
    b.cnt = reinterpret_cast<int *>(b.tau + BM);
    b.qinv = reinterpret_cast<float *>(b.cnt + BM);
    b.bound = b.qinv + BM;
    b.my_cand = DISTCOL ? nullptr : p.cand + (size_t)blockIdx.x * (BM / FOLD) * p.cap;
    block_state_init<T>(p, b, 0, tid, FOLD);
    float *xch = b.bound + BM;  // DISTCOL with two wave rows: [WN][NT][32] minima handed from wave row 1 to wave row 0
    static_assert(!DISTCOL || (T::WM <= 2 && BN <= 256), "the exchange area holds one wave row's minima of 256 corpus rows");

    f32x16 acc[MT][NT];
    float invn2[NT];
    float colmin[NT];  // DISTCOL: this lane's rows' minima over the wave's A rows, kept across the step barrier
    int64_t col_tile = 0;
    zero_acc(acc);
    const int64_t total_steps = (tile_hi - tile_lo) * p.nks;
    TileSrcT<T::NA> srcA;
    TileSrcT<T::NB> srcB;
    const int tid_a = tid % T::A_THREADS, wave_a = __builtin_amdgcn_readfirstlane(tid_a >> 6);
    stream_src_init<T::NA>(srcA, p.A, 0, p.nA < BM ? p.nA : BM, p.lda_bytes, tid_a, T::A_THREADS);
    auto init_b = [&](int64_t tile) {
        const int64_t r0 = p.row_begin + tile * BN;
        const int64_t valid = p.row_end - r0;
        stream_src_init<T::NB>(srcB, p.X, r0, valid < BN ? (int)valid : BN, p.ld_bytes, tid, T::GT);
    };
    // All LDS-DMA instructions of one k-step into a stage.  DUAL: steps [0, S) fetch (A hi, A lo | X hi), steps
    // [S, 2 S) fetch (A hi, A hi again (unused) | X lo); the instruction count per step is constant, which the vmcnt
    // arithmetic below relies on.
    auto fetch = [&](char *stage, int ks) {
        int off_a = ks * KB, off_a2 = ks * KB, off_b = ks * KB;
        if (!T::DUAL && p.split_steps) {  // split operands, one query image: hi.hi, hi.lo, lo.hi as in mfma_search_kernel
            const int slab = ks >= 2 * p.split_steps ? 2 : (ks >= p.split_steps ? 1 : 0), w = ks - slab * p.split_steps;
            off_a = ((slab == 2 ? p.split_steps : 0) + w) * KB;
            off_b = ((slab == 1 ? p.split_steps : 0) + w) * KB;
        }
        if (T::DUAL) {
            const int second = ks >= p.split_steps ? 1 : 0, w = ks - second * p.split_steps;
            off_a = w * KB;
            off_a2 = second ? off_a : (p.split_steps + w) * KB;
            off_b = (second * p.split_steps + w) * KB;
        }
#pragma unroll
        for (int i = 0; i < T::NA; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA.rsrc, (lds_void *)(stage + (T::A_THREADS * i + wave_a * 64) * 16), 16, srcA.voff[i], off_a, 0,
                                                     0);
        if (T::DUAL) {
#pragma unroll
            for (int i = 0; i < T::NA; i++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srcA.rsrc, (lds_void *)(stage + BM * KB + (T::A_THREADS * i + wave_a * 64) * 16), 16,
                                                         srcA.voff[i], off_a2, 0, 0);
        }
        // the corpus passes through once and is read by this block alone: non-temporal (aux bit 1), so that it does not push
        // the query tile out of L2 nor wait for room there -- a plain streaming read of this shape reaches 6.2 TB/s, a
        // non-temporal one 7.0 (tools/dma_bw.hip)
#pragma unroll
        for (int i = 0; i < T::NB; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srcB.rsrc, (lds_void *)(stage + X_IMAGE + (T::GT * i + wave * 64) * 16), 16, srcB.voff[i],
                                                     off_b, 0, 2);
    };

    // fetch cursor: the step whose fetch is issued next (runs NS - 1 steps ahead of the compute cursor)
    int64_t f_tile = tile_lo;
    int f_ks = 0;
    int64_t f_step = 0;
    auto fetch_next = [&](int slot) {
        if (f_step < total_steps && f_ks == 0) init_b(f_tile);
        // past the end the last valid step is fetched again into a stage nobody reads: keeps the instruction count
        // per step (and with it the vmcnt arithmetic) constant
        fetch(smem + slot * STAGE_BYTES, f_ks);
        if (f_step < total_steps - 1) {
            f_ks++;
            if (f_ks == p.nks) { f_ks = 0; f_tile++; }
        }
        f_step++;
    };
    if (total_steps > 0) {
#pragma unroll
        for (int d = 0; d < NS - 1; d++) fetch_next(d);
        // the first step's fetch is the oldest: everything issued after it may still be outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (NS - 2)) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the per-query state written above
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const int r = lane & 31, h = lane >> 5;
    const int swz = lds_swz(r);
    int64_t tile = tile_lo;
    int ks = 0;
    int slot = 0;
    bool compact_due = false;
    for (int64_t g = 0; g < total_steps; g++) {
        // the stage read during the previous step is free: every wave has passed that step's barrier
        int fslot = slot + NS - 1;
        fslot = fslot >= NS ? fslot - NS : fslot;
        fetch_next(fslot);
        if (ks == 0) load_row_meta<T>(p, tile, wn, lane, invn2);
        {
            const char *As = smem + slot * STAGE_BYTES + (wm * (MT * 32) + r) * KB;
            const char *Bs = smem + slot * STAGE_BYTES + X_IMAGE + (wn * (NT * 32) + r) * KB;
            auto frag = [&](const char *base, int t, int kc) {
                return *reinterpret_cast<const f32x4 *>(base + t * (32 * KB) + ((kc ^ swz) << 4));
            };
            const bool both = T::DUAL && ks < p.split_steps;  // the second query image holds the lo slab
            f32x4 fa[2][MT], fa2[2][MT], fb[2][NT];
#pragma unroll
            for (int t = 0; t < MT; t++) {
                fa[0][t] = frag(As, t, h);
                if (T::DUAL) fa2[0][t] = frag(As + BM * KB, t, h);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) fb[0][t] = frag(Bs, t, h);
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                if (sl < 3) {  // next slice's fragments land while this slice's MFMAs run
#pragma unroll
                    for (int t = 0; t < MT; t++) {
                        fa[(sl + 1) & 1][t] = frag(As, t, 2 * (sl + 1) + h);
                        if (T::DUAL) fa2[(sl + 1) & 1][t] = frag(As + BM * KB, t, 2 * (sl + 1) + h);
                    }
#pragma unroll
                    for (int t = 0; t < NT; t++) fb[(sl + 1) & 1][t] = frag(Bs, t, 2 * (sl + 1) + h);
                }
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma16<DT>(fa[sl & 1][mt], fb[sl & 1][nt], acc[mt][nt]);
                if (both) {
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
#pragma unroll
                        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma16<DT>(fa2[sl & 1][mt], fb[sl & 1][nt], acc[mt][nt]);
                }
            }
        }
        const bool tile_end = ks == p.nks - 1;
        if (tile_end) {
            if (DISTCOL) {
                // rows on lanes, A rows on accumulator registers: fminf drops the NaN of a zero-norm embedding, a
                // zero-norm row stays NaN; A rows past nA are padding and take no part
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    float m = __builtin_nanf("");
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
#pragma unroll
                        for (int gq = 0; gq < 4; gq++) {
                            const int a_loc = wm * (MT * 32) + mt * 32 + 8 * gq + 4 * h;
                            const f32x4 qi = *reinterpret_cast<const f32x4 *>(b.qinv + a_loc);
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (a_loc + e < p.nA) m = fminf(m, hvs_cos_dist(acc[mt][nt][4 * gq + e], invn2[nt], qi[e]));
                        }
                    m = fminf(m, __shfl_xor(m, 32));  // the other lane half holds the other A rows of the same corpus row
                    colmin[nt] = m;
                    if (T::WM == 2 && wm == 1 && h == 0) xch[(wn * NT + nt) * 32 + r] = m;
                }
                col_tile = tile;
            } else {
                tile_epilogue<FOLD, T, true>(p, b, acc, invn2, tile, tile_lo, wm, wn, lane);
            }
            zero_acc(acc);
        }
        // The next step's fetch (issued NS - 2 steps ago) has landed for this wave, then -- barrier -- for all of them.
        // Vector memory operations retire in order, so allowing only the youngest fetches to be outstanding also
        // covers what was issued before them.  The candidate keys a tile's epilogue stores are ranked one step
        // later (compact_due): by then only the fetch issued after them may still be in flight, and the ring never
        // drains.  A K loop of a single step has no later step before the next epilogue and drains instead.
        if (DISTCOL) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES * (NS - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tile_end && wm == 0 && h == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    float m = colmin[nt];
                    if (T::WM == 2) m = fminf(m, xch[(wn * NT + nt) * 32 + r]);
                    const int64_t row = p.row_begin + col_tile * BN + wn * (NT * 32) + nt * 32 + r;
                    if (row < p.row_end) p.dist_out[row - p.row_begin] = m;
                }
            }
            if (tile_end && p.nks == 1) {  // the exchange area is rewritten a whole K loop later otherwise
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        } else if (p.nks == 1) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            compact_lists<FOLD, T>(p, b, wave, lane);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            if (compact_due) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES * (NS - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (compact_due) compact_lists<FOLD, T>(p, b, wave, lane);
            compact_due = tile_end;
        }
        ks++;
        if (ks == p.nks) { ks = 0; tile++; }
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the block
    __syncthreads();
    if (!DISTCOL) {
        if (p.pooled_count) publish_pooled<FOLD, T>(p, b, 0, wave, lane);
        else publish_lists<FOLD, T>(p, b, 0, s, wave, lane);
    }
#endif
}

// stages + per-query state + (distance-column kernels) the 1 KB exchange area
template <class T> static size_t stream_lds_bytes() { return (size_t)stream_stages<T>() * T::STAGE_BYTES + T::BM * 16 + 1024; }

template <int DT, class T>
static int launch_stream(int pf, int grid, hipStream_t st, const SearchParams &p) {
    const size_t lds = stream_lds_bytes<T>();
#define HVS_LAUNCH_STREAM(F)                                                                                         \
    do {                                                                                                             \
        HVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mfma_stream_kernel<DT, F, T>),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
        mfma_stream_kernel<DT, F, T><<<grid, T::GT, lds, st>>>(p);                                                   \
    } while (0)
    if (pf == 1) HVS_LAUNCH_STREAM(1);
    else if (pf == 2) HVS_LAUNCH_STREAM(2);
    else HVS_LAUNCH_STREAM(4);
#undef HVS_LAUNCH_STREAM
    HVS_HIP(hipGetLastError());
    return 0;
}

// 8 waves per block in every shape (two per SIMD): 32 rows -> 1 x 8 waves of 32 x 32, 64 -> 2 x 4 of 32 x 64, 128 -> 2 x 4 of 64 x 64
template <int DT>
static int launch_stream_bm(int bm, bool dual, int pf, int grid, hipStream_t st, const SearchParams &p) {
    if (dual) {
        if (bm == 32) return launch_stream<DT, StreamTile<32, true, 1, 8>>(pf, grid, st, p);
        return launch_stream<DT, StreamTile<64, true, 2, 4>>(pf, grid, st, p);
    }
    if (bm == 32) return launch_stream<DT, StreamTile<32, false, 1, 8>>(pf, grid, st, p);
    if (bm == 64) return launch_stream<DT, StreamTile<64, false, 2, 4>>(pf, grid, st, p);
    return launch_stream<DT, StreamTile<128, false, 2, 4>>(pf, grid, st, p);
}

template <int DT, class T>
static int launch_distcol(int grid, hipStream_t st, const SearchParams &p) {
    const size_t lds = stream_lds_bytes<T>();
    HVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mfma_stream_kernel<DT, 1, T, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    mfma_stream_kernel<DT, 1, T, true><<<grid, T::GT, lds, st>>>(p);
    HVS_HIP(hipGetLastError());
    return 0;
}

template <int DT>
static int launch_distcol_bm(int bm, bool dual, int grid, hipStream_t st, const SearchParams &p) {
    if (dual) {
        if (bm == 32) return launch_distcol<DT, StreamTile<32, true, 1, 8>>(grid, st, p);
        return launch_distcol<DT, StreamTile<64, true, 2, 4>>(grid, st, p);
    }
    if (bm == 32) return launch_distcol<DT, StreamTile<32, false, 1, 8>>(grid, st, p);
    if (bm == 64) return launch_distcol<DT, StreamTile<64, false, 2, 4>>(grid, st, p);
    return launch_distcol<DT, StreamTile<128, false, 2, 4>>(grid, st, p);
}

#define HVS_COMPACT_MAX_BYTES ((size_t)16 << 30)  // largest dense copy a filtered search may make

// ---------------------------------------------------------------------------------------------
// host side of the search
// ---------------------------------------------------------------------------------------------
int hvs_search_scan(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int32_t group_filter,
                    const uint32_t *d_mask, float thr, int64_t row_begin, int64_t row_end, float *d_dist, int64_t *d_ids,
                    int32_t *d_count, void *lists = nullptr, size_t lists_bytes = 0);  // hvs_scan.hip
int hvs_reduce_lists(hvs_index *idx, uint64_t *bufA, uint64_t *bufB, int nq, int P, int k, float *d_dist,
                     int64_t *d_ids, int32_t *d_count, bool positions);

template <class T> static size_t search_lds_bytes() { return (size_t)2 * T::STAGE_BYTES + T::BM * 16; }

static int padded_fold(int fold) { return fold <= 1 ? 1 : (fold == 2 ? 2 : 4); }

bool hvs_mfma_supported(hvs_index *idx, int nq, int fold, int k) {
    (void)nq;
    (void)idx;
    return fold >= 1 && fold <= 4 && k >= 1 && k <= MFMA_WIDE_K;
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

// tau0[q] = the smallest float above the k-th distance of a finished pass (the main pass admits strictly below its
// bound, and the k-th row itself must pass again), or thr where the pass found fewer than k rows
__global__ void tau_from_topk_kernel(const float *__restrict__ dist, const int32_t *__restrict__ count, int nq, int k, float thr,
                                     float *__restrict__ tau0) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float t = thr;
    if (count[q] >= k) {
        const float d = dist[(size_t)q * k + k - 1];
        const float up = d >= 0.0f ? __uint_as_float(__float_as_uint(d) + 1u) : __uint_as_float(__float_as_uint(d) - 1u);  // next float up
        t = fminf(thr, d == 0.0f ? 1.4e-45f : up);
    }
    tau0[q] = t;
}

template <int DT, class T, bool XCHG = false>
static int launch_search(int pf, int grid, hipStream_t st, const SearchParams &p) {
    const size_t lds = search_lds_bytes<T>();
#define HVS_LAUNCH_SEARCH(F)                                                                                         \
    do {                                                                                                             \
        HVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mfma_search_kernel<DT, F, T, XCHG>),             \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
        mfma_search_kernel<DT, F, T, XCHG><<<grid, T::GT, lds, st>>>(p);                                             \
    } while (0)
    if (pf == 1) HVS_LAUNCH_SEARCH(1);
    else if (pf == 2) HVS_LAUNCH_SEARCH(2);
    else HVS_LAUNCH_SEARCH(4);
#undef HVS_LAUNCH_SEARCH
    HVS_HIP(hipGetLastError());
    return 0;
}

// Corpus splits of a two-stage launch over `tiles` corpus tiles with qt query tiles: fill every block slot of the chip
// (2 per CU for the small tiles, 1 for the large), keep the count a multiple of 8 (XCD affinity) and every split non-empty.
static int two_stage_splits(int qt, int64_t tiles, int slots) {
    int S = (slots + qt - 1) / qt;
    S = (S + 7) / 8 * 8;
    if ((int64_t)S > tiles) S = (int)tiles;
    if (S >= 8) S = S / 8 * 8;
    return S < 1 ? 1 : S;
}
// Can the blocks of such a launch pool their first tiles (exchange_bound)?  Every block resident at once, at most 1024 and
// at least 1.5 k published values per query (wn per block, or one per block where S * wn would pass 1024).  (Any k values of
// distinct rows bound the k-th distance; how many more there are decides how good the bound is.  With 2 k asked for, 4096
// queries with the 40 candidates of the fp32 filter route -- 16 splits, 64 values -- took a pass over a sample instead; with
// 1.5 k they pool: 500 k / 1M / 2M / 4M x 1536 rows 7.38 -> 6.56, 13.55 -> 12.36, 25.76 -> 24.32, 49.31 -> 47.19 ms per batch.
// With k values, 64 of 64, the bound is too loose: 47.8 -> 50.6 ms at k = 64 on 4M rows.)
static bool two_stage_can_pool(int qt, int S, int slots, int wn, int k) {
    int fit = 150;  // published values per query, in percent of k
    if (const char *e = getenv("HVS_XCHG_FIT")) fit = atoi(e) >= 100 ? atoi(e) : fit;  // experiments
    return (int64_t)qt * S <= slots && S <= 1024 && (int64_t)S * wn * 100 >= (int64_t)fit * k && (S * wn <= 1024 || S >= 2 * k);
}

// LEAST over m <= 128 prepared embeddings of the distance to every row of [row_begin, row_end), in one pass over the
// rows on the matrix pipe.  *used = false when this route does not apply (fp32 rows whose split copy is neither
// current nor worth building); the caller then streams the rows once per 8 embeddings.
int hvs_distances_mfma(hvs_index *idx, const float *d_q, const float *d_qinv, int m, int64_t row_begin, int64_t row_end, float *d_out,
                       bool *used) {
    *used = false;
    if (m < 1 || m > 128 || row_end <= row_begin || getenv("HVS_NO_MFMA_DIST")) return 0;
    bool use_split = false;
    if (idx->dtype == HVS_F32) {
        if (getenv("HVS_NO_SPLIT") || !(m >= 16 || (idx->split && idx->split_n == idx->n))) return 0;
        HVS_TRY(hvs_split_ensure(idx, &use_split));
        if (!use_split) return 0;
    }
    const int esz = idx->elem_size;
    const int ld_bytes = idx->ld * esz;
    const int lda_bytes = (ld_bytes + KB - 1) / KB * KB;
    const int lda = lda_bytes / esz;
    const int ld16 = (idx->dim + 63) / 64 * 64;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t a_bytes = up((size_t)m * lda_bytes), a_inv_bytes = up((size_t)m * 4);
    void *a_ws = nullptr;
    HVS_TRY(hvs_ws_get(idx, 4, a_bytes + a_inv_bytes + (use_split ? up((size_t)m * ld16 * 4) + up((size_t)m * 4) : 0), &a_ws));
    float *A_inv = (float *)((char *)a_ws + a_bytes);
    if (idx->dtype == HVS_F32) build_a_kernel<float><<<m, 256, 0, idx->stream>>>(d_q, d_qinv, (float *)a_ws, A_inv, 1, 1, idx->ld, lda);
    else if (idx->dtype == HVS_F16) build_a_kernel<__half><<<m, 256, 0, idx->stream>>>(d_q, d_qinv, (__half *)a_ws, A_inv, 1, 1, idx->ld, lda);
    else build_a_kernel<uint16_t><<<m, 256, 0, idx->stream>>>(d_q, d_qinv, (uint16_t *)a_ws, A_inv, 1, 1, idx->ld, lda);
    HVS_HIP(hipGetLastError());

    SearchParams p;
    memset(&p, 0, sizeof(p));
    p.A = a_ws;
    p.q_inv = A_inv;
    p.nA = m;
    p.lda_bytes = lda_bytes;
    p.X = idx->rows;
    p.inv_norm = idx->inv_norm;
    p.ld_bytes = ld_bytes;
    p.nks = lda_bytes / KB;
    p.group_filter = -1;
    p.row_begin = row_begin;
    p.row_end = row_end;
    p.thr = INFINITY;
    p.dist_out = d_out;
    const int bm = m <= 32 ? 32 : (m <= 64 ? 64 : 128);
    bool dual = false;
    if (use_split) {
        char *a_split = (char *)a_ws + a_bytes + a_inv_bytes;
        float *a_split_inv = (float *)(a_split + up((size_t)m * ld16 * 4));
        HVS_TRY(hvs_split_rows(idx->stream, (const float *)a_ws, A_inv, 0, m, lda, lda, ld16, a_split, a_split_inv, nullptr, nullptr));
        p.A = a_split;
        p.q_inv = a_split_inv;
        p.lda_bytes = p.ld_bytes = ld16 * 4;
        p.X = idx->split;
        p.inv_norm = idx->split_inv;
        p.split_steps = ld16 * 2 / KB;
        dual = bm <= 64;
        p.nks = (dual ? 2 : 3) * p.split_steps;
    }
    p.QT = 1;
    p.T = (row_end - row_begin + 255) / 256;
    p.S = (int)(p.T < 256 ? p.T : 256);
    HVS_TRY(hvs_prof_begin(idx));
    if (idx->dtype == HVS_BF16) HVS_TRY(launch_distcol_bm<HVS_BF16>(bm, false, p.S, idx->stream, p));
    else HVS_TRY(launch_distcol_bm<HVS_F16>(bm, dual, p.S, idx->stream, p));
    HVS_TRY(hvs_prof_end(idx, HVS_PROF_MFMA_SEARCH));
    *used = true;
    return 0;
}

int hvs_search_mfma(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int32_t group_filter,
                    const uint32_t *d_mask, float thr, int64_t row_begin, int64_t row_end, float *d_dist,
                    int64_t *d_ids, int32_t *d_count, int split_mode) {
    // split_mode 3 (fp32 rows with a current split copy; hvs_search_rescore): the filter pass -- one product over the
    // hi slabs alone (approximate scores), d_ids receives row positions
    const bool hi_only = split_mode == 3;
    const int pf = padded_fold(fold);
    const int nA = nq * pf;
    const int esz = idx->elem_size;
    int ld_bytes = idx->ld * esz;
    int lda_bytes = (ld_bytes + KB - 1) / KB * KB;
    const int lda = lda_bytes / esz;
    // fp32 rows on the 16-bit pipe: both operands as [hi | lo] fp16 slabs (hvs_split.hip)
    const int ld16 = (idx->dim + 63) / 64 * 64;
    const size_t a_split_bytes = idx->dtype == HVS_F32 ? (((size_t)nA * ld16 * 4 + 255) & ~(size_t)255) + (((size_t)nA * 4 + 255) & ~(size_t)255) : 0;
    void *a_ws = nullptr;
    const size_t a_bytes = ((size_t)nA * lda_bytes + 255) & ~(size_t)255;
    const size_t a_inv_bytes = ((size_t)nA * 4 + 255) & ~(size_t)255;
    const size_t tau0_bytes = ((size_t)nq * 4 + 255) & ~(size_t)255;  // tau0 of a sample pass
    HVS_TRY(hvs_ws_get(idx, 4, a_bytes + a_inv_bytes + a_split_bytes + tau0_bytes + a_inv_bytes, &a_ws));  // ... + q_rel of split queries
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
    p.groups = idx->groups;
    p.mask = d_mask;
    p.deleted = idx->n_deleted > 0 ? idx->deleted : nullptr;
    p.group_filter = group_filter;
    p.row_begin = row_begin;
    p.row_end = row_end;

    // fp32 rows on the 16-bit pipe: decided first, because a filtered search gathers from whichever copy is contracted
    bool use_split = false;
    if (idx->dtype == HVS_F32 && split_mode != 0 && !getenv("HVS_NO_SPLIT") &&
        (split_mode == 1 || split_mode == 3 || nA >= 16 || (idx->split && idx->split_n == idx->n)))
        HVS_TRY(hvs_split_ensure(idx, &use_split));
    const void *src_rows = use_split ? idx->split : idx->rows;
    const float *src_inv = use_split ? idx->split_inv : idx->inv_norm;
    if (use_split) ld_bytes = ld16 * 4;
    p.X = src_rows;
    p.inv_norm = src_inv;

    // A WHERE on the group code or a row bitmask: count the live rows and, when they are a minority,
    // contract over a dense copy of them (tombstones alone are sparse and stay in the epilogue): the
    // contraction then costs n_live instead of n rows, and the copy is 2 * n_live * row bytes of traffic.
    // (A short range -- a small table, or the tail behind the ordered part of a large one -- is cheaper to contract whole
    // with the test in the epilogue than to count, list and gather: the count alone is a synchronisation.  One query tile,
    // 75 embeddings, 768 f16 columns, 1 / 8 of the rows live: 20 000 rows 0.125 -> 0.082 ms, 100 000 rows 0.133 -> 0.095 ms;
    // from 400 000 rows on the copy wins, 0.156 against 0.213 ms.)
    const int64_t n_range = row_end - row_begin;
    const bool short_range = nA <= 256 && (double)n_range * ld_bytes <= 256.0e6;
    if ((group_filter >= 0 || d_mask) && n_range >= 512 && !short_range && row_end <= (int64_t)0xFFFFFFFFll && !getenv("HVS_NO_COMPACT")) {
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
            HVS_HIP(hipMemsetAsync(c_rows + copy_bytes, 0, HVS_ROWS_SLACK, idx->stream));  // see hvs_index_reserve
            if (n_live > 0) {
                HVS_TRY(hvs_live_write(idx, lp, positions));
                HVS_TRY(hvs_live_gather(idx, src_rows, src_inv, ld_bytes, positions, n_live, c_rows, c_inv));
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
    int kdt = idx->dtype;  // element type the kernel contracts in
    if (use_split) {
        char *a_split = (char *)a_ws + a_bytes + a_inv_bytes;
        float *a_split_inv = (float *)(a_split + (((size_t)nA * ld16 * 4 + 255) & ~(size_t)255));
        float *a_rel = (float *)((char *)a_ws + a_bytes + a_inv_bytes + a_split_bytes + tau0_bytes);
        HVS_TRY(hvs_split_rows(idx->stream, (const float *)a_ws, A_inv, 0, nA, lda, lda, ld16, a_split, a_split_inv, a_rel, nullptr));
        idx->q_rel = a_rel;
        kdt = HVS_F16;
        p.A = a_split;
        p.q_inv = a_split_inv;
        p.lda_bytes = lda_bytes = ld16 * 4;
        p.split_steps = hi_only ? 0 : ld16 * 2 / KB;  // hi slabs alone: plain operands that end half way through a row
    }
    HVS_REQUIRE(!hi_only || use_split, "hvs_search: the filter pass needs the split copy");
    p.ld_bytes = ld_bytes;
    p.nks = p.split_steps ? 3 * p.split_steps : (hi_only ? ld16 * 2 / KB : lda_bytes / KB);
    p.nq_out = nq;
    p.k = k;
    p.thr = thr;
    // one query tile of 16-bit operands and enough rows to keep every CU streaming: the ring kernel
    // (a 256-row query tile against 128 corpus rows, the shape whose three stages still fit in LDS, was measured
    // slower than the two-stage 256 x 256 kernel: 64 x 64 wave tiles read as many LDS bytes per MFMA as the pipe retires)
    const int stream_max = 128;
    bool stream = kdt != HVS_F32 && nA <= stream_max && row_end - row_begin >= 256 * 256;
    if (const char *e = getenv("HVS_STREAM")) stream = kdt != HVS_F32 && nA <= stream_max && atoi(e) != 0;  // 1 forces it at any size, 0 disables it
    else if (stream && !getenv("HVS_NO_XCHG")) {
        // ... unless the stream is short: below about 4 GB of rows the two-stage kernel, whose blocks pool their first tiles
        // (exchange_bound), is as fast or faster at every batch size (195 k x 768 bf16, k = 100: 8 / 64 / 128 queries 0.077 /
        // 0.096 / 0.108 ms against 0.073 / 0.194 / 0.320 ms on the ring; 3.1M rows: 0.83 / 1.06 / 1.22 against 0.80 / 0.98 /
        // 1.27; on 12.5M rows the ring leads by 5 - 10 % up to 64 queries)
        const bool sm = nA <= 32;
        const int64_t rows = row_end - row_begin, t2 = (rows + (sm ? 64 : 128) - 1) / (sm ? 64 : 128);
        const int64_t s2 = t2 < (sm ? 1024 : 512) ? (t2 >= 8 ? t2 / 8 * 8 : t2) : (sm ? 1024 : 512);
        const bool fits = s2 * 2 >= 2 * k && (s2 * 2 <= 1024 || s2 >= 2 * k);
        if (fits && (int64_t)k * 8192 > rows / s2 && (double)rows * ld_bytes < 4.0e9) stream = false;
    }
    // tile shape of the two-stage kernel: 16-bit rows with more than one 128-row tile of queries take the 256 x 256 block tile
    bool big = kdt != HVS_F32 && nA > Tile128::BM;
    if (const char *e = getenv("HVS_TILE")) big = kdt != HVS_F32 && atoi(e) == 256;
    // fp32 rows take the 256 x 256 tile too when it pads no more query rows than the 128 tile would: half the barriers
    // and LDS-DMA issues per MFMA and two blocks per corpus stream instead of four (4096 queries x 2M x 1536: 177.4 ms
    // against 182.5 ms, 111 GB of fabric reads per launch against 164 GB)
    if (kdt == HVS_F32) big = nA > Tile128::BM && (nA + Tile256::BM - 1) / Tile256::BM * Tile256::BM == (nA + Tile128::BM - 1) / Tile128::BM * Tile128::BM;
    if (const char *e = getenv("HVS_F32_TILE")) big = kdt == HVS_F32 ? atoi(e) == 256 && nA > Tile128::BM : big;  // 128 / 256 forces the shape for fp32 rows
    bool small = kdt != HVS_F32 && nA <= 32;  // measured: 10-13 % faster than Tile128 up to 16 embeddings, even at 32, slower at 64
    if (const char *e = getenv("HVS_TILE")) small = kdt != HVS_F32 && atoi(e) == 64;
    // A range so short that the blocks of the tile at hand could not pool their first tiles (fewer than 1.5 k published
    // values per query) takes a smaller tile that can: without a common bound every score of such a launch is a survivor
    // (the dense copy of 9 800 live rows of a tail, 256 queries, k = 100: 0.39 ms on 32 blocks of the 256 x 256 tile).
    if (!stream && kdt != HVS_F32 && !getenv("HVS_TILE") && !getenv("HVS_NO_XCHG")) {
        const int64_t rows = row_end - row_begin;
        auto is_short = [&](int bm_, int slots_) { return (int64_t)k * 8192 > rows / ((slots_ + (nA + bm_ - 1) / bm_ - 1) / ((nA + bm_ - 1) / bm_)); };
        auto can_pool = [&](int bm_, int bn_, int slots_, int wn_) {
            const int qt = (nA + bm_ - 1) / bm_;
            return two_stage_can_pool(qt, two_stage_splits(qt, (rows + bn_ - 1) / bn_, slots_), slots_, wn_, k);
        };
        if (big && is_short(Tile256::BM, 256) && !can_pool(Tile256::BM, Tile256::BN, 256, Tile256::WN)) {
            if (can_pool(Tile128::BM, Tile128::BN, 512, Tile128::WN)) big = false;
            else if (can_pool(Tile64::BM, Tile64::BN, 1024, Tile64::WN)) big = false, small = true;
        } else if (!big && !small && is_short(Tile128::BM, 512) && !can_pool(Tile128::BM, Tile128::BN, 512, Tile128::WN) &&
                   can_pool(Tile64::BM, Tile64::BN, 1024, Tile64::WN))
            small = true;
    }
    const int bm = stream ? (nA <= 32 ? 32 : (nA <= 64 ? 64 : 128)) : (big ? Tile256::BM : (small ? Tile64::BM : Tile128::BM));
    const int bn = stream ? 256 : (big ? Tile256::BN : (small ? Tile64::BN : Tile128::BN));
    // ring kernel on the split copy: up to 64 rows both query slabs ride in a stage -- (A hi, A lo | X hi) steps, then
    // (A hi | X lo) steps; two query images of 128 rows would leave no room for a third stage, so that tile keeps the
    // three-product order
    const bool dual = stream && p.split_steps != 0 && bm <= 64;
    if (dual) p.nks = 2 * p.split_steps;
    p.cap = k > 128 ? CAND_MAX : CAND_SMALL;  // >= k + bn with room to let an array grow to 2 k between rankings: the survivors of one more tile always fit
    p.dbg = 0;
    if (const char *e = getenv("HVS_DBG")) p.dbg = stream ? 0 : atoi(e);
#ifndef HVS_KERNEL_HOOKS
    HVS_REQUIRE(p.dbg == 0, "HVS_DBG=%d: this libhvs.so was built without the kernel's timing hooks (make HOOKS=1 builds them in)", p.dbg);
#endif
    p.QT = stream ? 1 : (nA + bm - 1) / bm;

    bool want_xchg = false;  // set below, before the main pass
    const int slots_of_tile = stream ? 256 : (big ? 256 : (small ? 1024 : 512));  // blocks the chip holds at once
    const int wn_of_tile = big ? Tile256::WN : (small ? Tile64::WN : Tile128::WN);
    // corpus splits for a range of T tiles
    auto splits_of = [&](int64_t T) -> int {
        if (stream) return (int)(T < 256 ? T : 256);
        return two_stage_splits(p.QT, T, slots_of_tile);
    };
    // can the blocks of such a launch pool their first tiles (exchange_bound)?  16-bit operands on the two-stage kernel,
    // every block resident at once, at most 1024 and at least 1.5 k published values per query
    auto xchg_fits = [&](int S) -> bool { return !stream && kdt != HVS_F32 && two_stage_can_pool(p.QT, S, slots_of_tile, wn_of_tile, k); };
    // One pass of the kernel over rows [rb, re) and the merge of its per-block lists into out_*.
    auto run_pass = [&](SearchParams pp, int64_t rb, int64_t re, const float *tau0, float *o_dist, int64_t *o_ids, int32_t *o_count) -> int {
        pp.row_begin = rb;
        pp.row_end = re;
        pp.tau0 = tau0;
        pp.T = (re - rb + bn - 1) / bn;
        const int S = splits_of(pp.T);
        pp.S = S;
        const int grid = pp.QT * S;
        void *cand = nullptr, *bufB = nullptr;
        const size_t cand_bytes = (size_t)grid * (bm / pf) * pp.cap * 8;
        const size_t part_bytes = (size_t)nq * S * k * 8;
        const int P2 = (S + MERGE_FAN - 1) / MERGE_FAN;
        const int slots = slots_of_tile;
        // arrival counters of lockstep_arrive: when the grid fits the chip at once (it is sized to) and blocks share streams
        const bool lockstep = !stream && (S & 7) == 0 && grid <= slots && !getenv("HVS_NO_LOCKSTEP");
        const size_t ls_bytes = lockstep ? (size_t)8 * S * 4 : 0;
        // the blocks pool what their first tile shows (exchange_bound): 16-bit operands, streams so short that a block's own
        // bound gets nowhere (the condition a pass over a sample is taken under, which this replaces), every block resident
        // and with a tile of its own, and at least 1.5 k published values per query
        const int xw = S * wn_of_tile <= 1024 ? wn_of_tile : 1;
        const int xm = S * xw;
        pp.xchg_w = xw;
        const bool xchg = want_xchg && tau0 == nullptr && (pp.dbg & ~512) == 0 && xchg_fits(S);
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t xc_bytes = 0;
        // later rounds: when every block still has at least as many tiles ahead of it as behind it at that point
        const int64_t tiles_per_block = pp.T / S;
        int rounds = 1;
        int t2 = XCHG_TILE_2, t3 = XCHG_TILE_3;
        if (const char *e = getenv("HVS_XCHG_T2")) t2 = atoi(e) > 0 ? atoi(e) : t2;  // experiments
        if (const char *e = getenv("HVS_XCHG_T3")) t3 = atoi(e) > t2 ? atoi(e) : t3;
        if (xchg && !getenv("HVS_XCHG_ONE_ROUND")) rounds += (tiles_per_block >= 2 * t2 ? 1 : 0) + (tiles_per_block >= 2 * t2 && tiles_per_block >= 2 * t3 ? 1 : 0);
        pp.xchg_t2 = t2;
        pp.xchg_t3 = t3;
        int xr = 1;
        while (xr < 16 && S * xr * 2 <= 1024) xr *= 2;
        const size_t x1_bytes = up((size_t)nq * xm * 4) + up((size_t)nq * 4);
        const size_t x2_bytes = rounds > 1 ? up((size_t)2 * 1024 * nq * 4) + up((size_t)2 * nq * 4) : 0;
        const size_t xv_bytes = xchg ? x1_bytes + x2_bytes : 0;
        // one list per query that every block appends to, ranked by select_pooled_kernel: where the blocks pool their bounds,
        // where k is beyond the sorted lists, and wherever a block's stream is so short that most of what it scores survives --
        // sorting a full candidate array per query and block then costs more than the contraction (a dense copy of 9 800 rows,
        // 256 queries, k = 100: 39 blocks of one tile each spent 0.43 ms on it)
        const bool pooled = xchg || k > MFMA_MAX_K || tiles_per_block <= 8;
        const size_t pc_bytes = pooled ? up((size_t)nq * 4) : 0;  // its counters, zeroed with the overflow flag
        const size_t keys_b = up((size_t)nq * P2 * k * 8);
        HVS_TRY(hvs_ws_get(idx, 1, cand_bytes + part_bytes + 512, &cand));
        HVS_TRY(hvs_ws_get(idx, 2, keys_b + up(64 + ls_bytes) + xc_bytes + pc_bytes + xv_bytes, &bufB));
        void *bufA = (char *)cand + ((cand_bytes + 255) & ~(size_t)255);
        int *overflow = (int *)((char *)bufB + keys_b);
        char *xc = (char *)overflow + up(64 + ls_bytes);
        {
            // flags and counters start at zero, the exchange tables behind them at all ones: one launch for both
            const size_t zero16 = (up(64 + ls_bytes) + xc_bytes + pc_bytes) / 16, all16 = zero16 + xv_bytes / 16;
            const int blocks = (int)((all16 + 255) / 256 < 512 ? (all16 + 255) / 256 : 512);
            init_state_kernel<<<blocks, 256, 0, idx->stream>>>((uint4 *)overflow, zero16, all16);
            HVS_HIP(hipGetLastError());
        }
        pp.prof = (unsigned long long *)(overflow + 4);
        pp.lockstep = lockstep ? (unsigned int *)(overflow + 16) : nullptr;
        pp.pooled_count = pooled ? (unsigned int *)(xc + xc_bytes) : nullptr;
        pp.xchg_min = xchg ? (uint32_t *)(xc + xc_bytes + pc_bytes) : nullptr;
        pp.xchg_tau = xchg ? (uint32_t *)(xc + xc_bytes + pc_bytes + up((size_t)nq * xm * 4)) : nullptr;
        pp.xchg_rounds = xchg ? rounds : 0;
        pp.xchg_r = xr;
        pp.xchg_min2 = (xchg && rounds > 1) ? (uint32_t *)(xc + xc_bytes + pc_bytes + x1_bytes) : nullptr;
        pp.xchg_tau2 = (xchg && rounds > 1) ? (uint32_t *)(xc + xc_bytes + pc_bytes + x1_bytes + up((size_t)2 * 1024 * nq * 4)) : nullptr;
        pp.cand = (uint64_t *)cand;
        pp.partial = (uint64_t *)bufA;
        idx->h_flag[0] = 0;  // (the stream is idle: every pass ends with a synchronisation)
        pp.overflow = idx->h_flag;
        HVS_TRY(hvs_prof_begin(idx));
        if (stream) {
            if (kdt == HVS_F16) HVS_TRY(launch_stream_bm<HVS_F16>(bm, dual, pf, grid, idx->stream, pp));
            else HVS_TRY(launch_stream_bm<HVS_BF16>(bm, false, pf, grid, idx->stream, pp));
        } else if (kdt == HVS_F32 && big) HVS_TRY((launch_search<HVS_F32, Tile256>(pf, grid, idx->stream, pp)));
        else if (kdt == HVS_F32) HVS_TRY((launch_search<HVS_F32, Tile128>(pf, grid, idx->stream, pp)));
        else if (xchg) {
            if (kdt == HVS_F16 && small) HVS_TRY((launch_search<HVS_F16, Tile64, true>(pf, grid, idx->stream, pp)));
            else if (kdt == HVS_F16 && !big) HVS_TRY((launch_search<HVS_F16, Tile128, true>(pf, grid, idx->stream, pp)));
            else if (kdt == HVS_F16) HVS_TRY((launch_search<HVS_F16, Tile256, true>(pf, grid, idx->stream, pp)));
            else if (small) HVS_TRY((launch_search<HVS_BF16, Tile64, true>(pf, grid, idx->stream, pp)));
            else if (!big) HVS_TRY((launch_search<HVS_BF16, Tile128, true>(pf, grid, idx->stream, pp)));
            else HVS_TRY((launch_search<HVS_BF16, Tile256, true>(pf, grid, idx->stream, pp)));
        } else if (kdt == HVS_F16 && small) HVS_TRY((launch_search<HVS_F16, Tile64>(pf, grid, idx->stream, pp)));
        else if (kdt == HVS_F16 && !big) HVS_TRY((launch_search<HVS_F16, Tile128>(pf, grid, idx->stream, pp)));
        else if (kdt == HVS_F16) HVS_TRY((launch_search<HVS_F16, Tile256>(pf, grid, idx->stream, pp)));
        else if (small) HVS_TRY((launch_search<HVS_BF16, Tile64>(pf, grid, idx->stream, pp)));
        else if (!big) HVS_TRY((launch_search<HVS_BF16, Tile128>(pf, grid, idx->stream, pp)));
        else HVS_TRY((launch_search<HVS_BF16, Tile256>(pf, grid, idx->stream, pp)));
        HVS_TRY(hvs_prof_end(idx, HVS_PROF_MFMA_SEARCH));
        HVS_TRY(hvs_prof_begin(idx));
        if (pooled) {
            const int64_t *id_map = (hi_only || idx->emit_positions) ? nullptr : idx->ids;
            const size_t slots_q = (size_t)S * k;
            if (k <= 64) select_pooled_kernel<1, CAND_MAX><<<nq, 64, 0, idx->stream>>>((uint64_t *)bufA, pp.pooled_count, slots_q, k, id_map, o_dist, o_ids, o_count);
            else if (k <= 128) select_pooled_kernel<2, CAND_MAX><<<nq, 64, 0, idx->stream>>>((uint64_t *)bufA, pp.pooled_count, slots_q, k, id_map, o_dist, o_ids, o_count);
            else if (k <= 256) select_pooled_kernel<4, CAND_MAX><<<nq, 64, 0, idx->stream>>>((uint64_t *)bufA, pp.pooled_count, slots_q, k, id_map, o_dist, o_ids, o_count);
            else select_pooled_kernel<16, 2 * CAND_MAX><<<nq, 64, 0, idx->stream>>>((uint64_t *)bufA, pp.pooled_count, slots_q, k, id_map, o_dist, o_ids, o_count);
            HVS_HIP(hipGetLastError());
        } else
            HVS_TRY(hvs_reduce_lists(idx, (uint64_t *)bufA, (uint64_t *)bufB, nq, S, k, o_dist, o_ids, o_count, hi_only));
        HVS_TRY(hvs_prof_end(idx, HVS_PROF_MERGE));
        HVS_HIP(hipStreamSynchronize(idx->stream));
        const int h_overflow = idx->h_flag[0];
        if (pp.dbg & 512) {
            unsigned long long h_prof[6] = {0, 0, 0, 0, 0, 0};
            HVS_HIP(hipMemcpy(h_prof, pp.prof, sizeof(h_prof), hipMemcpyDeviceToHost));
            const double waves = (double)grid * (big ? Tile256::NW : (small ? Tile64::NW : Tile128::NW));
            fprintf(stderr, "hvs prof (clock64 ticks per wave): k-loop+wait %.0f  epilogue %.0f  tile-end barrier %.0f  compaction %.0f  "
                            "first-tile exchange %.0f  publishing %.0f\n",
                    h_prof[0] / waves, h_prof[1] / waves, h_prof[2] / waves, h_prof[3] / waves, h_prof[4] / waves, h_prof[5] / waves);
        }
        HVS_REQUIRE(h_overflow == 0, "hvs_search: internal candidate array overflow (results would be incomplete)");
        return 0;
    };

    // A block tightens its admission bound from the rows it has seen itself; when the table is cut into so many
    // splits that a block's own stream is short next to k (its k-th distance then sits at the k / rows-per-block
    // quantile for the whole launch, and most score groups take the slow path of the epilogue), a first pass over a
    // sample of the rows -- 1/96 of them, at least 1280 k (measured optimum, flat) -- gives every query an upper bound on its k-th distance
    // that all blocks start from.  Any subset of the rows gives a valid bound; the sample's rows are scored again
    // by the main pass.
    const int64_t n_rows = row_end - row_begin;
    const int slots_main = stream ? 256 : (big ? 256 : (small ? 1024 : 512));
    const int64_t rows_per_block = n_rows / ((slots_main + p.QT - 1) / p.QT);
    // short streams: the two-stage kernels on 16-bit operands pool their first tiles inside the launch (exchange_bound) where
    // that applies; the pass over a sample stays for the ring kernel and for launches that cannot pool
    const bool short_streams = (int64_t)k * 8192 > rows_per_block;
    want_xchg = short_streams && !getenv("HVS_NO_XCHG");
    if (const char *e = getenv("HVS_XCHG")) want_xchg = atoi(e) != 0;
    const bool can_xchg = want_xchg && (p.dbg & ~512) == 0 && xchg_fits(splits_of((n_rows + bn - 1) / bn));
    bool prepass = short_streams && n_rows >= 262144 && (!stream || k >= 64) && !can_xchg;
    if (const char *e = getenv("HVS_PREPASS")) prepass = atoi(e) != 0 && n_rows >= 4 * bn;
    const float *tau0 = nullptr;
    if (prepass) {
        int64_t n_s = n_rows / 96 > (int64_t)k * 1280 ? n_rows / 96 : (int64_t)k * 1280;
        if (n_s > n_rows / 4) n_s = n_rows / 4;
        if (const char *e = getenv("HVS_PREPASS_ROWS")) n_s = atoll(e) < n_rows ? atoll(e) : n_rows;
        n_s = (n_s + bn - 1) / bn * bn;
        HVS_TRY(run_pass(p, row_begin, row_begin + n_s, nullptr, d_dist, d_ids, d_count));
        float *t0 = (float *)((char *)a_ws + a_bytes + a_inv_bytes + a_split_bytes);
        tau_from_topk_kernel<<<(nq + 255) / 256, 256, 0, idx->stream>>>(d_dist, d_count, nq, k, thr, t0);
        HVS_HIP(hipGetLastError());
        tau0 = t0;
    }
    return run_pass(p, row_begin, row_end, tau0, d_dist, d_ids, d_count);
}

// ---------------------------------------------------------------------------------------------
// fp32 tables, small k: filter on the hi slabs, exact scores for the candidates
// ---------------------------------------------------------------------------------------------
// The three products of the split route buy 22-bit operands for every one of the nq x rows scores, of which k per
// query are kept.  One product over the hi slabs alone gives every similarity to within a known bound:
//     q s_q = qh + ql, x s_x = xh + xl with |ql_i| <= 2^-11 |q s_q|_i, |xl_i| <= 2^-11 |x s_x|_i (fp16 rounding), so
//     |q.x s_q s_x - qh.xh| <= sum |qh_i xl_i| + |ql_i xh_i| + |ql_i xl_i| <= 2^-10 (1 + 2^-10) |q s_q| |x s_x|,
// i.e. 9.78e-4 of similarity, plus the fp32 accumulation of n exact products (<= n 2^-24 whatever the order) and
// three roundings of the distance formula.  (Elements more than 2^27 below a row's largest are fp16-subnormal after
// scaling and err by 2^-25 absolutely, which is below 2^-38 sqrt(n) of the row's norm.)
// So: pass 1 keeps the kc > k rows of smallest approximate distance below thr + eps; pass 2 scores those rows from
// the fp32 rows with fp32 FMAs, applies thr, orders them by (distance, row position) and keeps k.  A row outside
// the candidate list has approximate distance >= a_last (the list's largest), hence exact distance >= a_last - eps:
// the k kept are THE k smallest whenever the k-th exact distance lies below a_last - eps (or the list was not cut
// at all, or a_last - eps >= thr).  A query for which none of this holds is flagged and searched again by a route
// that scores every row in full (one flagged query: an fp32 scan; several: a batch of their own with all three
// products) -- the result is the exact top-k in every case, what changes is the time.
__global__ __launch_bounds__(256) void rescore_kernel(const float *__restrict__ rows, int ld, const float *__restrict__ inv_norm,
                                                       const int64_t *__restrict__ ids, const float *__restrict__ q,
                                                       const float *__restrict__ q_inv, int fold, const float *__restrict__ cand_dist,
                                                       const int64_t *__restrict__ cand_pos, const int32_t *__restrict__ cand_cnt, int kc,
                                                       int k, float thr, float eps, const float *__restrict__ q_rel, int q_rel_stride,
                                                       const uint32_t *__restrict__ x_rel_max, float eps_tail,
                                                       float *__restrict__ out_dist, int64_t *__restrict__ out_ids,
                                                       int32_t *__restrict__ out_count, int32_t *__restrict__ redo) {
    __shared__ uint64_t keys[MFMA_MAX_K], sorted[MFMA_MAX_K];
    const int qi = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_c = cand_cnt[qi];
    for (int c = wave; c < kc; c += 4) {
        uint64_t key = HVS_KEY_EMPTY;
        if (c < n_c) {
            const int64_t pos = cand_pos[(size_t)qi * kc + c];
            const float *x = rows + (size_t)pos * ld;
            const float inv_x = inv_norm[pos];
            float best = __builtin_nanf("");
            for (int j = 0; j < fold; j++) {
                const float *e = q + ((size_t)qi * fold + j) * ld;
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
                for (int i = lane * 4; i < ld; i += 256) {  // ld is a multiple of 4, pad columns are zero on both sides
                    const float4 a = *reinterpret_cast<const float4 *>(x + i);
                    const float4 b = *reinterpret_cast<const float4 *>(e + i);
                    s0 = fmaf(a.x, b.x, s0);
                    s1 = fmaf(a.y, b.y, s1);
                    s2 = fmaf(a.z, b.z, s2);
                    s3 = fmaf(a.w, b.w, s3);
                }
                float sum = (s0 + s1) + (s2 + s3);
                for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
                best = fminf(best, hvs_cos_dist(sum, inv_x, q_inv[qi * fold + j]));  // LEAST: NaN only if every term is
            }
            if (best < thr) key = hvs_make_key(hvs_ord(best), (uint32_t)pos);
        }
        if (lane == 0) keys[c] = key;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < kc) {  // rank by counting: kc <= MFMA_MAX_K = the block's threads, keys distinct wherever they are not empty
        const uint64_t me = keys[t];
        int rank = 0;
        for (int i = 0; i < kc; i++) {
            const uint64_t o = keys[i];
            rank += (o < me || (o == me && i < t)) ? 1 : 0;
        }
        sorted[rank] = me;
    }
    __syncthreads();
    if (t < k) {
        const uint64_t key = sorted[t];  // k <= kc
        const bool have = key != HVS_KEY_EMPTY;
        out_dist[(size_t)qi * k + t] = have ? hvs_unord((uint32_t)(key >> 32)) : __builtin_inff();
        out_ids[(size_t)qi * k + t] = have ? (ids ? ids[(uint32_t)key] : (int64_t)(uint32_t)key) : -1;  // ids == nullptr: row positions
    }
    if (t == 0) {
        int n_valid = 0;
        while (n_valid < kc && sorted[n_valid] != HVS_KEY_EMPTY) n_valid++;
        out_count[qi] = n_valid < k ? n_valid : k;
        bool ok = true;
        if (n_c >= kc) {  // the list was cut at kc
            // the bound of these operands (split_rows_kernel measured how far each lies from its hi slab), never more than
            // the bound of the rounding unit
            float rq = 0.0f;
            for (int j = 0; j < fold; j++) rq = fmaxf(rq, q_rel[(size_t)qi * q_rel_stride + j]);
            const float rx = __uint_as_float(*x_rel_max);
            const float eps_q = fminf(eps, rx + rq * (1.0f + rx) + eps_tail);
            const float bound = cand_dist[(size_t)qi * kc + kc - 1] - eps_q;  // no row outside the list is closer than this
            const bool proven = n_valid >= k && hvs_unord((uint32_t)(sorted[k - 1] >> 32)) + 2e-6f < bound;
            ok = proven || bound >= thr;
        }
        redo[qi] = ok ? 0 : 1;
    }
}

// the flagged queries of a batch as a batch of their own: embeddings and inverse norms gathered ...
__global__ void gather_queries_kernel(const float *__restrict__ q, const float *__restrict__ q_inv, const int32_t *__restrict__ list,
                                      int fold, int ld, float *__restrict__ q_sub, float *__restrict__ q_inv_sub) {
    const int j = blockIdx.x;
    const float *src = q + (size_t)list[j] * fold * ld;
    float *dst = q_sub + (size_t)j * fold * ld;
    for (int c = threadIdx.x; c < fold * ld; c += blockDim.x) dst[c] = src[c];
    if (threadIdx.x < fold) q_inv_sub[j * fold + threadIdx.x] = q_inv[list[j] * fold + threadIdx.x];
}

// ... and their answers put back in their places
__global__ void scatter_results_kernel(const float *__restrict__ sub_dist, const int64_t *__restrict__ sub_ids,
                                       const int32_t *__restrict__ sub_cnt, const int32_t *__restrict__ list, int k,
                                       float *__restrict__ dist, int64_t *__restrict__ ids, int32_t *__restrict__ cnt) {
    const int j = blockIdx.x, qi = list[j];
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        dist[(size_t)qi * k + i] = sub_dist[(size_t)j * k + i];
        ids[(size_t)qi * k + i] = sub_ids[(size_t)j * k + i];
    }
    if (threadIdx.x == 0) cnt[qi] = sub_cnt[j];
}

// One filter pass + exact rescoring of a batch with kc candidates per query; h_redo[i] = 1 for the queries whose list
// proves nothing.  Candidates live in workspace 7 (sized by the caller for the largest kc).
static int rescore_pass(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int kc, int32_t group_filter,
                        const uint32_t *d_mask, float thr, float eps, float eps_tail, int64_t row_begin, int64_t row_end, float *c_dist,
                        int64_t *c_pos, int32_t *c_cnt, int32_t *d_redo, float *d_dist, int64_t *d_ids, int32_t *d_count, std::vector<int32_t> &h_redo,
                        int *n_redo) {
    HVS_TRY(hvs_search_mfma(idx, d_q, d_qinv, nq, fold, kc, group_filter, d_mask, thr + eps, row_begin, row_end, c_dist, c_pos, c_cnt, 3));
    HVS_REQUIRE(idx->q_rel && idx->split_rel, "hvs_search: the filter pass left no error measure of its operands");
    HVS_TRY(hvs_prof_begin(idx));
    rescore_kernel<<<nq, 256, 0, idx->stream>>>((const float *)idx->rows, idx->ld, idx->inv_norm, idx->emit_positions ? nullptr : idx->ids, d_q, d_qinv, fold, c_dist, c_pos,
                                                c_cnt, kc, k, thr, eps, idx->q_rel, padded_fold(fold), idx->split_rel, eps_tail, d_dist, d_ids,
                                                d_count, d_redo);
    HVS_HIP(hipGetLastError());
    HVS_TRY(hvs_prof_end(idx, HVS_PROF_MERGE));
    h_redo.assign(nq, 0);
    void *pinned = nullptr;  // (a copy into pageable memory goes through the runtime's own staging and costs a second wait)
    HVS_TRY(hvs_pinned_get(idx, (size_t)nq * 4, &pinned));
    HVS_HIP(hipMemcpyAsync(pinned, d_redo, (size_t)nq * 4, hipMemcpyDeviceToHost, idx->stream));
    HVS_HIP(hipStreamSynchronize(idx->stream));
    memcpy(h_redo.data(), pinned, (size_t)nq * 4);
    *n_redo = 0;
    for (int i = 0; i < nq; i++) *n_redo += h_redo[i] != 0;
    return 0;
}

// *used = false: the route does not apply (k too large for a useful candidate list, or no split copy) and nothing
// was written.
// Two list lengths: 4 k candidates (at least 32) first; the queries whose list proves nothing -- a cloud of rows
// closer to each other than the bound, as tables of text embeddings have around most queries -- go through the filter
// again as a batch of their own with MFMA_MAX_K candidates (one more product over the hi slabs, against three for
// scoring every row in full; rescoring 256 rows per query is a few MB), and only what that leaves unproven is searched
// again in full.  A table on which the short lists fail for most of a batch starts with the long ones from then on.
int hvs_search_rescore(hvs_index *idx, const float *d_q, const float *d_qinv, int nq, int fold, int k, int32_t group_filter,
                       const uint32_t *d_mask, float thr, int64_t row_begin, int64_t row_end, float *d_dist, int64_t *d_ids,
                       int32_t *d_count, bool *used) {
    *used = false;
    if (idx->dtype != HVS_F32 || k > MFMA_MAX_K / 2 || fold > 4 || getenv("HVS_NO_SPLIT") || getenv("HVS_NO_FILTER")) return 0;
    bool have_split = false;
    HVS_TRY(hvs_split_ensure(idx, &have_split));
    if (!have_split) return 0;
    const bool adapt = !getenv("HVS_NO_ADAPT");
    if (adapt && idx->filter_skip > 0) {  // recent filter passes on this table proved nothing: do not pay for another yet
        idx->filter_skip--;
        return 0;
    }
    const int kc_long = MFMA_MAX_K;
    int kc = 4 * k < 32 ? 32 : (4 * k > kc_long ? kc_long : 4 * k);  // candidates per query: 4 k, at least 32, at most what a list holds
    if (const char *e = getenv("HVS_FILTER_KC")) kc = atoi(e) < k ? k : (atoi(e) > kc_long ? kc_long : atoi(e));
    else if (adapt && idx->filter_long) kc = kc_long;
    const int ld16 = (idx->dim + 63) / 64 * 64;
    const float eps_tail = (float)ld16 * 6.0e-8f + 1e-6f;  // fp32 accumulation of the products, roundings of the distance formula
    const float eps = 9.78e-4f + eps_tail;                // whatever the operands; the rescoring kernel tightens it to theirs
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_pos = up((size_t)nq * kc_long * 4), o_cnt = o_pos + up((size_t)nq * kc_long * 8), o_redo = o_cnt + up((size_t)nq * 4);
    // behind the candidates: room to search the flagged queries again as a batch of their own (list, embeddings,
    // inverse norms, outputs), unless that would be a workspace of its own size class
    const size_t sub_q_bytes = (size_t)nq * fold * idx->ld * 4;
    const bool sub_ok = sub_q_bytes <= ((size_t)256 << 20);
    const size_t o_list = o_redo + up((size_t)nq * 4), o_subq = o_list + up((size_t)nq * 4), o_subinv = o_subq + (sub_ok ? up(sub_q_bytes) : 0),
                 o_subd = o_subinv + up((size_t)nq * fold * 4), o_subi = o_subd + up((size_t)nq * k * 4), o_subc = o_subi + up((size_t)nq * k * 8),
                 ws_bytes = sub_ok ? o_subc + up((size_t)nq * 4) : o_list;
    void *ws = nullptr;
    HVS_TRY(hvs_ws_get(idx, 7, ws_bytes, &ws));
    float *c_dist = (float *)ws;
    int64_t *c_pos = (int64_t *)((char *)ws + o_pos);
    int32_t *c_cnt = (int32_t *)((char *)ws + o_cnt), *d_redo = (int32_t *)((char *)ws + o_redo);
    std::vector<int32_t> redo;
    int n_redo = 0;
    HVS_TRY(rescore_pass(idx, d_q, d_qinv, nq, fold, k, kc, group_filter, d_mask, thr, eps, eps_tail, row_begin, row_end, c_dist, c_pos, c_cnt, d_redo,
                         d_dist, d_ids, d_count, redo, &n_redo));
    *used = true;
    if (adapt && kc < kc_long && 2 * n_redo > nq) idx->filter_long = true;  // this table wants the long lists
    if (n_redo == 0) {
        idx->filter_backoff = 0;
        return 0;
    }
    int32_t *d_list = (int32_t *)((char *)ws + o_list);
    float *sub_q = (float *)((char *)ws + o_subq), *sub_inv = (float *)((char *)ws + o_subinv), *sub_d = (float *)((char *)ws + o_subd);
    int64_t *sub_i = (int64_t *)((char *)ws + o_subi);
    int32_t *sub_c = (int32_t *)((char *)ws + o_subc);
    std::vector<int32_t> list;
    for (int i = 0; i < nq; i++)
        if (redo[i]) list.push_back(i);
    auto gather_flagged = [&]() -> int {
        HVS_HIP(hipMemcpyAsync(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice, idx->stream));
        gather_queries_kernel<<<(int)list.size(), 256, 0, idx->stream>>>(d_q, d_qinv, d_list, fold, idx->ld, sub_q, sub_inv);
        HVS_HIP(hipGetLastError());
        HVS_HIP(hipStreamSynchronize(idx->stream));  // `list` is pageable host memory
        return 0;
    };
    auto scatter_flagged = [&]() -> int {
        scatter_results_kernel<<<(int)list.size(), 64, 0, idx->stream>>>(sub_d, sub_i, sub_c, d_list, k, d_dist, d_ids, d_count);
        HVS_HIP(hipGetLastError());
        return 0;
    };
    // second level: the flagged queries with the long lists (a single flagged query: one exact scan instead)
    if (kc < kc_long && sub_ok && n_redo > 1) {
        HVS_TRY(gather_flagged());
        std::vector<int32_t> redo2;
        int n_redo2 = 0;
        const int n_sub = (int)list.size();
        HVS_TRY(rescore_pass(idx, sub_q, sub_inv, n_sub, fold, k, kc_long, group_filter, d_mask, thr, eps, eps_tail, row_begin, row_end, c_dist, c_pos,
                             c_cnt, d_redo, sub_d, sub_i, sub_c, redo2, &n_redo2));
        HVS_TRY(scatter_flagged());  // unproven answers travel too and are overwritten below
        std::vector<int32_t> still;
        for (int j = 0; j < n_sub; j++)
            if (redo2[j]) still.push_back(list[j]);
        list.swap(still);
        n_redo = n_redo2;
        if (n_redo == 0) {
            idx->filter_backoff = 0;
            return 0;
        }
        HVS_HIP(hipStreamSynchronize(idx->stream));  // the scatter reads d_list, which the next level rewrites
    }
    idx->prof_launches[HVS_PROF_REDO] += n_redo;
    if (adapt) {
        if (2 * n_redo > nq) {
            idx->filter_backoff = idx->filter_backoff == 0 ? 4 : (idx->filter_backoff >= 32 ? 64 : 2 * idx->filter_backoff);
            idx->filter_skip = idx->filter_backoff;
        } else
            idx->filter_backoff = 0;
    }
    if (n_redo == 1 || (!sub_ok && n_redo <= 8)) {  // one exact scan per flagged query
        for (int i : list)
            HVS_TRY(hvs_search_scan(idx, d_q + (size_t)i * fold * idx->ld, d_qinv + (size_t)i * fold, 1, fold, k, group_filter, d_mask, thr,
                                    row_begin, row_end, d_dist + (size_t)i * k, d_ids + (size_t)i * k, d_count + i));
        return 0;
    }
    if (n_redo == nq || !sub_ok)  // every list unproven (a table of near-duplicates): the whole batch with all three products
        return hvs_search_mfma(idx, d_q, d_qinv, nq, fold, k, group_filter, d_mask, thr, row_begin, row_end, d_dist, d_ids, d_count, 1);
    // the flagged queries as a batch of their own, all three products: one more pass over the rows whatever their number
    HVS_TRY(gather_flagged());
    HVS_TRY(hvs_search_mfma(idx, sub_q, sub_inv, n_redo, fold, k, group_filter, d_mask, thr, row_begin, row_end, sub_d, sub_i, sub_c, 1));
    HVS_TRY(scatter_flagged());
    return 0;
}
