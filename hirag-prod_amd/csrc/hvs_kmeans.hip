// k-means E-step (argmin over centroids) on the MFMA tile machinery; the M-step partial sums are in
// hvs_kmeans_update.hip.
//
// assign[r] = argmin_j |x_r - c_j|^2 = argmin_j (|c_j|^2 - 2 x_r . c_j), lowest j on ties.
// Blocks own a tile of centroids and stream the corpus (the blocks of the 8 XCD groups that share a
// corpus split read the same tiles through one L2, so HBM sees every row once per iteration); each
// block writes, per corpus row, the best (value, centroid) of its tile as one 64-bit key, and a
// second kernel takes the minimum key over the centroid tiles.  No atomics, bit-reproducible.
//
// Two instantiations: fp32 operands on v_mfma_f32_32x32x2_f32 (128 x 128 tile), and the [hi | lo] fp16
// split copies of rows and centroids (hvs_split.hip) on v_mfma_f32_32x32x16_f16 (256 x 256 tile, three
// slab products per element), where the power-of-two scales of both operands are divided out of the
// accumulator before the comparison.
//
// Default for fp32 tables from 64 centroids on: the hi slabs alone as a filter (TOP2 instantiation).  One product
// instead of three gives every |c|^2 - 2 x.c to within a known bound (fp16 rounding of both operands: 2^-10 (1 + 2^-10)
// |x| |c| on the dot product); each block keeps, per corpus row, the best TWO values of its centroid tile, and
// kmeans_pick_kernel then looks at the 2 CT values of a row: a centroid whose lower bound lies above the smallest upper
// bound cannot be the argmin.  One centroid left: that is the assignment.  Several tile-bests left: their exact
// fp32 values decide (a few dot products from the fp32 rows).  A tile's second-best left: the tile may hide a third
// one, the row is flagged and goes through the three products with the other flagged rows.  The assignment is the
// exact-arithmetic argmin in every case; what varies is the time.
#include "hvs_gemm_tile.h"

struct AssignParams {
    const void *C;         // centroids [K] rows of ldc_bytes (pad columns zero): fp32, or [hi | lo] halves
    int ldc_bytes;
    const float *c_norm;   // [K] |c|^2
    const float *c_iscale; // split operands: [K] 1 / scale of the centroid's split row; else NULL
    int K;
    const void *X;         // corpus rows of ld_bytes: fp32, or the split copy
    const float *x_inv;    // split operands: inverse norms of the fp32 rows and of the split rows (their ratio is 1 / scale)
    const float *x_split_inv;
    int64_t n;             // rows
    int ld_bytes, nks;
    int split_steps;       // 0 = plain; else k-steps per slab (the K loop runs 3 of them: hi.hi, hi.lo, lo.hi)
    int hi_only;           // split operands, filter pass: the K loop runs over the hi slabs alone (nks = split_steps)
    int CT, S;             // centroid tiles, corpus splits
    int64_t T;             // corpus tiles
    uint64_t *best;        // [CT, n] keys
    unsigned int *lockstep; // optional: one arrival counter per corpus split (its CT blocks share an XCD); see lockstep_arrive
};

// Largest finite sentinel whose low byte is clear: TOP2 packs a centroid's index within its tile into the low 8 bits of
// the value's float image (truncation: 2^-15 relative, which the pick kernel adds to the bound), and an infinity with
// index bits would read as NaN.
#define KM_BIG_BITS 0x7F000000u

template <int DT, class T, bool TOP2 = false>
__global__ __launch_bounds__(T::GT, T::GT == 256 ? 2 : 1) void kmeans_assign_kernel(AssignParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // device-only types (buffer descriptors) inside
    static_assert(T::WM == 2, "the exchange below pairs wave row 1 with wave row 0");
    constexpr int MT = T::MT, NT = T::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *stages = smem;
    float *cn_lds = reinterpret_cast<float *>(smem + 2 * T::STAGE_BYTES);        // [BM] |c|^2
    float *cs_lds = cn_lds + T::BM;                                               // [BM] -2 / scale (split) or -2
    uint64_t *xch = reinterpret_cast<uint64_t *>(cs_lds + T::BM);                // [WN][NT][32] keys (wm = 1 -> wm = 0)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int h = lane >> 5;

    int ct, s;
    bool in_step = false;
    if ((p.S & 7) == 0) {
        const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
        ct = t % p.CT;
        s = xcd + 8 * (t / p.CT);
        in_step = p.lockstep != nullptr && p.CT > 1;
    } else {
        ct = blockIdx.x % p.CT;
        s = blockIdx.x / p.CT;
    }
    const int64_t tile_lo = p.T * s / p.S, tile_hi = p.T * (s + 1) / p.S;
    const int64_t a_row0 = (int64_t)ct * T::BM;
    for (int i = tid; i < T::BM; i += T::GT) {
        const int64_t c = a_row0 + i;
        cn_lds[i] = c < p.K ? p.c_norm[c] : (TOP2 ? __uint_as_float(KM_BIG_BITS) : __builtin_inff());  // padding centroids never win
        cs_lds[i] = c < p.K && p.c_iscale ? -2.0f * p.c_iscale[c] : -2.0f;
    }

    f32x16 acc[MT][NT];
    zero_acc(acc);
    const int64_t total_steps = (tile_hi - tile_lo) * p.nks;
    TileSrcT<T::DMA_N> srcA, srcB;
    {
        const int64_t a_valid = p.K - a_row0;
        tile_src_init<T>(srcA, p.C, a_row0, a_valid < T::BM ? (int)a_valid : T::BM, p.ldc_bytes, tid);
    }
    auto init_b = [&](int64_t tile) {
        const int64_t r0 = tile * T::BN;
        const int64_t valid = p.n - r0;
        tile_src_init<T>(srcB, p.X, r0, valid < T::BN ? (int)valid : T::BN, p.ld_bytes, tid);
    };
    if (total_steps > 0) {
        init_b(tile_lo);
        dma_tiles<T>(srcA, srcB, stages, koff_plain(0), wave);
    }
    __syncthreads();

    float x_is[NT];  // split operands: 1 / scale of this lane's rows
    // (centroid, row) byte offsets of k-step k inside a row.  All three products of split operands run three slab pairs,
    // (hi, hi), (hi, lo), (lo, hi), of split_steps steps each; anything else is the first pair of a split that never ends
    const int slab_steps = p.split_steps && !p.hi_only ? p.split_steps : 0x3fffffff;
    auto step_off = [&](int k) {
        const int slab = (k >= 2 * slab_steps ? 1 : 0) + (k >= slab_steps ? 1 : 0), w = k - slab * slab_steps;
        return KOff{((slab == 2 ? slab_steps : 0) + w) * KB, ((slab == 1 ? slab_steps : 0) + w) * KB};
    };
    // A tile loop around a k loop without tile bookkeeping in it; the tile's last step (whose LDS-DMA fetches step 0 of the
    // next tile) is written out after it.
    int cur = 0;
    for (int64_t tile = tile_lo; tile < tile_hi; tile++) {
        if (tid == 0 && (p.S & 7) == 0 && p.lockstep != nullptr && p.CT > 1)
            lockstep_arrive(p.lockstep + s, (unsigned int)(tile - tile_lo + 1), p.CT, in_step);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            x_is[nt] = 1.0f;
            if (p.x_split_inv) {
                const int64_t row = tile * T::BN + wn * (NT * 32) + nt * 32 + (lane & 31);
                if (row < p.n) {
                    const float inv = p.x_inv[row];
                    x_is[nt] = inv < __builtin_inff() ? p.x_split_inv[row] / inv : 1.0f;  // exact: the ratio is a power of two
                }
            }
        }
        for (int ks = 0; ks + 1 < p.nks; ks++) {
            compute_step<DT, T>(acc, stages + cur * T::STAGE_BYTES, wm, wn, lane, srcA, srcB, stages + (cur ^ 1) * T::STAGE_BYTES,
                                step_off(ks + 1), wave);
            __syncthreads();
            cur ^= 1;
        }
        if (tile + 1 < tile_hi) init_b(tile + 1);  // after the last tile: its own step 0 is fetched again (unused)
        compute_step<DT, T>(acc, stages + cur * T::STAGE_BYTES, wm, wn, lane, srcA, srcB, stages + (cur ^ 1) * T::STAGE_BYTES,
                            step_off(0), wave);
        cur ^= 1;

        uint64_t bestk[NT];
        if (TOP2) {
            static_assert(!TOP2 || T::BM <= 256, "the index within a tile is packed into 8 bits");
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                // the lane's two smallest values, index in the low byte: min + median-of-three per value
                float b1 = __uint_as_float(KM_BIG_BITS), b2 = b1;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const int a_loc = wm * (MT * 32) + mt * 32 + 8 * gq + 4 * h;
                        const f32x4 cn = *reinterpret_cast<const f32x4 *>(cn_lds + a_loc);
                        const f32x4 cs = *reinterpret_cast<const f32x4 *>(cs_lds + a_loc);
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float v = fmaf(acc[mt][nt][4 * gq + e] * x_is[nt], cs[e], cn[e]);
                            // bits of wm and h are added once per lane below
                            const float pk = __uint_as_float((__float_as_uint(v) & 0xFFFFFF00u) | (uint32_t)(mt * 32 + 8 * gq + e));
                            const float n1 = fminf(b1, pk);
                            b2 = __builtin_amdgcn_fmed3f(b1, b2, pk);
                            b1 = n1;
                        }
                    }
                const uint32_t lb = (uint32_t)(wm * (MT * 32) + 4 * h);
                b1 = __uint_as_float(__float_as_uint(b1) | lb);
                b2 = __uint_as_float(__float_as_uint(b2) | lb);
                // the other lane half holds the other centroid rows of the same corpus row
                const float o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
                const float n1 = fminf(b1, o1), n2 = fminf(fmaxf(b1, o1), fminf(b2, o2));
                bestk[nt] = ((uint64_t)__float_as_uint(n2) << 32) | __float_as_uint(n1);
            }
            zero_acc(acc);
            if (wm == 1 && h == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) xch[(wn * NT + nt) * 32 + (lane & 31)] = bestk[nt];
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                // per lane: strict "<" over increasing centroid index keeps the lowest index on ties
                float bv = __builtin_inff();
                int bi = -1;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const int a_loc = wm * (MT * 32) + mt * 32 + 8 * gq + 4 * h;
                        const f32x4 cn = *reinterpret_cast<const f32x4 *>(cn_lds + a_loc);
                        const f32x4 cs = *reinterpret_cast<const f32x4 *>(cs_lds + a_loc);
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            // |c|^2 - 2 x.c; with split operands the accumulator holds x.c times both scales
                            const float v = DT == HVS_F32 ? fmaf(-2.0f, acc[mt][nt][4 * gq + e], cn[e])
                                                          : fmaf(acc[mt][nt][4 * gq + e] * x_is[nt], cs[e], cn[e]);
                            const bool lt = v < bv;  // false for NaN
                            bv = lt ? v : bv;
                            bi = lt ? a_loc + e : bi;
                        }
                    }
                uint64_t b = bi >= 0 ? hvs_make_key(hvs_ord(bv), (uint32_t)(a_row0 + bi)) : HVS_KEY_EMPTY;
                // the other lane half holds the other centroid rows of the same corpus row
                const uint32_t olo = __shfl_xor((uint32_t)b, 32);
                const uint32_t ohi = __shfl_xor((uint32_t)(b >> 32), 32);
                const uint64_t o = ((uint64_t)ohi << 32) | olo;
                bestk[nt] = o < b ? o : b;
            }
            zero_acc(acc);
            if (wm == 1 && h == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) xch[(wn * NT + nt) * 32 + (lane & 31)] = bestk[nt];
            }
        }

        __syncthreads();

        if (wm == 0 && h == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const uint64_t o = xch[(wn * NT + nt) * 32 + (lane & 31)];
                uint64_t b;
                if (TOP2) {  // merge two sorted pairs
                    const float a1 = __uint_as_float((uint32_t)bestk[nt]), a2 = __uint_as_float((uint32_t)(bestk[nt] >> 32));
                    const float o1 = __uint_as_float((uint32_t)o), o2 = __uint_as_float((uint32_t)(o >> 32));
                    const float n1 = fminf(a1, o1), n2 = fminf(fmaxf(a1, o1), fminf(a2, o2));
                    b = ((uint64_t)__float_as_uint(n2) << 32) | __float_as_uint(n1);
                } else {
                    b = o < bestk[nt] ? o : bestk[nt];
                }
                const int64_t row = tile * T::BN + wn * (NT * 32) + nt * 32 + (lane & 31);
                if (row < p.n) p.best[(size_t)ct * p.n + row] = b;
            }
        }
        if (p.nks == 1) __syncthreads();  // xch is rewritten a whole K loop later otherwise
    }
#endif
}

// assign[r] = centroid of the minimum key over the CT tiles; deleted rows -> -1
__global__ void kmeans_reduce_kernel(const uint64_t *__restrict__ best, int CT, int64_t n, const uint32_t *__restrict__ deleted,
                                     int32_t *__restrict__ assign) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        uint64_t b = HVS_KEY_EMPTY;
        for (int t = 0; t < CT; t++) {
            const uint64_t k = best[(size_t)t * n + r];
            b = k < b ? k : b;
        }
        int32_t a = b == HVS_KEY_EMPTY ? -1 : (int32_t)(uint32_t)b;
        if (deleted && ((deleted[r >> 5] >> (r & 31)) & 1u)) a = -1;
        assign[r] = a;
    }
}

// ---- filter route: the rows' assignments from the best two values per (centroid tile, row) of the hi-slab pass ----
struct PickParams {
    const float *best;      // [CT, n, 2] packed values (index within the tile in the low byte), smaller first
    int CT, bm;             // centroid tiles, centroids per tile
    int64_t n;
    const float *x_inv;     // [n] 1 / |x| of the fp32 rows (+inf for a zero row)
    const float *rows;      // fp32 rows, pitch ld floats
    int ld;
    const float *c_pad;     // fp32 centroids, pitch ldc floats (pad columns zero), ldc >= ld
    int ldc;
    const float *c_norm;    // [K] |c|^2
    const float *cmax;      // [1] largest |c|
    float eps2;             // 2 * (bound of the hi-slab dot product relative to |x| |c|) for any operands
    float eps_tail;         // the part of that bound that does not come from rounding the operands to fp16
    const uint32_t *x_rel_max, *c_rel_max;  // [1] each, float bits: largest |v - hi slab| / |v| over the rows / the centroids
    const uint32_t *deleted;
    int32_t *assign;        // [n]
    uint32_t *flag_list;    // rows that need every product: [n] positions, any order
    uint32_t *flag_count;
};

// A block takes PICK_ROWS rows at a time: the 2 CT values of each are read tile by tile (rows on lanes: the layout the
// assignment kernel writes, 512 contiguous bytes per wave and tile) into LDS, then one wave per row goes over them with
// value j of the row (j = 2 * tile + which) in lane j & 63 of register j >> 6.
#define PICK_ROWS 64
template <int R> __global__ __launch_bounds__(256) void kmeans_pick_kernel(PickParams q) {
    extern __shared__ __attribute__((aligned(16))) float pick_vals[];  // [PICK_ROWS][2 CT + 2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float cmax = *q.cmax;
    const float rx = __uint_as_float(*q.x_rel_max), rc = __uint_as_float(*q.c_rel_max);
    const float eps2 = fminf(q.eps2, 2.0f * (rx + rc * (1.0f + rx) + q.eps_tail));  // x.(c - ch) + (x - xh).ch by Cauchy-Schwarz
    const int nv = 2 * q.CT, pitch = nv + 2;
    for (int64_t row0 = blockIdx.x * (int64_t)PICK_ROWS; row0 < q.n; row0 += (int64_t)gridDim.x * PICK_ROWS) {
        for (int i = threadIdx.x; i < q.CT * PICK_ROWS; i += 256) {
            const int ct = i / PICK_ROWS, r = i % PICK_ROWS;
            if (row0 + r < q.n)
                *reinterpret_cast<float2 *>(pick_vals + r * pitch + 2 * ct) =
                    *reinterpret_cast<const float2 *>(q.best + ((size_t)ct * q.n + row0 + r) * 2);
        }
        __syncthreads();
        for (int r = wave; r < PICK_ROWS; r += 4) {
            const int64_t row = row0 + r;
            if (row >= q.n) break;
            if (q.deleted && ((q.deleted[row >> 5] >> (row & 31)) & 1u)) {
                if (lane == 0) q.assign[row] = -1;
                continue;
            }
            float v[R], lo[R];
            const float inv = q.x_inv[row];
            const float xn = inv < __builtin_inff() ? 1.0f / inv : 0.0f;  // a zero row: every value is |c|^2 exactly
            const float eb = eps2 * xn * cmax * 1.0001f;
            float hi_min = __builtin_inff();
#pragma unroll
            for (int e = 0; e < R; e++) {
                const int j = e * 64 + lane;
                v[e] = j < nv ? pick_vals[r * pitch + j] : __builtin_nanf("");
                const float err = eb + fabsf(v[e]) * 3.1e-5f;  // + the truncated low byte (2^-15 relative)
                lo[e] = v[e] - err;
                hi_min = fminf(hi_min, v[e] + err);  // fminf drops NaN
            }
            for (int off = 32; off > 0; off >>= 1) hi_min = fminf(hi_min, __shfl_xor(hi_min, off));
            uint64_t m[R];
            int count = 0;
            bool second = false;
#pragma unroll
            for (int e = 0; e < R; e++) {
                m[e] = __ballot(lo[e] <= hi_min);  // false for NaN
                count += __popcll(m[e]);
                second |= (m[e] & 0xAAAAAAAAAAAAAAAAull) != 0;  // odd j: a tile's second best
            }
            if (count == 0 || second) {  // no finite value at all, or a tile that may hide a third contender
                if (lane == 0) {
                    q.flag_list[atomicAdd(q.flag_count, 1u)] = (uint32_t)row;  // integer: the list's order does not reach the result
                    q.assign[row] = -1;
                }
                continue;
            }
            uint64_t best_key = HVS_KEY_EMPTY;
#pragma unroll
            for (int e = 0; e < R; e++) {
                uint64_t pending = m[e];
                while (pending) {
                    const int src = __ffsll((unsigned long long)pending) - 1;
                    pending &= pending - 1;
                    const uint32_t bits = __builtin_amdgcn_readlane(__float_as_uint(v[e]), src);
                    const uint32_t c = (uint32_t)(((e * 64 + src) >> 1) * q.bm) + (bits & 0xFFu);
                    if (count == 1) {
                        best_key = c;
                        break;
                    }
                    // several tile-bests within the bound of each other: exact fp32 values decide
                    const float *x = q.rows + (size_t)row * q.ld, *cc = q.c_pad + (size_t)c * q.ldc;
                    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
                    for (int i = lane * 4; i < q.ld; i += 256) {  // ld is a multiple of 4, pad columns are zero on both sides
                        const float4 a = *reinterpret_cast<const float4 *>(x + i);
                        const float4 b = *reinterpret_cast<const float4 *>(cc + i);
                        s0 = fmaf(a.x, b.x, s0);
                        s1 = fmaf(a.y, b.y, s1);
                        s2 = fmaf(a.z, b.z, s2);
                        s3 = fmaf(a.w, b.w, s3);
                    }
                    float dot = (s0 + s1) + (s2 + s3);
                    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
                    const float val = fmaf(-2.0f, dot, q.c_norm[c]);
                    if (val == val) {
                        const uint64_t key = hvs_make_key(hvs_ord(val), c);  // ties go to the lower centroid
                        best_key = key < best_key ? key : best_key;
                    }
                }
            }
            if (lane == 0) q.assign[row] = best_key == HVS_KEY_EMPTY ? -1 : (int32_t)(uint32_t)best_key;
        }
        __syncthreads();  // the values are overwritten by the next group of rows
    }
}

// cmax[0] = largest |c| (one block)
__global__ __launch_bounds__(256) void kmeans_cmax_kernel(const float *__restrict__ c_norm, int K, float *__restrict__ cmax) {
    __shared__ float part[4];
    float m = 0.0f;
    for (int i = threadIdx.x; i < K; i += 256) m = fmaxf(m, c_norm[i]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) cmax[0] = sqrtf(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))) * 1.000001f;
}

// the flagged rows as a table of their own: split rows and both inverse norms gathered (one wave per row) ...
__global__ __launch_bounds__(256) void kmeans_gather_kernel(const char *__restrict__ split, int pitch16, const float *__restrict__ x_inv,
                                                            const float *__restrict__ x_split_inv, const uint32_t *__restrict__ list,
                                                            uint32_t n_list, char *__restrict__ dst, float *__restrict__ dst_inv,
                                                            float *__restrict__ dst_split_inv) {
    const int lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (i >= n_list) return;
    const uint32_t r = list[i];
    const uint4 *s = reinterpret_cast<const uint4 *>(split + (size_t)r * pitch16 * 16);
    uint4 *d = reinterpret_cast<uint4 *>(dst + (size_t)i * pitch16 * 16);
    for (int c = lane; c < pitch16; c += 64) d[c] = s[c];
    if (lane == 0) {
        dst_inv[i] = x_inv[r];
        dst_split_inv[i] = x_split_inv[r];
    }
}

// ... and their assignments put back in their places
__global__ void kmeans_scatter_kernel(const int32_t *__restrict__ sub, const uint32_t *__restrict__ list, uint32_t n_list,
                                      int32_t *__restrict__ assign) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_list) assign[list[i]] = sub[i];
}

// c_pad[j][0..ld) = centroid j zero-padded; c_norm[j] = |c_j|^2.  One wave per centroid.
__global__ void prep_centroids_kernel(const float *__restrict__ src, float *__restrict__ dst, float *__restrict__ c_norm,
                                      int K, int dim, int ldc) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= K) return;
    float s = 0.0f;
    for (int c = lane; c < ldc; c += 64) {
        const float v = c < dim ? src[(size_t)j * dim + c] : 0.0f;
        dst[(size_t)j * ldc + c] = v;
        s = fmaf(v, v, s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) c_norm[j] = s;
}

int hvs_kmeans_update(hvs_index *idx, const int32_t *d_assign, int64_t n, int K, float *d_sums, int64_t *d_counts);  // hvs_kmeans_update.hip

template <class T> static size_t assign_lds_bytes() { return (size_t)2 * T::STAGE_BYTES + T::BM * 8 + T::WN * T::NT * 32 * 8; }

__global__ void fill_ones_kernel(float *dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = 1.0f;
}

template <int DT, class T, bool TOP2 = false> static int launch_assign(AssignParams &p, int64_t n, hipStream_t st) {
    p.CT = (p.K + T::BM - 1) / T::BM;
    p.T = (n + T::BN - 1) / T::BN;
    int S = ((T::GT == 256 ? 512 : 256) + p.CT - 1) / p.CT;  // fill every block slot of the chip
    S = (S + 7) / 8 * 8;
    if ((int64_t)S > p.T) S = (int)p.T;
    if (S >= 8) S = S / 8 * 8;
    if (S < 1) S = 1;
    p.S = S;
    if (p.lockstep) HVS_HIP(hipMemsetAsync(p.lockstep, 0, 4096 * 4, st));
    if ((S & 7) != 0 || S > 4096 || p.CT * S > (T::GT == 256 ? 512 : 256)) p.lockstep = nullptr;  // only when every block is resident at once
    const size_t lds = assign_lds_bytes<T>();
    HVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&kmeans_assign_kernel<DT, T, TOP2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    kmeans_assign_kernel<DT, T, TOP2><<<p.CT * S, T::GT, lds, st>>>(p);
    HVS_HIP(hipGetLastError());
    return 0;
}

extern "C" int hvs_kmeans_assign(hvs_index *idx, const float *centroids, int n_centroids, int32_t *out_assign,
                                 float *out_sums, int64_t *out_counts, int on_device) {
    HVS_REQUIRE(idx != nullptr, "hvs_kmeans_assign: idx is NULL");
    HVS_REQUIRE(idx->dtype == HVS_F32, "hvs_kmeans_assign: the index must store f32 rows");
    HVS_REQUIRE(centroids && (out_assign || idx->n == 0), "hvs_kmeans_assign: NULL argument");
    HVS_REQUIRE(n_centroids >= 1, "hvs_kmeans_assign: need at least one centroid");
    HVS_REQUIRE((out_sums == nullptr) == (out_counts == nullptr), "hvs_kmeans_assign: pass both out_sums and out_counts or neither");
    const int64_t n = idx->n;
    if (n == 0) {
        // an empty shard still owes its all-reduce partner zeros, not whatever the buffers held
        if (out_sums) {
            const size_t sum_bytes = (size_t)n_centroids * idx->dim * 4, cnt_bytes = (size_t)n_centroids * 8;
            if (on_device) {
                HVS_HIP(hipSetDevice(idx->device));
                HVS_HIP(hipMemsetAsync(out_sums, 0, sum_bytes, idx->stream));
                HVS_HIP(hipMemsetAsync(out_counts, 0, cnt_bytes, idx->stream));
            } else {
                memset(out_sums, 0, sum_bytes);
                memset(out_counts, 0, cnt_bytes);
            }
        }
        return 0;
    }
    HVS_HIP(hipSetDevice(idx->device));
    const int K = n_centroids, dim = idx->dim, ld = idx->ld;
    const int ldc = (ld + 31) / 32 * 32;
    hipStream_t st = idx->stream;

    // the [hi | lo] fp16 copies of rows and centroids put the contraction on the 16-bit matrix pipe
    bool split = false;
    if (K >= 64 && !getenv("HVS_NO_SPLIT")) HVS_TRY(hvs_split_ensure(idx, &split));
    const int ld16 = (dim + 63) / 64 * 64;

    // workspace 0: padded centroids + norms (+ split centroids, their 1/scale, ones) (+ raw host copy);
    // workspace 1: best keys; workspace 3: host staging
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t cpad_bytes = (size_t)K * ldc * 4;
    const size_t norm_off = up(cpad_bytes);
    const size_t csplit_off = up(norm_off + (size_t)K * 4);
    const size_t cis_off = csplit_off + (split ? up((size_t)K * ld16 * 4) : 0);
    const size_t ones_off = cis_off + (split ? up((size_t)K * 4) : 0);
    const size_t raw_off = ones_off + (split ? up((size_t)K * 4) : 0);
    const size_t raw_bytes = on_device ? 0 : (size_t)K * dim * 4;
    const size_t rel_off = up(raw_off + raw_bytes);  // [1] largest |c - hi slab| / |c| of the split centroids (float bits)
    const size_t ls_off = rel_off + 256;
    void *ws0 = nullptr;
    HVS_TRY(hvs_ws_get(idx, 0, ls_off + 4096 * 4, &ws0));
    const float *src = centroids;
    if (!on_device) {
        HVS_HIP(hipMemcpyAsync((char *)ws0 + raw_off, centroids, raw_bytes, hipMemcpyHostToDevice, st));
        src = (const float *)((char *)ws0 + raw_off);
    }
    float *c_pad = (float *)ws0;
    float *c_norm = (float *)((char *)ws0 + norm_off);
    prep_centroids_kernel<<<(K + 3) / 4, 256, 0, st>>>(src, c_pad, c_norm, K, dim, ldc);
    HVS_HIP(hipGetLastError());

    AssignParams p;
    memset(&p, 0, sizeof(p));
    p.c_norm = c_norm;
    p.K = K;
    p.n = n;
    p.lockstep = getenv("HVS_NO_LOCKSTEP") ? nullptr : (unsigned int *)((char *)ws0 + ls_off);
    if (split) {
        float *c_is = (float *)((char *)ws0 + cis_off), *ones = (float *)((char *)ws0 + ones_off);
        fill_ones_kernel<<<(K + 255) / 256, 256, 0, st>>>(ones, K);
        HVS_HIP(hipGetLastError());
        // "inverse norm" 1 in, 1 / scale out
        HVS_HIP(hipMemsetAsync((char *)ws0 + rel_off, 0, 4, st));
        HVS_TRY(hvs_split_rows(st, c_pad, ones, 0, K, ldc, ldc, ld16, (char *)ws0 + csplit_off, c_is, nullptr, (uint32_t *)((char *)ws0 + rel_off)));
        p.C = (char *)ws0 + csplit_off;
        p.ldc_bytes = ld16 * 4;
        p.c_iscale = c_is;
        p.X = idx->split;
        p.x_inv = idx->inv_norm;
        p.x_split_inv = idx->split_inv;
        p.ld_bytes = ld16 * 4;
        p.split_steps = ld16 * 2 / KB;
        p.nks = 3 * p.split_steps;
    } else {
        p.C = c_pad;
        p.ldc_bytes = ldc * 4;
        p.X = idx->rows;
        p.ld_bytes = ld * 4;
        p.nks = ldc / 32;
    }
    // the 256 x 256 tile: split operands beyond one 128-centroid tile; fp32 operands when it pads no more centroids than the 128 tile
    const bool big = K > Tile128::BM && (split || (K + Tile256::BM - 1) / Tile256::BM * Tile256::BM == (K + Tile128::BM - 1) / Tile128::BM * Tile128::BM);
    const int bm = big ? Tile256::BM : Tile128::BM;
    void *best = nullptr;
    HVS_TRY(hvs_ws_get(idx, 1, (size_t)((K + bm - 1) / bm) * n * 8, &best));
    p.best = (uint64_t *)best;

    int32_t *d_assign = out_assign;
    float *d_sums = out_sums;
    int64_t *d_counts = out_counts;
    if (!on_device) {
        const size_t o_sums = ((size_t)n * 4 + 255) & ~(size_t)255;
        const size_t o_cnt = (o_sums + (out_sums ? (size_t)K * dim * 4 : 0) + 255) & ~(size_t)255;
        void *stage = nullptr;
        HVS_TRY(hvs_ws_get(idx, 3, o_cnt + (size_t)K * 8, &stage));
        d_assign = (int32_t *)stage;
        if (out_sums) {
            d_sums = (float *)((char *)stage + o_sums);
            d_counts = (int64_t *)((char *)stage + o_cnt);
        }
    }

    HVS_TRY(hvs_prof_begin(idx));
    // the hi slabs as a filter: best two values per (centroid tile, row) from one product, then kmeans_pick_kernel
    bool done = false;
    const int CT = (K + bm - 1) / bm;
    bool filter = split && 2 * CT <= 256 && n <= (int64_t)0xFFFFFFFFll && !getenv("HVS_NO_FILTER");
    if (filter && idx->km_filter_skip > 0 && !getenv("HVS_NO_ADAPT")) {  // recent filter passes on this table flagged too many rows
        idx->km_filter_skip--;
        filter = false;
    }
    if (filter) {
        AssignParams p1 = p;
        p1.hi_only = 1;
        p1.nks = p.split_steps;
        if (!big) HVS_TRY((launch_assign<HVS_F16, Tile128, true>(p1, n, st)));
        else HVS_TRY((launch_assign<HVS_F16, Tile256, true>(p1, n, st)));
        void *fl = nullptr;
        HVS_TRY(hvs_ws_get(idx, 2, 256 + (size_t)n * 4, &fl));
        uint32_t *flag_count = (uint32_t *)fl, *flag_list = (uint32_t *)((char *)fl + 256);
        float *cmax = (float *)((char *)fl + 64);
        HVS_HIP(hipMemsetAsync(fl, 0, 256, st));
        kmeans_cmax_kernel<<<1, 256, 0, st>>>(c_norm, K, cmax);
        HVS_HIP(hipGetLastError());
        PickParams q;
        memset(&q, 0, sizeof(q));
        q.best = (const float *)p.best;
        q.CT = p1.CT;
        q.bm = bm;
        q.n = n;
        q.x_inv = idx->inv_norm;
        q.rows = (const float *)idx->rows;
        q.ld = ld;
        q.c_pad = c_pad;
        q.ldc = ldc;
        q.c_norm = c_norm;
        q.cmax = cmax;
        // |x.c - xh.ch / scales| <= (2^-10 (1 + 2^-10) + n 2^-24 [fp32 accumulation of exact products]) |x| |c|, twice that on |c|^2 - 2 x.c;
        // the kernel replaces the first term by what split_rows_kernel measured on these rows and centroids where that is less
        q.eps_tail = (float)ld16 * 6.0e-8f + 1e-6f;
        q.eps2 = 2.0f * (9.78e-4f + q.eps_tail);
        q.x_rel_max = idx->split_rel;
        q.c_rel_max = (const uint32_t *)((char *)ws0 + rel_off);
        q.deleted = idx->n_deleted > 0 ? idx->deleted : nullptr;
        q.assign = d_assign;
        q.flag_list = flag_list;
        q.flag_count = flag_count;
        int pg = (int)((n + PICK_ROWS - 1) / PICK_ROWS);
        if (pg > 256 * 16) pg = 256 * 16;
        const int nv = 2 * p1.CT;
        const size_t pick_lds = (size_t)PICK_ROWS * (nv + 2) * 4;  // at most 64 x 258 floats
        if (nv <= 64) kmeans_pick_kernel<1><<<pg, 256, pick_lds, st>>>(q);
        else if (nv <= 128) kmeans_pick_kernel<2><<<pg, 256, pick_lds, st>>>(q);
        else {
            HVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&kmeans_pick_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pick_lds));
            kmeans_pick_kernel<4><<<pg, 256, pick_lds, st>>>(q);
        }
        HVS_HIP(hipGetLastError());
        uint32_t n_flag = 0;
        HVS_HIP(hipMemcpyAsync(&n_flag, flag_count, 4, hipMemcpyDeviceToHost, st));
        HVS_HIP(hipStreamSynchronize(st));
        idx->prof_launches[HVS_PROF_REDO] += n_flag;
        if ((int64_t)n_flag * 4 > n) {  // a table of tight clusters: the whole table through the three products, and not this way again for a while
            idx->km_filter_backoff = idx->km_filter_backoff == 0 ? 4 : (idx->km_filter_backoff >= 32 ? 64 : 2 * idx->km_filter_backoff);
            idx->km_filter_skip = idx->km_filter_backoff;
        } else {
            idx->km_filter_backoff = 0;
            done = true;
            if (n_flag > 0) {  // the flagged rows as a table of their own, all three products
                const int pitch = ld16 * 4;
                const size_t o_inv = up((size_t)n_flag * pitch + HVS_ROWS_SLACK), o_sinv = o_inv + up((size_t)n_flag * 4),
                             o_sub = o_sinv + up((size_t)n_flag * 4);
                void *g = nullptr, *sub_best = nullptr;
                HVS_TRY(hvs_ws_get(idx, 7, o_sub + up((size_t)n_flag * 4), &g));
                HVS_TRY(hvs_ws_get(idx, 4, (size_t)CT * n_flag * 8, &sub_best));
                HVS_HIP(hipMemsetAsync((char *)g + (size_t)n_flag * pitch, 0, HVS_ROWS_SLACK, st));
                float *g_inv = (float *)((char *)g + o_inv), *g_sinv = (float *)((char *)g + o_sinv);
                int32_t *sub_assign = (int32_t *)((char *)g + o_sub);
                kmeans_gather_kernel<<<(n_flag + 3) / 4, 256, 0, st>>>((const char *)idx->split, pitch / 16, idx->inv_norm, idx->split_inv, flag_list,
                                                                       n_flag, (char *)g, g_inv, g_sinv);
                HVS_HIP(hipGetLastError());
                AssignParams p3 = p;
                p3.X = g;
                p3.x_inv = g_inv;
                p3.x_split_inv = g_sinv;
                p3.n = n_flag;
                p3.best = (uint64_t *)sub_best;
                if (!big) HVS_TRY((launch_assign<HVS_F16, Tile128>(p3, n_flag, st)));
                else HVS_TRY((launch_assign<HVS_F16, Tile256>(p3, n_flag, st)));
                kmeans_reduce_kernel<<<(n_flag + 255) / 256, 256, 0, st>>>(p3.best, p3.CT, n_flag, nullptr, sub_assign);
                kmeans_scatter_kernel<<<(n_flag + 255) / 256, 256, 0, st>>>(sub_assign, flag_list, n_flag, d_assign);
                HVS_HIP(hipGetLastError());
            }
        }
    }
    if (!done) {
        if (!split && big) HVS_TRY((launch_assign<HVS_F32, Tile256>(p, n, st)));
        else if (!split) HVS_TRY((launch_assign<HVS_F32, Tile128>(p, n, st)));
        else if (!big) HVS_TRY((launch_assign<HVS_F16, Tile128>(p, n, st)));
        else HVS_TRY((launch_assign<HVS_F16, Tile256>(p, n, st)));
        int rg = (int)((n + 255) / 256);
        if (rg > 256 * 8) rg = 256 * 8;
        kmeans_reduce_kernel<<<rg, 256, 0, st>>>(p.best, p.CT, n, idx->n_deleted > 0 ? idx->deleted : nullptr, d_assign);
        HVS_HIP(hipGetLastError());
    }
    HVS_TRY(hvs_prof_end(idx, HVS_PROF_KMEANS_ASSIGN));

    if (out_sums) {
        HVS_TRY(hvs_prof_begin(idx));
        HVS_TRY(hvs_kmeans_update(idx, d_assign, n, K, d_sums, d_counts));
        HVS_TRY(hvs_prof_end(idx, HVS_PROF_KMEANS_UPDATE));
    }
    if (!on_device) {
        HVS_HIP(hipMemcpyAsync(out_assign, d_assign, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        if (out_sums) {
            HVS_HIP(hipMemcpyAsync(out_sums, d_sums, (size_t)K * dim * 4, hipMemcpyDeviceToHost, st));
            HVS_HIP(hipMemcpyAsync(out_counts, d_counts, (size_t)K * 8, hipMemcpyDeviceToHost, st));
        }
        HVS_HIP(hipStreamSynchronize(st));
    }
    return 0;
}
