// Shared tile machinery of the MFMA contraction (search and k-means kernels): 128 x 128 or 256 x 256 block tile,
// k-steps of 128 bytes per row (32 fp32 or 64 fp16/bf16 elements), double-buffered XOR-swizzled LDS
// image filled by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write, no per-step
// address arithmetic), v_mfma_f32_32x32x2_f32 for fp32 rows and v_mfma_f32_32x32x16_{f16,bf16} for
// 16-bit rows (fp32 accumulation in both).
#pragma once
#include "hvs_internal.h"

#define KB 128        // bytes of one tile row per k-step
#define CAND_MAX 1024   // capacity of a candidate array (entries) for k > 128: 16 keys per lane when a wave holds one in registers
#define CAND_SMALL 512  // ... for k <= 128: 8 keys per lane
#define MFMA_MAX_K 256   // largest k whose per-block lists are sorted inside the search kernels and merged list by list
#define MFMA_WIDE_K 768  // largest k the batched kernels take: CAND_MAX less one 256-row tile of survivors; above MFMA_MAX_K the
                         // blocks append their best k, unsorted, to one list per query that select_pooled_kernel ranks

// Block-tile shapes.  BM x BN outputs per block, WM x WN waves, each wave (BM / WM) x (BN / WN) as
// MT x NT MFMA tiles of 32 x 32.
//   Tile128: 128 x 128, 4 waves of 64 x 64, two blocks per CU.  Enough for fp32 rows, where a k-step is
//            64 MFMA-cycles per 16 bytes of fragment and the LDS / LDS-DMA paths idle most of the time.
//   Tile256: 256 x 256, 8 waves of 128 x 64, one block per CU.  For 16-bit rows, whose MFMAs retire 8 x
//            the elements per cycle: per MFMA the block moves half the global->LDS bytes of Tile128
//            (0.25 vs 0.5 KB; the vector-memory path delivers 64 B/clk/CU, which Tile128 needs in full)
//            and reads 0.75 instead of 1 KB of fragments.
template <int BM_, int BN_, int WM_, int WN_> struct Tile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int NW = WM * WN;      // waves per block
    static constexpr int GT = 64 * NW;      // threads per block
    static constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
    static constexpr int STAGE_BYTES = (BM + BN) * KB;  // one stage: A image then X image
    static constexpr int DMA_N = BM * KB / 16 / GT;     // LDS-DMA instructions per thread, operand and k-step
    static_assert(BM == BN, "the A and X pieces of a k-step are issued alternately, one each");
    static_assert(BM * KB / 16 % GT == 0, "whole pieces per thread");
};
typedef Tile<128, 128, 2, 2> Tile128;
typedef Tile<256, 256, 2, 4> Tile256;
//   Tile64:  64 x 64, 2 waves of 64 x 32, four blocks per CU.  For small batches on the 16-bit pipe: a block always
//            contracts BM query rows, so up to 64 embeddings cost half the MFMA work of Tile128 and the search
//            falls back onto the HBM stream.
typedef Tile<64, 64, 1, 2> Tile64;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

// LDS image of one operand tile: [128 rows][8 slots of 16 B].  The slot holding 16-byte k-chunk `kc`
// of `row` is kc ^ ((row >> 1) & 7), chosen so that the ds_read_b128 fragment reads (32 rows x one
// chunk per lane half; 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}) touch 16 distinct
// 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int lds_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int lds_off(int row, int kc) { return (row * 8 + (kc ^ lds_swz(row))) * 16; }  // bytes

// An LDS-DMA wave-instruction writes 64 consecutive 16-byte slots (wave-uniform base + lane * 16), so
// the swizzle is applied to the SOURCE address: thread t, instruction i fills LDS slot P = GT i + t,
// i.e. tile row P >> 3, and must fetch that row's k-chunk (P & 7) ^ swz(row).
// Returns the byte offset of that chunk inside a tile whose rows are `pitch_bytes` apart; rows past
// `last_row` (ragged last tile) re-read the last valid row.
__device__ __forceinline__ int dma_src_offset(int tid, int i, int pitch_bytes, int last_row, int threads) {
    const int P = threads * i + tid;
    int row = P >> 3;
    const int kc = (P & 7) ^ lds_swz(row);
    row = row > last_row ? last_row : row;
    return row * pitch_bytes + kc * 16;
}

template <int N> struct TileSrcT {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[N];
};
typedef TileSrcT<Tile128::DMA_N> TileSrc;

// Descriptor over rows [row0, row0 + rows_valid) of a row-major matrix; reads past the end return 0.
template <class T = Tile128>
__device__ __forceinline__ void tile_src_init(TileSrcT<T::DMA_N> &t, const void *base, int64_t row0, int rows_valid,
                                              int pitch_bytes, int tid) {
    const char *p = reinterpret_cast<const char *>(base) + row0 * pitch_bytes;
    // base and size are wave-uniform by construction; say so, or every DMA is wrapped in a waterfall loop
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane(rows_valid * pitch_bytes);
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < T::DMA_N; i++) t.voff[i] = dma_src_offset(tid, i, pitch_bytes, rows_valid - 1, T::GT);
}

// Where the 128-byte k-slice of one k-step starts inside a row of each operand (bytes).  Plain operands:
// both = step * KB.  Split operands (see KSplit) read different slabs of their rows in the same step.
struct KOff {
    int a, b;
};
__device__ __forceinline__ KOff koff_plain(int step) { return KOff{step * KB, step * KB}; }

// One LDS-DMA instruction of a k-step: j in [0, 2 DMA_N): even j -> A piece j/2, odd j -> X piece j/2.
template <class T = Tile128>
__device__ __forceinline__ void dma_piece(const TileSrcT<T::DMA_N> &a, const TileSrcT<T::DMA_N> &b, char *stage, KOff off,
                                          int wave, int j) {
    const int i = j >> 1;
    char *d = stage + (T::GT * i + wave * 64) * 16 + ((j & 1) ? T::BM * KB : 0);
    const TileSrcT<T::DMA_N> &t = (j & 1) ? b : a;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(t.rsrc, (lds_void *)d, 16, t.voff[i], (j & 1) ? off.b : off.a, 0, 0);
}

// Issue all LDS-DMA instructions of one k-step into `stage` (prologue use).
template <class T = Tile128>
__device__ __forceinline__ void dma_tiles(const TileSrcT<T::DMA_N> &a, const TileSrcT<T::DMA_N> &b, char *stage, KOff off,
                                          int wave) {
#pragma unroll
    for (int j = 0; j < 2 * T::DMA_N; j++) dma_piece<T>(a, b, stage, off, wave, j);
}

// acc[mt][nt] += A(MT*32 rows x 128 B) * X(NT*32 rows x 128 B)^T for this wave's sub-tile: 16 MT NT MFMAs
// for fp32 rows, 4 MT NT for 16-bit rows.  The 2 DMA_N LDS-DMA pieces of the next k-step (at `next_off`) are issued
// into `next_stage` one at a time between groups of MFMAs, so that each piece's issue cost hides under
// the execution of the MFMAs before it instead of stalling the wave in one burst.  They are issued
// unconditionally (no branch in the loop): after the last k-step of a block the caller passes a valid
// (offset, source) pair again and the fetched stage is simply never read.
// Fragment addressing: lane (r = lane & 31, h = lane >> 5) reads the 16 bytes of k-chunk 2 s + h of its
// row; for fp32 those 4 floats feed MFMA steps j = 0..3 of "octet" s (lane half h supplies k = 8 s + 4 h + j
// to both operands, so the products pair up); for 16-bit types the 8 elements are exactly the
// k = 16 s + 8 h + j fragment of one 32x32x16 MFMA.
template <int DT> __device__ __forceinline__ f32x16 mfma16(const f32x4 &a, const f32x4 &b, const f32x16 &c) {
    if (DT == HVS_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int DT, class T = Tile128>
__device__ __forceinline__ void compute_step(f32x16 (&acc)[T::MT][T::NT], const char *stage, int wm, int wn, int lane,
                                             const TileSrcT<T::DMA_N> &sa, const TileSrcT<T::DMA_N> &sb, char *next_stage,
                                             KOff next_off, int wave) {
    constexpr int MT = T::MT, NT = T::NT, PIECES = 2 * T::DMA_N;
    const int r = lane & 31;
    const int h = lane >> 5;
    const char *As = stage + (wm * (MT * 32) + r) * KB;  // rows t * 32 further down share this lane's swizzle
    const char *Bs = stage + T::BM * KB + (wn * (NT * 32) + r) * KB;
    const int swz = lds_swz(r);
    // 16 bytes of k-chunk kc of this lane's row in 32-row block t (32 rows = 4 KiB of LDS image)
    auto frag = [&](const char *base, int t, int kc) {
        return *reinterpret_cast<const f32x4 *>(base + t * (32 * KB) + ((kc ^ swz) << 4));
    };
    if (DT == HVS_F32) {
        // 16 groups (s, j) of MT NT MFMAs per k-step; the pieces go after every (16 / PIECES)-th group
        constexpr int EVERY = 16 / PIECES;  // (all pieces in the first half of the step: 177.8 - 180.4 ms against 176.3 on 4096 x 2M fp32 -- a step is long enough here)
        static_assert(EVERY >= 1 && 16 % PIECES == 0, "pieces must divide the MFMA groups");
#pragma unroll
        for (int s = 0; s < 4; s++) {
            f32x4 fa[MT], fb[NT];
#pragma unroll
            for (int t = 0; t < MT; t++) fa[t] = frag(As, t, 2 * s + h);
#pragma unroll
            for (int t = 0; t < NT; t++) fb[t] = frag(Bs, t, 2 * s + h);
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][j], fb[nt][j], acc[mt][nt], 0, 0, 0);
                if ((s * 4 + j) % EVERY == 0) {
                    dma_piece<T>(sa, sb, next_stage, next_off, wave, (s * 4 + j) / EVERY);
                    __builtin_amdgcn_sched_barrier(0);  // pin the piece between these MFMA groups
                }
            }
        }
    } else {
        // 16-bit rows: an MFMA retires 16 k-elements in 32 cycles, so fragment latency no longer hides by
        // itself.  The slice loop is software-pipelined over half-slices: while the MFMAs of the first MT/2
        // row blocks run, the fragments of the second half arrive; while those run, the first half of the
        // NEXT slice arrives (A fragments reuse the registers just consumed, X fragments alternate
        // between two sets).  The LDS-DMA pieces of the next step go out after the first half-slices.
        constexpr int HALF = MT / 2;
        static_assert(MT % 2 == 0 && PIECES == 8, "half-slice pipeline: even MT, one piece per half-slice");
        // LDS-DMA pieces of the next step per half-slice gap: 4, i.e. all 8 are out after the first two gaps.  One per gap
        // (the last one issued just ahead of the step barrier) left a good part of its latency exposed at every step:
        // 2M x 1536 f16 x 4096 queries 970 -> 1 005 TFLOP/s, K loop alone 1 085 -> 1 123; 2 per gap 1 002, all 8 in the first gap 999
        constexpr int DMA_PER_GAP = 4;
        f32x4 fa[MT], fb[2][NT];
#pragma unroll
        for (int t = 0; t < HALF; t++) fa[t] = frag(As, t, h);
#pragma unroll
        for (int t = 0; t < NT; t++) fb[0][t] = frag(Bs, t, h);
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int t = HALF; t < MT; t++) fa[t] = frag(As, t, 2 * s + h);
            __builtin_amdgcn_sched_barrier(0);  // reads first: they land while this half's MFMAs run
#pragma unroll
            for (int mt = 0; mt < HALF; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma16<DT>(fa[mt], fb[s & 1][nt], acc[mt][nt]);
            // LDS-DMA pieces of the next step: DMA_PER_GAP after each half-slice until all 8 are out
#pragma unroll
            for (int i = 0; i < DMA_PER_GAP; i++)
                if ((2 * s) * DMA_PER_GAP + i < PIECES) dma_piece<T>(sa, sb, next_stage, next_off, wave, (2 * s) * DMA_PER_GAP + i);
            __builtin_amdgcn_sched_barrier(0);
            if (s < 3) {
#pragma unroll
                for (int t = 0; t < HALF; t++) fa[t] = frag(As, t, 2 * (s + 1) + h);
#pragma unroll
                for (int t = 0; t < NT; t++) fb[(s + 1) & 1][t] = frag(Bs, t, 2 * (s + 1) + h);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = HALF; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma16<DT>(fa[mt], fb[s & 1][nt], acc[mt][nt]);
#pragma unroll
            for (int i = 0; i < DMA_PER_GAP; i++)
                if ((2 * s + 1) * DMA_PER_GAP + i < PIECES) dma_piece<T>(sa, sb, next_stage, next_off, wave, (2 * s + 1) * DMA_PER_GAP + i);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Blocks that read the same corpus stream through one XCD's L2 start every corpus tile together: left alone they drift
// apart by more tiles than the L2 holds, and every one of them then fetches the stream from the fabric for itself
// (fp32 search, 4096 queries x 2M rows: 528 GB per launch without this, 166 GB with it, and 2.8 % less time).
// One arrival counter per group in global memory (zeroed before the launch); thread 0 of a block adds one at a tile's
// start and waits until all `members` blocks have started that tile.  The wait is bounded -- a group whose blocks are
// not all resident (a grid larger than the chip, another kernel holding CUs) must not hang -- and a block that ever
// runs into the bound stops waiting for good (it still counts itself in, so that the others are not held up by it).
__device__ __forceinline__ void lockstep_arrive(unsigned int *counter, unsigned int tiles_started, int members, bool &enabled) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!enabled) return;
    const unsigned int want = tiles_started * (unsigned int)members;
    for (int spin = 0; spin < 4096; spin++) {
        if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return;
        __builtin_amdgcn_s_sleep(2);
    }
    enabled = false;
}

template <int MT, int NT> __device__ __forceinline__ void zero_acc(f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mt][nt][e] = 0.0f;
}
