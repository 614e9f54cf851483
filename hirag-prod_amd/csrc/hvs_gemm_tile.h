// Shared tile machinery of the fp32 MFMA contraction (search and k-means kernels): 128 x 128 block
// tile, 32-float k-steps, double-buffered XOR-swizzled LDS image filled by LDS-DMA
// (buffer_load_dwordx4 ... lds: no staging registers, no ds_write, no per-step address arithmetic),
// v_mfma_f32_32x32x2_f32.
#pragma once
#include "hvs_internal.h"

#define GT 256
#define BM 128
#define BN 128
#define STAGE_FLOATS(BK) ((BM + BN) * (BK))
#define CAND_MAX 384  // capacity of a candidate array (entries), also the per-wave ranking scratch
#define MFMA_MAX_K 128

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// LDS image of one operand tile: [128 rows][BK / 4 slots of 16 B].  The slot holding k-chunk `kc` of
// `row` is kc ^ swz(row), chosen so that the ds_read_b128 fragment reads (32 rows x one chunk per lane
// half, 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}) touch 16 distinct 16-byte slots of
// the 256-byte bank row: 8 slots per row -> swz = (row >> 1) & 7; 4 slots per row -> swz = (row >> 2) & 3.
template <int BK> __device__ __forceinline__ int lds_swz(int row) { return BK == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
template <int BK> __device__ __forceinline__ int lds_off(int row, int kc) { return (row * (BK / 4) + (kc ^ lds_swz<BK>(row))) * 4; }

// An LDS-DMA wave-instruction writes 64 consecutive 16-byte slots (wave-uniform base + lane * 16), so
// the swizzle is applied to the SOURCE address: thread t, instruction i fills LDS slot P = 256 i + t,
// i.e. tile row P / (BK/4), and must fetch that row's k-chunk (P % (BK/4)) ^ swz(row).
// Returns the byte offset of that chunk inside a tile whose rows are `pitch_bytes` apart; rows past
// `last_row` (ragged last tile) re-read the last valid row.
template <int BK> __device__ __forceinline__ int dma_src_offset(int tid, int i, int pitch_bytes, int last_row) {
    constexpr int SL = BK / 4;
    const int P = GT * i + tid;
    int row = P / SL;
    const int kc = (P % SL) ^ lds_swz<BK>(row);
    row = row > last_row ? last_row : row;
    return row * pitch_bytes + kc * 16;
}

template <int BK> struct TileSrc {
    static constexpr int N = BM * BK / 4 / GT;  // DMA instructions per thread, operand and k-step
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[N];
};

// Descriptor over rows [row0, row0 + rows_valid) of a row-major matrix; reads past the end return 0.
template <int BK>
__device__ __forceinline__ void tile_src_init(TileSrc<BK> &t, const float *base, int64_t row0, int rows_valid,
                                              int pitch_floats, int tid) {
    const float *p = base + row0 * pitch_floats;
    // base and size are wave-uniform by construction; say so, or every DMA is wrapped in a waterfall loop
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane(rows_valid * pitch_floats * 4);
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < TileSrc<BK>::N; i++) t.voff[i] = dma_src_offset<BK>(tid, i, pitch_floats * 4, rows_valid - 1);
}

// One LDS-DMA instruction of a k-step: j in [0, 2N): even j -> A piece j/2, odd j -> X piece j/2.
template <int BK>
__device__ __forceinline__ void dma_piece(const TileSrc<BK> &a, const TileSrc<BK> &b, float *stage, int k0, int wave, int j) {
    const int i = j >> 1;
    float *d = stage + (GT * i + wave * 64) * 4 + ((j & 1) ? BM * BK : 0);
    const TileSrc<BK> &t = (j & 1) ? b : a;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(t.rsrc, (lds_void *)d, 16, t.voff[i], k0 * 4, 0, 0);
}

// Issue all LDS-DMA instructions of one k-step into `stage` (prologue use).
template <int BK>
__device__ __forceinline__ void dma_tiles(const TileSrc<BK> &a, const TileSrc<BK> &b, float *stage, int k0, int wave) {
#pragma unroll
    for (int j = 0; j < 2 * TileSrc<BK>::N; j++) dma_piece<BK>(a, b, stage, k0, wave, j);
}

// 2 BK MFMAs: acc[mt][nt] += A(64 x BK) * X(64 x BK)^T for this wave's sub-tile.  When `prefetch` is
// set, the next k-step's LDS-DMA pieces are issued one at a time between groups of MFMAs, so that
// each piece's issue cost hides under the 64-cycle execution of the MFMA before it instead of
// stalling the wave in one burst.
template <int BK>
__device__ __forceinline__ void compute_step(f32x16 (&acc)[2][2], const float *stage, int wm, int wn, int lane,
                                             bool prefetch, const TileSrc<BK> &sa, const TileSrc<BK> &sb,
                                             float *next_stage, int next_k0, int wave) {
    const int r = lane & 31;
    const int h = lane >> 5;
    const float *As = stage;
    const float *Bs = stage + BM * BK;
    constexpr int NP = 2 * TileSrc<BK>::N;          // DMA pieces per k-step
    constexpr int GROUPS = (BK / 8) * 4;            // groups of 4 MFMAs per k-step
    constexpr int EVERY = GROUPS / NP;              // one piece every EVERY groups
#pragma unroll
    for (int o = 0; o < BK / 8; o++) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            a[t] = *reinterpret_cast<const f32x4 *>(As + lds_off<BK>(wm * 64 + t * 32 + r, 2 * o + h));
            b[t] = *reinterpret_cast<const f32x4 *>(Bs + lds_off<BK>(wn * 64 + t * 32 + r, 2 * o + h));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
            const int grp = o * 4 + j;
            if (grp % EVERY == 0 && grp / EVERY < NP) {
                if (prefetch) dma_piece<BK>(sa, sb, next_stage, next_k0, wave, grp / EVERY);
                __builtin_amdgcn_sched_barrier(0);  // pin the piece between these MFMA groups
            }
        }
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mt][nt][e] = 0.0f;
}
