// Shared tile machinery of the MFMA contraction (search and k-means kernels): 128 x 128 block tile,
// k-steps of 128 bytes per row (32 fp32 or 64 fp16/bf16 elements), double-buffered XOR-swizzled LDS
// image filled by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write, no per-step
// address arithmetic), v_mfma_f32_32x32x2_f32 for fp32 rows and v_mfma_f32_32x32x16_{f16,bf16} for
// 16-bit rows (fp32 accumulation in both).
#pragma once
#include "hvs_internal.h"

#define GT 256
#define BM 128
#define BN 128
#define KB 128                              // bytes of one tile row per k-step
#define STAGE_BYTES ((BM + BN) * KB)        // one stage: A image then X image
#define DMA_N (BM * KB / 16 / GT)           // LDS-DMA instructions per thread, operand and k-step (= 4)
#define CAND_MAX 384  // capacity of a candidate array (entries), also the per-wave ranking scratch
#define MFMA_MAX_K 128

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
// the swizzle is applied to the SOURCE address: thread t, instruction i fills LDS slot P = 256 i + t,
// i.e. tile row P >> 3, and must fetch that row's k-chunk (P & 7) ^ swz(row).
// Returns the byte offset of that chunk inside a tile whose rows are `pitch_bytes` apart; rows past
// `last_row` (ragged last tile) re-read the last valid row.
__device__ __forceinline__ int dma_src_offset(int tid, int i, int pitch_bytes, int last_row) {
    const int P = GT * i + tid;
    int row = P >> 3;
    const int kc = (P & 7) ^ lds_swz(row);
    row = row > last_row ? last_row : row;
    return row * pitch_bytes + kc * 16;
}

struct TileSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[DMA_N];
};

// Descriptor over rows [row0, row0 + rows_valid) of a row-major matrix; reads past the end return 0.
__device__ __forceinline__ void tile_src_init(TileSrc &t, const void *base, int64_t row0, int rows_valid, int pitch_bytes,
                                              int tid) {
    const char *p = reinterpret_cast<const char *>(base) + row0 * pitch_bytes;
    // base and size are wave-uniform by construction; say so, or every DMA is wrapped in a waterfall loop
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane(rows_valid * pitch_bytes);
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < DMA_N; i++) t.voff[i] = dma_src_offset(tid, i, pitch_bytes, rows_valid - 1);
}

// One LDS-DMA instruction of k-step `step`: j in [0, 2 DMA_N): even j -> A piece j/2, odd j -> X piece j/2.
__device__ __forceinline__ void dma_piece(const TileSrc &a, const TileSrc &b, char *stage, int step, int wave, int j) {
    const int i = j >> 1;
    char *d = stage + (GT * i + wave * 64) * 16 + ((j & 1) ? BM * KB : 0);
    const TileSrc &t = (j & 1) ? b : a;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(t.rsrc, (lds_void *)d, 16, t.voff[i], step * KB, 0, 0);
}

// Issue all LDS-DMA instructions of one k-step into `stage` (prologue use).
__device__ __forceinline__ void dma_tiles(const TileSrc &a, const TileSrc &b, char *stage, int step, int wave) {
#pragma unroll
    for (int j = 0; j < 2 * DMA_N; j++) dma_piece(a, b, stage, step, wave, j);
}

// acc[mt][nt] += A(64 rows x 128 B) * X(64 rows x 128 B)^T for this wave's sub-tile: 64 MFMAs for fp32
// rows, 16 for 16-bit rows.  When `prefetch` is set, the next k-step's 8 LDS-DMA pieces are issued one
// at a time between groups of MFMAs, so that each piece's issue cost hides under the execution of the
// MFMA before it instead of stalling the wave in one burst.
// Fragment addressing: lane (r = lane & 31, h = lane >> 5) reads the 16 bytes of k-chunk 2 s + h of its
// row; for fp32 those 4 floats feed MFMA steps j = 0..3 of "octet" s (lane half h supplies k = 8 s + 4 h + j
// to both operands, so the products pair up); for 16-bit types the 8 elements are exactly the
// k = 16 s + 8 h + j fragment of one 32x32x16 MFMA.
template <int DT>
__device__ __forceinline__ void compute_step(f32x16 (&acc)[2][2], const char *stage, int wm, int wn, int lane,
                                             bool prefetch, const TileSrc &sa, const TileSrc &sb, char *next_stage,
                                             int next_step, int wave) {
    const int r = lane & 31;
    const int h = lane >> 5;
    const char *As = stage;
    const char *Bs = stage + BM * KB;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        struct { f32x4 a[2], b[2]; } f;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f.a[t] = *reinterpret_cast<const f32x4 *>(As + lds_off(wm * 64 + t * 32 + r, 2 * s + h));
            f.b[t] = *reinterpret_cast<const f32x4 *>(Bs + lds_off(wn * 64 + t * 32 + r, 2 * s + h));
        }
        if (DT == HVS_F32) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[mt][j], f.b[nt][j], acc[mt][nt], 0, 0, 0);
                if ((j & 1) == 0) {  // 8 pieces over 16 groups of 4 MFMAs
                    if (prefetch) dma_piece(sa, sb, next_stage, next_step, wave, s * 2 + (j >> 1));
                    __builtin_amdgcn_sched_barrier(0);  // pin the piece between these MFMA groups
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    if (DT == HVS_F16)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.a[mt]),
                                                                             __builtin_bit_cast(f16x8, f.b[nt]), acc[mt][nt], 0, 0, 0);
                    else
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[mt]),
                                                                              __builtin_bit_cast(bf16x8, f.b[nt]), acc[mt][nt], 0, 0, 0);
                    if (nt == 1) {  // 8 pieces over 8 pairs of MFMAs
                        if (prefetch) dma_piece(sa, sb, next_stage, next_step, wave, s * 2 + mt);
                        __builtin_amdgcn_sched_barrier(0);
                    }
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
