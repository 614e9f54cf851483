// Shared tile machinery of the fp32 MFMA contraction (search and k-means kernels): 128 x 128 block
// tile, 32-float k-steps, double-buffered XOR-swizzled LDS image, v_mfma_f32_32x32x2_f32.
#pragma once
#include "hvs_internal.h"

#define GT 256
#define BM 128
#define BN 128
#define BK 32
#define STAGE_FLOATS ((BM + BN) * BK)
#define CAND_MAX 384  // capacity of a candidate array (entries), also the per-wave ranking scratch
#define MFMA_MAX_K 128

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TileRegs {
    f32x4 a[4];
    f32x4 b[4];
};

// float offset of 16-byte slot `slot` (0..7) of tile row `row` in a [128][32] fp32 LDS image
__device__ __forceinline__ int lds_off(int row, int slot) { return (row * 8 + (slot ^ ((row >> 1) & 7))) * 4; }

// Issue the global loads of one k-step: thread t moves chunks c = t + 256 i (row c >> 3, slot c & 7) of A and B.
__device__ __forceinline__ void load_tiles(TileRegs &r, const float *__restrict__ A, int64_t a_row0, int64_t a_last,
                                           const float *__restrict__ B, int64_t b_row0, int64_t b_last, int ld, int k0,
                                           int tid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = tid + GT * i;
        const int row = c >> 3;
        const int kk = k0 + (c & 7) * 4;
        int64_t ar = a_row0 + row;
        ar = ar > a_last ? a_last : ar;
        int64_t br = b_row0 + row;
        br = br > b_last ? b_last : br;
        if (kk < ld) {
            r.a[i] = *reinterpret_cast<const f32x4 *>(A + ar * ld + kk);
            r.b[i] = *reinterpret_cast<const f32x4 *>(B + br * ld + kk);
        } else {
            r.a[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            r.b[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
}

__device__ __forceinline__ void store_tiles(const TileRegs &r, float *stage, int tid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = tid + GT * i;
        const int off = lds_off(c >> 3, c & 7);
        *reinterpret_cast<f32x4 *>(stage + off) = r.a[i];
        *reinterpret_cast<f32x4 *>(stage + BM * BK + off) = r.b[i];
    }
}

// 64 MFMAs: acc[mt][nt] += A(64 x 32) * X(64 x 32)^T for this wave's sub-tile
__device__ __forceinline__ void compute_step(f32x16 (&acc)[2][2], const float *stage, int wm, int wn, int lane) {
    const int r = lane & 31;
    const int h = lane >> 5;
    const float *As = stage;
    const float *Bs = stage + BM * BK;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            a[t] = *reinterpret_cast<const f32x4 *>(As + lds_off(wm * 64 + t * 32 + r, 2 * o + h));
            b[t] = *reinterpret_cast<const f32x4 *>(Bs + lds_off(wn * 64 + t * 32 + r, 2 * o + h));
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
        if (o & 1) __builtin_amdgcn_sched_barrier(0);  // at most two octets of fragments live at once
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

