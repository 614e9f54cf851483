// fp32 rows as pairs of fp16 slabs, so that the batched search of an fp32 table can run on the 16-bit
// MFMA pipe (16x the rate of the fp32 MFMA) without giving up fp32-grade scores.
//
// A row a is scaled by a power of two s so that max|a s| lies in [2^13, 2^14) and stored as
//     hi = fp16(a s),   lo = fp16(a s - hi)            (the subtraction is exact in fp32)
// in one row of 2 * ld16 halves: [hi slab | lo slab], ld16 = dim rounded up to 64 (one 128-byte k-step).
// hi + lo carries 22 significant bits of every element that is not more than 2^16 times smaller than
// the row's largest.  The search kernel contracts three slab pairs into one fp32 accumulator,
//     hi.hi + hi.lo + lo.hi   (lo.lo <= 2^-22 of each product is dropped),
// every fp16 x fp16 product being exact in fp32.  The scale leaves through the inverse norm:
// inv_split = inv_norm / s, so cos = acc * inv_split_row * inv_split_query.
// Error of a similarity against exact arithmetic: about 2^-22 |a_i b_i| per term with random signs, i.e.
// ~1e-8 for dense embeddings and < 1e-6 in the worst case -- below the fp32 accumulation error of the
// scalar formula the reference's database evaluates.
#include "hvs_internal.h"

__device__ __forceinline__ float split_scale(float m, int *exp_out) {
    int ex = 0;
    if (m > 0.0f && m < __builtin_inff()) {
        (void)frexpf(m, &ex);  // m = f 2^ex, f in [0.5, 1)
        ex = 14 - ex;
        ex = ex > 126 ? 126 : (ex < -126 ? -126 : ex);
    }
    *exp_out = ex;
    return ldexpf(1.0f, ex);
}

// One wave per row.  src: fp32 rows of pitch ld_src floats (columns >= cols are ignored); dst: [n, 2 ld16] halves.
// dst_rel (may be null): [n] |a s - hi| / |a s| of every row, the relative distance between a row and its hi slab
// alone, rounded up; dst_rel_max (may be null): raised to the largest of them (float bits; the values are not
// negative, so their order is the order of their bits).  A product over hi slabs alone errs by at most
// rel_x + rel_q (1 + rel_x) of |q| |x| (Cauchy-Schwarz on q.(x - xh) + (q - qh).xh), which for rows that are not
// built to sit at rounding midpoints is less than half of the 2^-10 the rounding unit alone guarantees.
__global__ void __launch_bounds__(256) split_rows_kernel(const float *__restrict__ src, const float *__restrict__ src_inv, int64_t first,
                                                         int64_t n, int cols, int ld_src, int ld16, __half *__restrict__ dst,
                                                         float *__restrict__ dst_inv, float *__restrict__ dst_rel,
                                                         uint32_t *__restrict__ dst_rel_max) {
    const int lane = threadIdx.x & 63;
    const int64_t r = first + blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= first + n) return;
    const float *a = src + (size_t)r * ld_src;
    float m = 0.0f;
    for (int c = lane * 4; c < cols; c += 256) {  // cols and ld_src are multiples of 4
        const float4 v = *reinterpret_cast<const float4 *>(a + c);
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));  // fmaxf drops NaN: the products carry it
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    int ex;
    const float s = split_scale(m, &ex);
    __half *hi = dst + (size_t)r * 2 * ld16, *lo = hi + ld16;
    float se = 0.0f, sx = 0.0f;  // sums of (a s - hi)^2 and (a s)^2
    for (int c = lane * 4; c < ld16; c += 256) {
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (c < cols) v = *reinterpret_cast<const float4 *>(a + c);
        const float e[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
        __half h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            h[i] = __float2half_rn(e[i]);
            const float d = e[i] - __half2float(h[i]);
            l[i] = __float2half_rn(d);
            se = fmaf(d, d, se);
            sx = fmaf(e[i], e[i], sx);
        }
        *reinterpret_cast<uint2 *>(hi + c) = *reinterpret_cast<const uint2 *>(h);
        *reinterpret_cast<uint2 *>(lo + c) = *reinterpret_cast<const uint2 *>(l);
    }
    if (dst_rel || dst_rel_max) {
        for (int off = 32; off > 0; off >>= 1) {
            se += __shfl_xor(se, off);
            sx += __shfl_xor(sx, off);
        }
        // both sums carry a relative rounding error below cols 2^-24 each; 1.001 covers it up to a million columns'
        // worth of it and the rounding of the root.  A zero row, or one that holds a NaN or an infinity, has no score.
        float rel = 0.0f;
        if (sx > 0.0f && sx < __builtin_inff() && se == se) rel = sqrtf(se / sx) * 1.001f;
        if (lane == 0) {
            if (dst_rel) dst_rel[r - first] = rel;
            if (dst_rel_max) atomicMax(dst_rel_max, __float_as_uint(rel));
        }
    }
    if (lane == 0) dst_inv[r] = ldexpf(src_inv[r], -ex);
}

int hvs_split_rows(hipStream_t st, const float *src, const float *src_inv, int64_t first, int64_t n, int cols, int ld_src, int ld16,
                   void *dst, float *dst_inv, float *dst_rel, uint32_t *dst_rel_max) {
    if (n <= 0) return 0;
    const int64_t blocks = (n + 3) / 4;
    HVS_REQUIRE(blocks <= 0x7FFFFFFFll, "hvs_split_rows: too many rows for one launch");
    split_rows_kernel<<<(int)blocks, 256, 0, st>>>(src, src_inv, first, n, cols, ld_src, ld16, (__half *)dst, dst_inv, dst_rel, dst_rel_max);
    HVS_HIP(hipGetLastError());
    return 0;
}

// Bring the split copy of an fp32 index up to date (rows [split_n, n)).  *ok = false when device memory does
// not allow the copy (the caller then uses the fp32 MFMA kernel).
int hvs_split_ensure(hvs_index *idx, bool *ok) {
    *ok = false;
    if (idx->dtype != HVS_F32 || idx->split_failed) return 0;
    if (!idx->split_allowed) {
        idx->split_why = "the search-side copy is switched off for this table (hvs_index_search_copy); searches stay on the fp32 kernels";
        return 0;
    }
    const int ld16 = (idx->dim + 63) / 64 * 64;
    const size_t pitch = (size_t)ld16 * 4;  // two slabs of halves
    if (idx->split_cap < idx->n) {
        const int64_t cap = idx->cap > idx->n ? idx->cap : idx->n;
        const size_t need = (size_t)cap * pitch + HVS_ROWS_SLACK + (size_t)cap * 4;
        size_t free_b = 0, total_b = 0;
        HVS_HIP(hipMemGetInfo(&free_b, &total_b));
        // leave room for the searches' own buffers: candidate lists and staging are a few hundred MB even for
        // 4096-query batches; 1/16 of the copy, at least 1 GiB
        const size_t headroom = need / 16 > ((size_t)1 << 30) ? need / 16 : ((size_t)1 << 30);
        if (free_b < need + headroom) {
            idx->split_failed = true;
            char why[256];
            snprintf(why, sizeof(why), "the split copy needs %.2f GB plus %.2f GB of headroom, the device has %.2f GB free; searches stay on the fp32 kernels",
                     need / 1e9, headroom / 1e9, free_b / 1e9);
            idx->split_why = why;
            return 0;
        }
        char *rows = nullptr;
        float *inv = nullptr;
        if (hipMalloc((void **)&rows, (size_t)cap * pitch + HVS_ROWS_SLACK) != hipSuccess ||
            hipMalloc((void **)&inv, (size_t)cap * 4) != hipSuccess) {
            (void)hipGetLastError();
            if (rows) (void)hipFree(rows);
            idx->split_failed = true;
            idx->split_why = "hipMalloc of the split copy failed; searches stay on the fp32 kernels";
            return 0;
        }
        const int64_t keep = idx->split ? idx->split_n : 0;
        if (keep > 0) {
            HVS_HIP(hipMemcpyAsync(rows, idx->split, (size_t)keep * pitch, hipMemcpyDeviceToDevice, idx->stream));
            HVS_HIP(hipMemcpyAsync(inv, idx->split_inv, (size_t)keep * 4, hipMemcpyDeviceToDevice, idx->stream));
        }
        HVS_HIP(hipMemsetAsync(rows + (size_t)keep * pitch, 0, (size_t)(cap - keep) * pitch + HVS_ROWS_SLACK, idx->stream));
        HVS_HIP(hipStreamSynchronize(idx->stream));
        if (idx->split) (void)hipFree(idx->split);
        if (idx->split_inv) (void)hipFree(idx->split_inv);
        idx->split = rows;
        idx->split_inv = inv;
        idx->split_cap = cap;
        idx->split_n = keep;
    }
    if (!idx->split_rel) {  // largest relative distance between a row and its hi slab; only ever raised
        HVS_HIP(hipMalloc((void **)&idx->split_rel, 256));
        HVS_HIP(hipMemsetAsync(idx->split_rel, 0, 256, idx->stream));
    }
    if (idx->split_n == 0) HVS_HIP(hipMemsetAsync(idx->split_rel, 0, 4, idx->stream));  // every row is examined again
    if (idx->split_n < idx->n) {
        HVS_TRY(hvs_split_rows(idx->stream, (const float *)idx->rows, idx->inv_norm, idx->split_n, idx->n - idx->split_n, idx->ld, idx->ld,
                               ld16, idx->split, idx->split_inv, nullptr, idx->split_rel));
        idx->split_n = idx->n;
    }
    *ok = true;
    return 0;
}
