// Small dense cosine-similarity matrix: out[i, j] = a_i . b_j / (|a_i| |b_j|), 0 where a norm is 0.
//
// This is scikit-learn's cosine_similarity(A, B) (rows L2-normalised with zero rows left at zero,
// then A B^T), the routine behind the reference's small similarity sites: calculate_similarity
// (hirag.py:571-588), validate_similarity and the synonym filter (cross_language_search/
// functions.py:85-113, 152-165) and the cosine block of dpr_recall_chunks (hirag.py:1305-1314).
// Sizes there are a few hundred rows at most, so the kernel is latency-shaped, not a tiled GEMM:
// one block per row of A (held in LDS), one wave per row of B, fp32 accumulation.
#include "hvs_internal.h"

#define COS_THREADS 256
#define COS_WAVES (COS_THREADS / 64)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(COS_THREADS) void cosine_matrix_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                    int n, int dim, float *__restrict__ out) {
    extern __shared__ float arow[];  // [dim] + COS_WAVES partial sums
    float *part = arow + dim;
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float aa = 0.0f;
    for (int c = threadIdx.x; c < dim; c += COS_THREADS) {
        const float v = a[(int64_t)i * dim + c];
        arow[c] = v;
        aa = fmaf(v, v, aa);
    }
    aa = wave_sum(aa);
    if (lane == 0) part[wave] = aa;
    __syncthreads();
    aa = part[0] + part[1] + part[2] + part[3];
    const float na = sqrtf(aa);
    for (int j = wave; j < n; j += COS_WAVES) {
        const float *brow = b + (int64_t)j * dim;
        float ab = 0.0f, bb = 0.0f;
        for (int c = lane; c < dim; c += 64) {
            const float v = brow[c];
            ab = fmaf(arow[c], v, ab);
            bb = fmaf(v, v, bb);
        }
        ab = wave_sum(ab);
        bb = wave_sum(bb);
        if (lane == 0) {
            const float nb = sqrtf(bb);
            out[(int64_t)i * n + j] = (na > 0.0f && nb > 0.0f) ? (ab / na) / nb : 0.0f;
        }
    }
}

extern "C" int hvs_cosine_matrix(int device_id, void *stream, const float *a, int m, const float *b, int n, int dim,
                                 float *out, int on_device) {
    HVS_REQUIRE(m >= 0 && n >= 0, "hvs_cosine_matrix: negative row count");
    if (m == 0 || n == 0) return 0;
    HVS_REQUIRE(dim >= 1 && dim <= 16384, "hvs_cosine_matrix: dim %d not in [1, 16384]", dim);
    HVS_REQUIRE((int64_t)m * n <= ((int64_t)1 << 28), "hvs_cosine_matrix: %d x %d outputs exceed 2^28", m, n);
    HVS_REQUIRE(a && b && out, "hvs_cosine_matrix: NULL argument");
    HVS_HIP(hipSetDevice(device_id));
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = ((size_t)dim + COS_WAVES) * sizeof(float);
    if (on_device) {
        cosine_matrix_kernel<<<m, COS_THREADS, lds, s>>>(a, b, n, dim, out);
        HVS_HIP(hipGetLastError());
        return 0;
    }
    // host buffers: stage through one temporary device allocation and wait for the result
    const size_t ab = (size_t)m * dim * 4, bb = (size_t)n * dim * 4, ob = (size_t)m * n * 4;
    char *tmp = nullptr;
    HVS_HIP(hipMalloc((void **)&tmp, ab + bb + ob));
    int rc = 0;
    do {
        hipError_t e;
        if ((e = hipMemcpyAsync(tmp, a, ab, hipMemcpyHostToDevice, s)) != hipSuccess ||
            (e = hipMemcpyAsync(tmp + ab, b, bb, hipMemcpyHostToDevice, s)) != hipSuccess) {
            hvs_set_error("hvs_cosine_matrix: upload failed: %s", hipGetErrorString(e));
            rc = 1;
            break;
        }
        cosine_matrix_kernel<<<m, COS_THREADS, lds, s>>>((const float *)tmp, (const float *)(tmp + ab), n, dim,
                                                         (float *)(tmp + ab + bb));
        if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpyAsync(out, tmp + ab + bb, ob, hipMemcpyDeviceToHost, s)) != hipSuccess ||
            (e = hipStreamSynchronize(s)) != hipSuccess) {
            hvs_set_error("hvs_cosine_matrix: %s", hipGetErrorString(e));
            rc = 1;
        }
    } while (0);
    (void)hipFree(tmp);
    return rc;
}
