/*
 * CPU restatement, in plain C, of the scan PostgreSQL + pgvector run for
 * PGVector.query (reference: src/hirag_prod/storage/pgvector.py:561-585).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ as a second checker next to the
 * numpy oracle and by bench.py's cpu_baseline leg.  The product never links
 * or loads it.  Parity unpinned: the pgvector extension itself is not in the
 * reference tree (server-side dependency, version not pinned); this follows
 * its published algorithm:
 *
 *   cosine_distance (vector.c / halfvec.c): one pass with three float
 *   accumulators (dot, |a|^2, |b|^2); similarity = dot / sqrt(|a|^2 * |b|^2)
 *   evaluated in double; clamped to [-1, 1]; distance = 1 - similarity.
 *   pgvector builds with -ftree-vectorize -fassociative-math
 *   -fno-signed-zeros -fno-trapping-math, i.e. the accumulation order is the
 *   compiler's; the Makefile next to this file uses the same flags.
 *
 *   The executor's "ORDER BY distance LIMIT k" over a sequential scan is a
 *   bounded heap (tuplesort's top-N heapsort); rows failing the WHERE clause
 *   (filters, distance < threshold) never reach it.  NaN distances fail the
 *   "<" comparison.  Ties: lower row index first (PostgreSQL leaves it open).
 *
 *   LEAST over m query embeddings (pgvector.py:565-569): minimum, with NaN
 *   ordered above every number.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float half_to_float(uint16_t h)
{
    /* IEEE binary16 -> binary32, every case (subnormals, inf, NaN). */
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline float bf16_to_float(uint16_t h)
{
    uint32_t bits = (uint32_t)h << 16;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

double pgv_cosine_distance_f32(int dim, const float *a, const float *b)
{
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (int i = 0; i < dim; i++) {
        dot += a[i] * b[i];
        na += a[i] * a[i];
        nb += b[i] * b[i];
    }
    double sim = (double)dot / sqrt((double)na * (double)nb);
    if (sim > 1.0) sim = 1.0;
    else if (sim < -1.0) sim = -1.0;
    return 1.0 - sim;
}

/* rows stored as 16-bit (kind 1 = fp16 as halfvec, kind 2 = bf16): widened per element, float accumulators */
static double cosine_distance_16(int dim, const uint16_t *a, const float *b, int kind)
{
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (int i = 0; i < dim; i++) {
        float x = kind == 1 ? half_to_float(a[i]) : bf16_to_float(a[i]);
        dot += x * b[i];
        na += x * x;
        nb += b[i] * b[i];
    }
    double sim = (double)dot / sqrt((double)na * (double)nb);
    if (sim > 1.0) sim = 1.0;
    else if (sim < -1.0) sim = -1.0;
    return 1.0 - sim;
}

typedef struct { double d; int64_t i; } cand_t;

/* "a sorts after b" under (distance asc, row index asc) */
static inline int after(cand_t a, cand_t b) { return a.d > b.d || (a.d == b.d && a.i > b.i); }

static void heap_push(cand_t *h, int *n, int k, cand_t c)
{
    if (*n < k) {
        int i = (*n)++;
        h[i] = c;
        while (i > 0) {
            int p = (i - 1) / 2;
            if (!after(h[i], h[p])) break;
            cand_t t = h[i]; h[i] = h[p]; h[p] = t;
            i = p;
        }
        return;
    }
    if (!after(h[0], c)) return; /* c is not better than the current worst */
    h[0] = c;
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < k && after(h[l], h[m])) m = l;
        if (r < k && after(h[r], h[m])) m = r;
        if (m == i) break;
        cand_t t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

static int cmp_cand(const void *x, const void *y)
{
    cand_t a = *(const cand_t *)x, b = *(const cand_t *)y;
    return after(a, b) ? 1 : (after(b, a) ? -1 : 0);
}

/*
 * One PGVector.query over rows [row_begin, row_end).
 *   rows        row-major [n, dim]; elem_kind 0 = float, 1 = fp16 bits, 2 = bf16 bits
 *   queries     [m, dim] float (already rounded to the column's type by the caller)
 *   group_codes optional int32 per row, group_filter < 0 disables (workspaceId/knowledgeBaseId)
 *   keep        optional uint8 per row, 0 = filtered out (uri IN / private ==)
 *   threshold   strict "<"; pass +inf for none
 * Returns the number of results written to out_idx/out_dist (sorted ascending).
 */
int pgv_scan_topk(const void *rows, int elem_kind, int64_t row_begin, int64_t row_end, int dim,
                  const float *queries, int m,
                  const int32_t *group_codes, int32_t group_filter, const uint8_t *keep,
                  double threshold, int k, int64_t *out_idx, double *out_dist)
{
    if (k <= 0) return 0;
    cand_t *heap = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
    int n = 0;
    for (int64_t r = row_begin; r < row_end; r++) {
        if (group_filter >= 0 && group_codes && group_codes[r] != group_filter) continue;
        if (keep && !keep[r]) continue;
        double best = NAN;
        for (int j = 0; j < m; j++) {
            double d = elem_kind == 0
                ? pgv_cosine_distance_f32(dim, (const float *)rows + (size_t)r * dim, queries + (size_t)j * dim)
                : cosine_distance_16(dim, (const uint16_t *)rows + (size_t)r * dim, queries + (size_t)j * dim, elem_kind);
            if (isnan(best) || d < best) best = d; /* LEAST: NaN only if all NaN */
        }
        if (!(best < threshold)) continue; /* also drops NaN */
        cand_t c = { best, r };
        heap_push(heap, &n, k, c);
    }
    qsort(heap, (size_t)n, sizeof(cand_t), cmp_cand);
    for (int i = 0; i < n; i++) { out_idx[i] = heap[i].i; out_dist[i] = heap[i].d; }
    free(heap);
    return n;
}

/*
 * k-means assignment restatement (no counterpart in the reference; the build
 * defines it): assign[i] = argmin_j |x_i - c_j|^2, lowest j on ties, computed
 * as |c_j|^2 - 2 x_i.c_j in double.
 */
void pgv_kmeans_assign(const float *rows, int64_t n, int dim, const float *centroids, int kc, int32_t *assign)
{
    double *cn = (double *)malloc(sizeof(double) * (size_t)kc);
    for (int j = 0; j < kc; j++) {
        double s = 0.0;
        for (int t = 0; t < dim; t++) s += (double)centroids[(size_t)j * dim + t] * centroids[(size_t)j * dim + t];
        cn[j] = s;
    }
    for (int64_t i = 0; i < n; i++) {
        const float *x = rows + (size_t)i * dim;
        double best = INFINITY;
        int bj = 0;
        for (int j = 0; j < kc; j++) {
            const float *c = centroids + (size_t)j * dim;
            double dot = 0.0;
            for (int t = 0; t < dim; t++) dot += (double)x[t] * c[t];
            double v = cn[j] - 2.0 * dot;
            if (v < best) { best = v; bj = j; }
        }
        assign[i] = bj;
    }
    free(cn);
}
