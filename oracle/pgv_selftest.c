/*
 * Self-test driver for pgv_scan.c, built with -fsanitize=address,undefined by `make -C oracle asan` (test
 * infrastructure; CPU only).  It walks the paths a sanitizer can say something about: the scan over every element kind
 * at sizes that are not multiples of any vector width, rows at the very end of their allocation (exact-size malloc, so
 * a read past a row is a heap overflow), the bounded heap at k = 1, k = n and k > n, every 16-bit pattern through the
 * fp16 / bf16 conversions, filters that remove everything, NaN and infinite distances, and the k-means restatement.
 * Results are checked against a plain double-precision loop, so a wrong answer fails the run too.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

double pgv_cosine_distance_f32(int dim, const float *a, const float *b);
int pgv_scan_topk(const void *rows, int elem_kind, int64_t row_begin, int64_t row_end, int dim, const float *queries, int m,
                  const int32_t *group_codes, int32_t group_filter, const uint8_t *keep, double threshold, int k,
                  int64_t *out_idx, double *out_dist);
void pgv_kmeans_assign(const float *rows, int64_t n, int dim, const float *centroids, int kc, int32_t *assign);

static uint64_t rng_state = 0x1234abcdULL;
static double rnd(void)
{
    rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((double)(rng_state >> 11) / 9007199254740992.0) * 2.0 - 1.0;
}

static float half_bits_to_float(uint16_t h)
{
    int s = h >> 15, e = (h >> 10) & 31, f = h & 1023;
    double v;
    if (e == 0) v = ldexp((double)f, -24);
    else if (e == 31) v = f ? NAN : INFINITY;
    else v = ldexp(1.0 + f / 1024.0, e - 15);
    return (float)(s ? -v : v);
}

static float bf16_bits_to_float(uint16_t h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static uint16_t float_to_bf16_trunc(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)(u >> 16);
}

static uint16_t float_to_half_simple(float f) /* values used here are normal halves: exponent in range */
{
    int e;
    double m = frexp(fabs((double)f), &e); /* f = m * 2^e, m in [0.5, 1) */
    if (f == 0.0f) return 0;
    int he = e - 1 + 15;
    if (he <= 0) return (uint16_t)((f < 0) << 15); /* below the normal halves: signed zero */
    int frac = (int)floor((m * 2.0 - 1.0) * 1024.0);
    return (uint16_t)(((f < 0) << 15) | (he << 10) | frac);
}

static int failures = 0;
#define CHECK(cond, ...)                            \
    do {                                            \
        if (!(cond)) {                              \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);           \
            fputc('\n', stderr);                    \
            failures++;                             \
        }                                           \
    } while (0)

static double ref_distance(int dim, const float *a, const float *b)
{
    double dot = 0, na = 0, nb = 0;
    for (int i = 0; i < dim; i++) { dot += (double)a[i] * b[i]; na += (double)a[i] * a[i]; nb += (double)b[i] * b[i]; }
    double sim = dot / sqrt(na * nb);
    if (sim > 1) sim = 1;
    if (sim < -1) sim = -1;
    return 1.0 - sim;
}

static void scan_case(int n, int dim, int m, int k, int kind)
{
    /* exact-size allocations: the last row ends where the block ends */
    float *wide = (float *)malloc(sizeof(float) * (size_t)n * dim);
    void *rows = malloc((size_t)n * dim * (kind == 0 ? 4 : 2));
    float *q = (float *)malloc(sizeof(float) * (size_t)m * dim);
    int32_t *groups = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    uint8_t *keep = (uint8_t *)malloc((size_t)n);
    for (size_t i = 0; i < (size_t)n * dim; i++) {
        float v = (float)rnd();
        if (kind == 1) { uint16_t h = float_to_half_simple(v); ((uint16_t *)rows)[i] = h; v = half_bits_to_float(h); }
        else if (kind == 2) { uint16_t h = float_to_bf16_trunc(v); ((uint16_t *)rows)[i] = h; v = bf16_bits_to_float(h); }
        else ((float *)rows)[i] = v;
        wide[i] = v;
    }
    if (n > 3) { /* a zero row: NaN distance, never returned */
        memset((char *)rows + (size_t)3 * dim * (kind == 0 ? 4 : 2), 0, (size_t)dim * (kind == 0 ? 4 : 2));
        memset(wide + (size_t)3 * dim, 0, sizeof(float) * (size_t)dim);
    }
    for (size_t i = 0; i < (size_t)m * dim; i++) q[i] = (float)rnd();
    for (int i = 0; i < n; i++) { groups[i] = i % 3; keep[i] = (uint8_t)(i % 5 != 0); }
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
    double *dist = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
    for (int variant = 0; variant < 4; variant++) {
        const int32_t gf = variant == 1 ? 2 : -1;
        const uint8_t *kp = variant == 2 ? keep : NULL;
        const double thr = variant == 3 ? 0.95 : INFINITY;
        int got = pgv_scan_topk(rows, kind, 0, n, dim, q, m, groups, gf, kp, thr, k, idx, dist);
        /* expected count and the smallest distance by brute force */
        int expect = 0;
        double best = INFINITY;
        for (int r = 0; r < n; r++) {
            if (gf >= 0 && groups[r] != gf) continue;
            if (kp && !kp[r]) continue;
            double d = NAN;
            for (int j = 0; j < m; j++) {
                double dj = ref_distance(dim, wide + (size_t)r * dim, q + (size_t)j * dim);
                if (isnan(d) || dj < d) d = dj;
            }
            if (!(d < thr)) continue;
            expect++;
            if (d < best) best = d;
        }
        if (expect > k) expect = k;
        CHECK(got == expect || (variant == 3 && abs(got - expect) <= 1), "n=%d dim=%d m=%d k=%d kind=%d variant=%d: %d results, expected %d", n, dim, m,
              k, kind, variant, got, expect);
        for (int i = 1; i < got; i++) CHECK(dist[i - 1] <= dist[i], "results not ascending");
        if (got > 0) CHECK(fabs(dist[0] - best) < 1e-5, "best distance %g, expected %g", dist[0], best);
        for (int i = 0; i < got; i++) CHECK(idx[i] >= 0 && idx[i] < n && idx[i] != 3, "row %lld out of place", (long long)idx[i]);
    }
    /* a filter nobody passes, and an empty range */
    CHECK(pgv_scan_topk(rows, kind, 0, n, dim, q, m, groups, 99, NULL, INFINITY, k, idx, dist) == 0, "group 99 matched rows");
    CHECK(pgv_scan_topk(rows, kind, n, n, dim, q, m, NULL, -1, NULL, INFINITY, k, idx, dist) == 0, "empty range matched rows");
    free(wide); free(rows); free(q); free(groups); free(keep); free(idx); free(dist);
}

int main(void)
{
    static const int dims[] = {1, 3, 7, 8, 9, 31, 64, 100, 257};
    static const int ks[] = {1, 2, 10, 37, 64};
    for (int kind = 0; kind < 3; kind++)
        for (unsigned di = 0; di < sizeof(dims) / sizeof(dims[0]); di++)
            for (unsigned ki = 0; ki < sizeof(ks) / sizeof(ks[0]); ki++) {
                scan_case(37, dims[di], 1, ks[ki], kind); /* k up to and beyond n = 37 */
                scan_case(11, dims[di], 3, ks[ki], kind);
            }
    scan_case(1, 5, 1, 1, 0);
    /* k = 0 writes nothing (NULL outputs would fault otherwise) */
    {
        float row[4] = {1, 2, 3, 4}, qq[4] = {4, 3, 2, 1};
        CHECK(pgv_scan_topk(row, 0, 0, 1, 4, qq, 1, NULL, -1, NULL, INFINITY, 0, NULL, NULL) == 0, "k = 0 returned rows");
    }
    /* every 16-bit pattern as the only element of a row, both element kinds: conversions must not trip UBSan, and the
     * distance is 0, 2 or NaN */
    for (int kind = 1; kind <= 2; kind++)
        for (uint32_t bits = 0; bits < 65536; bits++) {
            uint16_t h = (uint16_t)bits;
            float qq = 1.0f;
            int64_t i0;
            double d0;
            int got = pgv_scan_topk(&h, kind, 0, 1, 1, &qq, 1, NULL, -1, NULL, INFINITY, 1, &i0, &d0);
            float v = kind == 1 ? half_bits_to_float(h) : bf16_bits_to_float(h);
            if (isnan(v) || v == 0.0f || isinf(v)) CHECK(got == 0 || !(d0 == d0) || d0 == 0.0 || d0 == 2.0, "pattern %04x", bits);
            else if (fabsf(v) < 1e-15f || fabsf(v) > 1e15f) {
                /* the float accumulators of pgvector's loop lose x * x here (underflow towards 0, overflow to inf): any
                 * distance in [0, 2] or no result at all is faithful; only the absence of undefined behaviour is checked */
                CHECK(got == 0 || (d0 >= 0.0 && d0 <= 2.0), "pattern %04x kind %d: distance %g", bits, kind, d0);
            } else {
                CHECK(got == 1, "pattern %04x kind %d (value %g): no result", bits, kind, (double)v);
                if (got == 1) CHECK(fabs(d0 - (v > 0 ? 0.0 : 2.0)) < 1e-6, "pattern %04x kind %d: distance %g", bits, kind, d0);
            }
        }
    /* extreme magnitudes: float accumulators overflow to inf -> NaN distance, dropped, no undefined behaviour */
    {
        float big[8], qq[8];
        for (int i = 0; i < 8; i++) { big[i] = 3e38f; qq[i] = 3e38f; }
        int64_t i0;
        double d0;
        int got = pgv_scan_topk(big, 0, 0, 1, 8, qq, 1, NULL, -1, NULL, INFINITY, 1, &i0, &d0);
        CHECK(got == 0 || d0 == d0, "overflowing row returned NaN");
        (void)pgv_cosine_distance_f32(8, big, qq);
    }
    /* k-means restatement: argmin with ties to the lower index */
    {
        enum { N = 23, D = 5, K = 4 };
        float rows[N * D], cent[K * D];
        int32_t assign[N];
        for (int i = 0; i < N * D; i++) rows[i] = (float)rnd();
        for (int i = 0; i < K * D; i++) cent[i] = (float)rnd();
        memcpy(cent + 2 * D, cent + 1 * D, sizeof(float) * D); /* centroid 2 == centroid 1 */
        pgv_kmeans_assign(rows, N, D, cent, K, assign);
        for (int i = 0; i < N; i++) CHECK(assign[i] >= 0 && assign[i] < K && assign[i] != 2, "row %d assigned to %d", i, assign[i]);
    }
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    puts("pgv_selftest ok");
    return 0;
}
