/*
 * The C ABI of libhvs.so used from plain C99, nothing else linked: what a maintainer of the reference
 * would see from cgo / JNI / ctypes.  Builds a small table, searches it (scan and matrix paths), selects
 * by distance bound, runs a personalised PageRank on a tiny graph, and checks every answer against
 * values computed here on the host in double precision.  Exit status 0 = all checks passed.
 *
 *   gcc -std=c99 -O2 -I../../include abi_example.c -L. -lhvs -lm -o abi_example   (see the Makefile)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hvs.h"

#define N 3000
#define D 96
#define K 5

#define CK(call)                                                              \
    do {                                                                      \
        if ((call) != 0) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #call, hvs_last_error());      \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static unsigned long long lcg_state = 88172645463325252ull;
static float lcg_uniform(void) { /* in (-1, 1) */
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((double)(lcg_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

static double cos_dist(const float *a, const float *b) {
    double dot = 0, na = 0, nb = 0;
    for (int i = 0; i < D; i++) {
        dot += (double)a[i] * b[i];
        na += (double)a[i] * a[i];
        nb += (double)b[i] * b[i];
    }
    return 1.0 - dot / sqrt(na * nb);
}

int main(void) {
    int n_dev = 0;
    CK(hvs_device_count(&n_dev));
    if (n_dev < 1) {
        fprintf(stderr, "no GPU visible\n");
        return 1;
    }
    static float rows[N][D];
    static int32_t groups[N];
    for (int r = 0; r < N; r++) {
        for (int c = 0; c < D; c++) rows[r][c] = lcg_uniform();
        groups[r] = r % 4;
    }
    float queries[40][D];
    for (int q = 0; q < 40; q++)
        for (int c = 0; c < D; c++) queries[q][c] = q == 0 ? rows[17][c] : lcg_uniform();

    hvs_index *idx = NULL;
    CK(hvs_index_create(&idx, 0, D, HVS_F32, NULL));
    CK(hvs_index_add(idx, rows, HVS_F32, N, NULL, groups, 0));
    int64_t n = 0;
    CK(hvs_index_size(idx, &n));
    if (n != N) return 2;

    /* brute force on the host: the K nearest of every query */
    static double ref[40][N];
    for (int q = 0; q < 40; q++)
        for (int r = 0; r < N; r++) ref[q][r] = cos_dist(rows[r], queries[q]);

    /* the fp32 table's split copy ahead of the first search (MFMA then filters on it, MFMA_SPLIT contracts all of it) */
    int ready = 0;
    CK(hvs_index_prepare(idx, &ready));
    if (!ready) return 8;

    const int algos[5] = {HVS_ALGO_SCAN, HVS_ALGO_MFMA, HVS_ALGO_MFMA_F32, HVS_ALGO_MFMA_SPLIT, HVS_ALGO_AUTO};
    for (int a = 0; a < 5; a++) {
        float dist[40 * K];
        int64_t ids[40 * K];
        int32_t cnt[40];
        CK(hvs_search(idx, &queries[0][0], 40, 1, K, -1, NULL, INFINITY, 0, -1, dist, ids, cnt, 0, algos[a]));
        for (int q = 0; q < 40; q++) {
            if (cnt[q] != K) return 3;
            for (int j = 0; j < K; j++) {
                const int64_t id = ids[q * K + j];
                if (id < 0 || id >= N || fabs(ref[q][id] - dist[q * K + j]) > 1e-5) return 4;
                if (j && dist[q * K + j] < dist[q * K + j - 1]) return 5;
                int closer = 0; /* rows strictly closer than this hit (beyond rounding): exactly j of them */
                for (int r = 0; r < N; r++) closer += ref[q][r] < ref[q][id] - 2e-6;
                if (closer > j) return 6;
            }
        }
        if (ids[0] != 17 || dist[0] > 1e-5f) return 7;
    }

    /* WHERE group = 2 AND distance < 0.9, every matching row */
    float rd[N];
    int64_t ri[N], total = 0;
    CK(hvs_range_search(idx, &queries[1][0], 1, 2, NULL, 0.9, 0, -1, N, rd, ri, &total, 0));
    int64_t want = 0;
    for (int r = 0; r < N; r++) want += groups[r] == 2 && ref[1][r] < 0.9 - 2e-6;
    int64_t maybe = 0;
    for (int r = 0; r < N; r++) maybe += groups[r] == 2 && ref[1][r] < 0.9 + 2e-6;
    if (total < want || total > maybe) return 8;
    for (int64_t i = 0; i < total; i++)
        if (groups[ri[i]] != 2 || rd[i] >= 0.9f || (i && ri[i] <= ri[i - 1])) return 9;

    /* the table in group order: WHERE group = 2 becomes a row range; ids travel with their rows, so the answers name
       the same rows as a search of the table in arrival order */
    {
        float d_before[40 * K], d_after[40 * K];
        int64_t i_before[40 * K], i_after[40 * K];
        int32_t c_before[40], c_after[40];
        static int64_t old_pos[N];
        int64_t n_after = 0, ordered = 0, n_groups = 0;
        CK(hvs_search(idx, &queries[0][0], 40, 1, K, 2, NULL, INFINITY, 0, -1, d_before, i_before, c_before, 0, HVS_ALGO_MFMA));
        CK(hvs_index_partition(idx, 0, &n_after, old_pos));
        CK(hvs_index_partition_info(idx, &ordered, &n_groups));
        if (n_after != N || ordered != N || n_groups != 4) return 12;
        for (int64_t i = 1; i < N; i++)
            if (groups[old_pos[i]] < groups[old_pos[i - 1]] || (groups[old_pos[i]] == groups[old_pos[i - 1]] && old_pos[i] < old_pos[i - 1])) return 13;
        CK(hvs_search(idx, &queries[0][0], 40, 1, K, 2, NULL, INFINITY, 0, -1, d_after, i_after, c_after, 0, HVS_ALGO_MFMA));
        if (memcmp(i_before, i_after, sizeof(i_after)) || memcmp(d_before, d_after, sizeof(d_after)) || memcmp(c_before, c_after, sizeof(c_after))) return 14;
        for (int q = 0; q < 40; q++)
            for (int j = 0; j < K; j++)
                if (groups[i_after[q * K + j]] != 2 || fabs(ref[q][i_after[q * K + j]] - d_after[q * K + j]) > 1e-5) return 15;
    }
    CK(hvs_index_destroy(idx));

    /* a path 0 - 1 - 2 - 3 plus a self loop on 3; nodes 0 and 2 are chunks; seed on node 1 */
    const int32_t src[4] = {0, 1, 2, 3}, dst[4] = {1, 2, 3, 3};
    const uint32_t chunk_mask[1] = {0x5u};
    hvs_graph *g = NULL;
    CK(hvs_graph_create(0, NULL, 4, src, dst, 4, chunk_mask, 0, &g));
    const int32_t seed[1] = {1};
    const double weight[1] = {2.5};
    int iters = 0;
    CK(hvs_graph_pagerank(g, seed, weight, 1, 0.5, 100, 1e-6, &iters, NULL));
    double x[4];
    CK(hvs_graph_scores(g, 0, 4, x, 0));
    {   /* the same power iteration on the host */
        const double deg[4] = {1, 2, 2, 2}, p[4] = {0, 1, 0, 0};
        const int adj[4][2] = {{1, -1}, {0, 2}, {1, 3}, {2, 3}};
        double y[4] = {0.25, 0.25, 0.25, 0.25};
        for (int it = 0; it < 100; it++) {
            double z[4], err = 0;
            for (int j = 0; j < 4; j++) {
                double s = 0;
                for (int e = 0; e < 2; e++)
                    if (adj[j][e] >= 0) s += y[adj[j][e]] / deg[adj[j][e]];
                z[j] = 0.5 * s + 0.5 * p[j];
                err += fabs(z[j] - y[j]);
            }
            memcpy(y, z, sizeof(y));
            if (err < 4 * 1e-6) break;
        }
        for (int j = 0; j < 4; j++)
            if (fabs(x[j] - y[j]) > 1e-12) return 10;
    }
    int32_t top[2];
    double top_score[2];
    int n_top = 0;
    CK(hvs_graph_top_chunks(g, 2, top, top_score, &n_top));
    if (n_top != 2 || top[0] != (x[0] >= x[2] ? 0 : 2) || top_score[0] < top_score[1]) return 11;
    CK(hvs_graph_destroy(g));
    printf("abi_example ok (%d PageRank iterations)\n", iters);
    return 0;
}
