/*
 * hvs -- HIP vector store: C ABI of the MI355X (gfx950) dense-retrieval path.
 *
 * The reference (htlyu/hirag-prod) has no native boundary for this path: it
 * sends SQL to PostgreSQL and the pgvector extension computes
 * "vector <=> query ... ORDER BY distance LIMIT k" server-side.  Each entry
 * point below names the reference interface whose work it takes over; the
 * Python host side (hirag-prod_amd/_lib.py, used by index.py and store.py)
 * binds them with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message
 *     for the calling thread's last failure is hvs_last_error();
 *   - no C++ exceptions, torch types or HIP types cross the boundary (streams
 *     travel as void*);
 *   - the caller owns every buffer it passes; "on_device" says whether the
 *     caller's buffers are host memory (0) or memory of the index's GPU (1);
 *   - an hvs_index owns its device memory and works on one HIP stream; calls
 *     on one index must not overlap (the Python side serialises them);
 *   - functions return after their work on the stream has been submitted when
 *     on_device = 1 (call hvs_sync before reading results on another stream)
 *     and after it has completed when on_device = 0.
 */
#ifndef HVS_H
#define HVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hvs_index hvs_index;

/* element type of the stored rows; schema/vector_config.py:8-11 (HALFVEC = fp16, VECTOR = fp32) */
enum { HVS_F32 = 0, HVS_F16 = 1, HVS_BF16 = 2 };

/*
 * search algorithm selector.  SCAN: streaming dot products, one pass over the rows per 8 (fp32) / 4-8
 * (16-bit) embeddings.  MFMA: batched contraction on the matrix cores; for fp32 rows it runs on the
 * 16-bit pipe over a split copy of the rows (two fp16 slabs per row, three products, fp32-grade
 * scores; the copy doubles the table's device memory and is built on first use) and falls back to the
 * fp32 MFMA kernel when memory does not allow the copy.  For k <= 128 the split copy is first used as a
 * filter: one product over the hi slabs bounds every score to within 1.1e-3 (less where the rows and queries at hand
 * lie closer to their hi slabs, which is measured when they are split), the 4 k (at most 256) best candidates per
 * query are scored again from the fp32 rows in fp32, a query whose k-th exact score does not clear the bound of the
 * rows left out goes through the filter once more with 256 candidates, and what that leaves unproven is searched
 * again in full -- the exact top-k either way.  MFMA_SPLIT: the three products for every score, no filter.
 * MFMA_F32: the fp32 MFMA kernel, no copy.  The matrix-pipe routes take k <= 768 (larger k: SCAN, one pass per query).
 * AUTO picks by batch size.
 */
enum { HVS_ALGO_AUTO = 0, HVS_ALGO_SCAN = 1, HVS_ALGO_MFMA = 2, HVS_ALGO_MFMA_F32 = 3, HVS_ALGO_MFMA_SPLIT = 4 };

/* counters readable through hvs_profile_get */
enum { HVS_PROF_SCAN = 0, HVS_PROF_MFMA_SEARCH = 1, HVS_PROF_KMEANS_ASSIGN = 2, HVS_PROF_KMEANS_UPDATE = 3,
       HVS_PROF_MERGE = 4,
       HVS_PROF_REDO = 5, /* launches = queries whose filtered candidate list proved nothing and that were searched again in full (no time) */
       HVS_PROF_COUNT = 6 };

#define HVS_MAX_K 1024       /* largest top-k a search accepts */
#define HVS_MAX_FOLD 8       /* largest number of query embeddings folded by LEAST */
#define HVS_MAX_DIST_FOLD 128 /* ... by hvs_distances / hvs_range_search (no ranking: one pass on the matrix pipe, or one per 8) */

const char *hvs_last_error(void);
int hvs_version(void);
int hvs_device_count(int *out);

/*
 * Table life cycle.  Replaces PGVector._init_vdb / create_all
 * (src/hirag_prod/storage/pgvector.py:704-726): one index per embedded table
 * (Chunks, Items, Triplets).  stream = a hipStream_t to work on, or NULL for a
 * stream the index creates itself.
 */
int hvs_index_create(hvs_index **out, int device_id, int dim, int dtype, void *stream);
int hvs_index_destroy(hvs_index *idx);
int hvs_index_reserve(hvs_index *idx, int64_t capacity_rows);
int hvs_index_size(hvs_index *idx, int64_t *out_rows);
int hvs_index_info(hvs_index *idx, int *dim, int *dtype, int *device_id, int64_t *capacity_rows);
int hvs_sync(hvs_index *idx);

/*
 * Append rows.  Replaces the INSERT of PGVector.upsert_texts
 * (pgvector.py:128-172): rows arrive as src_dtype (HVS_F32 or the index's own
 * type) and are rounded to the column type on the way in, as storing into
 * HALFVEC does; the inverse L2 norm of every stored row is computed once here.
 * ids: external 64-bit id per row, or NULL for "first_id + position".
 * group_codes: one int32 per row standing for (workspaceId, knowledgeBaseId)
 * (the WHERE workspaceId = .. AND knowledgeBaseId = .. of pgvector.py:577-580),
 * or NULL for group 0.
 */
int hvs_index_add(hvs_index *idx, const void *rows, int src_dtype, int64_t n,
                  const int64_t *ids, const int32_t *group_codes, int on_device);

/* Overwrite stored rows in place (positions are row positions, not ids; they must be distinct within a call and are
 * range-checked, on the device when they arrive there). */
int hvs_index_update(hvs_index *idx, const int64_t *positions, const void *rows, int src_dtype,
                     int64_t n, int on_device);

/*
 * Mark rows deleted / live.  Replaces DELETE ... WHERE of PGVector.clean_table
 * (pgvector.py:452-473) for the vector side: a deleted row is never returned.
 */
int hvs_index_set_deleted(hvs_index *idx, const int64_t *positions, int64_t n, int deleted);

/* Number of rows currently marked deleted (tombstones not yet compacted away). */
int hvs_index_deleted_count(hvs_index *idx, int64_t *out_rows);

/*
 * Drop every deleted row and close the gaps, keeping the order of the
 * survivors; row positions change (position i afterwards = the i-th live row
 * before), ids and group codes move with their rows.  The space a DELETE of
 * PGVector.clean_table (pgvector.py:452-473) leaves is reclaimed by PostgreSQL's
 * vacuum; this is the same step for the HBM-resident table.  With renumber_ids
 * the ids become the new positions (for callers that use positions as ids).
 * *out_rows (may be NULL) receives the new row count.
 */
int hvs_index_compact(hvs_index *idx, int renumber_ids, int64_t *out_rows);

/*
 * Order the rows by group code, so that the WHERE workspaceId = .. AND knowledgeBaseId = .. every PGVector.query
 * carries (pgvector.py:577-580) selects a row range instead of a test per row: deleted rows are dropped as by
 * hvs_index_compact, the live rows are sorted by their group code (codes must be >= 0), rows of one code keep their
 * order, and the table remembers code -> [begin, end).  From then on a search, range search or distance filter with
 * group_filter >= 0 runs over that range with no code test, no live-row count and no copy; rows added afterwards form
 * a tail that is searched with the per-row test and merged in, until the next call orders them too.  Answers are the
 * same as before the call (ids travel with their rows; ties in distance still go to the lower row position, which
 * for rows of one group is the order they had).  Row positions change: out_old_positions (HOST array with room for
 * the row count before the call, or NULL) receives, for every new position, the position the row had.  renumber_ids:
 * ids become the new positions.  *out_rows (may be NULL) = the row count afterwards.  The rows move through new
 * arrays: the call needs as much free device memory as the table holds and fails, leaving the table as it was, when
 * that is not there.  A table without deleted rows whose codes already ascend (a bulk load ordered by workspace) is
 * not moved: the call only builds its range table.  Nearest thing in the reference: none (PostgreSQL would answer the same WHERE through a b-tree on
 * the two columns, which the reference does not declare either); the step is CLUSTER for the HBM-resident table.
 */
int hvs_index_partition(hvs_index *idx, int renumber_ids, int64_t *out_rows, int64_t *out_old_positions);
/* rows at the front of the table that are in group order (0: no partition) and the number of distinct codes among them */
int hvs_index_partition_info(hvs_index *idx, int64_t *ordered_rows, int64_t *n_groups);

/* Overwrite the external ids of row positions [first_pos, first_pos + n) from a HOST array (after a compaction or
 * partition whose caller keeps its own numbering of the rows).  No reference counterpart. */
int hvs_index_set_ids(hvs_index *idx, int64_t first_pos, int64_t n, const int64_t *ids);

/*
 * Bring the search-side copy of an fp32 table up to date ahead of the first search (it is otherwise built by the
 * first batched search): the [hi | lo] fp16 split copy that batched searches contract and that searches with
 * k <= 128 (single queries on large tables: k <= 64) use as a filter before exact fp32 scoring.  The nearest
 * thing in the reference is the index DDL of PGVector._init_vdb (pgvector.py:704-726): work done once so that
 * queries are fast.  *out_ready (may be NULL) = 1 when the copy is in place, 0 when the table is 16-bit (no
 * copy needed) or device memory does not allow it (searches then stay on the fp32 kernels).
 */
int hvs_index_prepare(hvs_index *idx, int *out_ready);

/*
 * That copy costs as much device memory again as the fp32 rows themselves (61 GB for 10M x 1536).  allow = 0: never
 * build it for this table and free it if it exists -- batched searches, large-table single queries and k-means then
 * stay on the fp32 kernels; allow = 1 (the state of a new index): build it when a search wants it; allow < 0: change
 * nothing.  *out_bytes (may be NULL) = device bytes the copy holds now, 0 for none.  No reference counterpart
 * (nearest: choosing not to CREATE INDEX on a table to save its space).
 */
int hvs_index_search_copy(hvs_index *idx, int allow, int64_t *out_bytes);

/*
 * Copy rows [first, first + n) to HOST buffers, for snapshots: rows as stored
 * (index dtype, n * dim elements, no padding; feeding them back through
 * hvs_index_add with src_dtype = the index dtype reproduces the table bit for
 * bit), ids, group codes, and one byte per row that is 1 for a deleted row.
 * Any output may be NULL.  The reference keeps its table in PostgreSQL and has
 * no counterpart; the format the vectors take there is the `vector` /
 * `halfvec` column of schema/vector_config.py:11.
 */
int hvs_index_export(hvs_index *idx, int64_t first, int64_t n, void *rows, int64_t *ids,
                     int32_t *group_codes, uint8_t *deleted);

/*
 * Fill the index with n synthetic rows generated on the device: row r (global
 * number first_global_row + r), column c is ((sum of the four 16-bit fields of
 * splitmix64((r*dim+c) ^ splitmix64(seed))) - 131070) * (1/37837), an
 * approximately N(0,1) value any host can regenerate bit for bit.  ids are the
 * global row numbers; group code = global row number mod n_groups for
 * n_groups > 0, 0 for n_groups = 0, and for n_groups < 0 the call's rows
 * arrive in group order: -n_groups runs of equal length with codes 0, 1, ...
 * (a bulk load ordered by workspace).
 * (Benchmark and parity-at-scale support; no reference counterpart.)
 */
int hvs_index_fill_synthetic(hvs_index *idx, int64_t n, uint64_t seed, int64_t first_global_row,
                             int32_t n_groups);

/*
 * The same for a clustered corpus, the shape text-embedding tables have: n_clusters centres (rows of the generator
 * above under seed + 1), global row r belongs to cluster splitmix64(r ^ splitmix64(seed ^ 0xC1)) mod n_clusters and is
 * centre + sigma * noise with noise = row r of the generator above under seed, the product and the sum each rounded to
 * float (no fused multiply-add); sigma^2 = one of {1, 1.4, 2, 2.8, 4, 5.6, 8, 10}e-3 chosen per cluster c by
 * splitmix64(c ^ splitmix64(seed ^ 0xC2)) & 7, so that two members of a cluster lie at cosine distance 1e-3 .. 1e-2 of
 * each other.  Queries of the same distribution: rows at global numbers past the corpus.  ids and group codes as above.
 * (Benchmark and parity-at-scale support; no reference counterpart.)
 */
int hvs_index_fill_clustered(hvs_index *idx, int64_t n, uint64_t seed, int64_t first_global_row,
                             int32_t n_groups, int64_t n_clusters);

/*
 * Exact cosine top-k.  Replaces the statement PGVector.query builds and
 * PostgreSQL + pgvector execute (pgvector.py:561-585):
 *     distance = LEAST_{j < fold}(vector <=> q[i*fold + j])
 *     WHERE [group = group_filter] AND [row_mask bit set] AND distance < max_distance
 *     ORDER BY distance ASC LIMIT k
 * queries: float [nq * fold, dim]; query i of the nq results folds embeddings
 * i*fold .. i*fold+fold-1.  group_filter < 0 = no group filter.  row_mask:
 * optional bitmask, bit (r & 31) of word r >> 5 set = row r may be returned
 * (the uri IN (...) / private = ... filters, evaluated by the host).
 * max_distance: strict upper bound, +inf for none.  [row_begin, row_end):
 * row positions to scan (row_end < 0 = all).
 * Outputs, caller-allocated: out_dist float [nq, k], out_ids int64 [nq, k]
 * (sorted by distance, ties by lower row position), out_count int32 [nq]
 * (entries beyond the count are +inf / -1).  NaN distances (zero-norm row or
 * query) are never returned.
 */
int hvs_search(hvs_index *idx, const float *queries, int nq, int fold, int k,
               int32_t group_filter, const uint32_t *row_mask, double max_distance,
               int64_t row_begin, int64_t row_end,
               float *out_dist, int64_t *out_ids, int32_t *out_count,
               int on_device, int algo);

/*
 * The WHERE clauses of PGVector.query beyond the (workspaceId, knowledgeBaseId) pair (pgvector.py:573-580), evaluated on
 * the GPU: every row carries four attributes next to its group code -- workspace code, knowledge-base code, uri code
 * (int32 each) and a private flag -- set with hvs_index_set_attributes after the rows were added (HOST arrays of n
 * entries for row positions [first_pos, first_pos + n); a NULL array leaves that attribute as it is, an attribute no
 * call has set reads -1 and satisfies no clause, so rows between hvs_index_add and this call pass no attribute filter;
 * attributes move with their rows in hvs_index_compact / hvs_index_partition).  An hvs_filter names the values a row must have;
 * the searches below evaluate it in one pass over the attribute columns (16 B per row) into a row bitmask on the
 * device and then run as hvs_search / hvs_range_search do with that mask: no per-query mask is built or uploaded by the
 * host.  The uri list is a HOST array of at most HVS_MAX_FILTER_URIS codes in any order.
 */
typedef struct hvs_filter {
    int32_t group;          /* code of (workspaceId, knowledgeBaseId), -1 = any */
    int32_t workspace;      /* WHERE workspaceId = ..      alone, -1 = any */
    int32_t knowledge_base; /* WHERE knowledgeBaseId = ..  alone, -1 = any */
    int32_t access;         /* WHERE private = ..: 1 = private rows only, 0 = public rows only, -1 = any */
    const int32_t *uris;    /* WHERE uri IN (..): codes, NULL = any */
    int32_t n_uris;
} hvs_filter;
#define HVS_MAX_FILTER_URIS 8192
int hvs_index_set_attributes(hvs_index *idx, int64_t first_pos, int64_t n, const int32_t *workspace,
                             const int32_t *knowledge_base, const int32_t *uri, const uint8_t *is_private);
/* hvs_search with the filter in place of group_filter / row_mask (filter NULL = no WHERE) */
int hvs_search_filtered(hvs_index *idx, const float *queries, int nq, int fold, int k,
                        const hvs_filter *filter, double max_distance, int64_t row_begin, int64_t row_end,
                        float *out_dist, int64_t *out_ids, int32_t *out_count, int on_device, int algo);
/* hvs_range_search likewise */
int hvs_range_search_filtered(hvs_index *idx, const float *queries, int fold, const hvs_filter *filter,
                              double max_distance, int64_t row_begin, int64_t row_end, int64_t capacity,
                              float *out_dist, int64_t *out_ids, int64_t *out_total, int on_device);

/*
 * hvs_search for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same statement and the same answers as hvs_search with
 * on_device = 1 and HVS_ALGO_SCAN (the streaming scan; every pointer is memory of the index's GPU; returns after
 * the work has been submitted to the index's stream), but the prepared queries, the block lists and the merge levels
 * live in one HVS_NOALLOC_SCRATCH_BYTES array inside the library's code object -- held by every GPU the library runs
 * on until the process ends -- and the call makes no hipMalloc / hipFree / allocating library call
 * (hvs_index_create looks the array's address up once per device).  Limits, refused with a message:
 * nq * fold * (4 * padded dim + 4) <= 1 MiB (170 embeddings of 1536 columns), and the lists of one query must fit
 * the rest: k * 8 bytes * (P + ceil(P / 32)) <= 7 MiB for the P lists of the scan's resident blocks and their first
 * merge level, i.e. k <= 868 on a full grid of P = 1024, any k <= HVS_MAX_K where fewer blocks are resident (small
 * tables; queries of several embeddings; k = 1024, whose lists leave room for three blocks per compute unit).  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU -- on different indexes or streams too -- race; serialise them.
 */
#define HVS_NOALLOC_SCRATCH_BYTES (8u << 20)
int hvs_search_noalloc(hvs_index *idx, const float *queries, int nq, int fold, int k,
                       int32_t group_filter, const uint32_t *row_mask, double max_distance,
                       int64_t row_begin, int64_t row_end,
                       float *out_dist, int64_t *out_ids, int32_t *out_count);

/*
 * Full distance column.  Replaces the projected LEAST(Item.vector <=> e_i)
 * column of cross-language search
 * (src/hirag_prod/cross_language_search/search_service.py:100-113):
 * out[r - row_begin] = LEAST_j(distance(row r, q_j)) for every row of the
 * range, NaN where undefined; no filtering, no ordering.  fold counts the
 * sentence embeddings of one query, 1 .. HVS_MAX_DIST_FOLD (the same limit
 * holds for hvs_range_search).
 */
int hvs_distances(hvs_index *idx, const float *queries, int fold,
                  int64_t row_begin, int64_t row_end, float *out, int on_device);

/*
 * Every row whose distance is below a bound: the (distance, id) pairs of the
 * rows of [row_begin, row_end) that pass the filters and have LEAST-folded
 * distance < max_distance (strict, NaN never passes), in ascending row
 * position.  At most `capacity` pairs are written; *out_total (a HOST int64)
 * receives the number of matching rows, which may exceed capacity.  The
 * consumer of the reference's projected distance column keeps exactly these
 * rows: `search_sentence_cosine_distance < 0.4`,
 * cross_language_search/functions.py:330-340 over the column selected at
 * cross_language_search/search_service.py:100-113.  The call returns after the
 * work has completed.
 */
int hvs_range_search(hvs_index *idx, const float *queries, int fold, int32_t group_filter,
                     const uint32_t *row_mask, double max_distance, int64_t row_begin, int64_t row_end,
                     int64_t capacity, float *out_dist, int64_t *out_ids, int64_t *out_total, int on_device);

/*
 * Fetch stored vectors by row position, widened to float.  Replaces selecting
 * the "vector" column in PGVector.query_by_keys (pgvector.py:606-660) as used
 * by QueryService.query_chunk_embeddings (storage/query_service.py:194-224).
 */
int hvs_gather(hvs_index *idx, const int64_t *positions, int64_t n, float *out, int on_device);

/*
 * Merge P sorted candidate lists per query into one top-k (the step after the
 * all-gather when the corpus is row-sharded over GPUs).  Device buffers on
 * device_id: in_dist float [nq, P, k], in_ids int64 [nq, P, k], in_count int32
 * [nq, P]; outputs as hvs_search.  Ordering: distance, then lower id.
 * No reference counterpart (the reference is single-node).
 */
int hvs_merge_topk(int device_id, void *stream, const float *in_dist, const int64_t *in_ids,
                   const int32_t *in_count, int nq, int P, int k,
                   float *out_dist, int64_t *out_ids, int32_t *out_count);

/*
 * ---- row shards over the GPUs of one node: one process per GPU, one communicator per process ----
 * The reference is single-node (one PostgreSQL server) and has no counterpart to any of these.  Each rank holds a
 * contiguous row range of the table in its own hvs_index, with global row numbers as ids; queries and centroids are
 * replicated.
 *
 * hvs_comm_unique_id: rank 0 makes the HVS_UNIQUE_ID_BYTES rendezvous token (ncclGetUniqueId) and the caller hands it
 * to every rank by whatever channel it has (a file, MPI, torch.distributed).  hvs_comm_init: ncclCommInitRank on
 * device_id -- RCCL over xGMI, the product transport; collective on all ranks.  librccl is opened on first use
 * (HVS_RCCL_PATH, $ROCM_PATH/lib, /opt/rocm/lib) and must be bound to the same HIP runtime as this library, which the
 * call checks (a process can hold two; HVS_RCCL_ANY_HIP=1 skips the check).
 * hvs_comm_init_host: the same exchanges through two caller-supplied functions over HOST buffers, for ranks that cannot
 * form an RCCL communicator (processes sharing one GPU, a process group without device collectives).
 *   allgather(ctx, send, recv, bytes): every rank contributes `bytes` bytes; recv = the W contributions in rank order.
 *   allreduce(ctx, buf, count, dtype): element-wise sum over the ranks, in place; dtype HVS_F32 (float) or
 *     HVS_REDUCE_I64 (int64_t).
 *   Both return 0 on success.
 */
typedef struct hvs_comm hvs_comm;
#define HVS_UNIQUE_ID_BYTES 128
#define HVS_REDUCE_I64 100
typedef int (*hvs_allgather_fn)(void *ctx, const void *send, void *recv, uint64_t bytes_per_rank);
typedef int (*hvs_allreduce_fn)(void *ctx, void *buf, uint64_t count, int dtype);
int hvs_comm_unique_id(void *out, int bytes);
int hvs_comm_init(hvs_comm **out, int rank, int world, const void *unique_id, int device_id);
int hvs_comm_init_host(hvs_comm **out, int rank, int world, hvs_allgather_fn allgather,
                       hvs_allreduce_fn allreduce, void *ctx);
int hvs_comm_destroy(hvs_comm *comm);
int hvs_comm_info(hvs_comm *comm, int *rank, int *world, int *is_rccl);

/*
 * hvs_search over a row-sharded table: the local search of this rank's index (arguments as hvs_search, whole index),
 * ONE all-gather of every rank's packed (distances, ids, counts) record -- k * 12 + 4 bytes per query and rank -- and a
 * k-way merge by (distance, id) on every rank, all on the index's stream.  Every rank calls it with the same queries and
 * gets the same answer, which equals the single-index answer over the whole table.  Collective.
 * A rank whose local search fails (device memory for a workspace, say) still enters the all-gather, with empty lists and
 * a status word in its record; every rank then returns an error (the failing rank its own, the others one that names
 * it), and the communicator stays usable.  What this cannot cover is a rank that fails before it holds the buffer the
 * all-gather receives into (arguments the ranks disagree on, no memory for the staging itself): the other ranks then
 * wait in the collective, and the communicator is lost.  To read the status words the call waits for the all-gather,
 * so with on_device = 1 only the merge is still in flight when it returns.
 */
int hvs_search_sharded(hvs_index *idx, hvs_comm *comm, const float *queries, int nq, int fold, int k,
                       int32_t group_filter, const uint32_t *row_mask, double max_distance,
                       float *out_dist, int64_t *out_ids, int32_t *out_count, int on_device, int algo);

/*
 * Sum the k-means partial sums [n_centroids, dim] and counts [n_centroids] of hvs_kmeans_assign over the ranks, in
 * place (25.2 MB at 4096 centroids of 1536 columns); on return with on_device = 1 the work is complete for the host
 * transport and submitted to the index's stream for RCCL.  Collective.
 */
int hvs_allreduce_centroids(hvs_index *idx, hvs_comm *comm, float *sums, int64_t *counts, int n_centroids, int on_device);

/*
 * Dense cosine-similarity matrix of two small row sets: out[i, j] =
 * a_i . b_j / (|a_i| |b_j|), and 0 where either norm is 0 (scikit-learn's
 * cosine_similarity).  a is float [m, dim], b float [n, dim], out float [m, n],
 * all row-major without padding; on_device selects device or host pointers for
 * all three.  Replaces the sklearn / numpy calls at HiRAG.calculate_similarity
 * (hirag.py:571-588), validate_similarity and the synonym filter
 * (cross_language_search/functions.py:85-113, 152-165) and the cosine block of
 * HiRAG.dpr_recall_chunks (hirag.py:1305-1314).
 */
int hvs_cosine_matrix(int device_id, void *stream, const float *a, int m, const float *b, int n, int dim,
                      float *out, int on_device);

/*
 * One k-means E-step plus the partial sums of the M-step over the index's rows
 * (index dtype must be HVS_F32): assign[r] = argmin_j |x_r - c_j|^2 (lowest j
 * on ties); sums[j] += x_r, counts[j] += 1.  out_sums / out_counts may be NULL
 * (assignment only); they are overwritten, not accumulated.  Deleted rows get
 * assignment -1 and add nothing.  No reference counterpart (the reference's
 * only clustering is ward agglomeration of <= 10 vectors,
 * src/hirag_prod/cluster/hierachical_cluster.py:42-46); the sums are what a
 * multi-GPU build all-reduces.
 */
int hvs_kmeans_assign(hvs_index *idx, const float *centroids, int n_centroids,
                      int32_t *out_assign, float *out_sums, int64_t *out_counts, int on_device);

/* ---- the step after retrieval: personalised PageRank over the knowledge graph ---- */

typedef struct hvs_graph hvs_graph;

/*
 * The undirected simple graph of one (workspace, knowledge base) as a CSR
 * adjacency in device memory: what PGVector.pagerank_top_chunks_with_reset
 * builds per query with networkx (DiGraph.add_edge over the Graph rows, then
 * to_undirected(), src/hirag_prod/storage/pgvector.py:395-412).  src / dst
 * hold node numbers in [0, n_nodes); parallel and antiparallel rows collapse
 * and a self loop is one diagonal entry.  chunk_mask has one bit per node
 * (bit j of word j/32): set when the node id starts with "chunk-"
 * (pgvector.py:435-439).  on_device selects device or host pointers for the
 * three arrays.  stream as in hvs_index_create.
 */
int hvs_graph_create(int device_id, void *stream, int64_t n_nodes, const int32_t *src, const int32_t *dst,
                     int64_t n_edges, const uint32_t *chunk_mask, int on_device, hvs_graph **out);
int hvs_graph_destroy(hvs_graph *g);
/* nodes, stored adjacency entries (both directions), nodes without neighbours; any pointer may be NULL */
int hvs_graph_info(hvs_graph *g, int64_t *n_nodes, int64_t *nnz, int64_t *n_dangling);
/* the adjacency to HOST memory: row_ptr [n_nodes + 1], col [nnz] ascending in each row (either may be NULL) */
int hvs_graph_export(hvs_graph *g, int64_t *row_ptr, int32_t *col);

/*
 * networkx.pagerank(G, alpha, personalization) as the reference calls it
 * (pgvector.py:432-434; algorithm: networkx 3.3 _pagerank_scipy): float64 power
 * iteration x <- alpha (x A + dangling(x) p) + (1 - alpha) p from x = 1/N, with
 * A the row-normalised adjacency and p the seed weights divided by their sum,
 * until sum|x - x_prev| < N * tol.  seed_nodes / seed_weights are HOST arrays;
 * weights must be finite and > 0 (the caller drops the others, pgvector.py:
 * 416-425).  The scores stay in the graph object for hvs_graph_scores /
 * hvs_graph_top_chunks.  Returns 3 (networkx: PowerIterationFailedConvergence)
 * when max_iter iterations do not reach the bound; the library's defaults are
 * max_iter 100 and tol 1e-6.  out_iters / out_err may be NULL.
 */
int hvs_graph_pagerank(hvs_graph *g, const int32_t *seed_nodes, const double *seed_weights, int64_t n_seeds,
                       double alpha, int max_iter, double tol, int *out_iters, double *out_err);
/* scores of nodes [first, first + n) of the last hvs_graph_pagerank */
int hvs_graph_scores(hvs_graph *g, int64_t first, int64_t n, double *out, int on_device);
/*
 * The k best chunk nodes of the last hvs_graph_pagerank by descending score,
 * equal scores in ascending node number (the reference's stable sort over the
 * graph's node order, pgvector.py:435-441).  HOST outputs of k entries;
 * *out_count = min(k, chunk nodes).
 */
int hvs_graph_top_chunks(hvs_graph *g, int k, int32_t *out_nodes, double *out_scores, int *out_count);

/* Per-kernel device time, measured with HIP events on the index's stream. */
int hvs_profile_enable(hvs_index *idx, int on);
int hvs_profile_get(hvs_index *idx, int which, double *total_ms, int64_t *launches);
int hvs_profile_reset(hvs_index *idx);

/* Generic timer on the index's stream (HIP events): start, later stop -> elapsed ms. */
int hvs_timer_start(hvs_index *idx);
int hvs_timer_stop(hvs_index *idx, double *elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* HVS_H */
