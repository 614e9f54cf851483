// Row-sharded search and k-means over the GPUs of one node, below Python: the communicator object and the two exchange
// steps the path has -- an all-gather of every rank's local top-k lists followed by a k-way merge, and an all-reduce of
// the k-means partial sums.  Two transports behind one interface:
//   * RCCL (ncclCommInitRank / ncclAllGather / ncclAllReduce on the index's own stream): the product configuration, one
//     process per GPU over xGMI.  librccl (half a gigabyte) is opened on the first hvs_comm_init, not at library load.
//   * host-staged callbacks: the lists leave through pinned memory, the caller's function exchanges host buffers (gloo,
//     MPI, a pipe), the result goes back up.  For ranks that cannot form an RCCL communicator (two processes sharing one
//     GPU in a test, a process group without device collectives).
// The merge orders by (distance, id), so the answer does not depend on the number of shards.
#include "hvs_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

// the launcher of the k-way merge (hvs_scan.hip): P lists per query, element strides between queries and between lists
int hvs_merge_launch(hipStream_t st, const float *in_dist, const int64_t *in_ids, const int32_t *in_count, int nq, int P, int k,
                     int64_t dist_q, int64_t dist_p, int64_t ids_q, int64_t ids_p, int64_t cnt_q, int64_t cnt_p, float *out_dist,
                     int64_t *out_ids, int32_t *out_count);

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;

// RTLD_DEEPBIND: a process may already hold another RCCL (PyTorch's wheel carries one bound to its own HIP runtime);
// this one must resolve its symbols inside itself and against the HIP runtime libhvs is linked to, whose streams it gets
int rccl_load() {
    std::lock_guard<std::mutex> guard(g_rccl_mutex);
    if (g_rccl.handle) return 0;
    std::string tried;
    void *h = nullptr;
    const char *env = getenv("HVS_RCCL_PATH");
    const char *rocm = getenv("ROCM_PATH");
    std::string candidates[4] = {env ? env : "", rocm ? std::string(rocm) + "/lib/librccl.so.1" : "", "/opt/rocm/lib/librccl.so.1",
                                 "librccl.so.1"};
    for (const std::string &c : candidates) {
        if (c.empty()) continue;
        h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (h) break;
        tried += c + " (" + (dlerror() ? "not loadable" : "?") + ") ";
    }
    HVS_REQUIRE(h != nullptr, "hvs_comm_init: RCCL not found; tried %s(set HVS_RCCL_PATH)", tried.c_str());
#define HVS_SYM(field, name)                                                              \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));              \
    HVS_REQUIRE(g_rccl.field != nullptr, "hvs_comm_init: %s missing from the RCCL library", name)
    HVS_SYM(GetUniqueId, "ncclGetUniqueId");
    HVS_SYM(CommInitRank, "ncclCommInitRank");
    HVS_SYM(CommDestroy, "ncclCommDestroy");
    HVS_SYM(AllGather, "ncclAllGather");
    HVS_SYM(AllReduce, "ncclAllReduce");
    HVS_SYM(GroupStart, "ncclGroupStart");
    HVS_SYM(GroupEnd, "ncclGroupEnd");
    HVS_SYM(GetErrorString, "ncclGetErrorString");
#undef HVS_SYM
    // the streams this library hands to RCCL belong to the HIP runtime libhvs is bound to: RCCL must be bound to the same
    // one (a process can hold two -- PyTorch's wheel carries its own -- and which of them a library gets depends on
    // SONAMEs and load order)
    if (!getenv("HVS_RCCL_ANY_HIP")) {
        Dl_info theirs, ours;
        void *sym = dlsym(h, "hipStreamCreate");
        if (sym && dladdr(sym, &theirs) && dladdr((void *)&hipStreamCreate, &ours) && theirs.dli_fname && ours.dli_fname &&
            strcmp(theirs.dli_fname, ours.dli_fname) != 0) {
            hvs_set_error("hvs_comm_init: RCCL is bound to the HIP runtime %s, this library to %s; their streams are not interchangeable "
                          "(HVS_RCCL_PATH picks another RCCL, HVS_RCCL_ANY_HIP=1 skips this check)", theirs.dli_fname, ours.dli_fname);
            dlclose(h);
            return 1;
        }
    }
    g_rccl.handle = h;
    return 0;
}

}  // namespace

#define HVS_NCCL(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) {                                                                             \
            hvs_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r_));     \
            return 1;                                                                                        \
        }                                                                                                    \
    } while (0)

struct hvs_comm {
    int rank = 0, world = 1, device = -1;
    ncclComm_t nccl = nullptr;  // RCCL transport
    hvs_allgather_fn allgather = nullptr;  // host-staged transport
    hvs_allreduce_fn allreduce = nullptr;
    void *ctx = nullptr;
    void *h_buf = nullptr;  // pinned staging of the host transport
    size_t h_bytes = 0;
};

static int comm_pinned(hvs_comm *c, size_t bytes, void **out) {
    if (c->h_bytes < bytes) {
        if (c->h_buf) HVS_HIP(hipHostFree(c->h_buf));
        c->h_buf = nullptr;
        c->h_bytes = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        HVS_HIP(hipHostMalloc(&c->h_buf, want, hipHostMallocDefault));
        c->h_bytes = want;
    }
    *out = c->h_buf;
    return 0;
}

extern "C" int hvs_comm_unique_id(void *out, int bytes) {
    HVS_REQUIRE(out != nullptr && bytes >= HVS_UNIQUE_ID_BYTES, "hvs_comm_unique_id: needs a buffer of %d bytes", HVS_UNIQUE_ID_BYTES);
    static_assert(HVS_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    HVS_TRY(rccl_load());
    ncclUniqueId id;
    HVS_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int hvs_comm_init(hvs_comm **out, int rank, int world, const void *unique_id, int device_id) {
    HVS_REQUIRE(out != nullptr, "hvs_comm_init: out is NULL");
    *out = nullptr;
    HVS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "hvs_comm_init: rank %d not in [0, %d)", rank, world);
    HVS_REQUIRE(unique_id != nullptr, "hvs_comm_init: unique_id is NULL (rank 0 makes one with hvs_comm_unique_id)");
    int ndev = 0;
    HVS_TRY(hvs_device_count(&ndev));
    HVS_REQUIRE(device_id >= 0 && device_id < ndev, "hvs_comm_init: device %d not in [0, %d)", device_id, ndev);
    HVS_TRY(rccl_load());
    HVS_HIP(hipSetDevice(device_id));
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    hvs_comm *c = new hvs_comm();
    c->rank = rank;
    c->world = world;
    c->device = device_id;
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        hvs_set_error("hvs_comm_init: ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device_id, g_rccl.GetErrorString(r));
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

extern "C" int hvs_comm_init_host(hvs_comm **out, int rank, int world, hvs_allgather_fn allgather, hvs_allreduce_fn allreduce,
                                  void *ctx) {
    HVS_REQUIRE(out != nullptr, "hvs_comm_init_host: out is NULL");
    *out = nullptr;
    HVS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "hvs_comm_init_host: rank %d not in [0, %d)", rank, world);
    HVS_REQUIRE(allgather != nullptr && allreduce != nullptr, "hvs_comm_init_host: both exchange functions are needed");
    hvs_comm *c = new hvs_comm();
    c->rank = rank;
    c->world = world;
    c->allgather = allgather;
    c->allreduce = allreduce;
    c->ctx = ctx;
    *out = c;
    return 0;
}

extern "C" int hvs_comm_destroy(hvs_comm *c) {
    if (!c) return 0;
    if (c->nccl) {
        (void)hipSetDevice(c->device);
        g_rccl.CommDestroy(c->nccl);
    }
    if (c->h_buf) hipHostFree(c->h_buf);
    delete c;
    return 0;
}

extern "C" int hvs_comm_info(hvs_comm *c, int *rank, int *world, int *is_rccl) {
    HVS_REQUIRE(c != nullptr, "hvs_comm_info: comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (is_rccl) *is_rccl = c->nccl ? 1 : 0;
    return 0;
}

static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

__global__ void any_status_kernel(const char *records, size_t rec, size_t o_status, int W, int *out) {
    int bad = -1;
    for (int r = threadIdx.x; r < W; r += blockDim.x)
        if (*reinterpret_cast<const int *>(records + (size_t)r * rec + o_status) != 0) bad = r;
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(bad, off);
        bad = o > bad ? o : bad;
    }
    if (threadIdx.x == 0) *out = bad;  // the highest rank that reported a failure, -1 for none
}

// One packed record per rank: [nq, k] distances | [nq, k] ids | [nq] counts | status, each part 256-byte aligned, so
// that ONE all-gather moves everything the merge needs.  A rank whose local search fails still enters the collective
// (with empty lists and a non-zero status word), so that the others are not left waiting in it; every rank then returns
// an error.  What cannot be covered: a rank that fails before it has the staging buffer the collective receives into.
extern "C" int hvs_search_sharded(hvs_index *idx, hvs_comm *comm, const float *queries, int nq, int fold, int k,
                                  int32_t group_filter, const uint32_t *row_mask, double max_distance, float *out_dist,
                                  int64_t *out_ids, int32_t *out_count, int on_device, int algo) {
    HVS_REQUIRE(idx != nullptr && comm != nullptr, "hvs_search_sharded: NULL index or communicator");
    HVS_REQUIRE(nq >= 0, "hvs_search_sharded: negative query count");
    if (nq == 0) return 0;
    HVS_REQUIRE(queries && out_dist && out_ids && out_count, "hvs_search_sharded: NULL argument");
    HVS_REQUIRE(k >= 1 && k <= HVS_MAX_K, "hvs_search_sharded: k %d not in [1, %d]", k, HVS_MAX_K);
    HVS_REQUIRE(!comm->nccl || comm->device == idx->device, "hvs_search_sharded: the communicator was made for device %d, the index lives on %d",
                comm->device, idx->device);
    HVS_HIP(hipSetDevice(idx->device));
    const int W = comm->world;
    const size_t o_ids = up256((size_t)nq * k * 4), o_cnt = o_ids + up256((size_t)nq * k * 8), o_status = o_cnt + up256((size_t)nq * 4),
                 rec = o_status + 256;
    // staging (workspace 3, which a search with device buffers leaves alone):
    // [W records][merged outputs, if they go to the host][queries and mask, if they come from the host]
    const size_t o_out = (size_t)W * rec;
    const size_t o_q = o_out + (on_device ? 0 : rec);
    const size_t q_bytes = on_device ? 0 : up256((size_t)nq * fold * idx->dim * 4);
    const size_t mask_bytes = (!on_device && row_mask) ? up256((size_t)((idx->n + 31) / 32) * 4) : 0;
    void *ws = nullptr;
    HVS_TRY(hvs_ws_get(idx, 3, o_q + q_bytes + mask_bytes, &ws));
    char *base = (char *)ws;
    const float *d_q = queries;
    const uint32_t *d_mask = row_mask;
    if (!on_device) {
        HVS_HIP(hipMemcpyAsync(base + o_q, queries, (size_t)nq * fold * idx->dim * 4, hipMemcpyHostToDevice, idx->stream));
        d_q = (const float *)(base + o_q);
        if (row_mask) {
            HVS_HIP(hipMemcpyAsync(base + o_q + q_bytes, row_mask, (size_t)((idx->n + 31) / 32) * 4, hipMemcpyHostToDevice, idx->stream));
            d_mask = (const uint32_t *)(base + o_q + q_bytes);
        }
    }
    // the local top-k lands in this rank's record
    char *mine = base + (size_t)comm->rank * rec;
    int local_rc = hvs_search(idx, d_q, nq, fold, k, group_filter, d_mask, max_distance, 0, -1, (float *)mine, (int64_t *)(mine + o_ids),
                              (int32_t *)(mine + o_cnt), 1, algo);
    if (const char *e = getenv("HVS_TEST_FAIL_RANK"))  // tests: this rank's local search counts as failed
        if (local_rc == 0 && atoi(e) == comm->rank) {
            hvs_set_error("hvs_search_sharded: local search failed on purpose (HVS_TEST_FAIL_RANK)");
            local_rc = 2;
        }
    const std::string local_err = local_rc ? hvs_last_error() : "";
    if (local_rc) HVS_HIP(hipMemsetAsync(mine + o_cnt, 0, (size_t)nq * 4, idx->stream));  // empty lists
    HVS_HIP(hipMemsetAsync(mine + o_status, local_rc ? 1 : 0, 4, idx->stream));
    int failed_rank = local_rc ? comm->rank : -1;
    if (W > 1 || comm->nccl) {  // a one-rank RCCL communicator still runs its (trivial) collective: same code path at every size
        if (comm->nccl) {
            HVS_NCCL(g_rccl.AllGather(mine, base, rec, ncclChar, comm->nccl, idx->stream));  // in place: send = recv + rank * rec
            void *h = nullptr;
            HVS_TRY(comm_pinned(comm, 256, &h));
            int *d_flag = (int *)(base + (size_t)comm->rank * rec + o_status + 64);  // spare bytes of this rank's own status slot
            any_status_kernel<<<1, 64, 0, idx->stream>>>(base, rec, o_status, W, d_flag);
            HVS_HIP(hipGetLastError());
            HVS_HIP(hipMemcpyAsync(h, d_flag, 4, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipStreamSynchronize(idx->stream));
            failed_rank = *(const int *)h;
        } else {
            void *h = nullptr;
            HVS_TRY(comm_pinned(comm, (size_t)(W + 1) * rec, &h));
            char *h_send = (char *)h + (size_t)W * rec;
            HVS_HIP(hipMemcpyAsync(h_send, mine, rec, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipStreamSynchronize(idx->stream));
            const int rc = comm->allgather(comm->ctx, h_send, h, rec);
            HVS_REQUIRE(rc == 0, "hvs_search_sharded: the all-gather callback failed with status %d", rc);
            for (int r = 0; r < W; r++)
                if (*(const int *)((const char *)h + (size_t)r * rec + o_status) != 0) failed_rank = r;
            HVS_HIP(hipMemcpyAsync(base, h, (size_t)W * rec, hipMemcpyHostToDevice, idx->stream));
        }
    }
    if (local_rc) {
        hvs_set_error("%s", local_err.c_str());
        return local_rc;
    }
    HVS_REQUIRE(failed_rank < 0, "hvs_search_sharded: the local search of rank %d failed (its own error message says why)", failed_rank);
    float *m_dist = on_device ? out_dist : (float *)(base + o_out);
    int64_t *m_ids = on_device ? out_ids : (int64_t *)(base + o_out + o_ids);
    int32_t *m_cnt = on_device ? out_count : (int32_t *)(base + o_out + o_cnt);
    HVS_TRY(hvs_prof_begin(idx));
    HVS_TRY(hvs_merge_launch(idx->stream, (const float *)base, (const int64_t *)(base + o_ids), (const int32_t *)(base + o_cnt), nq, W, k,
                             k, (int64_t)(rec / 4), k, (int64_t)(rec / 8), 1, (int64_t)(rec / 4), m_dist, m_ids, m_cnt));
    HVS_TRY(hvs_prof_end(idx, HVS_PROF_MERGE));
    if (!on_device) {
        HVS_HIP(hipMemcpyAsync(out_dist, m_dist, (size_t)nq * k * 4, hipMemcpyDeviceToHost, idx->stream));
        HVS_HIP(hipMemcpyAsync(out_ids, m_ids, (size_t)nq * k * 8, hipMemcpyDeviceToHost, idx->stream));
        HVS_HIP(hipMemcpyAsync(out_count, m_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, idx->stream));
        HVS_HIP(hipStreamSynchronize(idx->stream));
    }
    return 0;
}

extern "C" int hvs_allreduce_centroids(hvs_index *idx, hvs_comm *comm, float *sums, int64_t *counts, int n_centroids, int on_device) {
    HVS_REQUIRE(idx != nullptr && comm != nullptr, "hvs_allreduce_centroids: NULL index or communicator");
    HVS_REQUIRE(sums && counts && n_centroids >= 1, "hvs_allreduce_centroids: NULL or empty argument");
    if (comm->world == 1 && !comm->nccl) return 0;
    HVS_HIP(hipSetDevice(idx->device));
    const size_t n_sum = (size_t)n_centroids * idx->dim, sum_bytes = n_sum * 4, cnt_bytes = (size_t)n_centroids * 8;
    if (comm->nccl) {
        float *d_sums = sums;
        int64_t *d_counts = counts;
        if (!on_device) {
            void *ws = nullptr;
            HVS_TRY(hvs_ws_get(idx, 3, up256(sum_bytes) + cnt_bytes, &ws));
            d_sums = (float *)ws;
            d_counts = (int64_t *)((char *)ws + up256(sum_bytes));
            HVS_HIP(hipMemcpyAsync(d_sums, sums, sum_bytes, hipMemcpyHostToDevice, idx->stream));
            HVS_HIP(hipMemcpyAsync(d_counts, counts, cnt_bytes, hipMemcpyHostToDevice, idx->stream));
        }
        HVS_NCCL(g_rccl.GroupStart());
        HVS_NCCL(g_rccl.AllReduce(d_sums, d_sums, n_sum, ncclFloat32, ncclSum, comm->nccl, idx->stream));
        HVS_NCCL(g_rccl.AllReduce(d_counts, d_counts, (size_t)n_centroids, ncclInt64, ncclSum, comm->nccl, idx->stream));
        HVS_NCCL(g_rccl.GroupEnd());
        if (!on_device) {
            HVS_HIP(hipMemcpyAsync(sums, d_sums, sum_bytes, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipMemcpyAsync(counts, d_counts, cnt_bytes, hipMemcpyDeviceToHost, idx->stream));
            HVS_HIP(hipStreamSynchronize(idx->stream));
        }
        return 0;
    }
    // host-staged: the callback reduces host buffers in place
    float *h_sums = sums;
    int64_t *h_counts = counts;
    if (on_device) {
        void *h = nullptr;
        HVS_TRY(comm_pinned(comm, up256(sum_bytes) + cnt_bytes, &h));
        h_sums = (float *)h;
        h_counts = (int64_t *)((char *)h + up256(sum_bytes));
        HVS_HIP(hipMemcpyAsync(h_sums, sums, sum_bytes, hipMemcpyDeviceToHost, idx->stream));
        HVS_HIP(hipMemcpyAsync(h_counts, counts, cnt_bytes, hipMemcpyDeviceToHost, idx->stream));
        HVS_HIP(hipStreamSynchronize(idx->stream));
    }
    int rc = comm->allreduce(comm->ctx, h_sums, n_sum, HVS_F32);
    HVS_REQUIRE(rc == 0, "hvs_allreduce_centroids: the all-reduce callback failed with status %d (sums)", rc);
    rc = comm->allreduce(comm->ctx, h_counts, (size_t)n_centroids, HVS_REDUCE_I64);
    HVS_REQUIRE(rc == 0, "hvs_allreduce_centroids: the all-reduce callback failed with status %d (counts)", rc);
    if (on_device) {
        HVS_HIP(hipMemcpyAsync(sums, h_sums, sum_bytes, hipMemcpyHostToDevice, idx->stream));
        HVS_HIP(hipMemcpyAsync(counts, h_counts, cnt_bytes, hipMemcpyHostToDevice, idx->stream));
        HVS_HIP(hipStreamSynchronize(idx->stream));
    }
    return 0;
}
