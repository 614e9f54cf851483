// Stable counting sort of item numbers by an integer key, hand-written for wave64: the one sort this library needs
// on its paths (rows by k-means assignment for the reproducible centroid sums, rows by group code for the
// group-partitioned table order).  No reference counterpart (PostgreSQL orders rows itself).
//
// Least-significant-digit passes of SORT_BITS bits.  A pass cuts the item sequence into chunks of SORT_CHUNK
// consecutive slots, one chunk per wave:
//   count   - every wave counts the digits of its chunk in its own LDS histogram -> counts[bin][chunk]
//   total   - one wave per bin adds that bin's row of counts
//   scan    - one block: exclusive scan of the SORT_BINS bin totals
//   offsets - one wave per bin: counts[bin][chunk] becomes the first output slot of (bin, chunk)
//   scatter - every wave walks its chunk again in order, 64 slots at a time: lanes with equal digits find each other
//             with one ballot per digit bit, a lane's output slot is the bin's running offset plus the number of equal
//             lanes below it, and the highest of them moves the offset on.  Items of one bin therefore leave in slot
//             order inside a wave, chunks of one bin in chunk order: the pass is stable.
// Nothing is atomic across waves and no order depends on timing: the output is the same on every run.
#include "hvs_internal.h"

#define SORT_BITS 12
#define SORT_BINS (1 << SORT_BITS)
#define SORT_CHUNK 8192  // slots per wave
#define SORT_WAVES 4     // waves per block: 4 x 16 KB of LDS histograms

__global__ __launch_bounds__(64 * SORT_WAVES) void sort_count_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ order_in,
                                                                      int64_t n, int shift, int n_chunks, uint32_t *__restrict__ counts) {
    __shared__ uint32_t hist[SORT_WAVES][SORT_BINS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = (int64_t)blockIdx.x * SORT_WAVES + wave;
    for (int b = lane; b < SORT_BINS; b += 64) hist[wave][b] = 0;
    __syncthreads();
    if (chunk < n_chunks) {
        const int64_t lo = chunk * SORT_CHUNK, hi = lo + SORT_CHUNK < n ? lo + SORT_CHUNK : n;
        for (int64_t s = lo + lane; s < hi; s += 64) {
            const uint32_t item = order_in ? order_in[s] : (uint32_t)s;
            atomicAdd(&hist[wave][(keys[item] >> shift) & (SORT_BINS - 1)], 1u);  // integer, the wave's own counters
        }
    }
    __syncthreads();
    if (chunk < n_chunks)
        for (int b = lane; b < SORT_BINS; b += 64) counts[(size_t)b * n_chunks + chunk] = hist[wave][b];
}

__global__ __launch_bounds__(256) void sort_total_kernel(const uint32_t *__restrict__ counts, int n_chunks, uint32_t *__restrict__ totals) {
    const int lane = threadIdx.x & 63;
    const int bin = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t *row = counts + (size_t)bin * n_chunks;
    uint32_t s = 0;
    for (int c = lane; c < n_chunks; c += 64) s += row[c];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) totals[bin] = s;
}

// one block of 1024 threads, four bins each: starts[b] = items with a smaller digit
__global__ __launch_bounds__(1024) void sort_scan_kernel(const uint32_t *__restrict__ totals, uint32_t *__restrict__ starts) {
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    uint32_t v[4], sum = 0;
    for (int i = 0; i < 4; i++) {
        v[i] = totals[t * 4 + i];
        sum += v[i];
    }
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t x = t >= o ? part[t - o] : 0u;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int i = 0; i < 4; i++) {
        starts[t * 4 + i] = run;
        run += v[i];
    }
}

__global__ __launch_bounds__(256) void sort_offsets_kernel(uint32_t *__restrict__ counts, int n_chunks, const uint32_t *__restrict__ starts) {
    const int lane = threadIdx.x & 63;
    const int bin = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t *row = counts + (size_t)bin * n_chunks;
    uint32_t run = starts[bin];
    for (int c0 = 0; c0 < n_chunks; c0 += 64) {
        const int c = c0 + lane;
        const uint32_t v = c < n_chunks ? row[c] : 0u;
        uint32_t inc = v;  // inclusive scan over the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t x = __shfl_up(inc, off);
            if (lane >= off) inc += x;
        }
        if (c < n_chunks) row[c] = run + inc - v;
        run += __shfl(inc, 63);
    }
}

__global__ __launch_bounds__(64 * SORT_WAVES) void sort_scatter_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ order_in,
                                                                        int64_t n, int shift, int n_chunks, const uint32_t *__restrict__ offsets,
                                                                        uint32_t *__restrict__ order_out) {
    __shared__ uint32_t next[SORT_WAVES][SORT_BINS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = (int64_t)blockIdx.x * SORT_WAVES + wave;
    if (chunk < n_chunks)
        for (int b = lane; b < SORT_BINS; b += 64) next[wave][b] = offsets[(size_t)b * n_chunks + chunk];
    __syncthreads();
    if (chunk >= n_chunks) return;
    volatile uint32_t *mine = next[wave];
    const uint64_t below = (1ull << lane) - 1ull;
    const int64_t lo = chunk * SORT_CHUNK, hi = lo + SORT_CHUNK < n ? lo + SORT_CHUNK : n;
    for (int64_t s0 = lo; s0 < hi; s0 += 64) {
        const int64_t s = s0 + lane;
        const bool valid = s < hi;
        uint32_t item = 0, digit = 0;
        if (valid) {
            item = order_in ? order_in[s] : (uint32_t)s;
            digit = (keys[item] >> shift) & (SORT_BINS - 1);
        }
        uint64_t peers = __ballot(valid);  // lanes holding the same digit as this one
#pragma unroll
        for (int bit = 0; bit < SORT_BITS; bit++) {
            const bool one = (digit >> bit) & 1u;
            const uint64_t m = __ballot(one);
            peers &= one ? m : ~m;
        }
        if (valid) {
            const uint32_t base = mine[digit];
            const uint32_t rank = (uint32_t)__popcll(peers & below);
            order_out[base + rank] = item;
            if ((peers >> lane) == 1ull) mine[digit] = base + (uint32_t)__popcll(peers);  // the highest of its peers
        }
        __builtin_amdgcn_wave_barrier();  // the next group's reads of `mine` stay behind these writes
    }
}

size_t hvs_sort_temp_bytes(int64_t n) {
    const int64_t n_chunks = (n + SORT_CHUNK - 1) / SORT_CHUNK;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return up((size_t)SORT_BINS * (size_t)(n_chunks > 0 ? n_chunks : 1) * 4) + 2 * up((size_t)SORT_BINS * 4) + up((size_t)(n > 0 ? n : 1) * 4);
}

// order_out[0 .. n) = the item numbers 0 .. n-1 ordered by keys[item] (the low key_bits bits), equal keys in ascending
// item number.  keys, order_out, tmp: device memory; tmp holds hvs_sort_temp_bytes(n) bytes; n < 2^32.
int hvs_sort_by_key(hipStream_t st, const uint32_t *keys, int64_t n, int key_bits, uint32_t *order_out, void *tmp) {
    if (n <= 0) return 0;
    HVS_REQUIRE(n < (int64_t)0xFFFFFFFFll, "sort: %lld items, at most 2^32 - 2", (long long)n);
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 32) key_bits = 32;
    const int passes = (key_bits + SORT_BITS - 1) / SORT_BITS;
    const int n_chunks = (int)((n + SORT_CHUNK - 1) / SORT_CHUNK);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    char *w = (char *)tmp;
    uint32_t *counts = (uint32_t *)w;
    uint32_t *totals = (uint32_t *)(w + up((size_t)SORT_BINS * n_chunks * 4));
    uint32_t *starts = (uint32_t *)((char *)totals + up((size_t)SORT_BINS * 4));
    uint32_t *other = (uint32_t *)((char *)starts + up((size_t)SORT_BINS * 4));
    const int blocks = (n_chunks + SORT_WAVES - 1) / SORT_WAVES;
    // ping-pong so that the last pass writes order_out
    const uint32_t *in = nullptr;
    uint32_t *out = (passes & 1) ? order_out : other;
    for (int p = 0; p < passes; p++) {
        const int shift = p * SORT_BITS;
        sort_count_kernel<<<blocks, 64 * SORT_WAVES, 0, st>>>(keys, in, n, shift, n_chunks, counts);
        sort_total_kernel<<<SORT_BINS / 4, 256, 0, st>>>(counts, n_chunks, totals);
        sort_scan_kernel<<<1, 1024, 0, st>>>(totals, starts);
        sort_offsets_kernel<<<SORT_BINS / 4, 256, 0, st>>>(counts, n_chunks, starts);
        sort_scatter_kernel<<<blocks, 64 * SORT_WAVES, 0, st>>>(keys, in, n, shift, n_chunks, counts, out);
        HVS_HIP(hipGetLastError());
        in = out;
        out = out == order_out ? other : order_out;
    }
    return 0;
}
