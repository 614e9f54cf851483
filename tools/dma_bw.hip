// Streaming-read microbenchmark for one MI355X: how fast do 256 blocks of 8 waves pull a large buffer through
//   mode 0: LDS-DMA (buffer_load ... lds, 16 bytes per lane), contiguous 1 KB per wave instruction
//   mode 1: LDS-DMA, 8 rows x 128 bytes per wave instruction at a pitch of `pitch` bytes (the ring kernel's pattern)
//   mode 2: global_load_dwordx4 into registers, contiguous 1 KB per wave instruction (the scan kernel's pattern)
// each with the blocks streaming parts of their own or taking tiles round-robin, and with the loads plain or non-temporal.
// Each wave keeps DEPTH batches of 4 instructions in flight (vmcnt), no barriers.  Prints GB/s.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/dma_bw tools/dma_bw.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

template <int MODE, int DEPTH, bool INTER, int NT>
__global__ __launch_bounds__(512) void stream_kernel(const char *__restrict__ buf, size_t bytes_per_block, int pitch, uint32_t *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // INTER: the blocks take tiles (768 KB) / batches (32 KB) round-robin, so that at any time the grid reads one
    // contiguous window of the buffer; otherwise every block streams a contiguous part of its own
    const char *base = INTER ? buf : buf + (size_t)blockIdx.x * bytes_per_block;
    // a batch = 4 instructions of this wave = 4 KB; the block's 8 waves cover 32 KB per batch
    const size_t batches = bytes_per_block / ((pitch < 0 ? 64 : 32) * 1024);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)hi << 32) | lo), 0, INTER ? 0xFFFFFFFF : (int)bytes_per_block, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    char *my_lds = smem + wave * (DEPTH * 4096);
    // strided pattern: batch b covers rows [8 w + 64 i, +8) of a 256-row tile at byte column (b % cols) * 128, tile b / cols
    const int cols = (pitch < 0 ? -pitch / 2 : pitch) / 128;  // negative pitch: only the first half of every row is read
    pitch = pitch < 0 ? -pitch : pitch;
    auto offset = [&](size_t b, int i) -> uint32_t {
        if (MODE == 0 || MODE == 2) return (uint32_t)((INTER ? b * gridDim.x + blockIdx.x : b) * 32768 + (size_t)(wave * 4 + i) * 1024 + lane * 16);
        const size_t tile = INTER ? (b / cols) * gridDim.x + blockIdx.x : b / cols;
        const int kc = (int)(b % cols);
        const int row = 64 * i + 8 * wave + (lane >> 3);
        return (uint32_t)(tile * 256 * (size_t)pitch + (size_t)row * pitch + kc * 128 + (lane & 7) * 16);
    };
    auto issue = [&](size_t b, int slot, u32x4 (&regs)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t off = offset(b, i);
            if (MODE <= 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(my_lds + slot * 4096 + i * 1024), 16, off, 0, 0, NT);
            else regs[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(base + off)) : *reinterpret_cast<const u32x4 *>(base + off);
        }
    };
    if (MODE <= 1) {
        u32x4 dummy[4];
        for (int d = 0; d < DEPTH - 1 && (size_t)d < batches; d++) issue(d, d, dummy);
        for (size_t b = 0; b < batches; b++) {
            const size_t nb = b + DEPTH - 1 < batches ? b + DEPTH - 1 : batches - 1;
            issue(nb, (int)((b + DEPTH - 1) % DEPTH), dummy);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
            acc.x += *reinterpret_cast<const uint32_t *>(my_lds + (b % DEPTH) * 4096 + lane * 4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // registers: DEPTH batches in flight through a rotating set (unrolled by DEPTH)
        u32x4 regs[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH - 1; d++) issue(d < (int)batches ? d : 0, 0, regs[d]);
        for (size_t b0 = 0; b0 < batches; b0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const size_t b = b0 + d;
                const size_t nb = b + DEPTH - 1 < batches ? b + DEPTH - 1 : batches - 1;
                issue(nb, 0, regs[(d + DEPTH - 1) % DEPTH]);
                if (MODE == 3) {
#pragma unroll
                    for (int i = 0; i < 4; i++) *reinterpret_cast<u32x4 *>(my_lds + (d % 2) * 4096 + i * 1024 + lane * 16) = regs[d][i];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) acc ^= regs[d][i];
                }
            }
        }
        if (MODE == 3) acc.x += *reinterpret_cast<const uint32_t *>(my_lds + lane * 4);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int DEPTH, bool INTER = false, int NT = 0> static void run(const char *buf, size_t total, int pitch, uint32_t *sink, const char *what, int grid = 256) {
    const size_t apitch = pitch < 0 ? -pitch : pitch;
    size_t per_block = total / grid / (256 * apitch) * (256 * apitch);  // whole tiles
    per_block = per_block / 32768 * 32768;
    const size_t lds = 8 * DEPTH * 4096;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_kernel<MODE, DEPTH, INTER, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 6; it++) {
        CHECK(hipEventRecord(e0));
        stream_kernel<MODE, DEPTH, INTER, NT><<<grid, 512, lds>>>(buf, per_block, pitch, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    printf("%-62s %s aux %2d depth %d grid %4d: %7.3f ms  %7.1f GB/s\n", what, INTER ? "round-robin" : "own part   ", NT, DEPTH, grid, best,
           per_block * (double)grid / (pitch < 0 ? 2 : 1) / best / 1e6);
}

int main() {
    const size_t total = (size_t)12 << 30;
    const int pitch = 3072;
    char *buf = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc((void **)&buf, total));
    CHECK(hipMalloc((void **)&sink, 256));
    CHECK(hipMemset(buf, 1, total));
    CHECK(hipMemset(sink, 0, 256));
    run<0, 2>(buf, total, pitch, sink, "LDS-DMA b128, 1 KB contiguous per instruction");
    run<0, 4>(buf, total, pitch, sink, "LDS-DMA b128, 1 KB contiguous per instruction");
    run<1, 2>(buf, total, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)");
    run<1, 4>(buf, total, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)");
    run<2, 2>(buf, total, pitch, sink, "global_load_dwordx4, 1 KB contiguous per instruction");
    run<2, 4>(buf, total, pitch, sink, "global_load_dwordx4, 1 KB contiguous per instruction");
    // 4 GiB so that 32-bit offsets reach every tile of the round-robin order
    const size_t small = ((size_t)4 << 30) - (1 << 20);
    for (int grid : {256, 512}) {
        run<0, 2, false>(buf, small, pitch, sink, "LDS-DMA b128, 1 KB contiguous per instruction", grid);
        run<0, 2, true>(buf, small, pitch, sink, "LDS-DMA b128, 1 KB contiguous per instruction", grid);
        run<1, 2, false>(buf, small, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)", grid);
        run<1, 2, true>(buf, small, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)", grid);
        run<2, 4, false>(buf, small, pitch, sink, "global_load_dwordx4, 1 KB contiguous per instruction", grid);
        run<2, 4, true>(buf, small, pitch, sink, "global_load_dwordx4, 1 KB contiguous per instruction", grid);
    }
    for (int grid : {256, 512, 1024}) {  // the same loads marked non-temporal (aux bit 1 of the LDS-DMA instruction / __builtin_nontemporal_load)
        run<0, 2, false, 2>(buf, total, pitch, sink, "LDS-DMA b128, 1 KB contiguous per instruction", grid);
        run<1, 2, false, 2>(buf, total, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)", grid);
        run<2, 4, false, 2>(buf, total, pitch, sink, "global_load_dwordx4, 1 KB contiguous per instruction", grid);
    }
    // the first 3072 bytes of rows of 6144 (the hi slabs of a split copy), GB/s of the bytes asked for
    run<1, 2, false, 2>(buf, total, -6144, sink, "LDS-DMA b128, 8 rows x 128 B, first half of 6144-byte rows");
    run<1, 2, false, 0>(buf, total, -6144, sink, "LDS-DMA b128, 8 rows x 128 B, first half of 6144-byte rows");
    // other cache-policy bits of the LDS-DMA instruction (bit 0 sc0, bit 1 nt, bit 4 sc1)
    run<1, 2, false, 1>(buf, total, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)");
    run<1, 2, false, 3>(buf, total, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)");
    run<1, 2, false, 16>(buf, total, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)");
    run<1, 2, false, 18>(buf, total, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)");
    run<1, 2, false, 19>(buf, total, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)");
    for (int grid : {512, 768, 1024}) {  // more than one block per CU (depth 2: 64 KB of LDS per block, two fit; registers: more)
        run<0, 2>(buf, total, pitch, sink, "LDS-DMA b128, 1 KB contiguous per instruction", grid);
        run<1, 2>(buf, total, pitch, sink, "LDS-DMA b128, 8 rows x 128 B per instruction (pitch 3072)", grid);
        run<2, 2>(buf, total, pitch, sink, "global_load_dwordx4, 1 KB contiguous per instruction", grid);
        run<2, 4>(buf, total, pitch, sink, "global_load_dwordx4, 1 KB contiguous per instruction", grid);
    }
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
