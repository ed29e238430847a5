// GEMM operand-fetch probe (gfx950): the DMA traffic of the 8-wave 256 x 256 GEMM tile with nothing consuming it.
// A workgroup (8 waves) "owns" a 256-row tile of A [M][K] and a 256-row tile of B [N][K] (bf16, K-contiguous, row PITCH in bytes) and
// walks K in stages of KB bytes per row: per stage and operand 32 KB (KB = 128: a DMA instruction covers 8 rows x one full 128-B
// line) or 16 KB (KB = 64: 16 rows x half a line), through a ring of 32-KB slots with DEPTH items in flight (counted vmcnt).
//   pitch   : row pitch of both matrices (1536 = K 768 bf16 packed; 1664 = + one line of padding per row)
//   stagger : 0 = every workgroup starts at k = 0 (the GEMMs' lockstep), 1 = workgroup b starts at stage b mod stages (wraps)
// Reports bytes landed per second per CU.  hipcc -O3 --offload-arch=gfx950 tile_feed_probe.hip -o tile_feed_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14)); }

// KB: bytes per row per stage (128 or 64); DEPTH: items (one operand of one stage) in flight per wave, each 4 (KB = 128) / 2 (KB = 64) instructions
template <int KB, int DEPTH>
__global__ void __launch_bounds__(512) tile_feed_kernel(const char* A, const char* B, int pitch, int krow_bytes, int m_tiles, int n_tiles, int reps,
                                                        int stagger, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int ROWS_PER_INSTR = 1024 / KB;             // 8 or 16
    constexpr int IPW = 256 / ROWS_PER_INSTR / 8;         // instructions per wave per item: 4 or 2
    constexpr int ITEM = 256 * KB;                        // bytes of one item
    constexpr int NSLOT = 163840 / ITEM;                  // 5 or 10
    static_assert(DEPTH < NSLOT, "ring");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = blockIdx.x, jx = L >> 3;
    const int tile_n = jx % n_tiles, tile_m = ((jx / n_tiles) * 8 + (L & 7)) % m_tiles;
    const int stages = krow_bytes / KB;
    const int rr = lane / (KB / 16), cs = lane % (KB / 16);
    const char* ap[IPW]; const char* bp[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int r = ROWS_PER_INSTR * (wave * IPW + i) + rr;
        ap[i] = A + (size_t)(tile_m * 256 + r) * pitch + cs * 16;
        bp[i] = B + (size_t)(tile_n * 256 + r) * pitch + cs * 16;
    }
    const int total = reps * stages * 2;                  // items: A0 B0 A1 B1 ...
    int st = stagger ? (L % stages) : 0;                  // stage of the next item pair
    int slot = 0;
    auto issue = [&](int q) {
        const char* const* pp = (q & 1) ? bp : ap;
#pragma unroll
        for (int i = 0; i < IPW; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pp[i] + st * KB), (lds_ptr_t)(lds + slot * ITEM + (wave * IPW + i) * 1024), 16, 0, 0);
        slot = slot == NSLOT - 1 ? 0 : slot + 1;
        if (q & 1) st = st == stages - 1 ? 0 : st + 1;
    };
    int q = 0;
    for (; q < DEPTH && q < total; ++q) issue(q);
    for (; q < total; ++q) {
        wait_vmcnt<(DEPTH - 1) * IPW>();
        issue(q);
    }
    wait_vmcnt<0>();
    __syncthreads();
    unsigned v = (unsigned)lds[(threadIdx.x * 16) & 163839];
    if (v == 0x12345678u) sink[0] = v;
}

template <int KB, int DEPTH>
static void run(const char* name, const char* A, const char* B, unsigned* sink, int pitch, int krow_bytes, int M, int N, int stagger) {
    const int reps = 64;
    const int m_tiles = M / 256, n_tiles = N / 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(tile_feed_kernel<KB, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((tile_feed_kernel<KB, DEPTH>), dim3(256), dim3(512), 163840, 0, A, B, pitch, krow_bytes, m_tiles, n_tiles, 2, stagger, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((tile_feed_kernel<KB, DEPTH>), dim3(256), dim3(512), 163840, 0, A, B, pitch, krow_bytes, m_tiles, n_tiles, reps, stagger, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * reps * (krow_bytes / KB) * 2.0 * 256.0 * KB;
    printf("%-10s row bytes %4d pitch %4d stage %3d B/row depth %d items (%3d KB in flight) stagger %d: %6.1f GB/s per CU (%5.2f TB/s chip, %.3f ms)%s\n",
           name, krow_bytes, pitch, KB, DEPTH, DEPTH * 256 * KB / 1024, stagger, bytes / ms / 1e6 / 256, bytes / ms / 1e9, ms,
           hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
    const int M = 12032 / 256 * 256, N = 2304;
    char *A, *B; unsigned* sink;
    const size_t maxpitch = 2048;
    hipMalloc(&A, (size_t)(M + 256) * maxpitch); hipMemset(A, 1, (size_t)(M + 256) * maxpitch);
    hipMalloc(&B, (size_t)(N + 256) * maxpitch); hipMemset(B, 1, (size_t)(N + 256) * maxpitch);
    hipMalloc(&sink, 64);
    for (int rep = 0; rep < 2; ++rep) {
        // K = 768 bf16 (QKV, out-proj, SwiGLU)
        run<128, 3>("full-line", A, B, sink, 1536, 1536, M, N, 0);
        run<128, 4>("full-line", A, B, sink, 1536, 1536, M, N, 0);
        run<128, 4>("full-line", A, B, sink, 1536, 1536, M, N, 1);
        run<128, 4>("full-line", A, B, sink, 1664, 1536, M, N, 0);
        run<128, 4>("full-line", A, B, sink, 1664, 1536, M, N, 1);
        run<128, 4>("full-line", A, B, sink, 1600, 1536, M, N, 0);
        run<128, 4>("full-line", A, B, sink, 2048, 1536, M, N, 0);
        run<128, 2>("full-line", A, B, sink, 1536, 1536, M, N, 0);
        run<128, 2>("full-line", A, B, sink, 1664, 1536, M, N, 0);
        run<64, 8>("half-line", A, B, sink, 1536, 1536, M, N, 0);
        run<64, 8>("half-line", A, B, sink, 1664, 1536, M, N, 0);
        run<64, 8>("half-line", A, B, sink, 1536, 1536, M, N, 1);
        // K = 512 bf16 (routed w2)
        run<128, 4>("full-line", A, B, sink, 1024, 1024, M, 768, 0);
        run<128, 4>("full-line", A, B, sink, 1152, 1024, M, 768, 0);
        run<128, 4>("full-line", A, B, sink, 1024, 1024, M, 768, 1);
    }
    return 0;
}
