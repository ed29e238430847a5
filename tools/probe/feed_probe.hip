// L2 -> CU feed probe (gfx950): how many bytes per second one CU can pull from L2-resident data
//   dma   : global_load_lds (16 B per lane, 1 KB per instruction, straight into LDS), D pieces in flight per wave
//   vgpr  : global_load_dwordx4 into registers, lane-linear (1 KB per instruction), U loads in flight per wave
//   vrow  : the same into registers with the GEMM A-fragment pattern (32 rows x 32 B per instruction, row pitch 1536 B)
//   mixed : waves 0-1 run dma, waves 2-3 run vgpr  (are the two return paths additive?)
// Every workgroup of an XCD walks the same 1 MB window (8 windows = 8 MB in total: L2 hits after the first touch).
// hipcc -O3 --offload-arch=gfx950 feed_probe.hip -o feed_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14)); }

#define WINDOW (1u << 20)
// FAR = true: every workgroup walks its OWN 1 MB window (256 MB in all: L2 misses, served by the MALL / HBM)

template <int D>
__device__ __forceinline__ void dma_loop(const char* win, char* ring, int wave_slot, int lane, int iters, unsigned start) {
    // ring: D pieces of 1 KB owned by this wave; piece addresses walk the window with a per-wave phase
    unsigned off = start;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(win + off + lane * 16), (lds_ptr_t)(ring + d * 1024), 16, 0, 0);
        off = (off + 4096) & (WINDOW - 1);
    }
    for (int it = 0; it < iters; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            wait_vmcnt<D - 1>();
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(win + off + lane * 16), (lds_ptr_t)(ring + d * 1024), 16, 0, 0);
            off = (off + 4096) & (WINDOW - 1);
        }
    }
    wait_vmcnt<0>();
}

template <int U, bool ROWS>
__device__ __forceinline__ uint4 vgpr_loop(const char* win, int lane, int iters, unsigned start) {
    // rotating register ring, U loads in flight: use the oldest, refill its slot (loads return in order)
    // ROWS: lane (r = lane & 31, h = lane >> 5) reads 16 B at row r (pitch 1536 B), column byte h*16 + (piece & 3) * 32
    const unsigned lane_off = ROWS ? (unsigned)((lane & 31) * 1536 + (lane >> 5) * 16) : (unsigned)(lane * 16);
    const unsigned step = ROWS ? 32u : 4096u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 r[U];
    unsigned off = start & ~1023u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        r[u] = *reinterpret_cast<const uint4*>(win + ((off + lane_off) & (WINDOW - 1)));
        off += step;
    }
    for (int it = 0; it < iters; it += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc.x ^= r[u].x; acc.y ^= r[u].y; acc.z ^= r[u].z; acc.w ^= r[u].w;
            r[u] = *reinterpret_cast<const uint4*>(win + ((off + lane_off) & (WINDOW - 1)));
            off += step;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x ^= r[u].x; acc.y ^= r[u].y; acc.z ^= r[u].z; acc.w ^= r[u].w; }
    return acc;
}

// MODE 0 dma, 1 vgpr, 2 vrow, 3 mixed (waves 0-1 dma, 2-3 vgpr); NW = active waves (2 or 4); D = depth per wave
template <int MODE, int D, int NW, bool FAR = false>
__global__ void __launch_bounds__(256) feed_kernel(const char* src, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* win = src + (size_t)(FAR ? (blockIdx.x & 255) : (blockIdx.x & 7)) * WINDOW;
    const unsigned start = ((blockIdx.x >> 3) * 16384u + wave * 1024u) & (WINDOW - 1);
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (wave < NW) {
        if (MODE == 0 || (MODE == 3 && wave < 2)) dma_loop<D>(win, lds + wave * D * 1024, wave, lane, iters, start);
        else if (MODE == 2) acc = vgpr_loop<D, true>(win, lane, iters, start);
        else acc = vgpr_loop<D, false>(win, lane, iters, start);
    }
    __syncthreads();
    unsigned v = acc.x ^ acc.y ^ acc.z ^ acc.w ^ (unsigned)lds[(threadIdx.x * 16) & (NW * D * 1024 - 1)];
    if (v == 0x12345678u) sink[0] = v;
}

template <int MODE, int D, int NW, bool FAR = false>
static void run(const char* name, const char* src, unsigned* sink, int blocks) {
    const int iters = 16384;       // 1-KB pieces (or 1-KB load instructions) per active wave
    const size_t lds = (size_t)4 * D * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(feed_kernel<MODE, D, NW, FAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((feed_kernel<MODE, D, NW, FAR>), dim3(blocks), dim3(256), lds, 0, src, 256, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((feed_kernel<MODE, D, NW, FAR>), dim3(blocks), dim3(256), lds, 0, src, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * NW * iters * 1024.0;
    const int cus = blocks < 256 ? blocks : 256;
    printf("%s%-6s depth %2d KB/wave x %d waves, %4d blocks: %7.1f GB/s per CU  (%.2f TB/s chip, %.3f ms)%s\n", FAR ? "far-" : "", name, D, NW, blocks,
           bytes / ms / 1e6 / cus, bytes / ms / 1e9, ms, hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
    char* src; unsigned* sink;
    hipMalloc(&src, (size_t)256 * WINDOW); hipMemset(src, 1, (size_t)256 * WINDOW); hipMalloc(&sink, 64);
    for (int blocks : {256, 512}) {
        run<0, 4, 4>("dma", src, sink, blocks);
        run<0, 8, 4>("dma", src, sink, blocks);
        run<0, 16, 4>("dma", src, sink, blocks);
        run<0, 32, 4>("dma", src, sink, blocks);
        run<0, 16, 2>("dma", src, sink, blocks);
        run<1, 8, 4>("vgpr", src, sink, blocks);
        run<1, 16, 4>("vgpr", src, sink, blocks);
        run<1, 16, 2>("vgpr", src, sink, blocks);
        run<2, 16, 4>("vrow", src, sink, blocks);
        run<3, 16, 4>("mixed", src, sink, blocks);
    }
    for (int rep = 0; rep < 2; ++rep) {        // rep 1: the windows were just streamed once (what the MALL keeps of 256 MB)
        run<0, 4, 4, true>("dma", src, sink, 256);
        run<0, 12, 4, true>("dma", src, sink, 256);
        run<0, 24, 4, true>("dma", src, sink, 256);
        run<0, 32, 4, true>("dma", src, sink, 256);
        run<1, 16, 4, true>("vgpr", src, sink, 256);
    }
    return 0;
}
