// Which feature of the exact-fp32 convolution loop costs matrix-pipe time?  (hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form
// -I versband_amd/csrc)  The pure v_mfma_f32_32x32x2_f32 loop reaches the nominal rate (peak_probe: 155 TF/s at one to three waves per
// SIMD); conv1d_f32g_kernel with fragment reads, DMA and barriers ablated away still runs at 0.83 of it.  This probe starts from the pure
// loop - a 2 x 2 wave tile, 8 channel pairs per step like the kernel - and adds the loop's features one at a time:
//   F_VALU  1  LeakyReLU on the two B fragments of every pair (v_mul + 2 v_max each), operands change every pair
//   F_LDS   2  the pair's four fragments by inline-asm ds_read_b32 two pairs ahead with exact lgkmcnt waits (lds_asm.h)
//   F_BAR   4  one s_barrier per step (32 MFMAs)
//   F_SALU  8  ~40 scalar instructions of ring bookkeeping per step
//   F_DEFER 16 the step's last pair is multiplied after the next step's barrier and first fragment requests
//   F_BR    32 ~10 data-dependent scalar branches per step (the ring's wait-count selection)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>

#include "lds_asm.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

template <int FEAT, int NP = 8>      // NP channel pairs per step: 8 = 32 MFMAs (the convolution kernel, 32-channel pairs), 4 = 16 (64-channel pairs)
__global__ void __launch_bounds__(256) loop_kernel(float* out, int steps, float slope_in, int salt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 12288; i += 256) lds[i] = (float)((i * 7 + salt) & 255) * 0.01f - 1.f;
    __syncthreads();
    float slope = slope_in;
    asm volatile("v_mov_b32 %0, %0" : "+v"(slope));
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float da[2] = {0.f, 0.f}, db[2] = {0.f, 0.f};
    int ring = salt & 3, ring2 = 1, cnt = 0;
    for (int s = 0; s < steps; ++s) {
        if constexpr (FEAT & 4) __builtin_amdgcn_s_barrier();
        if constexpr (FEAT & 8) {          // ring bookkeeping: a few dependent scalar selects and adds
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                ring = ring == 2 ? 0 : ring + 1;
                ring2 = ring2 == 3 ? 0 : ring2 + 1;
                cnt += (ring == ring2) ? 1 : 0;
                asm volatile("" : "+s"(ring), "+s"(ring2), "+s"(cnt));
            }
        }
        if constexpr (FEAT & 32) {
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                if (((cnt + q) & 3) == (salt & 3)) { asm volatile("s_nop 0" ::: "memory"); cnt += 1; }
                else { asm volatile("s_nop 1" ::: "memory"); cnt += 3; }
                asm volatile("" : "+s"(cnt));
            }
        }
        const unsigned waddr = lds_u32(lds + (ring & 1) * 2048 + (lane & 31) + (lane >> 5) * 128);
        const unsigned xaddr = lds_u32(lds + 6144 + (s & 1) * 3072 + wave * 64 + (lane & 31) + (lane >> 5) * 192);
        float a[3][2], b[3][2];
#pragma unroll
        for (int q = 0; q < 3; ++q) { a[q][0] = a[q][1] = (float)lane * 0.01f; b[q][0] = b[q][1] = 0.5f - (float)lane * 0.02f; }
        auto fload = [&](auto kc) {
            constexpr int KK = decltype(kc)::value, S = KK % 3;
            if constexpr (FEAT & 2) {
                lds_rd32<(2 * KK * 128) * 4>(a[S][0], waddr); lds_rd32<(2 * KK * 128 + 32) * 4>(a[S][1], waddr);
                lds_rd32<(2 * KK * 192) * 4>(b[S][0], xaddr); lds_rd32<(2 * KK * 192 + 32) * 4>(b[S][1], xaddr);
            }
        };
        fload(std::integral_constant<int, 0>{});
        fload(std::integral_constant<int, 1>{});
        if constexpr (FEAT & 16) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[i], db[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        sfor<0, NP>([&](auto kc) {
            constexpr int KK = decltype(kc)::value, S = KK % 3;
            if constexpr (FEAT & 2) {
                if constexpr (KK + 1 < NP) LDS_WAIT(4); else LDS_WAIT(0);
                lds_pin(a[S][0]); lds_pin(a[S][1]); lds_pin(b[S][0]); lds_pin(b[S][1]);
            } else {
                asm volatile("" : "+v"(a[S][0]), "+v"(a[S][1]), "+v"(b[S][0]), "+v"(b[S][1]));       // operands "change" every pair
            }
            if constexpr (KK + 2 < NP) fload(std::integral_constant<int, KK + 2>{});
            float bv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = (FEAT & 1) ? fmaxf(b[S][j], b[S][j] * slope) : b[S][j];
            if constexpr ((FEAT & 16) && KK == NP - 1) {
                da[0] = a[S][0]; da[1] = a[S][1]; db[0] = bv[0]; db[1] = bv[1];
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][i], bv[j], acc[i][j], 0, 0, 0);
            }
        });
    }
    float sum = (float)cnt;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = sum + da[0] + db[0];
}

template <int FEAT, int NP = 8>
static void run(float* out, const char* what, int blocks, int lds_bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)loop_kernel<FEAT, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int steps = 4000 * 8 / NP;
    loop_kernel<FEAT, NP><<<blocks, 256, lds_bytes>>>(out, 50, 0.1f, 3);
    hipEventRecord(e0); loop_kernel<FEAT, NP><<<blocks, 256, lds_bytes>>>(out, steps, 0.1f, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nm = 4.0 * NP;
    const double fl = (double)blocks * 4 * steps * nm * 2.0 * 32 * 32 * 2;
    printf("%-44s %4d blocks (%d KB LDS): %6.1f TFLOP/s\n", what, blocks, lds_bytes / 1024, fl / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    run<0>(out, "(warm-up)", 768, 49152);
    for (int blocks : {768, 512}) {
        const int lb = blocks == 768 ? 49152 : 72 * 1024;       // three / two workgroups per CU
        run<0>(out, "pure MFMA, 2 x 2 tile", blocks, lb);
        run<1>(out, "+ LeakyReLU VALU", blocks, lb);
        run<2>(out, "+ fragment ds_reads", blocks, lb);
        run<3>(out, "+ VALU + ds_reads", blocks, lb);
        run<4>(out, "+ barrier per step", blocks, lb);
        run<8>(out, "+ scalar bookkeeping", blocks, lb);
        run<7>(out, "+ VALU + ds_reads + barrier", blocks, lb);
        run<15>(out, "+ VALU + ds_reads + barrier + scalar", blocks, lb);
        run<31>(out, "... + deferred last pair", blocks, lb);
        run<6>(out, "+ ds_reads + barrier (no VALU)", blocks, lb);
        run<14>(out, "+ ds_reads + barrier + scalar", blocks, lb);
        run<38>(out, "+ ds_reads + barrier + branches", blocks, lb);
        run<46>(out, "+ ds_reads + barrier + scalar + branches", blocks, lb);
        run<22>(out, "+ ds_reads + barrier + deferred pair", blocks, lb);
    }
    printf("-- 16 MFMAs per step (4 channel pairs), three workgroups per CU\n");
    run<6, 4>(out, "ds_reads + barrier", 768, 49152);
    run<14, 4>(out, "+ scalar", 768, 49152);
    run<38, 4>(out, "+ branches", 768, 49152);
    run<46, 4>(out, "+ scalar + branches", 768, 49152);
    run<22, 4>(out, "ds_reads + barrier + deferred pair", 768, 49152);
    run<62, 4>(out, "+ scalar + branches + deferred pair", 768, 49152);
    return 0;
}
