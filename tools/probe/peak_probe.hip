// Hardware sanity probe: MFMA-only bf16 throughput and HBM copy bandwidth on this box (hipcc -O3 --offload-arch=gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f - threadIdx.x * 0.002f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the exact-fp32 MFMA of the VAE / vocoder kernels: NACC independent accumulators per wave (1 = a dependent chain, as in the 32-channel
// pair kernel), cycles per MFMA from the shader-clock counter and the clock itself from the 100-MHz wall counter
template <int NACC>
__global__ void __launch_bounds__(256) mfma_f32_loop(float* out, long long* clk, int iters) {
    float a = threadIdx.x * 0.001f, b = 0.5f - threadIdx.x * 0.002f;
    f32x16 c[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4 / NACC; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
    const long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int NACC>
static void run_f32(float* out, long long* clk, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    mfma_f32_loop<NACC><<<blocks, 256>>>(out, clk, 100);
    hipEventRecord(e0); mfma_f32_loop<NACC><<<blocks, 256>>>(out, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 2;
    printf("mfma f32 32x32x2, %d accumulators: %4d blocks x 4 waves: %6.1f TFLOP/s (%.2f ms); wave 0: %.1f counter ticks per MFMA, counter %.0f MHz\n",
           NACC, blocks, fl / ms / 1e9, ms, (double)h[0] / (4.0 * iters), h[1] ? 100.0 * h[0] / h[1] : 0.0);
}
__global__ void copy4(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) out[i] = in[i];
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024, 2048}) {
        int iters = 20000;
        mfma_loop<<<blocks, 256>>>(out, 100);
        hipEventRecord(e0); mfma_loop<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("mfma bf16 32x32x16: %4d blocks x 4 waves: %.1f TFLOP/s (%.2f ms)\n", blocks, fl / ms / 1e9, ms);
    }
    long long* clk; hipMalloc(&clk, 16);
    for (int blocks : {256, 512, 768}) { run_f32<4>(out, clk, blocks); run_f32<2>(out, clk, blocks); run_f32<1>(out, clk, blocks); }
    size_t n = (size_t)1 << 28;   // 4 GiB of float4? no: 2^28 float4 = 4 GiB; use 2^26 = 1 GiB
    n = (size_t)1 << 26;
    float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
    copy4<<<2048, 256>>>(a, b, n);
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) copy4<<<2048, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("copy: %.2f TB/s (read+write)\n", 5.0 * 2 * n * 16 / ms / 1e9);
    return 0;
}
