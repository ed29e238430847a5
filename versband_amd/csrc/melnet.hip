// Log-mel front-end (gfx950) - MelNet.forward of the reference (preprocess/NAT_mel.py:42-86), SURVEY 8(f) N4.
//
//   y = clamp(wav, -1, 1); reflect-pad (n_fft - hop)/2 on both sides (+ n_fft/2 when center); STFT (window folded into the basis);
//   mag = sqrt(re^2 + im^2 + 1e-9); mel = basis @ mag; out = log10(max(mel, 1e-5)).
//
// n_fft is a multiple of the hop (1280 = 4 x 320), so frame t is the concatenation of hop-blocks t .. t+taps-1 of the padded
// signal and the windowed DFT of all frames is ONE convolution over hop-blocks: input X[c][j] = padded[j*hop + c] (hop "channels",
// T + taps - 1 blocks), taps = n_fft / hop, weights W[q][c][o] = window[q*hop + c] * {cos, -sin}(2 pi f (q*hop + c) / n_fft).
// It runs on the exact-fp32 MFMA convolution kernel (conv1d_f32.hip; 2*1282*1280 flop per frame = 4.9 GFLOP per 20 s clip - a
// few tens of microseconds of v_mfma_f32_32x32x2f32), output transposed [b][t][Co4] so that the tail reads a frame's spectrum
// contiguously.  Two small kernels sit either side: frames_kernel (clamp + reflect + de-interleave into hop channels through an
// LDS tile so that both the read and the write are coalesced) and mel_tail_kernel (magnitude into LDS, mel filterbank, log10).
#include "kernels.h"

// X[b][c][j] = clamp(wav[b][r_L(r_L1(j*hop + c - pad2) - pad)]),  c < hop, j < J: the reference reflect-pads by `pad` (NAT_mel.py:71) and
// torch.stft(center=True) reflect-pads THAT signal (length L1 = L + 2 pad) by pad2 = n_fft/2 again - two reflections, not one of pad + pad2
#define FR_JT 16
__global__ void __launch_bounds__(256) frames_kernel(const float* __restrict__ wav, int L, int hop, int pad, int pad2, int J, float* __restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [FR_JT][hop + 1]
    const int b = blockIdx.y, j0 = blockIdx.x * FR_JT;
    const int pitch = hop + 1;
    const float* w = wav + (int64_t)b * L;
    for (int id = threadIdx.x; id < FR_JT * hop; id += 256) {
        const int jj = id / hop, c = id - jj * hop;
        const int j = j0 + jj;
        float v = 0.f;
        if (j < J) {
            const int L1 = L + 2 * pad;
            int k = j * hop + c - pad2;
            if (k < 0) k = -k;
            if (k >= L1) k = 2 * (L1 - 1) - k;
            k -= pad;
            if (k < 0) k = -k;
            if (k >= L) k = 2 * (L - 1) - k;
            k = k < 0 ? 0 : (k >= L ? L - 1 : k);       // only reachable when a padding exceeds its signal, which the launcher rejects
            v = fminf(fmaxf(w[k], -1.f), 1.f);
        }
        tile[jj * pitch + c] = v;
    }
    __syncthreads();
    float* xb = X + (int64_t)b * hop * J;
    for (int id = threadIdx.x; id < FR_JT * hop; id += 256) {
        const int c = id / FR_JT, jj = id - c * FR_JT;
        const int j = j0 + jj;
        if (j < J) xb[(int64_t)c * J + j] = tile[jj * pitch + c];
    }
}
int launch_stft_frames(const float* wav, int B, int L, int hop, int pad, int pad2, int J, float* X, hipStream_t st) {
    if (pad >= L || pad2 >= L + 2 * pad) VB_FAIL(VB_E_INVALID, "melnet: reflect padding %d (+ %d) needs more than %d samples", pad, pad2, L);
    const size_t lds = (size_t)FR_JT * (hop + 1) * sizeof(float);
    if (lds > 64 * 1024) VB_FAIL(VB_E_INVALID, "melnet: hop %d too large", hop);
    hipLaunchKernelGGL(frames_kernel, dim3(cdiv(J, FR_JT), B), dim3(256), lds, st, wav, L, hop, pad, pad2, J, X);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// spec f32 [B][T][Co4] (re at [0, nb), im at [im_off, im_off + nb)); basisT f32 [nb][n_mels]; mel f32 [B][n_mels][T]
// block = MT_FT frames of one clip; threads = n_mels x 4 frame groups (4 frames each)
#define MT_FT 16
__global__ void mel_tail_kernel(const float* __restrict__ spec, int T, int Co4, int nb, int im_off, const float* __restrict__ basisT,
                                int n_mels, float* __restrict__ mel) {
    extern __shared__ __attribute__((aligned(16))) float mag[];      // [nb][MT_FT]
    const int b = blockIdx.y, t0 = blockIdx.x * MT_FT;
    const float* sb = spec + ((int64_t)b * T + t0) * Co4;
    for (int id = threadIdx.x; id < MT_FT * nb; id += blockDim.x) {
        const int tt = id / nb, f = id - tt * nb;
        float v = 0.f;
        if (t0 + tt < T) {
            const float re = sb[(int64_t)tt * Co4 + f], im = sb[(int64_t)tt * Co4 + im_off + f];
            v = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)), 1e-9f));     // spec.pow(2).sum(-1) + 1e-9, :76
        }
        mag[f * MT_FT + tt] = v;
    }
    __syncthreads();
    const int m = threadIdx.x % n_mels, fg = threadIdx.x / n_mels;      // blockDim = 4 * n_mels
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int f = 0; f < nb; ++f) {
        const float w = basisT[(int64_t)f * n_mels + m];
        const float4 g = *reinterpret_cast<const float4*>(&mag[f * MT_FT + fg * 4]);
        a0 = fmaf(w, g.x, a0); a1 = fmaf(w, g.y, a1); a2 = fmaf(w, g.z, a2); a3 = fmaf(w, g.w, a3);
    }
    const float acc[4] = {a0, a1, a2, a3};
    float* mb = mel + ((int64_t)b * n_mels + m) * T;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + fg * 4 + i;
        if (t < T) mb[t] = log10f(fmaxf(acc[i], 1e-5f));      // dynamic_range_compression_torch, :26-27
    }
}
int launch_mel_tail(const float* spec, int B, int T, int Co4, int nb, int im_off, const float* basisT, int n_mels, float* mel, hipStream_t st) {
    if (n_mels < 1 || n_mels * 4 > 1024) VB_FAIL(VB_E_INVALID, "melnet: n_mels %d (1..256)", n_mels);
    const size_t lds = (size_t)nb * MT_FT * sizeof(float);
    if (lds > 64 * 1024) VB_FAIL(VB_E_INVALID, "melnet: %d frequency bins do not fit the magnitude tile", nb);
    hipLaunchKernelGGL(mel_tail_kernel, dim3(cdiv(T, MT_FT), B), dim3(4 * n_mels), lds, st, spec, T, Co4, nb, im_off, basisT, n_mels, mel);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
