// fp32 implicit-GEMM Conv1d on the f32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32) for the
// VAE decoder, the HiFi-GAN generator and the DiT stem (gfx950).
//
//   out[b][co][t] = beta*out + alpha*( act_out( acc_scale*sum_{ci,j} W[j][ci][co] * act_in(x[b][ci][t + j*dil - pad]) )
//                                      + bias[co] + res[b][co][t] )
//
//  * activations are [B][C][T] (T contiguous); weights are pre-packed [phase][tap][Ci][Co]
//    (Co contiguous) so both MFMA operands are read from LDS with unit lane stride
//    (A = W[ci][co..co+31], B = x[ci][t..t+31]) - conflict free without swizzles.
//  * per 16-channel chunk the input window (tile + (k-1)*dil halo) is staged ONCE and
//    re-used by all k taps; the pointwise input transform (LeakyReLU, or GroupNorm+swish
//    from precomputed statistics), zero padding and nearest x2 upsampling are applied
//    while staging, so no activated/upsampled tensor ever goes to HBM.
//  * ConvTranspose1d runs as `stride` polyphase sub-convolutions in one launch
//    (grid.z = batch x phase), each with ceil(k/stride) taps.
//  * per-batch "weights" (w_bstride) let the VAE's single-head attention (q^T k and P v)
//    run on the same kernel.
#include <stdlib.h>

#include "kernels.h"

#define CK 16
#define XHALO 64

#include "conv1d_dev.h"

template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(256) conv1d_f32_kernel(const ConvDev p) {
    constexpr int CO_TILE = WM * TM * 32;
    constexpr int T_TILE = WN * TN * 32;
    constexpr int XW = T_TILE + XHALO;
    __shared__ float xw[CK * XW];
    __shared__ float wl[2][CK * CO_TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int z = blockIdx.z;
    const int b = z / p.phases, ph = z - b * p.phases;
    const int n0 = blockIdx.x * T_TILE;
    const int co0 = blockIdx.y * CO_TILE;

    // polyphase geometry (phases == 1 -> in_off = -pad, out index = n)
    int in_off, out_off, out_stride, n_count;
    if (p.phases == 1) {
        in_off = -p.pad; out_off = 0; out_stride = 1; n_count = p.T_out;
    } else {
        const int u = p.phases;
        const int d = p.tr_pad - ph;
        const int q0 = d > 0 ? (d + u - 1) / u : 0;
        in_off = q0 - (p.ntaps - 1);
        out_off = q0 * u + ph - p.tr_pad;
        out_stride = u;
        n_count = (p.T_out - out_off + u - 1) / u;
    }
    if (n0 >= n_count) return;

    const int halo = (p.ntaps - 1) * p.dil;
    const int xw_used = T_TILE + halo;
    const int T_eff = p.upsample2 ? 2 * p.T_in : (p.T_in - p.in_phase + p.in_stride - 1) / p.in_stride;
    const int xb = p.x_bmod > 0 ? (b % p.x_bmod) : b;
    const float* xbase = p.x + (int64_t)xb * p.x_bstride;
    const float* wbase = p.w + (int64_t)b * p.w_bstride + (int64_t)ph * p.ntaps * p.Ci * p.Co;
    const int cpg = p.gn_groups > 0 ? (p.Ci / p.gn_groups) : 1;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int WPT = CK * CO_TILE / 256;   // weight elements per thread per tap tile
    float wreg[WPT];
    auto wload = [&](int c0, int j) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            int id = tid + i * 256;
            int ci = id / CO_TILE, col = id - ci * CO_TILE;
            int cig = c0 + ci, cog = co0 + col;
            wreg[i] = (cig < p.Ci && cog < p.Co) ? wbase[((int64_t)j * p.Ci + cig) * p.Co + cog] : 0.f;
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) wl[buf][tid + i * 256] = wreg[i];
    };

    const int nchunks = (p.Ci + CK - 1) / CK;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * CK;
        // ---- stage the activated input window of this channel chunk
        for (int ci = wave; ci < CK; ci += 4) {
            const int cig = c0 + ci;
            const bool cok = cig < p.Ci;
            float gm = 0.f, gr = 1.f, gg = 1.f, gb = 0.f;
            if (cok && (p.in_act == ACT_GN_SWISH || p.in_act == ACT_GN)) {
                const int grp = cig / cpg;
                gm = p.gn_mean[b * p.gn_groups + grp]; gr = p.gn_rstd[b * p.gn_groups + grp];
                gg = p.gn_gamma[cig]; gb = p.gn_beta[cig];
            }
            const float* xrow = xbase + (int64_t)(cok ? cig : 0) * p.T_in;
            for (int wpos = lane; wpos < xw_used; wpos += 64) {
                const int idx = n0 + in_off + wpos;
                float v = 0.f;
                if (cok && idx >= 0 && idx < T_eff) {
                    v = xrow[p.upsample2 ? (idx >> 1) : idx * p.in_stride + p.in_phase];
                    if (p.in_act == ACT_LRELU) {
                        v = v > 0.f ? v : v * p.in_slope;
                    } else if (p.in_act == ACT_GN_SWISH || p.in_act == ACT_GN) {
                        v = (v - gm) * gr * gg + gb;
                        if (p.in_act == ACT_GN_SWISH) v = v / (1.f + __expf(-v));
                    }
                }
                xw[ci * XW + wpos] = v;
            }
        }
        wload(c0, 0);
        wstore(0);
        __syncthreads();
        for (int j = 0; j < p.ntaps; ++j) {
            const int buf = j & 1;
            if (j + 1 < p.ntaps) wload(c0, j + 1);
            const int xoff = j * p.dil;
#pragma unroll
            for (int kk = 0; kk < CK / 2; ++kk) {
                const int ci = 2 * kk + g;
                float a[TM], bb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = wl[buf][ci * CO_TILE + (wm * TM + i) * 32 + l31];
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) bb[jn] = xw[ci * XW + (wn * TN + jn) * 32 + l31 + xoff];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[jn], acc[i][jn], 0, 0, 0);
            }
            if (j + 1 < p.ntaps) wstore(buf ^ 1);
            __syncthreads();
        }
    }

    conv_epilogue<WM, WN, TM, TN>(p, acc, b, n0, co0, n_count, out_stride, out_off);
}

template <int WM, int WN, int TM, int TN>
static void launch_cfg(const ConvDev& d, int n_count, int B, hipStream_t st) {
    dim3 grid(cdiv(n_count, WN * TN * 32), cdiv(d.Co, WM * TM * 32), B * d.phases);
    hipLaunchKernelGGL((conv1d_f32_kernel<WM, WN, TM, TN>), grid, dim3(256), 0, st, d);
}


// ---------------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant: same implicit GEMM, same fused staging and epilogue, but every fp32 operand is
// split on the fly into a bf16 hi/lo pair and each product runs as hi*hi + lo*hi + hi*lo on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Operand error 2^-17 (fp32-class results, ~3e-5 max relative vs
// the exact-f32 kernel) at 3 bf16 MFMAs per 16-deep k-step: 5.3x the f32-MFMA rate of gfx950 (157 TF vs 2.5 PF/3).
// The activation window is transposed while staging: LDS holds xT[plane][t][ci] (ci contiguous, pitch 80 B so the
// 16-B fragment reads of 16 consecutive t hit 16 distinct slots); weights are pre-packed [plane][tap][co][ci].
// ---------------------------------------------------------------------------------------------------------
#define CK3 32
#define CKP3 40      // bf16 elements per LDS row (32 + 8 pad)

// ABL (tuning only): 1 = window staged once, 2 = weights staged once, 3 = no MFMA, 4 = no epilogue
// ABL == 5 is not an ablation but the XT input mode: the window comes from pre-activated, transposed split planes (xt_planes_kernel)
// by DMA (global_load_lds) - no register staging, no per-tile transform/split; LDS rows are 64 B, XOR-swizzled instead of padded.
template <int WM, int WN, int TM, int TN, int ABL = 0>
__global__ void __launch_bounds__(256) conv1d_x3_kernel(const ConvDev p) {
    constexpr int CO_TILE = WM * TM * 32;
    constexpr int T_TILE = WN * TN * 32;
    constexpr int XW = T_TILE + XHALO;
    __shared__ __attribute__((aligned(16))) bf16_t xT[2][XW * CKP3];
    __shared__ __attribute__((aligned(16))) bf16_t wl[2][2][CO_TILE * CKP3];     // [buf][plane]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int z = blockIdx.z;
    const int b = z / p.phases, ph = z - b * p.phases;
    const int n0 = blockIdx.x * T_TILE;
    const int co0 = blockIdx.y * CO_TILE;

    int in_off, out_off, out_stride, n_count;
    if (p.phases == 1) {
        in_off = -p.pad; out_off = 0; out_stride = 1; n_count = p.T_out;
    } else {
        const int u = p.phases;
        const int d = p.tr_pad - ph;
        const int q0 = d > 0 ? (d + u - 1) / u : 0;
        in_off = q0 - (p.ntaps - 1);
        out_off = q0 * u + ph - p.tr_pad;
        out_stride = u;
        n_count = (p.T_out - out_off + u - 1) / u;
    }
    if (n0 >= n_count) return;

    const int halo = (p.ntaps - 1) * p.dil;
    const int xw_used = T_TILE + halo;
    const int T_eff = p.upsample2 ? 2 * p.T_in : (p.T_in - p.in_phase + p.in_stride - 1) / p.in_stride;
    const int xb = p.x_bmod > 0 ? (b % p.x_bmod) : b;
    const float* xbase = p.x + (int64_t)xb * p.x_bstride;
    const bf16_t* wbase = p.wp + (int64_t)b * p.wp_bstride + (int64_t)ph * p.ntaps * p.Co * p.Ci_pad;
    const int cpg = p.gn_groups > 0 ? (p.Ci / p.gn_groups) : 1;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // weight tile of one (tap, chunk): CO_TILE rows x 32 ci x 2 planes = CO_TILE*8 pieces of 16 B
    constexpr int WPT = CO_TILE * 8 / 256;
    uint4 wreg[WPT];
    auto wload = [&](int c0, int j) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int id = tid + i * 256;
            const int pl = id / (CO_TILE * 4), rem = id - pl * (CO_TILE * 4);
            const int co = rem >> 2, pc = rem & 3;
            const int cog = co0 + co;
            wreg[i] = (cog < p.Co) ? *reinterpret_cast<const uint4*>(wbase + pl * p.wp_plane + ((int64_t)j * p.Co + cog) * p.Ci_pad + c0 + pc * 8)
                                   : make_uint4(0, 0, 0, 0);
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int id = tid + i * 256;
            const int pl = id / (CO_TILE * 4), rem = id - pl * (CO_TILE * 4);
            const int co = rem >> 2, pc = rem & 3;
            *reinterpret_cast<uint4*>(&wl[buf][pl][co * CKP3 + pc * 8]) = wreg[i];
        }
    };

    // ---- activation window staging, split in two halves (async-STAGE): xload() issues ALL global loads of a chunk
    // into registers (raw values + the per-channel affine of the fused norm), xstore() later applies the pointwise
    // transform, splits to bf16 hi/lo and writes the transposed image xT[plane][t][ci].  The loads of chunk ch+1 are
    // in flight while the taps of chunk ch are multiplied.  A wave owns 4 channel pairs of the 32-channel chunk.
    // A wave owns 8 CONSECUTIVE channels of the 32-channel chunk and a lane one window position per pass: the global loads stay
    // coalesced along t (one channel row per instruction) and the transposed image takes ONE 16-byte LDS write per plane and
    // position (round 3; four 4-byte writes of channel pairs before - 4-way bank-conflicted at the 80-byte row pitch).
    constexpr int NIT = XW / 64;
    float raw[8][NIT];
    float nsc[8], nsh[8];
    auto xload = [&](int c0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = c0 + 8 * wave + e;
            const bool cok = ci < p.Ci;
            nsc[e] = 1.f; nsh[e] = 0.f;
            if (cok && (p.in_act == ACT_GN_SWISH || p.in_act == ACT_GN)) {
                const int grp = ci / cpg;
                const float rs = p.gn_rstd[b * p.gn_groups + grp] * p.gn_gamma[ci];
                nsc[e] = rs;
                nsh[e] = p.gn_beta[ci] - p.gn_mean[b * p.gn_groups + grp] * rs;
            }
            const float* xrow = xbase + (int64_t)(cok ? ci : 0) * p.T_in;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = n0 + in_off + lane + 64 * it;
                const bool ok = cok && (lane + 64 * it) < xw_used && idx >= 0 && idx < T_eff;
                raw[e][it] = ok ? xrow[p.upsample2 ? (idx >> 1) : idx * p.in_stride + p.in_phase] : 0.f;
            }
        }
    };
    auto xstore = [&](int c0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int wpos = lane + 64 * it;
            if (wpos >= xw_used) continue;
            const int idx = n0 + in_off + wpos;
            const bool inr = idx >= 0 && idx < T_eff;
            bf16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = raw[e][it];
                if (inr && (c0 + 8 * wave + e) < p.Ci) {         // zero padding stays zero: the conv pads the ACTIVATED tensor
                    if (p.in_act == ACT_LRELU) {
                        t = t > 0.f ? t : t * p.in_slope;
                    } else if (p.in_act == ACT_GN_SWISH || p.in_act == ACT_GN) {
                        t = t * nsc[e] + nsh[e];
                        if (p.in_act == ACT_GN_SWISH) t = t / (1.f + __expf(-t));
                    }
                } else {
                    t = 0.f;
                }
                hi[e] = f2bf(t);
                lo[e] = f2bf(t - bf2f(hi[e]));
            }
            *reinterpret_cast<bf16x8*>(&xT[0][wpos * CKP3 + 8 * wave]) = hi;
            *reinterpret_cast<bf16x8*>(&xT[1][wpos * CKP3 + 8 * wave]) = lo;
        }
    };

    const int nchunks = (p.Ci + CK3 - 1) / CK3;
    // XT mode: DMA of one chunk's window = xw_used rows x 64 B per plane, in 1-KB pieces of 16 rows; lane -> (row, 16-B slot),
    // the slot holds source chunk slot ^ ((row >> 2) & 3)
    auto xt_issue = [&](int c0) {
        typedef __attribute__((address_space(3))) void* lds_p;
        typedef const __attribute__((address_space(1))) void* glb_p;
        const int P = (xw_used + 15) >> 4;
        for (int q = wave; q < 2 * P; q += 4) {
            const int pl = q >= P, pr = q - pl * P;
            const int row = pr * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            const bf16_t* src = p.xt + pl * p.xt_plane + ((int64_t)xb * p.xt_Tp + (n0 + in_off + XT_HEAD + row)) * p.Ci + c0 + c * 8;
            __builtin_amdgcn_global_load_lds((glb_p)src, (lds_p)(&xT[pl][pr * 16 * 32]), 16, 0, 0);
        }
    };
    if constexpr (ABL != 5) xload(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * CK3;
        if constexpr (ABL == 5) {
            xt_issue(c0);                       // every wave is past the previous chunk's last tap (barrier below)
            wload(c0, 0); wstore(0);
            __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0): the window landed
        } else {
        if (ABL != 1 || ch == 0) xstore(c0);
        if (ABL != 2 || ch == 0) { wload(c0, 0); wstore(0); }
        }
        __syncthreads();
        if (ABL != 1 && ABL != 5 && ch + 1 < nchunks) xload(c0 + CK3);
        for (int j = 0; j < p.ntaps; ++j) {
            const int buf = (ABL == 2) ? 0 : (j & 1);
            if (ABL != 2 && j + 1 < p.ntaps) wload(c0, j + 1);
            const int xoff = j * p.dil;
#pragma unroll
            for (int ks = 0; ks < CK3 / 16; ++ks) {
                const int kofs = ks * 16 + g * 8;
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int o = ((wm * TM + i) * 32 + l31) * CKP3 + kofs;
                    ah[i] = *reinterpret_cast<const bf16x8*>(&wl[buf][0][o]);
                    al[i] = *reinterpret_cast<const bf16x8*>(&wl[buf][1][o]);
                }
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    const int row = (wn * TN + jn) * 32 + l31 + xoff;
                    const int o = (ABL == 5) ? row * 32 + (((kofs >> 3) ^ ((row >> 2) & 3)) << 3) : row * CKP3 + kofs;
                    bh[jn] = *reinterpret_cast<const bf16x8*>(&xT[0][o]);
                    bl[jn] = *reinterpret_cast<const bf16x8*>(&xT[1][o]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
                        if constexpr (ABL == 3) {
                            acc[i][jn][0] += (float)ah[i][0] * (float)bh[jn][0] + (float)al[i][1] * (float)bl[jn][1];
                        } else {
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[jn], acc[i][jn], 0, 0, 0);
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[jn], acc[i][jn], 0, 0, 0);
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[jn], acc[i][jn], 0, 0, 0);
                        }
                    }
            }
            if (ABL != 2 && j + 1 < p.ntaps) wstore(buf ^ 1);
            __syncthreads();
        }
    }
    if constexpr (ABL == 4) {
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) sink += acc[i][jn][0] + acc[i][jn][7];
        if (sink == 12345.678f) p.out[0] = sink;
    } else {
        // (the loop's last __syncthreads() is behind every wave: xT is free and holds the four wave-private staging patches)
        static_assert(sizeof(xT) >= 4 * 32 * CE_PITCH * sizeof(float), "staging patches must fit in the window buffer");
        if (p.stage_epi) conv_epilogue_staged<WM, WN, TM, TN>(p, acc, b, n0, co0, n_count, reinterpret_cast<float*>(&xT[0][0]));
        else conv_epilogue<WM, WN, TM, TN>(p, acc, b, n0, co0, n_count, out_stride, out_off);
    }
}

template <int WM, int WN, int TM, int TN>
static void launch_cfg_x3(const ConvDev& d, int n_count, int B, hipStream_t st) {
    dim3 grid(cdiv(n_count, WN * TN * 32), cdiv(d.Co, WM * TM * 32), B * d.phases);
#ifdef VB_EXPERIMENTS      // ablation instances exist in the experiments build only (tools/conv_bench.py)
    const int abl = vb_tune().conv_ablate;
    if (abl == 1) hipLaunchKernelGGL((conv1d_x3_kernel<WM, WN, TM, TN, 1>), grid, dim3(256), 0, st, d);
    else if (abl == 2) hipLaunchKernelGGL((conv1d_x3_kernel<WM, WN, TM, TN, 2>), grid, dim3(256), 0, st, d);
    else if (abl == 3) hipLaunchKernelGGL((conv1d_x3_kernel<WM, WN, TM, TN, 3>), grid, dim3(256), 0, st, d);
    else if (abl == 4) hipLaunchKernelGGL((conv1d_x3_kernel<WM, WN, TM, TN, 4>), grid, dim3(256), 0, st, d);
    else
#endif
    hipLaunchKernelGGL((conv1d_x3_kernel<WM, WN, TM, TN, 0>), grid, dim3(256), 0, st, d);
}
template <int WM, int WN, int TM, int TN>
static void launch_cfg_xt(const ConvDev& d, int n_count, int B, hipStream_t st) {
    dim3 grid(cdiv(n_count, WN * TN * 32), cdiv(d.Co, WM * TM * 32), B * d.phases);
    hipLaunchKernelGGL((conv1d_x3_kernel<WM, WN, TM, TN, 5>), grid, dim3(256), 0, st, d);
}

// ---------------------------------------------------------------------------------------------------------
// One output channel (HiFi-GAN conv_post: 32 -> 1 channels, k = 7, tanh; vocoder/hifigan/modules/hifigan.py:139-141).  On the MFMA kernels a
// 32-row tile computes one useful row: 860 us for 0.86 GFLOP at 8 clips.  Here a thread owns two output samples and runs the SAME fmaf
// chain the f32 MFMA does (16-channel chunks -> taps -> channels; v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain, bit for bit), on the
// activated window staged once per chunk in LDS: bit-identical to conv1d_f32_kernel, exact fp32 in both vocoder precisions.
// ---------------------------------------------------------------------------------------------------------
#define C1_TT 512
__global__ void __launch_bounds__(256) conv1d_co1_kernel(const ConvDev p) {
    __shared__ float xs[CK][C1_TT + XHALO];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, n0 = blockIdx.x * C1_TT;
    const int halo = (p.ntaps - 1) * p.dil, xw = C1_TT + halo;
    const float* xb = p.x + (int64_t)b * p.x_bstride;
    float acc[2] = {0.f, 0.f};
    for (int c0 = 0; c0 < p.Ci; c0 += CK) {
        for (int i = tid; i < CK * xw; i += 256) {
            const int ci = i / xw, wpos = i - ci * xw;
            const int idx = n0 - p.pad + wpos;
            float v = 0.f;
            if (c0 + ci < p.Ci && idx >= 0 && idx < p.T_in) {
                v = xb[(int64_t)(c0 + ci) * p.T_in + idx];
                if (p.in_act == ACT_LRELU) v = v > 0.f ? v : v * p.in_slope;
            }
            xs[ci][wpos] = v;
        }
        __syncthreads();
        for (int j = 0; j < p.ntaps; ++j) {
            const float* wj = p.w + (int64_t)j * p.Ci + c0;
#pragma unroll
            for (int ci = 0; ci < CK; ++ci) {
                const float wv = (c0 + ci < p.Ci) ? wj[ci] : 0.f;
                acc[0] = fmaf(wv, xs[ci][2 * tid + j * p.dil], acc[0]);
                acc[1] = fmaf(wv, xs[ci][2 * tid + 1 + j * p.dil], acc[1]);
            }
        }
        __syncthreads();
    }
    const float bias = p.bias ? p.bias[0] : 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int n = n0 + 2 * tid + e;
        if (n >= p.T_out) continue;
        const int64_t oi = (int64_t)b * p.out_bstride + n;
        const float res = p.res ? p.res[(int64_t)b * p.res_bstride + n] : 0.f;
        const float old = p.beta != 0.f ? p.out[oi] : 0.f;
        p.out[oi] = conv_out_value(p, acc[e], bias, res, old);
    }
}

int launch_conv1d(const ConvArgs& a, hipStream_t st) {
    ConvDev d;
    d.x = a.x; d.x_bstride = a.x_bstride; d.Ci = a.Ci; d.T_in = a.T_in; d.x_bmod = a.x_bmod;
    d.w = a.w; d.w_bstride = a.w_bstride; d.bias = a.bias; d.Co = a.Co; d.dil = a.dil; d.pad = a.pad;
    d.upsample2 = a.upsample2; d.in_act = a.in_act; d.in_slope = a.in_slope;
    d.in_stride = a.in_stride > 0 ? a.in_stride : 1; d.in_phase = a.in_phase;
    if (d.in_stride > 1 && (a.upsample2 || a.tr_stride > 1)) VB_FAIL(VB_E_INVALID, "conv1d: in_stride with upsample/transpose");
    d.gn_mean = a.gn_mean; d.gn_rstd = a.gn_rstd; d.gn_gamma = a.gn_gamma; d.gn_beta = a.gn_beta; d.gn_groups = a.gn_groups;
    d.out = a.out; d.out_bstride = a.out_bstride; d.T_out = a.T_out; d.res = a.res; d.res_bstride = a.res_bstride;
    d.alpha = a.alpha; d.beta = a.beta; d.acc_scale = a.acc_scale; d.out_act = a.out_act; d.out_slope = a.out_slope;
    d.out_transposed = a.out_transposed; d.add = a.add; d.add_bstride = a.add_bstride; d.add_bmod = a.add_bmod;
    d.wp = a.wp; d.wp_plane = a.wp_plane; d.Ci_pad = a.Ci_pad; d.wp_bstride = a.wp_bstride;
    d.xt = a.xt; d.xt_Tp = xt_rows(a.upsample2 ? 2 * a.T_in : a.T_in); d.xt_plane = (int64_t)a.B * d.xt_Tp * a.Ci;
    if (a.xt && (a.Ci % CK3 || a.tr_stride > 1 || a.in_stride > 1 || !a.wp || a.x_bmod || a.pad > XT_HEAD))
        VB_FAIL(VB_E_INVALID, "conv1d: XT input needs Ci %% 32 == 0, stride 1, split weights, pad <= %d", XT_HEAD);
    d.stage_epi = 0;
    int n_count;
    if (a.tr_stride > 1) {
        d.phases = a.tr_stride; d.tr_pad = a.tr_pad; d.ntaps = (a.tr_k + a.tr_stride - 1) / a.tr_stride; d.dil = 1;
        n_count = cdiv(a.T_out, a.tr_stride);
        if (a.out_transposed) VB_FAIL(VB_E_INVALID, "conv1d: transposed output not supported with tr_stride");
    } else {
        d.phases = 1; d.tr_pad = 0; d.ntaps = a.ksize; n_count = a.T_out;
    }
    if ((d.ntaps - 1) * d.dil > XHALO) VB_FAIL(VB_E_INVALID, "conv1d: halo %d exceeds %d", (d.ntaps - 1) * d.dil, XHALO);
    if (a.out_transposed && (a.Co % 4)) VB_FAIL(VB_E_INVALID, "conv1d: transposed output needs Co%%4==0");
    if ((a.in_act == ACT_GN_SWISH || a.in_act == ACT_GN) && (a.Ci % a.gn_groups)) VB_FAIL(VB_E_INVALID, "conv1d: Ci %% groups");
    // Wide layers over pre-activated transposed planes ARE GEMMs (rows = clip x time, K = taps x Ci, both operands K-contiguous
    // already): the DMA-fed 128 x 128 GEMM kernel walks K tap by tap (GemmArgs::conv_*), three bf16 passes for the split precision,
    // bias + residual in a channel-major epilogue.  Measured on these shapes (tools/conv_as_gemm_calib.py): 750-840 TFLOP/s of bf16
    // MFMA work against ~385 of this file's kernel, which stages weights through registers and synchronises per (tap, 32 channels).
    if (a.xt && a.wp && a.Co >= 384 && a.Ci % 64 == 0 && a.Ci_pad == a.Ci && a.Co % 4 == 0 && d.phases == 1 && !a.wp_bstride && !a.w_bstride &&
        a.alpha == 1.f && a.beta == 0.f && a.acc_scale == 1.f && a.out_act == ACT_NONE && !a.out_transposed && !a.add &&
        a.out_bstride == (int64_t)a.Co * a.T_out && (!a.res || a.res_bstride == a.out_bstride) &&
        a.T_out == (a.upsample2 ? 2 * a.T_in : a.T_in) && (int64_t)a.B * a.T_out < (1 << 21) && a.Co < (1 << 21) && !vb_tune().conv_gemm_off) {
        // (B * T_out / Co beyond launch_gemm's reciprocal-divide range fall through to the conv kernels - ADVICE r3)
        GemmArgs g;
        g.A = a.xt; g.a_plane = d.xt_plane; g.lda = a.Ci; g.B = a.wp; g.b_plane = a.wp_plane; g.ldb = a.Ci_pad;
        g.M = a.B * a.T_out; g.N = a.Co; g.K = d.ntaps * a.Ci; g.nseg = 3; g.ngroups = a.B;
        g.group_rows = a.T_out; g.T = a.T_out; g.epi = EPI_F32_CT; g.bias = a.bias; g.out32 = a.out; g.res32 = a.res;
        g.conv_ci = a.Ci; g.conv_dil = d.dil; g.conv_agrp = d.xt_Tp; g.conv_arow0 = XT_HEAD - a.pad; g.conv_btap = (int64_t)a.Co * a.Ci_pad;
        g.prof_class = 2;
        return launch_gemm(g, st);
    }
    // staged epilogue (split-bf16 kernel): plain [b][co][t] output whose rows start 16-B aligned
    d.stage_epi = (a.tr_stride <= 1 && !a.out_transposed && a.T_out % 4 == 0 && a.out_bstride % 4 == 0 &&
                   (!a.res || a.res_bstride % 4 == 0) && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                   (!a.res || (reinterpret_cast<uintptr_t>(a.res) & 15) == 0) && !vb_tune().conv_direct_epi) ? 1 : 0;
    // minimal filtering (conv1d_f32w_kernel): the DMA-fed fp32 kernel's conditions + stride 1, shared weights, the staged epilogue, wide layers
    const bool use_mf = a.w_mf && !a.wp && !a.w_bstride && !a.x_bmod && d.in_stride == 1 && d.phases == 1 && !a.upsample2 &&
                        (a.in_act == ACT_NONE || a.in_act == ACT_LRELU) && a.Ci % GK == 0 && a.Co % 4 == 0 && a.Co >= 32 && d.stage_epi &&
                        conv1d_f32w_supported(a.ksize, a.dil) && a.T_in % 4 == 0 && a.x_bstride % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w_mf) & 15) == 0 && !vb_tune().conv_mf_off;
    // (flops of the profiler's class table: EXECUTED MFMA work - the minimal-filtering kernel runs pseudo-taps / 2 products per output)
    const double taps_eff = use_mf ? 0.5 * conv1d_f32w_pseudo_taps(a.ksize) : (a.tr_stride > 1 ? (double)a.tr_k / a.tr_stride : (double)a.ksize);
    ProfScope prof(2, 2.0 * a.B * a.Co * a.Ci * (double)a.T_out * taps_eff,
                   4.0 * a.B * ((double)a.Ci * a.T_in * (a.xt ? 1.0 : 1.0) + (double)a.Co * a.T_out * (1.0 + (a.res ? 1.0 : 0.0) + (a.beta != 0.f ? 1.0 : 0.0)))
                       + 4.0 * (double)a.Co * a.Ci * (a.tr_stride > 1 ? a.tr_k : a.ksize), st);
    if (use_mf) {
        d.ww = a.w_mf;
        launch_conv1d_f32w(d, a.B, st);
    } else if (a.Co == 1 && a.w && !a.w_bstride && !a.x_bmod && d.phases == 1 && d.in_stride == 1 && !a.upsample2 && !a.out_transposed &&
        (a.in_act == ACT_NONE || a.in_act == ACT_LRELU) && !vb_tune().conv_f32_old) {
        hipLaunchKernelGGL(conv1d_co1_kernel, dim3(cdiv(a.T_out, C1_TT), a.B), dim3(256), 0, st, d);       // (VB_CONV_F32_OLD=1: the MFMA kernels)
    } else if (a.wp && (!a.w_bstride || a.wp_bstride)) {
        if (a.Ci_pad % CK3) VB_FAIL(VB_E_INVALID, "conv1d: split weights need Ci_pad %% %d == 0", CK3);
        // one workgroup of the 128co x 256t tile per CU (92 KB LDS): a grid a little over 256 workgroups (every VAE level at
        // B = 8 makes 288) runs as two rounds at 56 % - the 128co x 128t tile (71 KB, two per CU) halves the granule
        const int64_t blocks = (int64_t)cdiv(n_count, 256) * cdiv(a.Co, 128) * a.B * d.phases;
        const double eff = (double)blocks / (double)(cdiv(blocks, 256) * 256);
        if (a.xt) {
            if (a.Co <= 64) VB_FAIL(VB_E_INVALID, "conv1d: XT input is built for Co > 64 (wide layers)");
            if (eff < 0.7) launch_cfg_xt<2, 2, 2, 1>(d, n_count, a.B, st);
            else launch_cfg_xt<2, 2, 2, 2>(d, n_count, a.B, st);
        } else if (a.Co > 64 && eff < 0.7) launch_cfg_x3<2, 2, 2, 1>(d, n_count, a.B, st);
        else if (a.Co > 64) launch_cfg_x3<2, 2, 2, 2>(d, n_count, a.B, st);
        else if (a.Co > 32) launch_cfg_x3<2, 2, 1, 2>(d, n_count, a.B, st);
        else launch_cfg_x3<1, 4, 1, 2>(d, n_count, a.B, st);
    } else if (!a.wp && a.w_bstride % 4 == 0 && !a.x_bmod && d.in_stride == 1 && (a.in_act == ACT_NONE || a.in_act == ACT_LRELU) && a.Ci % GK == 0 &&
               a.Co % 4 == 0 && (d.ntaps - 1) * d.dil <= 60 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 &&
               (a.upsample2 ? (a.Co > 64 && d.phases == 1)
                            : (a.T_in % 4 == 0 && a.x_bstride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0)) &&
               !vb_tune().conv_f32_old) {
        // exact fp32, DMA-fed (conv1d_f32g_kernel); VB_CONV_F32_OLD=1 keeps the register-staged kernel (bit-identical, tests compare the two)
        launch_conv1d_f32g(d, n_count, a.B, a.upsample2, st);
    } else if (a.Co > 64) launch_cfg<2, 2, 2, 2>(d, n_count, a.B, st);
    else if (a.Co > 32) launch_cfg<2, 2, 1, 2>(d, n_count, a.B, st);
    else launch_cfg<1, 4, 1, 2>(d, n_count, a.B, st);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
