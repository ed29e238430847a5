// Fused HiFi-GAN ResBlock1 "pair" for the narrow, long stages of the generator (gfx950):
//
//   out[b][c][t] = beta*out + alpha*( x + b2 + conv2_{k,dil=1}( lrelu( b1 + conv1_{k,dil=d}( lrelu(x) ) ) ) )
//
// (vocoder/hifigan/modules/hifigan.py ResBlock1.forward: xt = lrelu(x); xt = c1(xt); xt = lrelu(xt); xt = c2(xt); x = xt + x)
//
// At C = 64 / 32 channels and T = 240k / 481k samples per clip the two convolutions of a pair are HBM-bound as separate
// launches: x is read, the intermediate written, read again, the residual read and the result written - 5 tensor passes.
// Here one workgroup produces TT = 128-(k-1) output samples of ALL channels and keeps the intermediate in LDS: 2 passes.
//   * window of x (TT + (k-1)(d+1) samples) -> LeakyReLU -> bf16 hi/lo split -> LDS xT[plane][t][ci]  (per 32-ci chunk)
//   * conv1 as split-bf16 ("bf16x3") MFMA implicit GEMM over exactly 128 intermediate positions (one 32-wide MFMA column
//     tile per wave), + b1, LeakyReLU, zero outside [0,T) (conv2 pads the ACTIVATED intermediate), split -> LDS hT
//   * conv2 over hT, + b2 + residual x, alpha/beta accumulation into the MRF sum, coalesced stores (a lane owns one t).
// Same operand precision as conv1d_x3_kernel (operand error 2^-17, fp32 accumulate).
#include "kernels.h"

#define RP_T 128            // intermediate positions per workgroup (4 waves x 32)
#define RP_HALO 64          // max (k-1)*dil of conv1
#define RP_XW (RP_T + RP_HALO)
#define RP_P 40             // bf16 elements per LDS row of a 32-channel chunk (32 + 8 pad: conflict-free 16-B fragment reads)

struct PairDev {
    const float* x; float* out; int64_t bstride; int T;
    int k, dil;
    const bf16_t* w1; const bf16_t* w2; int64_t w_plane;      // [2 planes][k][C][C] each, ci contiguous
    const float* b1; const float* b2;
    float slope, alpha, beta;
    int staged;               // 16-B (staged) epilogue: T % 4 == 0 and 16-B aligned tensors
};

// one output element (pinned arithmetic: the direct and the staged epilogue must round alike)
__device__ __forceinline__ float pair_out_value(const PairDev& p, float acc, float bias, float res, float old) {
#pragma clang fp contract(off)
    float val = acc + bias;
    val = val + res;
    return fmaf(val, p.alpha, p.beta * old);
}

template <int CH>      // C = 32*CH channels
__global__ void __launch_bounds__(256) respair_x3_kernel(const PairDev p) {
    constexpr int C = 32 * CH;
    // xT (one ci chunk of the activated window, conv1 only) and hT (activated intermediate, all chunks, conv2 only) share
    // storage: hT is written after the barrier that ends conv1's last tap.  41 KB + weights instead of 72 KB: 3 workgroups
    // per CU at 32 channels, 2 at 64 - these kernels are a chain of short phases and live off co-resident workgroups.
    constexpr int XT_EL = 2 * RP_XW * RP_P, HT_EL = CH * 2 * RP_T * RP_P;
    __shared__ __attribute__((aligned(16))) bf16_t xh[XT_EL > HT_EL ? XT_EL : HT_EL];
    bf16_t (*xT)[RP_XW * RP_P] = reinterpret_cast<bf16_t (*)[RP_XW * RP_P]>(xh);
    bf16_t (*hT)[2][RP_T * RP_P] = reinterpret_cast<bf16_t (*)[2][RP_T * RP_P]>(xh);
    __shared__ __attribute__((aligned(16))) bf16_t wl[2][2][C * RP_P];             // [buf][plane] one (tap, ci chunk) of weights

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int h2 = (p.k - 1) / 2, h1 = (p.k - 1) * p.dil / 2;
    const int TT = (RP_T - (p.k - 1)) & ~3;          // outputs per workgroup (a multiple of 4: the epilogue moves 16-B quads)
    const int n0 = blockIdx.x * TT;                  // first output sample
    const int m0 = n0 - h2;                          // first intermediate position
    const int x0 = m0 - h1;                          // first window sample
    const int xw_used = RP_T + (p.k - 1) * p.dil;
    const float* xb = p.x + (int64_t)b * p.bstride;

    // weight tile of one (tap, ci chunk): C rows x 32 ci x 2 planes = C*8 pieces of 16 B
    constexpr int WPT = C * 8 / 256;
    uint4 wreg[WPT];
    auto wload = [&](const bf16_t* wsrc, int c0, int j) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int id = tid + i * 256;
            const int pl = id / (C * 4), rem = id - pl * (C * 4);
            const int co = rem >> 2, pc = rem & 3;
            wreg[i] = *reinterpret_cast<const uint4*>(wsrc + pl * p.w_plane + ((int64_t)j * C + co) * C + c0 + pc * 8);
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int id = tid + i * 256;
            const int pl = id / (C * 4), rem = id - pl * (C * 4);
            const int co = rem >> 2, pc = rem & 3;
            *reinterpret_cast<uint4*>(&wl[buf][pl][co * RP_P + pc * 8]) = wreg[i];
        }
    };
    // activation window staging in two halves (loads of the next chunk fly while the taps of this one are multiplied)
    // (a wave owns 8 CONSECUTIVE channels of the chunk, a lane one window position per pass: coalesced loads along t, ONE 16-byte
    //  LDS write per plane and position - four bank-conflicted 4-byte writes of channel pairs before round 3)
    constexpr int NIT = RP_XW / 64;
    float raw[8][NIT];
    auto xload = [&](int c0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float* xrow = xb + (int64_t)(c0 + 8 * wave + e) * p.T;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int wpos = lane + 64 * it;
                const int idx = x0 + wpos;
                const bool ok = wpos < xw_used && idx >= 0 && idx < p.T;
                raw[e][it] = ok ? xrow[idx] : 0.f;
            }
        }
    };
    auto xstore = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int wpos = lane + 64 * it;
            if (wpos >= xw_used) continue;
            bf16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = raw[e][it];                   // out-of-range samples were loaded as 0 and lrelu(0) = 0
                const float v = t > 0.f ? t : t * p.slope;
                hi[e] = f2bf(v);
                lo[e] = f2bf(v - bf2f(hi[e]));
            }
            *reinterpret_cast<bf16x8*>(&xT[0][wpos * RP_P + 8 * wave]) = hi;
            *reinterpret_cast<bf16x8*>(&xT[1][wpos * RP_P + 8 * wave]) = lo;
        }
    };

    // two accumulator sets (hi*hi terms / cross terms): consecutive MFMAs never wait for each other's result
    f32x16 acc[CH], acx[CH];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; acx[i][r] = 0.f; }
    };
    auto fold_acc = [&]() {
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] += acx[i][r];
    };
    // all taps of one ci chunk: B fragments from `src` rows (32*wave + l31 + j*step), A fragments from the weight tile
    auto taps = [&](const bf16_t* wsrc, int c0, const bf16_t* s0, const bf16_t* s1, int step) {
        for (int j = 0; j < p.k; ++j) {
            const int buf = j & 1;
            if (j + 1 < p.k) wload(wsrc, c0, j + 1);
            const int row = 32 * wave + l31 + j * step;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int kofs = ks * 16 + g * 8;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(s0 + row * RP_P + kofs);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(s1 + row * RP_P + kofs);
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int o = (i * 32 + l31) * RP_P + kofs;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&wl[buf][0][o]);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(&wl[buf][1][o]);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
                    acx[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acx[i], 0, 0, 0);
                    acx[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acx[i], 0, 0, 0);
                }
            }
            if (j + 1 < p.k) wstore(buf ^ 1);
            __syncthreads();
        }
    };

    // ---- conv1 (dilated) over the activated window -> intermediate positions m0 + [0,128)
    zero_acc();
    xload(0);
#pragma unroll 1
    for (int ch = 0; ch < CH; ++ch) {
        xstore();
        wload(p.w1, ch * 32, 0);
        wstore(0);
        __syncthreads();
        if (ch + 1 < CH) xload((ch + 1) * 32);
        taps(p.w1, ch * 32, &xT[0][0], &xT[1][0], p.dil);
    }
    fold_acc();
    {   // + b1, LeakyReLU, zero outside [0,T), split, to hT[chunk][plane][t][c]
        const int m = m0 + 32 * wave + l31;
        const bool inr = m >= 0 && m < p.T;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = 8 * rg + 4 * g;          // channel inside chunk i
                bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][rg * 4 + e] + p.b1[i * 32 + c + e];
                    v = v > 0.f ? v : v * p.slope;
                    if (!inr) v = 0.f;
                    hi[e] = f2bf(v);
                    lo[e] = f2bf(v - bf2f(hi[e]));
                }
                *reinterpret_cast<bf16x4*>(&hT[i][0][(32 * wave + l31) * RP_P + c]) = hi;
                *reinterpret_cast<bf16x4*>(&hT[i][1][(32 * wave + l31) * RP_P + c]) = lo;
            }
    }
    // ---- conv2 (dil 1) over the intermediate -> outputs n0 + [0,TT)
    zero_acc();
#pragma unroll 1
    for (int ch = 0; ch < CH; ++ch) {
        wload(p.w2, ch * 32, 0);
        wstore(0);             // wl[0]'s last readers passed the barrier that ends taps()
        __syncthreads();       // (ch == 0: also publishes hT)
        taps(p.w2, ch * 32, &hT[ch][0][0], &hT[ch][1][0], 1);
    }
    fold_acc();
    // ---- epilogue.  The accumulator gives a lane ONE output sample and 16 channels; moved like that every residual /
    // accumulate-into load and every store is a 4-byte lane access.  When the rows are 16-B aligned (T % 4 == 0) each wave passes
    // its 32 x 32 tiles through a private LDS patch instead (xh is free: conv2's last barrier is behind every wave) and comes
    // back with 4 consecutive samples of one channel per lane - 16-byte accesses, whole 128-B lines per 8 lanes; same
    // arithmetic per element (see conv_epilogue_staged in conv1d_f32.hip).
    constexpr int EP = 36;
    static_assert(sizeof(xh) >= 4 * 32 * EP * sizeof(float), "staging patches must fit");
    if (p.staged) {
        float* patch = reinterpret_cast<float*>(xh) + wave * (32 * EP);
        const int rr = lane >> 3, t4 = (lane & 7) * 4;
        const int nl = 32 * wave + t4;
        const int n = n0 + nl;
        const bool nok = nl < TT && n < p.T;
        float* ob = p.out + (int64_t)b * p.bstride;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[(4 * g + 8 * (r >> 2) + (r & 3)) * EP + l31] = acc[i][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            float4 v[4], rv[4], ov[4];
            float bv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = i * 32 + rr + 8 * k;
                v[k] = *reinterpret_cast<const float4*>(patch + (rr + 8 * k) * EP + t4);
                const int64_t oi = (int64_t)co * p.T + (nok ? n : 0);
                rv[k] = nok ? *reinterpret_cast<const float4*>(xb + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
                ov[k] = (nok && p.beta != 0.f) ? *reinterpret_cast<const float4*>(ob + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[k] = p.b2[co];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (nok) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int co = i * 32 + rr + 8 * k;
                    const float a4[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, r4[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
                    const float o4[4] = {ov[k].x, ov[k].y, ov[k].z, ov[k].w};
                    float q[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = pair_out_value(p, a4[e], bv[k], r4[e], o4[e]);
                    *reinterpret_cast<float4*>(ob + (int64_t)co * p.T + n) = make_float4(q[0], q[1], q[2], q[3]);
                }
            }
        }
    } else {
        const int nl = 32 * wave + l31;
        const int n = n0 + nl;
        const bool nok = nl < TT && n < p.T;
        float* ob = p.out + (int64_t)b * p.bstride;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            float rv[16], ov[16], bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = i * 32 + 4 * g + 8 * (r >> 2) + (r & 3);
                rv[r] = nok ? xb[(int64_t)co * p.T + n] : 0.f;
                ov[r] = (nok && p.beta != 0.f) ? ob[(int64_t)co * p.T + n] : 0.f;
                bv[r] = p.b2[co];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = i * 32 + 4 * g + 8 * (r >> 2) + (r & 3);
                if (!nok) continue;
                ob[(int64_t)co * p.T + n] = pair_out_value(p, acc[i][r], bv[r], rv[r], ov[r]);
            }
        }
    }
}


int launch_respair(const RespairArgs& a, hipStream_t st) {
    if (a.C != 32 && a.C != 64) VB_FAIL(VB_E_INVALID, "respair: C=%d (32 or 64)", a.C);
    if (a.k < 1 || (a.k & 1) == 0 || (a.k - 1) * a.dil > RP_HALO || a.k > 33) VB_FAIL(VB_E_INVALID, "respair: k=%d dil=%d", a.k, a.dil);
    if (a.x == a.out) VB_FAIL(VB_E_INVALID, "respair: x and out must be distinct buffers (neighbouring workgroups re-read the halo)");
    PairDev d;
    d.x = a.x; d.out = a.out; d.bstride = (int64_t)a.C * a.T; d.T = a.T; d.k = a.k; d.dil = a.dil;
    d.w1 = a.w1; d.w2 = a.w2; d.w_plane = (int64_t)a.k * a.C * a.C; d.b1 = a.b1; d.b2 = a.b2;
    d.slope = a.slope; d.alpha = a.alpha; d.beta = a.beta;
    const int TT = (RP_T - (a.k - 1)) & ~3;
    d.staged = (a.T % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                !vb_tune().conv_direct_epi) ? 1 : 0;
    dim3 grid(cdiv(a.T, TT), 1, a.B);
    // two convolutions' worth of flops (the recomputed halo of conv1 is not counted)
    ProfScope prof(3, 2.0 * 2.0 * a.B * (double)a.C * a.C * a.k * (double)a.T,
                   4.0 * a.B * (double)a.C * a.T * (2.0 + (a.beta != 0.f ? 1.0 : 0.0)) + 2.0 * 4.0 * a.k * a.C * a.C, st);
    // (a persistent variant with both convolutions' weights resident in LDS and the next window prefetched was measured slower,
    //  460 / 900 / 1100 us against 440 / 620 / 830 us for k = 3 / 7 / 11: one workgroup per CU cannot hide its own phase latencies)
    if (a.C == 32) hipLaunchKernelGGL(respair_x3_kernel<1>, grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL(respair_x3_kernel<2>, grid, dim3(256), 0, st, d);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
