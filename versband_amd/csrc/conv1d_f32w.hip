// Reduced-multiplication exact-fp32 Conv1d (v_mfma_f32_32x32x2_f32): 1-D minimal filtering F(2,3) over the DMA-fed rings of conv1d_f32g.hip.
// Built with -mllvm -amdgpu-mfma-vgpr-form like that file (versband_amd/build.py).
//
// Why: the fp32 VAE / vocoder convolutions run at 0.77 of the f32-MFMA peak and are 47 % of a pass - past that wall only fewer
// multiplications help.  The HiFi-GAN ResBlock convolutions (vocoder/hifigan/modules/hifigan.py:27-64: k = 3 / 7 / 11, dilation 1 / 3 / 5, Ci = Co)
// are stride-1 FIR filters per channel pair, so two neighbouring outputs of one dilation class share their inputs:
//
//   y0 = w0 x0 + w1 x1 + w2 x2        m0 = (x0 - x2) w0               y0 = m0 + m1 + m2
//   y1 = w0 x1 + w1 x2 + w2 x3        m1 = (x1 + x2) (w0 + w1 + w2)/2  y1 = m1 - m2 - m3
//                                     m2 = (x2 - x1) (w0 - w1 + w2)/2
//                                     m3 = (x1 - x3) w2
//
// (Winograd's F(2,3): 4 products for 2 outputs of a 3-tap filter instead of 6.)  The sum over input channels commutes with the output
// combination, so the four m's are four ACCUMULATORS over (ci, tap group): every "pseudo-tap" is one ring step exactly like a tap of the direct
// kernel - a [16 ci][128 co] weight tile (pre-combined on the host, pack.py:pack_conv_mf) against ONE value per lane built from two window reads
// by one v_add / v_sub - and the outputs are formed once, in the epilogue.  A k-tap filter is floor(k/3) such groups plus a remainder: one tap
// = 2 pseudo-taps (into m0 and, negated, m3), two taps = 3 (Karatsuba: (x0 - x1) w0 -> m0, x1 (w0 + w1) -> m1, (x2 - x1)(-w1) -> m3):
//
//   k = 3:  4 pseudo-taps per output PAIR = 2.0 per output (direct 3)      1.50x fewer MFMAs
//   k = 7: 10                                5.0            (direct 7)      1.40x
//   k = 11: 15                               7.5            (direct 11)     1.47x
//
// All coefficients are +-1 and 1/2: the transform adds one rounding per operand, the results agree with the direct kernel to fp32 roundoff
// (<= 3e-7 of the tensor's max, tests/test_gpu_kernels.py), NOT bit for bit - so this is a separate precision mode ("fp32mf") and the direct kernel stays.
//
// Dilation d: outputs u and u + d pair up.  A wave's 32 MFMA columns are v = (q, r), r < d, q < 32 / d; column v owns outputs 2 d q + r and
// 2 d q + r + d of the wave's 2 * VW positions, VW = (32 / d) d (32 for d = 1, 30 for d = 3 / 5: two idle columns).  Tile = 128 co x 4 VW positions,
// 4 waves as 2 x 2, wave = 64 co x 2 VW positions, accumulators 4 m's x 2 co tiles x 16 = 128 VGPRs (two workgroups per CU).
// Ring, DMA pieces, counted vmcnt, fragment pipeline and XCD numbering: conv1d_f32g_kernel's unrolled-tap form with NT = pseudo-taps.
#include <stdlib.h>
#include <type_traits>

#include "conv1d_dev.h"
#include "lds_asm.h"

template <int I, int N, class F> __device__ __forceinline__ void w_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); w_static_for<I + 1, N>(f); }
}
typedef __attribute__((address_space(3))) void* w_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* w_glb_ptr_t;
template <int N> __device__ __forceinline__ void w_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}

#include "mf_taps.h"

#define MF_PITCH 68          // floats per staged channel row (64 positions + 4)
#ifndef MF_MFMA_FORM
#define MF_MFMA_FORM 0
#endif
#if MF_MFMA_FORM == 0
#define MFMA_ACC(c, a, b) (c) = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#elif MF_MFMA_FORM == 1
#define MFMA_ACC(c, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#else
#define MFMA_ACC(c, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#endif

// WN = waves along time (2: tile 128 co x 4 VW positions; 4: 64 co x 8 VW positions - the 64-channel layers)
template <int K, int OCC, int WN>
__global__ void __launch_bounds__(256, OCC) conv1d_f32w_kernel(const ConvDev p) {
    // ring depth = conv1d_f32g_kernel's: two window stages, three weight stages (49 KB; 60 KB on the 64-channel tile).  (Round 6, measured and not
    // adopted: five weight stages in the two-per-CU build - vmcnt retires in order, so every load's lead, the window's too, is NSW - 1 ring steps -
    // no faster on the layer shapes and 1.5 % slower to equal end to end: 1215 / 1193 against 1199 / 1205 mel-s/s, profiles/r06_mf_ring_depth.txt.)
    constexpr int TM = 2, WM = 4 / WN, NXS = 2, NSW = 3;
    constexpr int P = mf_ntaps(K);
    constexpr int CO_TILE = WM * TM * 32, XP = WN * 64 + 64, NP = XP / 64;
    constexpr int XST = GK * XP, WT = GK * CO_TILE;
    constexpr int NWI = CO_TILE / 16, WPW = NWI / 4, XPW = NP;
    static_assert(P >= NSW - 1, "a step issues the tile NSW - 1 ahead: it lies in this chunk or the next one");
    static_assert((NXS * XST + NSW * WT) >= 4 * 32 * MF_PITCH, "the staging patches must fit in the rings");
    extern __shared__ __attribute__((aligned(16))) float w_lds[];
    float* lx = w_lds;
    float* lw = w_lds + NXS * XST;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int dil = p.dil;
    const int VW = (32 / dil) * dil;                       // MFMA columns in use per wave; the wave owns 2 * VW consecutive positions
    const int T_TILE = WN * 2 * VW;
    int b, n0, co0;
    {
        const int L = blockIdx.x, j = L >> 3;
        const int ct = j / p.g_tbx, ul = j - ct * p.g_tbx;
        const int u = ul * 8 + (L & 7);
        if (u >= p.g_ntb) return;
        b = u / p.g_nt;
        n0 = (u - b * p.g_nt) * T_TILE;
        co0 = ct * CO_TILE;
    }
    const int n_count = p.T_out;
    if (n0 >= n_count) return;

    const int start = n0 - p.pad;
    const int start_al = start & ~3;
    const int aoff = start - start_al;
    const float* xbase = p.x + (int64_t)b * p.x_bstride;
    const float* wbase = p.ww;
    float slope = p.in_act == ACT_LRELU ? p.in_slope : 1.f;
    asm volatile("v_mov_b32 %0, %0" : "+v"(slope));

    // ---- window DMA (conv1d_f32g_kernel: 16-B lanes, four rows of 64 positions per 1-KB piece)
    int xsrc[XPW];
    unsigned xoob = 0;
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int ii = wave * XPW + i;
        const int q = ii * 4 + (lane >> 4);
        const int ci = q / NP, pos = (q - ci * NP) * 64 + (lane & 15) * 4;
        const int idx = start_al + pos;
        const bool ok = idx >= 0 && idx < p.T_in;
        xsrc[i] = ci * p.T_in + (ok ? idx : 0);
        xoob |= ok ? 0u : (1u << i);
    }
    auto issue_x = [&](int ch, int stage) {
        const float* src = xbase + (int64_t)ch * GK * p.T_in;
        float* dst = lx + stage * XST;
#pragma unroll
        for (int i = 0; i < XPW; ++i)
            __builtin_amdgcn_global_load_lds((w_glb_ptr_t)(src + xsrc[i]), (w_lds_ptr_t)(dst + (wave * XPW + i) * 256), 16, 0, 0);
    };
    const bool act = p.in_act == ACT_LRELU;
    auto fix_x = [&](int stage) {     // zero padding + LeakyReLU in place, by the lanes whose own DMA brought the quads
        const unsigned a0 = lds_u32(lx + stage * XST + wave * XPW * 256 + lane * 4);
        lds_u32x4 v[XPW];
        const lds_u32x4 zero = {0u, 0u, 0u, 0u};
        if (act) {
            w_static_for<0, XPW>([&](auto ic) { constexpr int I = decltype(ic)::value; lds_rd128<I * 1024>(v[I], a0); });
            LDS_WAIT(0);
        }
        w_static_for<0, XPW>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            const bool oob = (xoob >> I) & 1;
            if (act) { lds_pin(v[I]); lds_wr128<I * 1024>(a0, oob ? zero : lds_lrelu128_apply(v[I], slope)); }
            else if (oob) lds_wr128<I * 1024>(a0, zero);
        });
    };
    // ---- weight DMA: tile (chunk, pseudo-tap) = 16 rows of 128 floats
    constexpr int RPI = 256 / CO_TILE, LPR = CO_TILE / 4;
    int wsrc0;                            // piece 0 of this wave; piece i lies RPI * i rows further on (a scalar offset)
    {
        const int row = wave * WPW * RPI + lane / LPR;
        int cog = co0 + (lane % LPR) * 4;
        if (cog >= p.Co) cog = 0;
        wsrc0 = row * p.Co + cog;
    }
    auto issue_w = [&](int ch, int j, int slot) {
        const float* src = wbase + ((int64_t)j * p.Ci + ch * GK) * p.Co;
        float* dst = lw + slot * WT;
#pragma unroll
        for (int i = 0; i < WPW; ++i)
            __builtin_amdgcn_global_load_lds((w_glb_ptr_t)(src + (int64_t)i * RPI * p.Co + wsrc0), (w_lds_ptr_t)(dst + (wave * WPW + i) * 256), 16, 0, 0);
    };

    f32x16 acc[4][TM];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;

    const int nchunks = p.Ci / GK;
#pragma unroll
    for (int c = 0; c < NXS - 1; ++c) issue_x(c < nchunks ? c : 0, c);      // windows first: they retire first
#pragma unroll
    for (int t = 0; t < NSW - 1; ++t) issue_w(0, t, t);       // P >= NSW - 1: the first tiles all lie in chunk 0

    // column v = (q, r) -> the even output's position inside the wave's range; idle columns (v >= VW) repeat column 0
    const int vcol = l31 < VW ? l31 : 0;
    const int vq = vcol / dil;
    const int bpos = 2 * dil * vq + (vcol - vq * dil);
    const unsigned xa0 = lds_u32(lx + aoff + wn * 2 * VW + bpos + g * XP);
    const unsigned wa0 = lds_u32(lw + wm * TM * 32 + l31 + g * CO_TILE);
    const int dil4 = dil * 4;

    float da[TM], dbv = 0.f;             // held-back operands of a step's last channel pair (conv1d_f32g_kernel)
#pragma unroll
    for (int i = 0; i < TM; ++i) da[i] = 0.f;

    // ONE loop body for every chunk (no separate last-chunk variant: two copies of the unrolled steps made the accumulators' live ranges
    // meet in 128-register PHIs the allocator could not coalesce).  The last chunk therefore issues what a middle chunk would - the "next"
    // window and the first NSW - 1 tiles of the "next" chunk, re-read from chunk 0 into slots nobody reads again - so that every counted
    // wait sees the same pieces in flight; they are drained in front of the epilogue.
    int slot = 0, nslot = NSW - 1;
    int xs = 0, xsn = NXS - 1;            // window stage of chunk ch; stage the window issued at this chunk's first step goes to
    for (int ch = 0; ch < nchunks; ++ch) {
        const unsigned xa_ch = xa0 + xs * (XST * 4);
        const int chn = ch + 1 < nchunks ? ch + 1 : 0;          // source of the weight tiles a step issues for the next chunk
        const int chx = ch + NXS - 1 < nchunks ? ch + NXS - 1 : 0;      // ... and of the window it issues (NXS - 1 chunks ahead)
        w_static_for<0, P>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr MfTap TP = mf_tap(K, J);
            constexpr int PACC = mf_tap(K, (J + P - 1) % P).acc;        // the accumulator of the held-back pair (previous step)
            constexpr int AH = NSW - 2;                                   // younger weight tiles that may fly
            if constexpr (J == 0) {
                w_wait_vmcnt<AH * WPW>();
#ifdef VB_EXPERIMENTS
                if (!(p.mf_abl & 1))
#endif
                if (act || xoob) { fix_x(xs); LDS_WAIT(0); }
            } else {
                w_wait_vmcnt<AH * WPW + (J <= NSW - 2 ? XPW : 0)>();
            }
            __builtin_amdgcn_s_barrier();
            if constexpr (J == 0) {       // window ch + NXS - 1 -> the stage chunk ch - 1 has just left
#ifdef VB_EXPERIMENTS
                if (!(p.mf_abl & 4))
#endif
                {                const float* src = xbase + (int64_t)chx * GK * p.T_in;
                float* dst = lx + xsn * XST;
#pragma unroll
                for (int i = 0; i < XPW; ++i)
                    __builtin_amdgcn_global_load_lds((w_glb_ptr_t)(src + xsrc[i]), (w_lds_ptr_t)(dst + (wave * XPW + i) * 256), 16, 0, 0);
                }
            }
            if constexpr (J + NSW - 1 < P) issue_w(ch, J + NSW - 1, nslot);
            else issue_w(chn, J + NSW - 1 - P, nslot);
            // ---- one ring step: 8 channel pairs x TM MFMAs into accumulator TP.acc
            {
                const unsigned waddr = wa0 + slot * (WT * 4);
                const unsigned xaddr_a = xa_ch + TP.oa * dil4;
                const unsigned xaddr_b = xa_ch + TP.ob * dil4;
                constexpr int NR = TM + (TP.op == 2 ? 1 : 2);
                float a[3][TM], xa[3], xb[3];
                auto fload = [&](auto kc) {
                    constexpr int KK = decltype(kc)::value, S = KK % 3;
                    w_static_for<0, TM>([&](auto ic) { constexpr int I = decltype(ic)::value; lds_rd32<(2 * KK * CO_TILE + I * 32) * 4>(a[S][I], waddr); });
                    lds_rd32<(2 * KK * XP) * 4>(xa[S], xaddr_a);
                    if constexpr (TP.op != 2) lds_rd32<(2 * KK * XP) * 4>(xb[S], xaddr_b);
                };
                fload(std::integral_constant<int, 0>{});
                fload(std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i) MFMA_ACC(acc[PACC][i], da[i], dbv);
                __builtin_amdgcn_sched_barrier(0);
                w_static_for<0, GK / 2>([&](auto kc) {
                    constexpr int KK = decltype(kc)::value, S = KK % 3;
                    if constexpr (KK + 1 < GK / 2) LDS_WAIT(NR); else LDS_WAIT(0);
#pragma unroll
                    for (int i = 0; i < TM; ++i) lds_pin(a[S][i]);
                    lds_pin(xa[S]);
                    if constexpr (TP.op != 2) lds_pin(xb[S]);
                    if constexpr (KK + 2 < GK / 2) fload(std::integral_constant<int, KK + 2>{});
                    float bv;
                    if constexpr (TP.op == 0) bv = xa[S] - xb[S];
                    else if constexpr (TP.op == 1) bv = xa[S] + xb[S];
                    else bv = xa[S];
                    if constexpr (KK + 1 < GK / 2) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) MFMA_ACC(acc[TP.acc][i], a[S][i], bv);
                    } else {
#pragma unroll
                        for (int i = 0; i < TM; ++i) da[i] = a[S][i];
                        dbv = bv;
                    }
                });
            }
            if (++slot == NSW) slot = 0;
            if (++nslot == NSW) nslot = 0;
        });
        if (++xs == NXS) xs = 0;
        if (++xsn == NXS) xsn = 0;
    }
    w_wait_vmcnt<0>();                   // the loads the last chunk issued for a chunk that does not exist
    {
        constexpr int LACC = mf_tap(K, P - 1).acc;
#pragma unroll
        for (int i = 0; i < TM; ++i) MFMA_ACC(acc[LACC][i], da[i], dbv);
    }
    __syncthreads();                     // the rings are free: they hold the four wave-private staging patches now
#ifdef VB_EXPERIMENTS
    if (p.mf_abl & 2) {                  // timing only: no epilogue
        float sum = 0.f;
        for (int a = 0; a < 4; ++a) for (int i = 0; i < TM; ++i) for (int r = 0; r < 16; ++r) sum += acc[a][i][r];
        if (sum == 1.2345e-33f) p.out[0] = sum;
        return;
    }
#endif

    // ---- epilogue: y0 = m0 + m1 + m2, y1 = m1 - m2 - m3 (fixed order), staged through a wave-private patch [32 co][2 VW positions] so that
    // bias / residual / accumulate-into / output move as 16-byte lane accesses; conv_out_value is the direct kernels' arithmetic
    {
#pragma clang fp contract(off)
        // lane-derived values are recomputed here instead of being kept alive (or spilled) across the main loop
        int tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int lane2 = tid2 & 63, g2 = lane2 >> 5, l31b = lane2 & 31;
        const bool vok = l31b < VW;
        const int vq2 = (vok ? l31b : 0) / dil;
        const int bpos2 = 2 * dil * vq2 + ((vok ? l31b : 0) - vq2 * dil);
        float* patch = w_lds + wave * (32 * MF_PITCH);
        const int PW = 2 * VW;
        const int rr = lane2 >> 4, t4 = (lane2 & 15) * 4;
        const int nb = n0 + wn * PW + t4;
        const bool tok = t4 < PW && nb < n_count;          // (PW % 4 == 0, n_count % 4 == 0: quads never straddle an end)
        float* ob = p.out + (int64_t)b * p.out_bstride;
        const float* rb = p.res ? p.res + (int64_t)b * p.res_bstride : nullptr;
        const bool has_old = p.beta != 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int cb = co0 + (wm * TM + i) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y0 = (acc[0][i][r] + acc[1][i][r]) + acc[2][i][r];
                const float y1 = (acc[1][i][r] - acc[2][i][r]) - acc[3][i][r];
                float* row = patch + (4 * g2 + 8 * (r >> 2) + (r & 3)) * MF_PITCH + bpos2;
                if (vok) { row[0] = y0; row[dil] = y1; }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                float4 q[4], rv[4], ov[4];
                float bvv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int co = cb + rr + 4 * (kh * 4 + k);
                    const bool ok = tok && co < p.Co;
                    const int64_t oi = (int64_t)(ok ? co : 0) * p.T_out + (ok ? nb : 0);
                    rv[k] = (ok && rb) ? *reinterpret_cast<const float4*>(rb + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
                    ov[k] = (ok && has_old) ? *reinterpret_cast<const float4*>(ob + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
                    bvv[k] = (ok && p.bias) ? p.bias[co] : 0.f;
                    q[k] = *reinterpret_cast<const float4*>(patch + (rr + 4 * (kh * 4 + k)) * MF_PITCH + t4);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    q[k].x = conv_out_value(p, q[k].x, bvv[k], rv[k].x, ov[k].x); q[k].y = conv_out_value(p, q[k].y, bvv[k], rv[k].y, ov[k].y);
                    q[k].z = conv_out_value(p, q[k].z, bvv[k], rv[k].z, ov[k].z); q[k].w = conv_out_value(p, q[k].w, bvv[k], rv[k].w, ov[k].w);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int co = cb + rr + 4 * (kh * 4 + k);
                    if (tok && co < p.Co) *reinterpret_cast<float4*>(ob + (int64_t)co * p.T_out + nb) = q[k];
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);          // the patch is read before the next tile overwrites it
        }
    }
}

template <int K, int OCC, int WN>
static void launch_w(ConvDev& d, int B, hipStream_t st) {
    constexpr int CO_TILE = (4 / WN) * 64, XP = WN * 64 + 64;
    constexpr int NXS = 2, NSW = 3;      // (the kernel's ring depths)
    constexpr int BYTES = (NXS * GK * XP + NSW * GK * CO_TILE) * (int)sizeof(float);
    const int T_TILE = WN * 2 * ((32 / d.dil) * d.dil);
    d.g_nt = cdiv(d.T_out, T_TILE); d.g_nco = cdiv(d.Co, CO_TILE);
    d.g_ntb = d.g_nt * B; d.g_tbx = cdiv(d.g_ntb, 8);
#ifdef VB_EXPERIMENTS
    d.mf_abl = getenv("VB_F32W_ABL") ? atoi(getenv("VB_F32W_ABL")) : 0;
#endif
    static OnceFlags once;
    vb_set_max_lds_once(once, (const void*)conv1d_f32w_kernel<K, OCC, WN>, BYTES);
    hipLaunchKernelGGL((conv1d_f32w_kernel<K, OCC, WN>), dim3(8 * d.g_tbx * d.g_nco), dim3(256), BYTES, st, d);
}
template <int OCC, int WN>
static void launch_w_k(ConvDev& d, int B, hipStream_t st) {
    switch (d.ntaps) {
        case 3: launch_w<3, OCC, WN>(d, B, st); break;
        case 5: launch_w<5, OCC, WN>(d, B, st); break;
        case 7: launch_w<7, OCC, WN>(d, B, st); break;
        default: launch_w<11, OCC, WN>(d, B, st); break;
    }
}

bool conv1d_f32w_supported(int ksize, int dil) { return (ksize == 3 || ksize == 5 || ksize == 7 || ksize == 11) && dil >= 1 && dil <= 8 && (ksize - 1) * dil <= 60; }
int conv1d_f32w_pseudo_taps(int ksize) { return mf_ntaps(ksize); }

// the caller (launch_conv1d) has checked the conditions it shares with conv1d_f32g_kernel.  Two builds of the same code: <= 168 VGPRs (three
// workgroups per CU, 49 KB of LDS each) and <= 256 (two).  Two per CU measured 1 % faster end to end (1174 against 1163 mel-s/s, same box):
// the default; VB_MF_OCC=3 selects the other (A/B knob)
void launch_conv1d_f32w(ConvDev& d, int B, hipStream_t st) {
    // 64 co x 8 VW tiles where a 128-channel tile would be half empty (Co <= 64) or leave a half-empty last tile (Co % 128 in (0, 64])
    const bool narrow = d.Co <= 64 || (d.Co % 128 != 0 && d.Co % 128 <= 64 && d.Co < 256);
    if (narrow) launch_w_k<2, 4>(d, B, st);          // (five window pieces per wave: 177 VGPRs - the two-per-CU build only)
    else if (vb_tune().conv_mf_occ == 3) launch_w_k<3, 2>(d, B, st);
    else launch_w_k<2, 2>(d, B, st);
}
