// Fused HiFi-GAN ResBlock1 pair in fp32 with F(2,3) minimal filtering, 32 channels (gfx950; round 6).
//
//   out[b][c][t] = beta*out + alpha*( x + b2 + conv2_{k,1}( lrelu( b1 + conv1_{k,d}( lrelu(x) ) ) ) )      (vocoder/hifigan/modules/hifigan.py:27-64)
//
// respair_f32.hip's idea - one workgroup produces a run of output samples of ALL channels, the intermediate never leaves LDS - with
// conv1d_f32w.hip's arithmetic: every 3-tap group of a filter is 4 pseudo-taps for 2 neighbouring outputs of a dilation class (mf_taps.h), the
// four product sums m0..m3 are four MFMA accumulators, y0 = (m0 + m1) + m2 and y1 = (m1 - m2) - m3 are formed once per convolution.  Same chunk ->
// pseudo-tap -> channel-pair order as conv1d_f32w_kernel and the same epilogue arithmetic - but NOT the same bits as two conv1d_f32w launches: the
// even and the odd output of a pair are different sums, and the intermediate run of a workgroup starts (k - 1) / 2 positions in front of its outputs,
// so a position can be the even member here and the odd one there (one ulp on ~1/3 of the elements; both equally close to float64, tests/test_gpu_kernels.py).
// A clip's bits depend on its positions only, never on the batch.
//
// Shape, C = 32.  4 waves, all along time: wave w owns MFMA columns v = 0..31 = 2 x 32 positions.  conv1 (dilation d): column v = (q, r), r < d, owns the
// intermediate positions 2dq + r and 2dq + r + d of the wave's 2 VW, VW = (32 / d) d; the workgroup's intermediate run is M = 8 VW (256 at d = 1, 240 at
// d = 3 / 5).  conv2 (dilation 1): column v owns outputs 2v, 2v + 1 of the wave's 64; the workgroup stores TT = (M - (k - 1)) & ~3 outputs.
// LDS: window ring 2 x [16 ci][320] (both 16-channel chunks of x are requested up front), the intermediate h[32][272] aliases it, weight ring of 3 tiles
// [TPS pseudo-taps][16 ci][32 co] (TPS = 4: 32 MFMAs per ring step and wave; k = 3 has 4 pseudo-taps in all and runs TPS = 2) = 64 KB, two workgroups
// per CU.  The whole schedule - 2 convolutions x 2 chunks x ceil(P / TPS) ring steps - is unrolled: every wait count, tap offset and tile index is an
// immediate.  Ring / fragment pipeline / counted vmcnt as in respair_f32_kernel.
// C = 64: the waves are 2 (time) x 2 (channel halves) - a wave computes 32 of the 64 output channels of its 64 positions in BOTH convolutions, so the
// accumulators stay 4 x 16 registers; runs of M = 4 VW = 128 / 120 intermediate positions, window pitch 192, h[64][144] = 37 KB, tiles of TPS = 2 pseudo-taps
// [2][16][64] = 8 KB: 61 KB, two workgroups per CU; the four 16-channel windows go through the two-stage ring (chunk c + 1 is requested at chunk c's first step).
#include <stdlib.h>
#include <type_traits>

#include "kernels.h"
#include "lds_asm.h"
#include "mf_taps.h"

#define PW_GK 16
#define PW_EP 68             // staged epilogue patch pitch (64 positions + 4)

typedef __attribute__((address_space(3))) void* pw_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* pw_glb_ptr_t;
template <int N> __device__ __forceinline__ void pw_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}
template <int I, int N, class F> __device__ __forceinline__ void pw_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); pw_static_for<I + 1, N>(f); }
}

struct PairWDev {
    const float* x; float* out; int64_t bstride; int T;
    int dil;
    const float* w1; const float* w2;      // minimal-filtering pseudo-taps [P][32 ci][32 co] fp32 each (pack.py:pack_conv_mf)
    const float* b1; const float* b2;
    float slope, alpha, beta;
};

__device__ __forceinline__ float pairw_out_value(const PairWDev& p, float acc, float bias, float res, float old) {
#pragma clang fp contract(off)
    float val = acc + bias;          // conv_out_value (conv1d_dev.h) with acc_scale = 1, no output activation
    val = val + res;
    return fmaf(val, p.alpha, p.beta * old);
}

template <int K, int TPS, int C>
__global__ void __launch_bounds__(256, 2) respair_f32w_kernel(const PairWDev p) {
    constexpr int NSW = 3;
    constexpr int WC = C / 32, WTW = 4 / WC;            // waves along the channels / along time
    constexpr int NCH = C / PW_GK;                      // 16-channel chunks per convolution
    constexpr int XP = WTW * 64 + 64;                   // window pitch: the run + halo (<= 60) + alignment slack (<= 3)
    constexpr int HP = WTW * 64 + 16;                   // intermediate pitch: the run + (k - 1 <= 16)
    constexpr int P = mf_ntaps(K);
    constexpr int KNS = (P + TPS - 1) / TPS;            // ring steps per 16-channel chunk
    constexpr int NT = 2 * NCH * KNS;                   // conv1 chunks, then conv2 chunks
    constexpr int XST = PW_GK * XP;                     // floats per window stage
    constexpr int WT = TPS * PW_GK * C;                 // floats per weight tile [TPS][16 ci][C co]
    constexpr int NPIECE = WT / 256;                    // 1-KB DMA pieces per weight tile
    static_assert(NPIECE % 4 == 0, "every wave issues the same number of pieces");
    constexpr int WPW = NPIECE / 4;
    constexpr int NP = XP / 64, XPW = NP;               // window pieces per wave and chunk (16 rows x XP / 256 floats / 4 waves)
    constexpr int XH = (2 * XST > C * HP ? 2 * XST : C * HP) > 4 * 32 * PW_EP ? (2 * XST > C * HP ? 2 * XST : C * HP) : 4 * 32 * PW_EP;
    static_assert(NT >= NSW, "ring");
    extern __shared__ __attribute__((aligned(16))) float pw_lds[];
    float* lx = pw_lds;                                 // window ring; then h[c][HP]; then the epilogue patches
    float* lw = pw_lds + XH;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave % WTW, wc = wave / WTW;          // this wave's time slot and channel half
    const int g = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int dil = p.dil;
    const int VW = (32 / dil) * dil;
    const int M = WTW * 2 * VW;                         // intermediate positions per workgroup
    const int h2 = (K - 1) / 2, h1 = (K - 1) * dil / 2;
    const int TT = (M - (K - 1)) & ~3;                  // outputs per workgroup
    const int n0 = blockIdx.x * TT;
    const int m0 = n0 - h2;
    const int x0 = m0 - h1;
    const int start_al = x0 & ~3;
    const int aoff = x0 - start_al;
    const float* xb = p.x + (int64_t)b * p.bstride;
    float slope = p.slope;
    asm volatile("v_mov_b32 %0, %0" : "+v"(slope));

    // ---- window DMA: chunk c -> stage c & 1, 16-B lanes, four rows of 64 positions per piece; chunks 0 and 1 now
    unsigned xoob = 0;
    int xsrc[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int ii = wave * XPW + i;
        const int q = ii * 4 + (lane >> 4);
        const int ci = q / NP, pos = (q - ci * NP) * 64 + (lane & 15) * 4;
        const int idx = start_al + pos;
        const bool ok = idx >= 0 && idx < p.T;
        xsrc[i] = ci * p.T + (ok ? idx : 0);
        xoob |= ok ? 0u : (1u << i);
    }
    auto issue_x = [&](int ch) {
#pragma unroll
        for (int i = 0; i < XPW; ++i)
            __builtin_amdgcn_global_load_lds((pw_glb_ptr_t)(xb + (int64_t)ch * PW_GK * p.T + xsrc[i]),
                                             (pw_lds_ptr_t)(lx + (ch & 1) * XST + (wave * XPW + i) * 256), 16, 0, 0);
    };
    issue_x(0);
    issue_x(1);
    auto fix_x = [&](int ch) {        // zero padding + LeakyReLU in place, by the lanes whose own DMA brought the quads
        const unsigned a0 = lds_u32(lx + (ch & 1) * XST + wave * XPW * 256 + lane * 4);
        lds_u32x4 v[XPW];
        const lds_u32x4 zero = {0u, 0u, 0u, 0u};
        pw_static_for<0, XPW>([&](auto ic) { constexpr int I = decltype(ic)::value; lds_rd128<I * 1024>(v[I], a0); });
        LDS_WAIT(0);
        pw_static_for<0, XPW>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            lds_pin(v[I]);
            lds_wr128<I * 1024>(a0, ((xoob >> I) & 1) ? zero : lds_lrelu128_apply(v[I], slope));
        });
    };
    // ---- weight DMA: tile t = (conv, chunk, step): pseudo-taps [s TPS, s TPS + TPS) of 16 input channels; a piece = 256 consecutive floats of the
    // pseudo-tap's [16 ci][C co] block
    constexpr int PPT = PW_GK * C / 256;                // pieces per pseudo-tap
    auto issue_w = [&](auto tc) {
        constexpr int T_ = decltype(tc)::value;
        constexpr int CONV = T_ / (NCH * KNS), CH = (T_ / KNS) % NCH, S = T_ % KNS, SLOT = T_ % NSW;
        const float* wsrc = CONV ? p.w2 : p.w1;
        float* dst = lw + SLOT * WT;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int piece = wave * WPW + i;
            const int tap = piece / PPT, sub = piece - tap * PPT;
            int j = S * TPS + tap;
            if (j > P - 1) j = P - 1;                  // pseudo-taps beyond P: a valid block, never multiplied
            const float* src = wsrc + ((int64_t)j * C + CH * PW_GK) * C + sub * 256 + lane * 4;
            __builtin_amdgcn_global_load_lds((pw_glb_ptr_t)src, (pw_lds_ptr_t)(dst + piece * 256), 16, 0, 0);
        }
    };
    issue_w(std::integral_constant<int, 0>{});
    issue_w(std::integral_constant<int, 1>{});

    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // column v -> the even position inside the wave's range (conv1: dilation classes; conv2: 2 v)
    const int vcol = l31 < VW ? l31 : 0;
    const int vq = vcol / dil;
    const int bpos = 2 * dil * vq + (vcol - vq * dil);
    const unsigned wa0 = lds_u32(lw + g * C + wc * 32 + l31);
    const unsigned xa0 = lds_u32(lx + g * XP + aoff + wt * 2 * VW + bpos);
    const unsigned ha0 = lds_u32(lx + g * HP + wt * 64 + 2 * l31);
    const int dil4 = dil * 4;

    // + b1, LeakyReLU, zero outside [0, T) (conv2 pads the ACTIVATED intermediate) -> h[c][m]; the window ring is dead by then
    auto middle = [&]() {
#pragma clang fp contract(off)
        const int mrel = wt * 2 * VW + bpos;
        const int ma = m0 + mrel;
        const bool ok0 = l31 < VW && ma >= 0 && ma < p.T, ok1 = l31 < VW && ma + dil >= 0 && ma + dil < p.T;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = wc * 32 + 8 * (r >> 2) + 4 * g + (r & 3);
            const float bb = p.b1[c];
            float y0 = (acc[0][r] + acc[1][r]) + acc[2][r];
            float y1 = (acc[1][r] - acc[2][r]) - acc[3][r];
            y0 = y0 + bb; y1 = y1 + bb;
            y0 = y0 > 0.f ? y0 : y0 * p.slope;
            y1 = y1 > 0.f ? y1 : y1 * p.slope;
            if (l31 < VW) {
                lx[c * HP + mrel] = ok0 ? y0 : 0.f;
                lx[c * HP + mrel + dil] = ok1 ? y1 : 0.f;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][r] = 0.f;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };

    // ---- the unrolled schedule.  Issue order: windows 0, 1, tiles 0, 1; then per step [window c + 1 at conv1 chunk c's first step, c >= 1], tile t + 2.
    // In front of step t the loads younger than tile t are: the window step t - 1 issued (if it did) and tile t + 1.
    pw_static_for<0, NT>([&](auto tc) {
        constexpr int T_ = decltype(tc)::value;
        constexpr int CONV = T_ / (NCH * KNS), CH = (T_ / KNS) % NCH, S = T_ % KNS, SLOT = T_ % NSW;
        constexpr int TP_ = T_ > 0 ? T_ - 1 : 0;        // the previous step: did it request a window?
        constexpr bool PREV_WIN = T_ > 0 && (TP_ / (NCH * KNS)) == 0 && (TP_ % KNS) == 0 && ((TP_ / KNS) % NCH) >= 1 && ((TP_ / KNS) % NCH) + 1 < NCH;
        constexpr int AHEAD = (NT - 1 - T_) < (NSW - 2) ? (NT - 1 - T_) : (NSW - 2);       // younger weight tiles that may fly
        pw_wait_vmcnt<AHEAD * WPW + (PREV_WIN ? XPW : 0)>();
        if constexpr (CONV == 0 && S == 0) { fix_x(CH); LDS_WAIT(0); }
        __builtin_amdgcn_s_barrier();          // tile T_ (and its window) landed everywhere; everyone finished step T_ - 1
        if constexpr (CONV == 0 && S == 0 && CH >= 1 && CH + 1 < NCH) issue_x(CH + 1);      // -> the stage chunk CH - 1 has left
        if constexpr (T_ + NSW - 1 < NT) issue_w(std::integral_constant<int, T_ + NSW - 1>{});
        if constexpr (CONV == 1 && CH == 0 && S == 0) middle();
        constexpr int J0 = S * TPS;
        constexpr int NTAP = (P - J0) < TPS ? (P - J0) : TPS;
        constexpr int NM = NTAP * 8;
        const unsigned waddr = wa0 + SLOT * (WT * 4);
        const unsigned baddr = CONV ? ha0 + CH * (PW_GK * HP * 4) : xa0 + (CH & 1) * (XST * 4);
        float fa[3], xa[3], xb2[3];
        auto fload = [&](auto mc) {
            constexpr int MM = decltype(mc)::value, TP = MM / 8, KK = MM % 8, SS = MM % 3;
            constexpr MfTap MT = mf_tap(K, J0 + TP);
            lds_rd32<((TP * PW_GK + 2 * KK) * C) * 4>(fa[SS], waddr);
            if constexpr (CONV) {
                lds_rd32<(2 * KK * HP + MT.oa) * 4>(xa[SS], baddr);
                if constexpr (MT.op != 2) lds_rd32<(2 * KK * HP + MT.ob) * 4>(xb2[SS], baddr);
            } else {
                lds_rd32<(2 * KK * XP) * 4>(xa[SS], baddr + MT.oa * dil4);
                if constexpr (MT.op != 2) lds_rd32<(2 * KK * XP) * 4>(xb2[SS], baddr + MT.ob * dil4);
            }
        };
        fload(std::integral_constant<int, 0>{});
        fload(std::integral_constant<int, 1>{});
        pw_static_for<0, NM>([&](auto mc) {
            constexpr int MM = decltype(mc)::value, SS = MM % 3;
            constexpr MfTap MT = mf_tap(K, J0 + MM / 8);
            if constexpr (MM + 1 < NM) {
                constexpr MfTap MN = mf_tap(K, J0 + (MM + 1) / 8);
                LDS_WAIT(MN.op == 2 ? 2 : 3);
            } else {
                LDS_WAIT(0);
            }
            lds_pin(fa[SS]); lds_pin(xa[SS]);
            if constexpr (MT.op != 2) lds_pin(xb2[SS]);
            if constexpr (MM + 2 < NM) fload(std::integral_constant<int, MM + 2>{});
            float bv;
            if constexpr (MT.op == 0) bv = xa[SS] - xb2[SS];
            else if constexpr (MT.op == 1) bv = xa[SS] + xb2[SS];
            else bv = xa[SS];
            acc[MT.acc] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SS], bv, acc[MT.acc], 0, 0, 0);
        });
    });
    __syncthreads();                     // h is dead: its storage holds the four wave-private staging patches now

    // ---- epilogue: y0 / y1, staged through a wave-private patch [32 co][64 positions], + b2 + residual x, alpha / beta, 16-byte stores
    {
#pragma clang fp contract(off)
        float* patch = lx + wave * (32 * PW_EP);
        float* ob = p.out + (int64_t)b * p.bstride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float y0 = (acc[0][r] + acc[1][r]) + acc[2][r];
            const float y1 = (acc[1][r] - acc[2][r]) - acc[3][r];
            float* row = patch + (4 * g + 8 * (r >> 2) + (r & 3)) * PW_EP + 2 * l31;
            row[0] = y0; row[1] = y1;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int rr = lane >> 4, t4 = (lane & 15) * 4;
        const int nl = wt * 64 + t4;
        const int n = n0 + nl;
        const bool nok = nl < TT && n < p.T;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            float4 v[4], rv[4], ov[4];
            float bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = rr + 4 * (kh * 4 + q), co = wc * 32 + cl;
                v[q] = *reinterpret_cast<const float4*>(patch + cl * PW_EP + t4);
                const int64_t oi = (int64_t)co * p.T + (nok ? n : 0);
                rv[q] = nok ? *reinterpret_cast<const float4*>(xb + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
                ov[q] = (nok && p.beta != 0.f) ? *reinterpret_cast<const float4*>(ob + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[q] = p.b2[co];
            }
            if (nok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = wc * 32 + rr + 4 * (kh * 4 + q);
                    float4 o;
                    o.x = pairw_out_value(p, v[q].x, bv[q], rv[q].x, ov[q].x); o.y = pairw_out_value(p, v[q].y, bv[q], rv[q].y, ov[q].y);
                    o.z = pairw_out_value(p, v[q].z, bv[q], rv[q].z, ov[q].z); o.w = pairw_out_value(p, v[q].w, bv[q], rv[q].w, ov[q].w);
                    *reinterpret_cast<float4*>(ob + (int64_t)co * p.T + n) = o;
                }
            }
        }
    }
}

template <int K, int TPS, int C>
static void launch_pair_w(const PairWDev& d, int B, hipStream_t st) {
    constexpr int WTW = 4 / (C / 32), XP = WTW * 64 + 64, HP = WTW * 64 + 16;
    constexpr int XH = (2 * PW_GK * XP > C * HP ? 2 * PW_GK * XP : C * HP) > 4 * 32 * PW_EP ? (2 * PW_GK * XP > C * HP ? 2 * PW_GK * XP : C * HP) : 4 * 32 * PW_EP;
    constexpr int BYTES = (XH + 3 * TPS * PW_GK * C) * (int)sizeof(float);
    const int VW = (32 / d.dil) * d.dil;
    const int TT = (WTW * 2 * VW - (K - 1)) & ~3;
    static OnceFlags once;
    vb_set_max_lds_once(once, (const void*)respair_f32w_kernel<K, TPS, C>, BYTES);
    hipLaunchKernelGGL((respair_f32w_kernel<K, TPS, C>), dim3(cdiv(d.T, TT), 1, B), dim3(256), BYTES, st, d);
}

bool respair_f32w_supported(const RespairF32Args& a) {
    return (a.C == 32 || a.C == 64) && (a.k == 3 || a.k == 7 || a.k == 11) && a.dil >= 1 && a.dil <= 8 && (a.k - 1) * a.dil <= 60 && a.T % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(a.w1) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w2) & 15) == 0;
}

// a.w1 / a.w2 are the minimal-filtering pseudo-tap weights of the two convolutions ([P][C][C], pack.py:pack_conv_mf)
int launch_respair_f32w(const RespairF32Args& a, hipStream_t st) {
    if (!respair_f32w_supported(a)) VB_FAIL(VB_E_INVALID, "respair_f32w: C=%d k=%d dil=%d T=%d (C 32 / 64, k 3/7/11, (k-1) dil <= 60, T %% 4 == 0, 16-B aligned)", a.C, a.k, a.dil, a.T);
    if (a.x == a.out) VB_FAIL(VB_E_INVALID, "respair_f32w: x and out must be distinct buffers (neighbouring workgroups re-read the halo)");
    PairWDev d;
    d.x = a.x; d.out = a.out; d.bstride = (int64_t)a.C * a.T; d.T = a.T; d.dil = a.dil;
    d.w1 = a.w1; d.w2 = a.w2; d.b1 = a.b1; d.b2 = a.b2; d.slope = a.slope; d.alpha = a.alpha; d.beta = a.beta;
    // executed flops: pseudo-taps / 2 products per output, two convolutions
    ProfScope prof(3, 2.0 * 2.0 * a.B * (double)a.C * a.C * (0.5 * mf_ntaps(a.k)) * (double)a.T,
                   4.0 * a.B * (double)a.C * a.T * (2.0 + (a.beta != 0.f ? 1.0 : 0.0)) + 2.0 * 4.0 * mf_ntaps(a.k) * a.C * a.C, st);
    if (a.C == 32) {
        if (a.k == 3) launch_pair_w<3, 2, 32>(d, a.B, st);
        else if (a.k == 7) launch_pair_w<7, 4, 32>(d, a.B, st);
        else launch_pair_w<11, 4, 32>(d, a.B, st);
    } else {
        if (a.k == 3) launch_pair_w<3, 2, 64>(d, a.B, st);
        else if (a.k == 7) launch_pair_w<7, 2, 64>(d, a.B, st);
        else launch_pair_w<11, 2, 64>(d, a.B, st);
    }
    VB_CHECK_LAUNCH();
    return VB_OK;
}
