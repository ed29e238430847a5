// Fused HiFi-GAN ResBlock1 "pair" in EXACT fp32 (v_mfma_f32_32x32x2_f32) for the narrow, long stages of the generator (gfx950):
//
//   out[b][c][t] = beta*out + alpha*( x + b2 + conv2_{k,dil=1}( lrelu( b1 + conv1_{k,dil=d}( lrelu(x) ) ) ) )
//
// (vocoder/hifigan/modules/hifigan.py:27-64 ResBlock1.forward: xt = lrelu(x); xt = c1(xt); xt = lrelu(xt); xt = c2(xt); x = xt + x -
//  the reference runs this in fp32, vocoder/hifigan/hifigan.py:20-30; BASELINE configs[1] "fp32 vocoder")
//
// As two launches of the fp32 convolution kernel a pair moves five tensor passes (x in, intermediate out and in, residual in, result out;
// 493 MB each at 64 / 32 channels and 8 clips) around 2 x 47 .. 174 GFLOP of f32-MFMA work: the unfused k = 3 convolutions are HBM-bound
// (192 flop per 12 bytes at 32 channels) and every layer pays its staging / epilogue once more.  Here one workgroup produces
// TT = 128 - (k-1) output samples of ALL channels and the intermediate never leaves LDS:
//   * conv1: the raw window of x (128 + (k-1) d positions) arrives by DMA (global_load_lds, 16-B lanes) one 16-channel chunk ahead, the
//     weight tiles [TPS taps][16 ci][C co] of BOTH convolutions stream through one 3- or 4-stage ring (counted vmcnt, one raw s_barrier per
//     ring step = TPS taps of a chunk: 32 / 16 MFMAs per wave at 32 / 64 channels); fragments by inline-asm ds_read_b32 two channel pairs ahead with exact lgkmcnt
//     (lds_asm.h); LeakyReLU is applied in place once per chunk by the lanes that DMA'd it; a wave owns 32 intermediate positions x all channels;
//   * + b1, LeakyReLU, zero outside [0,T) (conv2 pads the ACTIVATED intermediate) -> LDS h[c][m] (aliases the window ring);
//   * conv2 over h, + b2 + residual x, alpha / beta accumulation into the MRF sum through the staged 16-byte epilogue.
// Same chunk -> tap -> channel-pair accumulation order and the same epilogue arithmetic as conv1d_f32_kernel / conv1d_f32g_kernel:
// the fused pair equals the two unfused fp32 launches bit for bit (up to the sign of a zero intermediate) - the test compares them.
#include <stdlib.h>
#include <type_traits>

#include "kernels.h"
#include "lds_asm.h"

#define PF_T 128            // intermediate positions per workgroup (4 waves x 32)
#define PF_XP 192           // window pitch: 128 + halo (<= 60) + alignment slack (<= 3)
#define PF_HP 144           // intermediate pitch: 128 + (k-1 <= 16)
#define PF_GK 16
#define PF_EP 36

typedef __attribute__((address_space(3))) void* pf_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* pf_glb_ptr_t;
template <int N> __device__ __forceinline__ void pf_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}
template <int WPW, int XPW> __device__ __forceinline__ void pf_wait_tile(int ahead, bool xin) {
    if (xin) {
        if (ahead >= 2) pf_wait_vmcnt<2 * WPW + XPW>();
        else if (ahead == 1) pf_wait_vmcnt<WPW + XPW>();
        else pf_wait_vmcnt<XPW>();
    } else {
        if (ahead >= 2) pf_wait_vmcnt<2 * WPW>();
        else if (ahead == 1) pf_wait_vmcnt<WPW>();
        else pf_wait_vmcnt<0>();
    }
}
template <int I, int N, class F> __device__ __forceinline__ void pf_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); pf_static_for<I + 1, N>(f); }
}

struct PairF32Dev {
    const float* x; float* out; int64_t bstride; int T;
    int k, dil;
    const float* w1; const float* w2;      // [k][C ci][C co] fp32 each
    const float* b1; const float* b2;
    float slope, alpha, beta;
    int staged;
#ifdef VB_EXPERIMENTS
    int x_nt;          // window DMA with the non-temporal policy (VB_CONV_XNT, see conv1d_f32g.hip)
#endif
};

// one output element: the arithmetic of conv_out_value (conv1d_f32.hip) with acc_scale = 1 and no output activation
__device__ __forceinline__ float pairf_out_value(const PairF32Dev& p, float acc, float bias, float res, float old) {
#pragma clang fp contract(off)
    float val = acc + bias;
    val = val + res;
    return fmaf(val, p.alpha, p.beta * old);
}

// C = 32*CH channels; a ring step multiplies TPS taps of one 16-channel chunk (at 32 channels a one-tap step is 8 MFMAs = 512 cycles
// between two barriers, and the step's fixed costs - barrier, DMA issue, the first fragment round trip - were 40 % of it: four taps,
// 32 MFMAs; at 64 channels one tap = 16 MFMAs keeps the workgroup at 49 KB of LDS, see launch_respair_f32); NSW ring stages
// ABL (experiments build, timing only - results are wrong): 1 = no fragment reads, 2 = no DMA, 4 = no ring barriers, 8 = no MFMAs,
// 16 = no epilogue, 32 = no intermediate write
// K = kernel size as a template parameter (3 / 7 / 11; 0 = runtime): see the unrolled control flow below
template <int CH, int TPS, int NSW, int ABL = 0, int K = 0>
__global__ void __launch_bounds__(256) respair_f32_kernel(const PairF32Dev p) {
    constexpr int C = 32 * CH;
    constexpr int XST = PF_GK * PF_XP;                 // floats per window stage
    constexpr int WT = TPS * PF_GK * C;                // floats per weight tile: [tap][16 ci][C co], contiguous per tap in global memory
    constexpr int PPT = C / 16;                        // 1-KB DMA pieces per tap
    constexpr int NWI = TPS * PPT;                     // ... per tile
    static_assert(NWI % 4 == 0, "every wave issues the same number of pieces");
    constexpr int WPW = NWI / 4;
    constexpr int NP = PF_XP / 64;
    constexpr int XPW = NP;
    constexpr int XH = (2 * XST > C * PF_HP) ? 2 * XST : C * PF_HP;     // window ring and intermediate share storage
    static_assert(NSW == 3 || NSW == 4, "pf_wait_tile counts at most two tiles ahead");
    static_assert(XH * sizeof(float) >= 4 * 32 * PF_EP * sizeof(float), "staging patches must fit");
    extern __shared__ __attribute__((aligned(16))) float pf_lds[];
    float* lx = pf_lds;                                // conv1: window ring; then h[c][PF_HP]; then the epilogue patches
    float* lw = pf_lds + XH;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int h2 = (p.k - 1) / 2, h1 = (p.k - 1) * p.dil / 2;
    const int TT = (PF_T - (p.k - 1)) & ~3;           // outputs per workgroup (a multiple of 4: 16-B quads)
    const int n0 = blockIdx.x * TT;                   // first output sample
    const int m0 = n0 - h2;                           // first intermediate position
    const int x0 = m0 - h1;                           // first window sample
    const int start_al = x0 & ~3;
    const int aoff = x0 - start_al;
    const float* xb = p.x + (int64_t)b * p.bstride;
    float slope = p.slope;
    asm volatile("v_mov_b32 %0, %0" : "+v"(slope));       // VGPR copy: an SGPR operand makes hipcc re-wait lgkmcnt(0) in front of every use

    int xsrc[XPW];
    unsigned xoob = 0;
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
        if constexpr (ABL & 2) return;
        const float* src = xb + (int64_t)ch * PF_GK * p.T;
        float* dst = lx + (ch & 1) * XST;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
#ifdef VB_EXPERIMENTS
            if (p.x_nt) { __builtin_amdgcn_global_load_lds((pf_glb_ptr_t)(src + xsrc[i]), (pf_lds_ptr_t)(dst + (wave * XPW + i) * 256), 16, 0, 2); continue; }
#endif
            __builtin_amdgcn_global_load_lds((pf_glb_ptr_t)(src + xsrc[i]), (pf_lds_ptr_t)(dst + (wave * XPW + i) * 256), 16, 0, 0);
        }
    };
    // once per chunk, by the lanes that DMA'd the quads (after the wave's own DMA landed, in front of the publishing barrier): zeros over
    // the out-of-range quads and LeakyReLU in place (conv1d_f32g.hip: VALU instructions between a SIMD's MFMAs cost matrix-pipe time; 36 VALU
    // + 6 LDS instructions per thread and chunk instead of 3 per B fragment; inline-asm LDS accesses so that hipcc does not drain the DMA
    // ring with vmcnt(0) in front of them).  The residual is read from global memory, not from this window.
    auto fix_x = [&](int ch) {
        const unsigned a0 = lds_u32(lx + (ch & 1) * XST + wave * XPW * 256 + lane * 4);
        lds_u32x4 v[XPW];
        const lds_u32x4 zero = {0u, 0u, 0u, 0u};
        pf_static_for<0, XPW>([&](auto ic) { constexpr int I = decltype(ic)::value; lds_rd128<I * 1024>(v[I], a0); });
        LDS_WAIT(0);
        pf_static_for<0, XPW>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            lds_pin(v[I]);
            lds_wr128<I * 1024>(a0, ((xoob >> I) & 1) ? zero : lds_lrelu128_apply(v[I], slope));
        });
    };
    // ring tile = (convolution, chunk, step): taps [s TPS, s TPS + TPS) of 16 input channels; a tap's [16][C] block is contiguous
    constexpr int NCH = C / PF_GK;
    const int NS = (p.k + TPS - 1) / TPS;             // steps per chunk
    const int NT1 = NCH * NS, total = 2 * NT1;
    auto issue_w = [&](int conv, int ch, int s, int slot) {
        if constexpr (ABL & 2) return;
        const float* wsrc = conv ? p.w2 : p.w1;
        float* dst = lw + slot * WT;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int piece = wave * WPW + i;
            const int tap = piece / PPT, sub = piece - tap * PPT;
            const int j = min(s * TPS + tap, p.k - 1);          // taps beyond k: a valid block, never multiplied
            const float* src = wsrc + ((int64_t)j * C + ch * PF_GK) * C + sub * 256 + lane * 4;
            __builtin_amdgcn_global_load_lds((pf_glb_ptr_t)src, (pf_lds_ptr_t)(dst + piece * 256), 16, 0, 0);
        }
    };

    f32x16 acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    issue_x(0);
    int nconv = 0, nch = 0, ns = 0;      // tile t + NSW - 1
    auto next_tile = [&]() {
        if (++ns == NS) { ns = 0; if (++nch == NCH) { nch = 0; ++nconv; } }
    };
#pragma unroll
    for (int t = 0; t < NSW - 1; ++t) {
        if (t < total) issue_w(nconv, nch, ns, t);
        next_tile();
    }
    int conv = 0, ch = 0, s = 0;         // tile t
    int slot = 0, nslot = NSW - 1;
    auto pre = [&](int t) {
        const int ahead_all = min(total - 1, t + NSW - 2) - t;
        if (conv == 0 && s == 0) {
            // (the window went out NS steps ago in front of tile t - NS + NSW - 1: see conv1d_f32g.hip for the count; conv1's tiles follow
            //  conv0's, so the end-of-sequence case - ahead_all < NSW - 2 - cannot meet a window here, the form is kept the same anyway)
            const int lag = NSW - 2 - NS;
            pf_wait_tile<WPW, XPW>(ch == 0 || lag <= 0 ? ahead_all : max(ahead_all - lag, 0), false);
            fix_x(ch);
            LDS_WAIT(0);
        } else {
            pf_wait_tile<WPW, XPW>(ahead_all, conv == 0 && ch + 1 < NCH && s <= NSW - 2);
        }
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();    // tile t (and the window) landed everywhere; everyone finished tile t - 1
        if (conv == 0 && s == 0 && ch + 1 < NCH) issue_x(ch + 1);
        if (t + NSW - 1 < total) issue_w(nconv, nch, ns, nslot);
    };
    auto advance = [&]() {
        if (++s == NS) { s = 0; if (++ch == NCH) { ch = 0; ++conv; } }
        next_tile();
        if (++slot == NSW) slot = 0;
        if (++nslot == NSW) nslot = 0;
    };
    // one ring step: up to TPS taps x 8 channel pairs, fragments requested two pairs ahead of their MFMAs (lds_asm.h), three
    // register sets so that a request never lands in registers an MFMA in flight still reads
    // (waddr: this lane's byte address in the step's weight tile; b0: ... of the step's first tap in the window / intermediate; bstep: bytes per tap)
    auto body = [&](auto convc, auto ntc, const unsigned waddr, const unsigned b0, const unsigned bstep) {
        constexpr int CONV = decltype(convc)::value, NM = decltype(ntc)::value * 8;       // channel pairs of this step: compile-time,
        constexpr int PITCH = CONV ? PF_HP : PF_XP;                                          // so every wait count below is an immediate
        unsigned baddr[TPS];
#pragma unroll
        for (int tp = 0; tp < TPS; ++tp) baddr[tp] = b0 + tp * bstep;
        float fa[3][CH], fb[3];
        if constexpr (ABL & 1) {
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
                for (int i = 0; i < CH; ++i) fa[s3][i] = (float)lane;
                fb[s3] = (float)(lane + s3);
            }
        }
        auto fload = [&](auto mc) {
            constexpr int M = decltype(mc)::value, TP = M / 8, KK = M % 8, S = M % 3;
            if constexpr (ABL & 1) return;
            pf_static_for<0, CH>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                lds_rd32<((TP * PF_GK + 2 * KK) * C + I * 32) * 4>(fa[S][I], waddr);
            });
            lds_rd32<2 * KK * PITCH * 4>(fb[S], baddr[TP]);
        };
        fload(std::integral_constant<int, 0>{});
        fload(std::integral_constant<int, 1>{});
        pf_static_for<0, NM>([&](auto mc) {
            constexpr int M = decltype(mc)::value, S = M % 3;
            if constexpr (M + 1 < NM) LDS_WAIT(CH + 1); else LDS_WAIT(0);
#pragma unroll
            for (int i = 0; i < CH; ++i) lds_pin(fa[S][i]);
            lds_pin(fb[S]);
            if constexpr (M + 2 < NM) fload(std::integral_constant<int, M + 2>{});
            const float bv = fb[S];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if constexpr (ABL & 8) acc[i][0] += fa[S][i] * bv;
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S][i], bv, acc[i], 0, 0, 0);
            }
        });
    };
    auto compute = [&](auto convc) {
        constexpr int CONV = decltype(convc)::value;
        const int ntap = min(TPS, p.k - s * TPS);       // taps of this step (the last step of a chunk may hold fewer)
        const unsigned waddr = lds_u32(lw + slot * WT + l31 + g * C);
        const unsigned b0 = CONV ? lds_u32(lx + (ch * PF_GK + g) * PF_HP + s * TPS + 32 * wave + l31)
                                 : lds_u32(lx + (ch & 1) * XST + g * PF_XP + aoff + s * TPS * p.dil + 32 * wave + l31);
        pf_static_for<1, TPS + 1>([&](auto ntc) {
            if (ntap == decltype(ntc)::value) body(convc, ntc, waddr, b0, CONV ? 4u : (unsigned)(p.dil * 4));
        });
    };
    // + b1, LeakyReLU, zero outside [0,T), to h[c][m] (the window ring is dead: every wave is past conv1's last tile once it has crossed
    // the barrier of the first conv2 tile's step); a second barrier publishes it
    auto middle = [&]() {
        const int m = m0 + 32 * wave + l31;
        const bool inr = m >= 0 && m < p.T;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + 8 * (r >> 2) + 4 * g + (r & 3);
                float v = acc[i][r] + p.b1[c];
                v = fmaxf(v, v * p.slope);
                if constexpr (!(ABL & 32)) lx[c * PF_HP + 32 * wave + l31] = inr ? v : 0.f;
                else if (v == 1.2345e-33f) lx[c] = v;
                acc[i][r] = 0.f;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };
    if constexpr (K > 0) {
        // ---- kernel size as a template parameter (the launcher checks p.k == K): a chunk's NS steps are unrolled, every wait count, tap
        // offset and "which tile to request" is an immediate, and a step's control flow is the ring-slot update alone.  The runtime
        // loop below spends ~50 scalar instructions and ~10 branches per step on it - at 16-32 MFMAs per step that costs up to 18 % of
        // the matrix pipe (tools/probe/f32_loop_probe: 147 -> 121 TF/s at 16 MFMAs per step).  Same tiles in the same order: bit-identical.
        constexpr int KNS = (K + TPS - 1) / TPS;
        static_assert(KNS >= NSW - 1, "a step requests the tile NSW - 1 ahead: it lies in this chunk or the next one");
        const unsigned wa0 = lds_u32(lw + l31 + g * C);
        const unsigned xa0 = lds_u32(lx + g * PF_XP + aoff + 32 * wave + l31);
        const unsigned ha0 = lds_u32(lx + g * PF_HP + 32 * wave + l31);
        const unsigned dil4 = (unsigned)(p.dil * 4);
        int slot_b = 0, nslot = NSW - 1;                 // byte offset of tile t's ring slot; slot index of tile t + NSW - 1
        auto chunk = [&](auto convc, auto lastc, auto firstc, const int ch) {
            constexpr int CONV = decltype(convc)::value;
            constexpr bool LASTCH = decltype(lastc)::value, FIRST1 = decltype(firstc)::value;      // last chunk of this convolution; conv2's first chunk
            const unsigned bch = CONV ? ha0 + ch * (PF_GK * PF_HP * 4) : xa0 + (ch & 1) * (XST * 4);
            pf_static_for<0, KNS>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                constexpr int AH = (CONV == 1 && LASTCH) ? (KNS - 1 - S < NSW - 2 ? KNS - 1 - S : NSW - 2) : NSW - 2;   // younger weight tiles that may fly
                if constexpr (CONV == 0 && S == 0) {
                    pf_wait_vmcnt<(AH < KNS ? AH : KNS) * WPW>();
                    fix_x(ch);
                    LDS_WAIT(0);
                } else {
                    pf_wait_vmcnt<AH * WPW + ((CONV == 0 && !LASTCH && S <= NSW - 2) ? XPW : 0)>();
                }
                if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
                if constexpr (CONV == 0 && S == 0 && !LASTCH) issue_x(ch + 1);
                constexpr int SN = S + NSW - 1;
                if constexpr (SN < KNS) issue_w(CONV, ch, SN, nslot);
                else if constexpr (!LASTCH) issue_w(CONV, ch + 1, SN - KNS, nslot);
                else if constexpr (CONV == 0) issue_w(1, 0, SN - KNS, nslot);
                if constexpr (FIRST1 && S == 0) middle();
                constexpr int NTAP = (K - S * TPS) < TPS ? (K - S * TPS) : TPS;
                body(convc, std::integral_constant<int, NTAP>{}, wa0 + slot_b, bch + (CONV ? (unsigned)(S * TPS * 4) : (unsigned)(S * TPS) * dil4),
                     CONV ? 4u : dil4);
                slot_b = slot_b == (NSW - 1) * WT * 4 ? 0 : slot_b + WT * 4;
                if (++nslot == NSW) nslot = 0;
            });
        };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        for (int c = 0; c < NCH - 1; ++c) chunk(C0{}, std::false_type{}, std::false_type{}, c);
        chunk(C0{}, std::true_type{}, std::false_type{}, NCH - 1);
        chunk(C1{}, std::false_type{}, std::true_type{}, 0);
        for (int c = 1; c < NCH - 1; ++c) chunk(C1{}, std::false_type{}, std::false_type{}, c);
        chunk(C1{}, std::true_type{}, std::false_type{}, NCH - 1);
    } else {

    // ---- conv1 (dilated) over the activated window -> intermediate positions m0 + [0,128)
    int t = 0;
    for (; t < NT1; ++t) {
        pre(t);
        compute(std::integral_constant<int, 0>{});
        advance();
    }
    // ---- between the convolutions
    pre(t);
    middle();
    // ---- conv2 (dil 1) over the intermediate -> outputs n0 + [0,TT)
    compute(std::integral_constant<int, 1>{});
    advance();
    for (++t; t < total; ++t) {
        pre(t);
        compute(std::integral_constant<int, 1>{});
        advance();
    }
    }
    __syncthreads();                     // h is dead: its storage holds the four wave-private staging patches now

    // ---- epilogue (same two forms as respair_x3_kernel)
    float* ob = p.out + (int64_t)b * p.bstride;
    if constexpr (ABL & 16) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[i][r];
        if (sum == 1.2345e-33f) ob[0] = sum;
        return;
    }
    if (p.staged) {
        float* patch = lx + wave * (32 * PF_EP);
        const int rr = lane >> 3, t4 = (lane & 7) * 4;
        const int nl = 32 * wave + t4;
        const int n = n0 + nl;
        const bool nok = nl < TT && n < p.T;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[(4 * g + 8 * (r >> 2) + (r & 3)) * PF_EP + l31] = acc[i][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            float4 v[4], rv[4], ov[4];
            float bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = i * 32 + rr + 8 * q;
                v[q] = *reinterpret_cast<const float4*>(patch + (rr + 8 * q) * PF_EP + t4);
                const int64_t oi = (int64_t)co * p.T + (nok ? n : 0);
                rv[q] = nok ? *reinterpret_cast<const float4*>(xb + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
                ov[q] = (nok && p.beta != 0.f) ? *reinterpret_cast<const float4*>(ob + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[q] = p.b2[co];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (nok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = i * 32 + rr + 8 * q;
                    const float a4[4] = {v[q].x, v[q].y, v[q].z, v[q].w}, r4[4] = {rv[q].x, rv[q].y, rv[q].z, rv[q].w};
                    const float o4[4] = {ov[q].x, ov[q].y, ov[q].z, ov[q].w};
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pairf_out_value(p, a4[e], bv[q], r4[e], o4[e]);
                    *reinterpret_cast<float4*>(ob + (int64_t)co * p.T + n) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    } else {
        const int nl = 32 * wave + l31;
        const int n = n0 + nl;
        const bool nok = nl < TT && n < p.T;
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
                ob[(int64_t)co * p.T + n] = pairf_out_value(p, acc[i][r], bv[r], rv[r], ov[r]);
            }
        }
    }
}

template <int CH, int TPS, int NSW, int ABL = 0, int K = 0>
static void launch_pair_f32(const PairF32Dev& d, dim3 grid, hipStream_t st) {
    constexpr int C = 32 * CH, XST = PF_GK * PF_XP;
    constexpr int XH = (2 * XST > C * PF_HP) ? 2 * XST : C * PF_HP;
    constexpr int BYTES = (XH + NSW * TPS * PF_GK * C) * (int)sizeof(float);
    static OnceFlags once;
    vb_set_max_lds_once(once, (const void*)respair_f32_kernel<CH, TPS, NSW, ABL, K>, BYTES);
    hipLaunchKernelGGL((respair_f32_kernel<CH, TPS, NSW, ABL, K>), grid, dim3(256), BYTES, st, d);
}

bool respair_f32_supported(const RespairF32Args& a) {
    return (a.C == 32 || a.C == 64 || a.C == 128) && a.k >= 1 && (a.k & 1) && a.k <= 17 && (a.k - 1) * a.dil <= 60 && a.T % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w1) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(a.w2) & 15) == 0;
}

int launch_respair_f32(const RespairF32Args& a, hipStream_t st) {
    if (!respair_f32_supported(a)) VB_FAIL(VB_E_INVALID, "respair_f32: C=%d k=%d dil=%d T=%d (C 32/64/128, odd k <= 17, (k-1) dil <= 60, T %% 4 == 0, 16-B aligned)", a.C, a.k, a.dil, a.T);
    if (a.x == a.out) VB_FAIL(VB_E_INVALID, "respair_f32: x and out must be distinct buffers (neighbouring workgroups re-read the halo)");
    PairF32Dev d;
    d.x = a.x; d.out = a.out; d.bstride = (int64_t)a.C * a.T; d.T = a.T; d.k = a.k; d.dil = a.dil;
    d.w1 = a.w1; d.w2 = a.w2; d.b1 = a.b1; d.b2 = a.b2; d.slope = a.slope; d.alpha = a.alpha; d.beta = a.beta;
    const int TT = (PF_T - (a.k - 1)) & ~3;
    d.staged = ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && !vb_tune().conv_direct_epi) ? 1 : 0;
#ifdef VB_EXPERIMENTS
    d.x_nt = getenv("VB_CONV_XNT") ? 1 : 0;
#endif
    dim3 grid(cdiv(a.T, TT), 1, a.B);
    ProfScope prof(3, 2.0 * 2.0 * a.B * (double)a.C * a.C * a.k * (double)a.T,
                   4.0 * a.B * (double)a.C * a.T * (2.0 + (a.beta != 0.f ? 1.0 : 0.0)) + 2.0 * 4.0 * a.k * a.C * a.C, st);
    // 32 channels: 4 taps per ring step, 3 stages (49 KB: three workgroups per CU).  64 channels: ONE tap per step, 3 stages - 49 KB, three
    // workgroups per CU: 16 MFMAs per step and wave instead of 32, but a third wave per SIMD: 2046 -> 1787 us at k = 7, 3181 -> 2799 at k = 11,
    // 994 -> 850 at k = 3 (8 clips; two taps / 4 stages = 70 KB = two per CU was the first shape; four per CU at 32 channels measured
    // 1 % slower: tools/pair_cfg_bench.py).  128 channels: 1 tap, 106 KB, one per CU - not used by the builder.
#ifdef VB_EXPERIMENTS
    if (const char* e = getenv("VB_PAIRF_ABL")) {          // timing-only ablations (tools/conv_f32_ablate.py)
        const int v = atoi(e);
        bool done = true;
        auto go = [&](auto ablc) {
            constexpr int A = decltype(ablc)::value;
            if (a.C == 32) launch_pair_f32<1, 4, 3, A>(d, grid, st); else if (a.C == 64) launch_pair_f32<2, 2, 4, A>(d, grid, st); else done = false;
        };
        switch (v) {
            case 1: go(std::integral_constant<int, 1>{}); break;
            case 2: go(std::integral_constant<int, 2>{}); break;
            case 4: go(std::integral_constant<int, 4>{}); break;
            case 8: go(std::integral_constant<int, 8>{}); break;
            case 16: go(std::integral_constant<int, 16>{}); break;
            case 32: go(std::integral_constant<int, 32>{}); break;
            case 48: go(std::integral_constant<int, 48>{}); break;
            case 7: go(std::integral_constant<int, 7>{}); break;
            default: done = false;
        }
        if (done) { VB_CHECK_LAUNCH(); return VB_OK; }
    }
#endif
#ifdef VB_EXPERIMENTS
    if (const char* e = getenv("VB_PAIRF_CFG")) {           // ring shapes measured and not adopted (tools/conv_f32_ablate.py)
        const int v = atoi(e);
        if (a.C == 64 && v == 1) { launch_pair_f32<2, 1, 4>(d, grid, st); VB_CHECK_LAUNCH(); return VB_OK; }     // 53 KB: three workgroups per CU, one tap per step
        if (a.C == 64 && v == 2) { launch_pair_f32<2, 1, 3>(d, grid, st); VB_CHECK_LAUNCH(); return VB_OK; }     // 49 KB
        if (a.C == 32 && v == 3) { launch_pair_f32<1, 2, 4>(d, grid, st); VB_CHECK_LAUNCH(); return VB_OK; }
        if (a.C == 32 && v == 4) { launch_pair_f32<1, 2, 3>(d, grid, st); VB_CHECK_LAUNCH(); return VB_OK; }     // 36.5 KB: four workgroups per CU
        if (a.C == 64 && v == 5) { launch_pair_f32<2, 2, 4>(d, grid, st); VB_CHECK_LAUNCH(); return VB_OK; }     // round-4 shape: 70 KB, two per CU
    }
#endif
    // the generator's kernel sizes run the unrolled control flow (template parameter K) where a chunk has at least NSW - 1 steps;
    // VB_CONV_F32_RT_TAPS=1 keeps the runtime loop (the bit-identity test and the A/B)
    const int kk = vb_tune().conv_f32_rt_taps ? 0 : a.k;
    if (a.C == 32) {
        if (kk == 7) launch_pair_f32<1, 4, 3, 0, 7>(d, grid, st);
        else if (kk == 11) launch_pair_f32<1, 4, 3, 0, 11>(d, grid, st);
        else launch_pair_f32<1, 4, 3>(d, grid, st);            // (k = 3 is one step per chunk at four taps per step)
    } else if (a.C == 64) {
        if (kk == 3) launch_pair_f32<2, 1, 3, 0, 3>(d, grid, st);
        else if (kk == 7) launch_pair_f32<2, 1, 3, 0, 7>(d, grid, st);
        else if (kk == 11) launch_pair_f32<2, 1, 3, 0, 11>(d, grid, st);
        else launch_pair_f32<2, 1, 3>(d, grid, st);
    } else launch_pair_f32<4, 1, 4>(d, grid, st);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
