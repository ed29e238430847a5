// DMA-fed exact-fp32 Conv1d / ConvTranspose1d (v_mfma_f32_32x32x2_f32) - its own translation unit because it is built with
// -mllvm -amdgpu-mfma-vgpr-form (accumulators stay in VGPRs: the asm-pipelined loop below otherwise gets its accumulators copied between
// VGPRs and AGPRs around every ring step); see versband_amd/build.py.
#include <stdlib.h>
#include <type_traits>

#include "conv1d_dev.h"
#include "lds_asm.h"

template <int I, int N, class F> __device__ __forceinline__ void g_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); g_static_for<I + 1, N>(f); }
}

// ---------------------------------------------------------------------------------------------------------
// DMA-fed exact-fp32 kernel (round 4).  conv1d_f32_kernel above multiplies at 80-93 % of the f32 MFMA rate PER TAP (measured at 8
// clips: +54 us per tap at 32 channels = 146 TF/s, +114 at 64, +125 at 128) but every layer pays 650-800 us before its first tap and
// after its last one: the window of every 16-channel chunk goes global -> registers -> transform -> LDS with nothing in flight, weights
// go through registers once per tap, and the epilogue moves 4-byte lane accesses - the vocoder's 72 ResBlock convolutions spend as long
// there as in their MFMAs (profiles/r04_open_fp32voc_kernel_stats.csv).  This kernel keeps the arithmetic - same 16-channel chunks,
// chunk -> tap -> channel-pair order, one v_mfma_f32_32x32x2_f32 per pair: BIT-IDENTICAL results - and changes how operands arrive:
//   * the raw window of a chunk is DMA'd (global_load_lds, 16 B per lane) into a two-stage LDS ring one chunk ahead; LeakyReLU (in place)
//     and zero padding by the lanes that own the quads, once their own DMA has landed and in front of the barrier that publishes the chunk; GroupNorm + swish inputs are pre-activated by gn_apply_kernel (the builder
//     emits it in fp32 mode: the old kernel redid norm + swish + expf once per output-channel tile, 12x on the 1536-channel layers);
//   * weight tiles [16 ci][CO_TILE] stream through a 4-stage ring, three tiles in flight, counted vmcnt + one raw s_barrier per tap;
//   * the [b][co][t] result leaves through the staged 16-byte epilogue of the split-bf16 kernel (conv_epilogue_staged);
//   * workgroups are numbered so that one XCD keeps a (time tile, clip) unit for ALL its output-channel tiles: the window is fetched
//     into one L2, the (small) weights into all of them.
// Conditions (launch_conv1d falls back to conv1d_f32_kernel otherwise): shared or per-clip fp32 weights (16-B aligned), Ci % 16 == 0, Co % 4 == 0, in_act none / LeakyReLU,
// unit input stride, halo <= 60, 16-byte aligned rows (T_in % 4 == 0) unless the input is upsampled (UPS: 4-byte DMA pieces).
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* g_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* g_glb_ptr_t;
template <int N> __device__ __forceinline__ void g_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}
// counted wait in front of ring step t: `ahead` (<= 2) younger weight tiles (+ one window chunk when xin) may stay in flight
template <int WPW, int XPW> __device__ __forceinline__ void g_wait_tile(int ahead, bool xin) {
    if (xin) {
        if (ahead >= 2) g_wait_vmcnt<2 * WPW + XPW>();
        else if (ahead == 1) g_wait_vmcnt<WPW + XPW>();
        else g_wait_vmcnt<XPW>();
    } else {
        if (ahead >= 2) g_wait_vmcnt<2 * WPW>();
        else if (ahead == 1) g_wait_vmcnt<WPW>();
        else g_wait_vmcnt<0>();
    }
}

// NSW = weight-tile ring stages (4: three tiles in flight; 3: two - 49 KB of LDS with the 96-sample tile, three workgroups per CU)
// ABL (experiments build, timing only - results are wrong): 1 = no fragment reads, 2 = no DMA, 4 = no barrier, 8 = no MFMAs, 16 = no epilogue
// NT = taps per phase as a template parameter (0: runtime loop)
template <int WM, int WN, int TM, int TN, bool UPS, int NSW, int ABL = 0, int NT = 0>
__global__ void __launch_bounds__(256, 3) conv1d_f32g_kernel(const ConvDev p) {      // three waves per SIMD: <= 168 VGPRs (the unrolled-tap instances of the 128 x 128 tile came out at 171)
    constexpr int CO_TILE = WM * TM * 32;
    constexpr int T_TILE = WN * TN * 32;
    constexpr int XP = (T_TILE + 64 + 63) / 64 * 64;      // window pitch (positions): tile + halo (<= 60) + alignment slack (<= 3)
    constexpr int NP = XP / 64;
    constexpr int XST = GK * XP, WT = GK * CO_TILE;       // floats per window stage / weight tile
    constexpr int NWI = CO_TILE / 16;                     // 1-KB DMA pieces per weight tile
    constexpr int WPW = NWI >= 4 ? NWI / 4 : 1;           // ... per wave (narrow tiles: the waves repeat each other's pieces)
    constexpr int XPW = UPS ? 4 * NP : NP;                // window pieces per wave and chunk (16-B lanes: 4 rows of 64 positions per piece)
    static_assert(NSW == 3 || NSW == 4, "g_wait_tile counts at most two tiles ahead");
    static_assert((2 * XST + NSW * WT) * sizeof(float) >= 4 * 32 * CE_PITCH * sizeof(float), "staging patches must fit in the rings (contiguous)");
    extern __shared__ __attribute__((aligned(16))) float g_lds[];
    float* lx = g_lds;
    float* lw = g_lds + 2 * XST;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    // block -> (channel tile, unit): XCD x = L & 7 serves the units x, x + 8, ... for every channel tile
    int b, ph, n0, co0;
    {
        const int L = blockIdx.x, j = L >> 3;
        const int ct = j / p.g_tbx, ul = j - ct * p.g_tbx;
        const int u = ul * 8 + (L & 7);
        if (u >= p.g_ntb) return;
        const int z = u / p.g_nt;
        n0 = (u - z * p.g_nt) * T_TILE;
        b = z / p.phases; ph = z - b * p.phases;
        co0 = ct * CO_TILE;
    }
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

    const int T_eff = UPS ? 2 * p.T_in : p.T_in;
    const int start = n0 + in_off;
    const int start_al = UPS ? start : (start & ~3);      // 16-B pieces start on a quad of the row (rows are 16-B aligned)
    const int aoff = start - start_al;
    const float* xbase = p.x + (int64_t)b * p.x_bstride;
    const float* wbase = p.w + (int64_t)b * p.w_bstride + (int64_t)ph * p.ntaps * p.Ci * p.Co;      // (w_bstride: per-clip operands, VAE attention)
    float slope = p.in_act == ACT_LRELU ? p.in_slope : 1.f;            // max(v, 1 v) = v: no branch in the fragment path
    asm volatile("v_mov_b32 %0, %0" : "+v"(slope));                     // (kept in a VGPR: as an SGPR operand hipcc re-waits lgkmcnt(0) for
                                                                        //  its s_load in front of every use inside the loop)

    // ---- window DMA: piece i of this wave -> lane's source offset inside the clip (chunk 0) and whether it lies inside [0, T_eff)
    int xsrc[XPW];
    unsigned xoob = 0;
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int ii = wave * XPW + i;
        int ci, pos;
        if constexpr (UPS) { ci = ii / NP; pos = (ii - ci * NP) * 64 + lane; }
        else { const int q = ii * 4 + (lane >> 4); ci = q / NP; pos = (q - ci * NP) * 64 + (lane & 15) * 4; }
        const int idx = start_al + pos;
        const bool ok = idx >= 0 && idx < T_eff;           // (quads never straddle an end: T_eff % 4 == 0)
        xsrc[i] = ci * p.T_in + (ok ? (UPS ? (idx >> 1) : idx) : 0);
        xoob |= ok ? 0u : (1u << i);                       // per lane: a bit per piece
    }
    auto issue_x = [&](int ch) {
        if constexpr (ABL & 2) return;
        const float* src = xbase + (int64_t)ch * GK * p.T_in;
        float* dst = lx + (ch & 1) * XST;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int ii = wave * XPW + i;
#ifdef VB_EXPERIMENTS
            // (VB_CONV_XNT=1: the window - read by one or two workgroups - with the non-temporal policy, so that it does not displace the weights
            //  every workgroup re-reads from L2; A/B of round 5, profiles/r05_conv_window_nt.txt)
            if (p.x_nt) {
                if constexpr (UPS) __builtin_amdgcn_global_load_lds((g_glb_ptr_t)(src + xsrc[i]), (g_lds_ptr_t)(dst + ii * 64), 4, 0, 2);
                else __builtin_amdgcn_global_load_lds((g_glb_ptr_t)(src + xsrc[i]), (g_lds_ptr_t)(dst + ii * 256), 16, 0, 2);
                continue;
            }
#endif
            if constexpr (UPS) __builtin_amdgcn_global_load_lds((g_glb_ptr_t)(src + xsrc[i]), (g_lds_ptr_t)(dst + ii * 64), 4, 0, 0);
            else __builtin_amdgcn_global_load_lds((g_glb_ptr_t)(src + xsrc[i]), (g_lds_ptr_t)(dst + ii * 256), 16, 0, 0);
        }
    };
    // Once per chunk, by the lanes that DMA'd the quads, after the wave's own DMA has landed and in front of the barrier that publishes the
    // chunk: zeros over the out-of-range quads (padding) and LeakyReLU IN PLACE - not on the B fragments inside the MFMA loop: v_mul +
    // 2 v_max per fragment were 6 VALU instructions per 4 MFMAs, and VALU instructions issued between a SIMD's MFMAs cost matrix-pipe
    // time (tools/probe/f32_loop_probe: 149 -> 136 TF/s with them; here 36 VALU + 6 LDS instructions per thread and chunk replace 48 per
    // tap).  Same operation on the same values: bit-identical.  The LDS accesses come from inline asm: an ordinary one makes hipcc drain
    // the DMA ring with vmcnt(0) in front of it (lds_asm.h); the wave has waited for exactly these pieces itself.
    const bool act = p.in_act == ACT_LRELU;
    auto fix_x = [&](int ch) {
        if constexpr (UPS) {
            const unsigned a0 = lds_u32(lx + (ch & 1) * XST + wave * XPW * 64 + lane);
            float v[XPW];
            if (act) {
                g_static_for<0, XPW>([&](auto ic) { constexpr int I = decltype(ic)::value; lds_rd32<I * 256>(v[I], a0); });
                LDS_WAIT(0);
            }
            g_static_for<0, XPW>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                const bool oob = (xoob >> I) & 1;
                if (act) { lds_pin(v[I]); lds_wr32<I * 256>(a0, oob ? 0.f : fmaxf(v[I], v[I] * slope)); }
                else if (oob) lds_wr32<I * 256>(a0, 0.f);
            });
        } else {
            const unsigned a0 = lds_u32(lx + (ch & 1) * XST + wave * XPW * 256 + lane * 4);
            lds_u32x4 v[XPW];
            const lds_u32x4 zero = {0u, 0u, 0u, 0u};
            if (act) {
                g_static_for<0, XPW>([&](auto ic) { constexpr int I = decltype(ic)::value; lds_rd128<I * 1024>(v[I], a0); });
                LDS_WAIT(0);
            }
            g_static_for<0, XPW>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                const bool oob = (xoob >> I) & 1;
                if (act) { lds_pin(v[I]); lds_wr128<I * 1024>(a0, oob ? zero : lds_lrelu128_apply(v[I], slope)); }
                else if (oob) lds_wr128<I * 1024>(a0, zero);
            });
        }
    };
    // ---- weight DMA: tile (chunk, tap) = 16 rows of CO_TILE floats, 1-KB pieces of 256 / CO_TILE rows
    constexpr int RPI = 256 / CO_TILE, LPR = CO_TILE / 4;
    int wsrc[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int ii = (wave * WPW + i) % NWI;
        const int row = ii * RPI + lane / LPR;
        int cog = co0 + (lane % LPR) * 4;
        if (cog >= p.Co) cog = 0;                           // channels beyond Co: any valid quad, the rows are never stored
        wsrc[i] = row * p.Co + cog;
    }
    auto issue_w = [&](int ch, int j, int slot) {
        if constexpr (ABL & 2) return;
        const float* src = wbase + ((int64_t)j * p.Ci + ch * GK) * p.Co;
        float* dst = lw + slot * WT;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int ii = (wave * WPW + i) % NWI;
            __builtin_amdgcn_global_load_lds((g_glb_ptr_t)(src + wsrc[i]), (g_lds_ptr_t)(dst + ii * 256), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = p.Ci / GK;
    const int total = nchunks * p.ntaps;
    // prologue: window of chunk 0, then the first NSW - 1 weight tiles (issue order = retirement order)
    issue_x(0);
    int nch = 0, nj = 0;                 // tile t + NSW - 1
#pragma unroll
    for (int t = 0; t < NSW - 1; ++t) {
        if (t < total) issue_w(nch, nj, t);
        if (++nj == p.ntaps) { nj = 0; ++nch; }
    }
    float da[TM], dbv[TN];               // held-back operands of a step's last channel pair
#pragma unroll
    for (int i = 0; i < TM; ++i) da[i] = 0.f;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) dbv[jn] = 0.f;
    // one ring step's multiplications: fragments are requested two channel pairs ahead of their MFMAs from inline asm with exact lgkmcnt
    // waits (lds_asm.h: with LDS-DMA in flight hipcc only ever waits lgkmcnt(0), which exposed one LDS round trip per two pairs); three
    // register sets so that a request never lands in registers an MFMA in flight still reads
    auto multiply = [&](const unsigned xaddr, const unsigned waddr) {
        float a[3][TM], bb[3][TN];
        if constexpr (ABL & 1) {
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[s3][i] = (float)lane;
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) bb[s3][jn] = (float)(lane + s3);
            }
        }
        auto fload = [&](auto kc) {
            constexpr int KK = decltype(kc)::value, S = KK % 3;
            if constexpr (ABL & 1) return;
            g_static_for<0, TM>([&](auto ic) { constexpr int I = decltype(ic)::value; lds_rd32<(2 * KK * CO_TILE + I * 32) * 4>(a[S][I], waddr); });
            g_static_for<0, TN>([&](auto jc) { constexpr int J = decltype(jc)::value; lds_rd32<(2 * KK * XP + J * 32) * 4>(bb[S][J], xaddr); });
        };
        fload(std::integral_constant<int, 0>{});
        fload(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        // the LAST channel pair of the previous step was held back (operands in registers): its MFMAs are queued here, behind the barrier,
        // the DMA issue and the first two fragment requests of this step - 256 matrix-pipe cycles that cover the first LDS round trip
        // (same box: 810 -> 798 us on the 128 x 128 tile; at three workgroups per CU most of it was hidden already).
        // Same accumulation order (they still precede this step's first pair); the first step queues zeros.
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                if constexpr (ABL & 8) acc[i][jn][0] += da[i] * dbv[jn];
                else acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[i], dbv[jn], acc[i][jn], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        g_static_for<0, GK / 2>([&](auto kc) {
            constexpr int KK = decltype(kc)::value, S = KK % 3;
            if constexpr (KK + 1 < GK / 2) LDS_WAIT(TM + TN); else LDS_WAIT(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) lds_pin(a[S][i]);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) lds_pin(bb[S][jn]);
            if constexpr (KK + 2 < GK / 2) fload(std::integral_constant<int, KK + 2>{});
            if constexpr (KK + 1 < GK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
                        if constexpr (ABL & 8) acc[i][jn][0] += a[S][i] * bb[S][jn];
                        else acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S][i], bb[S][jn], acc[i][jn], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) da[i] = a[S][i];
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) dbv[jn] = bb[S][jn];
            }
        });
    };
    if constexpr (NT > 0) {
        // ---- the tap count is a template parameter (3 / 7 / 11: the generator's ResBlock layers; the launcher checks p.ntaps == NT): the
        // taps of a chunk are unrolled, every wait count is an immediate and a step's control flow is the ring-slot update alone.  The
        // runtime-tap loop below spends ~50 scalar instructions and ~10 branches per step on it, and scalar work between a barrier and
        // the step's first MFMA costs matrix-pipe time (tools/probe/f32_loop_probe: 151 -> 142 TF/s for ~50 dependent scalar
        // instructions per 32 MFMAs).  Same tiles in the same order: bit-identical.
        static_assert(NT >= NSW - 1, "a step issues the tile NSW - 1 ahead: it lies in this chunk or the next one");
        // (the same bound keeps the J == 0 wait below right: min(AH, NT) * WPW equals the runtime loop's (ahead_all - lag) form only while
        //  NT >= NSW - 2, i.e. lag <= 0 - an NT = 1 / 2 instance on the 4-stage ring would bring back the round-5 tail race)
        static_assert(NT >= NSW - 2, "unrolled-tap wait count assumes no lag between the window and its first weight tile");
        const unsigned xa0 = lds_u32(lx + aoff + wn * TN * 32 + l31 + g * XP);
        const unsigned wa0 = lds_u32(lw + wm * TM * 32 + l31 + g * CO_TILE);
        const int dil4 = p.dil * 4;
        int slot = 0, nslot = NSW - 1;
        for (int ch = 0; ch < nchunks; ++ch) {
            const unsigned xa_ch = xa0 + (ch & 1) * (XST * 4);
            auto chunk = [&](auto lastc) {
                constexpr bool LAST = decltype(lastc)::value;
                g_static_for<0, NT>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    constexpr int AH = LAST ? (NT - 1 - J < NSW - 2 ? NT - 1 - J : NSW - 2) : NSW - 2;      // younger weight tiles that may fly
                    if constexpr (J == 0) {
                        g_wait_vmcnt<(AH < NT ? AH : NT) * WPW>();
                        if (act || xoob) { fix_x(ch); LDS_WAIT(0); }
                    } else {
                        g_wait_vmcnt<AH * WPW + ((!LAST && J <= NSW - 2) ? XPW : 0)>();
                    }
                    if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
                    if constexpr (J == 0 && !LAST) issue_x(ch + 1);
                    if constexpr (J + NSW - 1 < NT) issue_w(ch, J + NSW - 1, nslot);
                    else if constexpr (!LAST) issue_w(ch + 1, J + NSW - 1 - NT, nslot);
                    multiply(xa_ch + J * dil4, wa0 + slot * (WT * 4));
                    if (++slot == NSW) slot = 0;
                    if (++nslot == NSW) nslot = 0;
                });
            };
            if (ch + 1 == nchunks) chunk(std::true_type{}); else chunk(std::false_type{});
        }
    } else {
    int ch = 0, j = 0;                   // tile t = (ch, j)
    int slot = 0, nslot = NSW - 1;
    for (int t = 0; t < total; ++t) {
        const int ahead_all = min(total - 1, t + NSW - 2) - t;
        if (j == 0) {
            // the chunk's window must have landed: it was issued at step t - ntaps in front of tile (t - ntaps + NSW - 1), so of the
            // ahead_all tiles younger than tile t only those from that one on may fly: ahead_all - (NSW - 2 - ntaps) of them when the
            // chunk has fewer taps than the ring runs ahead.  (Round 5: this read min(ahead_all, ntaps), which near the END of a 1-tap
            // layer on the 4-stage ring - no weight tile left to issue behind the window - let the window's last piece fly: rare wrong
            // tiles beside a second GPU process, profiles/r05_conv_tail_race.txt.)
            const int lag = NSW - 2 - p.ntaps;
#ifdef VB_EXPERIMENTS
            if (p.old_tail_wait) g_wait_tile<WPW, XPW>(ch == 0 ? ahead_all : min(ahead_all, p.ntaps), false);      // the round-4 count (tools/flake_conv.py)
            else
#endif
            g_wait_tile<WPW, XPW>(ch == 0 || lag <= 0 ? ahead_all : max(ahead_all - lag, 0), false);
            if (act || xoob) { fix_x(ch); LDS_WAIT(0); }
        } else {
            // window ch + 1 was issued at this chunk's first tap, in front of tile (t - j + NSW - 1): younger than tile t while j <= NSW - 2
            g_wait_tile<WPW, XPW>(ahead_all, ch + 1 < nchunks && j <= NSW - 2);
        }
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();    // tile t (and the window) landed everywhere; everyone finished tile t - 1
        if (j == 0 && ch + 1 < nchunks) issue_x(ch + 1);
        if (t + NSW - 1 < total) issue_w(nch, nj, nslot);
        multiply(lds_u32(lx + (ch & 1) * XST + aoff + j * p.dil + wn * TN * 32 + l31 + g * XP),
                 lds_u32(lw + slot * WT + wm * TM * 32 + l31 + g * CO_TILE));
        if (++j == p.ntaps) { j = 0; ++ch; }
        if (++nj == p.ntaps) { nj = 0; ++nch; }
        if (++slot == NSW) slot = 0;
        if (++nslot == NSW) nslot = 0;
    }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[i], dbv[jn], acc[i][jn], 0, 0, 0);       // the last step's held-back pair
    __syncthreads();                     // the window ring is free: it holds the four wave-private staging patches now
    if constexpr (ABL & 16) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][jn][r];
        if (sum == 1.2345e-33f) p.out[0] = sum;
        return;
    }
    if (p.stage_epi) conv_epilogue_staged<WM, WN, TM, TN>(p, acc, b, n0, co0, n_count, lx);
    else conv_epilogue<WM, WN, TM, TN>(p, acc, b, n0, co0, n_count, out_stride, out_off);
}

template <int WM, int WN, int TM, int TN, bool UPS, int NSW = 4, int ABL = 0, int NT = 0>
static void launch_cfg_g(ConvDev& d, int n_count, int B, hipStream_t st) {
    constexpr int CO_TILE = WM * TM * 32, T_TILE = WN * TN * 32, XP = (T_TILE + 64 + 63) / 64 * 64;
    constexpr int BYTES = (2 * GK * XP + NSW * GK * CO_TILE) * (int)sizeof(float);
    d.g_nt = cdiv(n_count, T_TILE); d.g_nco = cdiv(d.Co, CO_TILE);
    d.g_ntb = d.g_nt * B * d.phases; d.g_tbx = cdiv(d.g_ntb, 8);
    static OnceFlags once;
    vb_set_max_lds_once(once, (const void*)conv1d_f32g_kernel<WM, WN, TM, TN, UPS, NSW, ABL, NT>, BYTES);
    int bytes = BYTES;
#ifdef VB_EXPERIMENTS
    if (const char* e = getenv("VB_F32G_LDSPAD")) bytes += atoi(e);      // unused LDS: fewer workgroups per CU (tools/flake_conv.py; <= 64 KB in all)
    if (getenv("VB_F32G_NOSTAGE")) d.stage_epi = 0;
    d.old_tail_wait = getenv("VB_F32G_OLDWAIT") ? 1 : 0;
    d.x_nt = getenv("VB_CONV_XNT") ? 1 : 0;
#endif
    hipLaunchKernelGGL((conv1d_f32g_kernel<WM, WN, TM, TN, UPS, NSW, ABL, NT>), dim3(8 * d.g_tbx * d.g_nco), dim3(256), bytes, st, d);
}
// Tile choice for wide layers (Co > 64).  Every configuration walks chunks, taps and channel pairs in the same order - the choice never
// changes a bit of the result - so it is free to follow the launch's size: the busiest CU's load (workgroups on it x tile area) over a
// per-tile efficiency (MFMAs between two barriers: 32 / 24 / 16 / 8 per wave).  8 clips: the vocoder's 128 / 256-channel layers take
// 128 x 128, the 1536 / 768-channel VAE layers 128 x 96 (768 workgroups = one round at three per CU); one or two clips: the VAE layers make
// 48-96 workgroups of those tiles for 256 CUs and take 64 x 128 or 64 x 64 instead (one clip: fp32 conv class 14.6 -> 12.1 ms per pass).
static int g_pick_tile(int n_count, int Co, int B, int phases) {
#ifdef VB_EXPERIMENTS
    if (const char* e = getenv("VB_F32G_PICK")) return atoi(e);       // force a tile configuration (tools/flake_hunt_vae.py)
#endif
    struct Cand { int id, co, t; double eff; };
    static const Cand cands[4] = {{0, 128, 128, 1.00}, {1, 128, 96, 0.95}, {2, 64, 128, 0.85}, {3, 64, 64, 0.70}};
    int best = 0;
    double best_cost = 1e30;
    for (const Cand& c : cands) {
        const int64_t wgs = (int64_t)cdiv(n_count, c.t) * cdiv(Co, c.co) * B * phases;
        const double cost = (double)cdiv(wgs, 256) * c.co * c.t / c.eff;
        if (cost < best_cost * 0.999) { best_cost = cost; best = c.id; }
    }
    return best;
}

// the ResBlock tap counts run the loop with its taps unrolled (template parameter NT); VB_CONV_F32_RT_TAPS=1 keeps the runtime-tap loop
// (the bit-identity test and the A/B)
template <int WM, int WN, int TM, int TN, int NSW>
static void launch_cfg_g_taps(ConvDev& d, int n_count, int B, hipStream_t st) {
    const int nt = vb_tune().conv_f32_rt_taps ? 0 : d.ntaps;
    if (nt == 3) launch_cfg_g<WM, WN, TM, TN, false, NSW, 0, 3>(d, n_count, B, st);
    else if (nt == 7) launch_cfg_g<WM, WN, TM, TN, false, NSW, 0, 7>(d, n_count, B, st);
    else if (nt == 11) launch_cfg_g<WM, WN, TM, TN, false, NSW, 0, 11>(d, n_count, B, st);
    else launch_cfg_g<WM, WN, TM, TN, false, NSW>(d, n_count, B, st);
}
// picks the tile and launches; the caller (launch_conv1d) has checked the kernel's conditions
void launch_conv1d_f32g(ConvDev& d, int n_count, int B, int upsample2, hipStream_t st) {
    if (upsample2) {
        const int64_t w128 = (int64_t)cdiv(n_count, 128) * cdiv(d.Co, 128) * B, w96 = (int64_t)cdiv(n_count, 96) * cdiv(d.Co, 128) * B;
        if (cdiv(w96, 256) * 96 < cdiv(w128, 256) * 128) launch_cfg_g<4, 1, 1, 3, true, 3>(d, n_count, B, st);
        else launch_cfg_g<2, 2, 2, 2, true, 3>(d, n_count, B, st);
    } else if (d.Co > 64) {
#ifdef VB_EXPERIMENTS
        if (const char* e = getenv("VB_F32G_TILE")) {       // tile shapes measured and not adopted
            if (atoi(e) == 2) { launch_cfg_g_taps<2, 2, 1, 2, 3>(d, n_count, B, st); return; }      // 64 x 128, 3 stages: 36.5 KB, four workgroups per CU
            if (atoi(e) == 3) { launch_cfg_g_taps<1, 4, 2, 1, 3>(d, n_count, B, st); return; }      // 64 x 128 as 1 x 4 waves of 2 x 1 tiles
        }
#endif
        switch (g_pick_tile(n_count, d.Co, B, d.phases)) {
            case 1:
#ifdef VB_EXPERIMENTS
                if (const char* e = getenv("VB_F32G_ABL")) {      // timing-only ablations of the 128 x 96 tile (the VAE's layers; runtime-tap loop)
                    switch (atoi(e)) {
                        case 1: launch_cfg_g<4, 1, 1, 3, false, 3, 1>(d, n_count, B, st); return;
                        case 2: launch_cfg_g<4, 1, 1, 3, false, 3, 2>(d, n_count, B, st); return;
                        case 4: launch_cfg_g<4, 1, 1, 3, false, 3, 4>(d, n_count, B, st); return;
                        case 8: launch_cfg_g<4, 1, 1, 3, false, 3, 8>(d, n_count, B, st); return;
                        case 16: launch_cfg_g<4, 1, 1, 3, false, 3, 16>(d, n_count, B, st); return;
                        case 3: launch_cfg_g<4, 1, 1, 3, false, 3, 3>(d, n_count, B, st); return;
                        case 7: launch_cfg_g<4, 1, 1, 3, false, 3, 7>(d, n_count, B, st); return;
                        case 6: launch_cfg_g<4, 1, 1, 3, false, 3, 6>(d, n_count, B, st); return;
                        case 100: launch_cfg_g_taps<2, 2, 2, 2, 3>(d, n_count, B, st); return;      // the 128 x 128 tile on this launch
                        default: break;
                    }
                }
#endif
                launch_cfg_g_taps<4, 1, 1, 3, 3>(d, n_count, B, st); break;
            case 2: launch_cfg_g<2, 2, 1, 2, false, 4>(d, n_count, B, st); break;
            case 3: launch_cfg_g<2, 2, 1, 1, false, 3>(d, n_count, B, st); break;
#ifdef VB_EXPERIMENTS
            case 4: launch_cfg_g<2, 2, 1, 2, false, 3>(d, n_count, B, st); break;      // (VB_F32G_PICK only) 64 x 128 on a 3-stage weight ring
            case 5: launch_cfg_g<1, 4, 1, 2, false, 4>(d, n_count, B, st); break;      // (VB_F32G_PICK only) 32 x 256
#endif
            default:
#ifdef VB_EXPERIMENTS
                if (const char* e = getenv("VB_F32G_ABL")) {      // timing-only ablations of the 128 x 128 tile (tools/conv_f32_ablate.py)
                    switch (atoi(e)) {
                        case 1: launch_cfg_g<2, 2, 2, 2, false, 3, 1>(d, n_count, B, st); return;
                        case 2: launch_cfg_g<2, 2, 2, 2, false, 3, 2>(d, n_count, B, st); return;
                        case 4: launch_cfg_g<2, 2, 2, 2, false, 3, 4>(d, n_count, B, st); return;
                        case 8: launch_cfg_g<2, 2, 2, 2, false, 3, 8>(d, n_count, B, st); return;
                        case 16: launch_cfg_g<2, 2, 2, 2, false, 3, 16>(d, n_count, B, st); return;
                        case 3: launch_cfg_g<2, 2, 2, 2, false, 3, 3>(d, n_count, B, st); return;
                        case 7: launch_cfg_g<2, 2, 2, 2, false, 3, 7>(d, n_count, B, st); return;
                        case 6: launch_cfg_g<2, 2, 2, 2, false, 3, 6>(d, n_count, B, st); return;
                        default: break;
                    }
                }
#endif
                launch_cfg_g_taps<2, 2, 2, 2, 3>(d, n_count, B, st);     // 3-stage ring: 49 KB, three workgroups per CU (31.3 -> 30.4 ms per pass against 4 stages / two)
        }
    } else if (d.Co > 32) launch_cfg_g<2, 2, 1, 2, false>(d, n_count, B, st);
    else launch_cfg_g<1, 4, 1, 2, false>(d, n_count, B, st);
}
