// Device-side pieces shared by the convolution kernels (conv1d_f32.hip: register-staged fp32 + split-bf16 kernels; conv1d_f32g.hip: the
// DMA-fed exact-fp32 kernel): the launch descriptor and the two epilogues.
#pragma once
#include "kernels.h"

struct ConvDev {
    const float* x; int64_t x_bstride; int Ci, T_in, x_bmod;
    const float* w; int64_t w_bstride; const float* bias;
    int Co, ntaps, dil, pad, upsample2;
    int in_stride, in_phase;        // the convolution sees xd[i] = x[i * in_stride + in_phase] (strided down-convs as polyphase sums)
    int in_act; float in_slope;
    const float* gn_mean; const float* gn_rstd; const float* gn_gamma; const float* gn_beta; int gn_groups;
    float* out; int64_t out_bstride; int T_out;
    const float* res; int64_t res_bstride;
    float alpha, beta, acc_scale;
    int out_act; float out_slope;
    int out_transposed; const float* add; int64_t add_bstride; int add_bmod;
    int phases, tr_pad;      // phases == 1: ordinary convolution
    const bf16_t* wp; int64_t wp_plane; int Ci_pad;   // split-bf16 weights [2 planes][phase][tap][Co][Ci_pad]
    int64_t wp_bstride;
    const bf16_t* xt; int64_t xt_plane; int xt_Tp;     // pre-activated transposed split planes of the input (XT mode)
    const float* ww;         // conv1d_f32w_kernel: minimal-filtering (F(2,3)) pseudo-tap weights [P][Ci][Co], pack.py:pack_conv_mf
    int stage_epi;           // [b][co][t] output, stride 1, T_out % 4 == 0, 16-B aligned rows: the staged (16-B lane) epilogue
#ifdef VB_EXPERIMENTS
    int old_tail_wait;       // conv1d_f32g: the round-4 wait count in front of a chunk's first tap (A/B of the round-5 fix)
    int x_nt;                // conv1d_f32g: window DMA with the non-temporal policy (VB_CONV_XNT)
    int mf_abl;              // conv1d_f32w: timing-only ablations (VB_F32W_ABL: 1 = no in-place window pass, 2 = no epilogue, 4 = no window DMA after the first)
#endif
    int g_nt, g_nco, g_ntb, g_tbx;   // conv1d_f32g_kernel: time tiles, channel tiles, (time tile, clip, phase) units, units per XCD
};

// One output element of the [b][co][t] epilogues.  The arithmetic is pinned (no implicit contraction, one explicit fma): the direct
// and the staged epilogue - and every tile configuration - must round alike, bit for bit.
__device__ __forceinline__ float conv_out_value(const ConvDev& p, float acc, float bias, float res, float old) {
#pragma clang fp contract(off)
    float val = acc * p.acc_scale;
    val = val + bias;
    val = val + res;
    if (p.out_act == ACT_LRELU) val = val > 0.f ? val : val * p.out_slope;
    else if (p.out_act == ACT_TANH) val = tanhf(val);
    return fmaf(val, p.alpha, p.beta * old);
}

// Staged variant of the [b][co][t] epilogue.  The MFMA accumulator gives a lane ONE output position and 16 channels, so the direct
// epilogue below moves every residual / accumulate-into load and every store as 4-byte lane accesses (two 128-B row pieces per
// wave instruction) - on the narrow, long vocoder layers that epilogue was half of the kernel (tools/conv_bench.py noEpi column:
// 743 -> 367 us at 128 channels, 1075 -> 267 us at 32).  Here each wave passes its 32 x 32 tiles through a PRIVATE 4.5-KB LDS
// patch (no block barrier: only the wave's own writes precede its reads) and comes back with a lane owning 4 consecutive
// positions of one channel: residual, accumulate-into and output move as 16-byte lane accesses, 8 full 128-B lines per wave
// instruction.  Same arithmetic per element, same order: bit-identical to the direct form.
#define CE_PITCH 36          // floats per staged channel row (32 + 4: keeps the 16-B reads aligned, spreads the rows over banks)
// Round 5: the side loads (residual, accumulate-into, bias) of tile u+1 are requested BEFORE tile u's stores are issued.  vmcnt retires a
// wave's memory operations in order: requested behind the stores (rounds 1-4) a tile's loads also waited out the previous tile's store
// round trip - with every workgroup of a one-round launch in its epilogue at the same time that was ~10 % of a VAE layer
// (tools/conv_f32_ablate.py: 756 -> 682 us without the epilogue at 1536 channels).  Same loads, same arithmetic, other issue order.
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_epilogue_staged(const ConvDev& p, f32x16 (&acc)[TM][TN], int b, int n0, int co0, int n_count,
                                                     float* stage) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));        // lane-derived values are recomputed here instead of being kept alive (or spilled) across the caller's main loop
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    float* patch = stage + wave * (32 * CE_PITCH);
    const int rr = lane >> 3, t4 = (lane & 7) * 4;           // read-back: channel row rr + 8k, positions t4 .. t4+3
    float* ob = p.out + (int64_t)b * p.out_bstride;
    const float* rb = p.res ? p.res + (int64_t)b * p.res_bstride : nullptr;
    const bool has_old = p.beta != 0.f;
    struct Side { float4 rv[4], ov[4]; float bv[4]; };
    // tile u = jn * TM + i (the order the tiles are written in)
    auto side_loads = [&](int u, Side& sd) {
        const int jn = u / TM, i = u - jn * TM;
        const int nb = n0 + (wn * TN + jn) * 32, cb = co0 + (wm * TM + i) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = cb + rr + 8 * k, n = nb + t4;
            const bool ok = co < p.Co && n < n_count;        // (n_count % 4 == 0 is a launch condition of this variant)
            const int64_t oi = (int64_t)(ok ? co : 0) * p.T_out + (ok ? n : 0);
            sd.rv[k] = (ok && rb) ? *reinterpret_cast<const float4*>(rb + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
            sd.ov[k] = (ok && has_old) ? *reinterpret_cast<const float4*>(ob + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
            sd.bv[k] = (ok && p.bias) ? p.bias[co] : 0.f;
        }
    };
    Side sd;                             // ONE set of side values: a tile's results are computed first (they replace the staged values), then
    side_loads(0, sd);                   // the set is refilled for the next tile, then the results are stored
#pragma unroll
    for (int u = 0; u < TM * TN; ++u) {
        const int jn = u / TM, i = u - jn * TM;
        const int nb = n0 + (wn * TN + jn) * 32, cb = co0 + (wm * TM + i) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[(4 * g + 8 * (r >> 2) + (r & 3)) * CE_PITCH + l31] = acc[i][jn][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): own writes landed (wave-private patch)
        float4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = *reinterpret_cast<const float4*>(patch + (rr + 8 * k) * CE_PITCH + t4);
        __builtin_amdgcn_s_waitcnt(0xc07f);              // the patch is read before the next tile overwrites it
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            q[k].x = conv_out_value(p, q[k].x, sd.bv[k], sd.rv[k].x, sd.ov[k].x); q[k].y = conv_out_value(p, q[k].y, sd.bv[k], sd.rv[k].y, sd.ov[k].y);
            q[k].z = conv_out_value(p, q[k].z, sd.bv[k], sd.rv[k].z, sd.ov[k].z); q[k].w = conv_out_value(p, q[k].w, sd.bv[k], sd.rv[k].w, sd.ov[k].w);
        }
        if (u + 1 < TM * TN) {
            __builtin_amdgcn_sched_barrier(0);           // (the refill stays behind the arithmetic that reads the set and in front of the stores)
            side_loads(u + 1, sd);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = cb + rr + 8 * k, n = nb + t4;
            if (co < p.Co && n < n_count) *reinterpret_cast<float4*>(ob + (int64_t)co * p.T_out + n) = q[k];
        }
    }
}

template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvDev& p, f32x16 (&acc)[TM][TN], int b, int n0, int co0, int n_count,
                                              int out_stride, int out_off) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    // ---- epilogue: lane owns output position n, 4 consecutive co per accumulator quad.
    // All loads of a 32x32 accumulator tile (residual, accumulate-into, transposed add) are issued BEFORE its first
    // store: vmcnt retires loads and stores in order, so a load issued behind a store would wait for the store's
    // round trip as well - interleaving them serialises 16 memory round trips per tile (measured: 3-4x slower layers).
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + (wn * TN + jn) * 32 + l31;
        const bool nok = n < n_count;
        const int t = n * out_stride + out_off;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int cobase = co0 + (wm * TM + i) * 32 + 4 * g;
            if (p.out_transposed) {
                // out[b][t][co..co+3]   (Co % 4 == 0 enforced at launch)
                float4 ad[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int cob = cobase + 8 * rg;
                    ad[rg] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (nok && cob < p.Co && p.add) {
                        const int ab = p.add_bmod > 0 ? (b % p.add_bmod) : b;
                        ad[rg] = *reinterpret_cast<const float4*>(p.add + (int64_t)ab * p.add_bstride + (int64_t)t * p.Co + cob);
                    }
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int cob = cobase + 8 * rg;
                    if (!nok || cob >= p.Co) continue;
                    float4 o;
                    o.x = acc[i][jn][rg * 4 + 0] * p.acc_scale + (p.bias ? p.bias[cob + 0] : 0.f) + ad[rg].x;
                    o.y = acc[i][jn][rg * 4 + 1] * p.acc_scale + (p.bias ? p.bias[cob + 1] : 0.f) + ad[rg].y;
                    o.z = acc[i][jn][rg * 4 + 2] * p.acc_scale + (p.bias ? p.bias[cob + 2] : 0.f) + ad[rg].z;
                    o.w = acc[i][jn][rg * 4 + 3] * p.acc_scale + (p.bias ? p.bias[cob + 3] : 0.f) + ad[rg].w;
                    *reinterpret_cast<float4*>(p.out + (int64_t)b * p.out_bstride + (int64_t)t * p.Co + cob) = o;
                }
            } else {
                float rv[16], ov[16], bv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cobase + 8 * (r >> 2) + (r & 3);
                    const bool ok = nok && co < p.Co;
                    const int64_t oi = (int64_t)b * p.out_bstride + (int64_t)co * p.T_out + t;
                    rv[r] = (ok && p.res) ? p.res[(int64_t)b * p.res_bstride + (int64_t)co * p.T_out + t] : 0.f;
                    ov[r] = (ok && p.beta != 0.f) ? p.out[oi] : 0.f;
                    bv[r] = (ok && p.bias) ? p.bias[co] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cobase + 8 * (r >> 2) + (r & 3);
                    if (!nok || co >= p.Co) continue;
                    p.out[(int64_t)b * p.out_bstride + (int64_t)co * p.T_out + t] = conv_out_value(p, acc[i][jn][r], bv[r], rv[r], ov[r]);
                }
            }
        }
    }
}


// conv1d_f32g.hip: DMA-fed exact-fp32 kernel (shared weights, Ci % 16 == 0, Co % 4 == 0, in_act none / LeakyReLU, unit input stride,
// halo <= 60, 16-byte aligned rows unless upsample2)
#define GK 16                      // input channels per chunk (= CK: the accumulation order of conv1d_f32_kernel)
void launch_conv1d_f32g(ConvDev& d, int n_count, int B, int upsample2, hipStream_t st);
// conv1d_f32w.hip: the same rings with F(2,3) minimal filtering (k = 3 / 5 / 7 / 11, stride 1): fp32 products, ~1.4-1.5x fewer of them
bool conv1d_f32w_supported(int ksize, int dil);
int conv1d_f32w_pseudo_taps(int ksize);
void launch_conv1d_f32w(ConvDev& d, int B, hipStream_t st);
