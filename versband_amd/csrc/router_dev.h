// Device side of the Band-MoE router, shared by router_kernel (elementwise.hip) and the fused score + router kernel
// (score_router.hip): one wave decides RT_TPW consecutive tokens.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------
// Gumbel noise generator for the production path: G = -log(-log(1-u)), u from splitmix64
// keyed by (seed, stream, element) - independent of launch geometry and world size.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// Gumbel draw of element (token t, slot e) of stream (seed, clip, nfe, branch, block, gate): counter based, independent
// of launch geometry, batch slot and world size
__device__ __forceinline__ float gumbel_draw(uint64_t seed, int64_t clip, int nfe, int branch, int block, int gate, int t, int width, int e) {
    uint64_t key = splitmix64(seed ^ splitmix64((uint64_t)clip * 0x9E3779B97F4A7C15ull + 0x1234567ull));
    key = splitmix64(key + (((uint64_t)nfe * 2 + branch) << 20) + ((uint64_t)block << 8) + (uint64_t)gate);
    uint64_t z = splitmix64(key + ((uint64_t)t * width + e + 1) * 0x9E3779B97F4A7C15ull);
    float u = (float)((z >> 40) + 1) * (1.0f / 16777218.0f);      // (0,1)
    float ex = fmaxf(-log1pf(-u), 1e-30f);
    return -logf(ex);
}

// Sum EE per-lane partials over the wave with a halving butterfly: exchanging with lane^32 a lane keeps half of the
// values, with lane^16 a quarter, ...; the last value is then summed over the remaining lane bits.  Returns, on lane e,
// the full sum of value e (EE + log2(64/EE) - 1 shuffles instead of 6*EE).
template <int EE, int PE>
__device__ __forceinline__ float reduce_logits(const float (&part)[PE], int lane) {
    float v[EE];
#pragma unroll
    for (int e = 0; e < EE; ++e) v[e] = part[e];
    int off = 32;
#pragma unroll
    for (int width = EE; width > 1; width >>= 1) {
        const int half = width >> 1;
        const bool upper = (lane & off) != 0;
#pragma unroll
        for (int j = 0; j < half; ++j) {
            const float send = upper ? v[j] : v[j + half];
            const float keep = upper ? v[j + half] : v[j];
            v[j] = keep + __shfl_xor(send, off, 64);
        }
        off >>= 1;
    }
    float r = v[0];
    for (; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
    // value e sits on the lanes whose top log2(EE) bits spell e (bit 5 = most significant): fetch it to lane e
    int src = 0;
#pragma unroll
    for (int bit = 0, o = 32, w = EE; w > 1; w >>= 1, o >>= 1, ++bit) {
        const int nb = (EE == 4) ? 2 : 3;
        if ((lane >> (nb - 1 - bit)) & 1) src += o;
    }
    return __shfl(r, src, 64);
}

// ---------------------------------------------------------------------------
// Band-MoE router (vocal2music_moe.py:132-151): one wave per token.
//   lc = cq . Wg^T + bg ; ic = argmax(lc + G2) ; ia = argmax(la + G3) (first maximum wins, like
//   torch.max) ; (m_c, m_a) = softmax(hl[b] + G1).   G* are Gumbel draws (-log Exp(1)).
// ---------------------------------------------------------------------------
#define RT_TPW_MAX 4  // tokens per wave (4 when the launch fills the chip anyway; 1 for small batches: 4x the waves, a quarter of the latency)
// SC = true: "folded" caption gate.  The token features are not materialised at all: `sc` holds the token's attention
// SCORES against its clip's caption keys for all heads ([N][NS], NS = L * Hh, column = key * Hh + head; scale, q-projection
// and q-bias already inside - one grouped GEMM against per-clip folded keys), `Wg` holds per clip VW[key*Hh+head][e] =
// value_row(head) . (gate weight row e restricted to the head), so   logit_e = sum_heads sum_keys softmax(scores)_key VW_e.
struct RouterDev {
    Planes cq; const float* Wg; const float* bg; const float* la; int la_rows; const float* hl; int hl_ld;
    const float* g1; const float* g2; const float* g3; int N, T, D, E; int* ic; int* ia; float* mc; float* ma; float* lc_out; int B;
    uint64_t seed; int64_t clip_base; int nfe_base; const int* step; int block; const float* sc; int NS, Hh;
    // bucket counts as a side product (round 5): cnt[(n / RT_CNT_BLOCK) * cnt_G + group] += 1 for the token's expert pair (cnt_pairs) or its two
    // expert groups - what bucket_count_kernel computed in a launch of its own; the table must be zero on entry (launch_bucket's place kernel
    // clears the table of the NEXT launch).  Integer atomics: the sums do not depend on their order.
    int* cnt; int cnt_G; int cnt_pairs;
};
#define RT_CNT_BLOCK 256      // = BK_T of the bucket kernels (elementwise.hip)
// Phase B of the router for the RT_TPW tokens n0 .. of one wave: noise draws, arg-max, high-level gate.  A token needs 2E+2 "slots"
// (E caption-gate, E acoustic-gate, 2 high-level-gate values): PP tokens are laid side by side in the wave (SPT = 64/PP lanes each), so
// the counter-based noise generator, the index arithmetic and the arg-max loops run once per PP tokens.  logit_of(t0, tokq, sl) returns
// the caption-gate logit (before the bias) of token n0 + t0 + tokq and expert sl; it is called by ALL lanes (it may shuffle).
template <int PP, int RT_TPW, typename LogitFn>
__device__ __forceinline__ void router_phase_b(const RouterDev& a, const int n0, const int N, LogitFn&& logit_of) {
    const float* __restrict__ bg = a.bg; const float* __restrict__ la = a.la;
    const int la_rows = a.la_rows; const float* __restrict__ hl = a.hl; const int hl_ld = a.hl_ld;
    const float* __restrict__ g1 = a.g1; const float* __restrict__ g2 = a.g2; const float* __restrict__ g3 = a.g3;
    const int T = a.T, E = a.E, B = a.B, block = a.block;
    int* ic = a.ic; int* ia = a.ia; float* mc = a.mc; float* ma = a.ma; float* lc_out = a.lc_out;
    uint64_t seed = a.seed; int64_t clip_base = a.clip_base; int nfe_base = a.nfe_base; const int* step = a.step;
    const int lane = threadIdx.x & 63;
    constexpr int SPT = 64 / PP;
    const bool gen = g1 == nullptr;
    if (step) {
        // sampler path: the noise key lives in the device-side parameter block behind the step counter (launch_sampler_params), so
        // a captured graph of the step loop can be replayed for another seed / clip base without re-capturing
        const long long* prm = reinterpret_cast<const long long*>(step + 4);
        seed = (uint64_t)prm[0]; clip_base = prm[1]; nfe_base = (int)prm[2];
    }
    const int nfe = nfe_base + (step ? *step : 0);
    const int tokq = lane / SPT, sl = lane % SPT, lbase = lane - sl;
#pragma unroll
    for (int t0 = 0; t0 < RT_TPW; t0 += PP) {
        const int n = n0 + t0 + tokq;
        const bool valid = n < N;
        const int nn = valid ? n : N - 1;
        const int bb = nn / T, tt = nn - bb * T;
        const int branch = bb / B;
        const int64_t clip = clip_base + (bb - branch * B);
        const int gate = sl < E ? 1 : (sl < 2 * E ? 2 : 0);
        const int slot = sl < E ? sl : (sl < 2 * E ? sl - E : sl - 2 * E);
        // this lane's side input: caption gate bias / acoustic gate logit / high-level gate logit; and its noise value
        float sv = 0.f, nz = 0.f;
        if (sl < 2 * E + 2) {
            if (gate == 1) sv = bg[slot];
            else if (gate == 2) sv = la[(int64_t)(nn % la_rows) * E + slot];
            else sv = hl[bb * hl_ld + slot];
            if (gen) nz = gumbel_draw(seed, clip, nfe, branch, block, gate, tt, gate == 0 ? 2 : E, slot);
            else nz = gate == 1 ? g2[(int64_t)nn * E + slot] : (gate == 2 ? g3[(int64_t)nn * E + slot] : g1[(int64_t)nn * 2 + slot]);
        }
        const float mylogit = logit_of(t0, tokq, sl);     // caption-gate logit of this lane's (token, expert) before the bias; lanes with sl >= E: unused
        const float zl = (sl < E ? mylogit : 0.f) + sv;      // gate logit of this lane's (token, gate, slot)
        if (lc_out && valid && sl < E) lc_out[(int64_t)n * E + sl] = zl;
        const float z = zl + nz;
        float best = -INFINITY, bz = -INFINITY; int bi = 0, ba = 0;
        for (int e = 0; e < E; ++e) {
            const float zc = __shfl(z, lbase + e, 64), za = __shfl(z, lbase + E + e, 64);
            if (zc > best) { best = zc; bi = e; }
            if (za > bz) { bz = za; ba = e; }
        }
        const float z0 = __shfl(z, lbase + 2 * E, 64), z1 = __shfl(z, lbase + 2 * E + 1, 64);
        if (valid && sl == 0) {
            ic[n] = bi;
            ia[n] = ba;
            if (a.cnt) {
                int* row = a.cnt + (n / RT_CNT_BLOCK) * a.cnt_G;
                if (a.cnt_pairs) atomicAdd(row + bi * E + ba, 1);
                else { atomicAdd(row + bi, 1); atomicAdd(row + E + ba, 1); }
            }
            const float m = fmaxf(z0, z1);
            const float e0 = expf(z0 - m), e1 = expf(z1 - m);
            const float inv = 1.f / (e0 + e1);
            mc[n] = e0 * inv;
            ma[n] = e1 * inv;
        }
    }
}

// One wave, tokens n0 .. n0 + RT_TPW - 1 (n0 < N).  `N` bounds the tokens this call may touch (the launch's token count, or the end of
// the caller's clip tile); rows of the score matrix are read from sc_row0 + (n - n0) * sc_ld (global memory or LDS: a flat pointer);
// rt_ws = gate weights staged in LDS by the caller (SC = false only).
// KPL: score columns per lane (NS / 64) as a compile-time constant - 10 for 80 caption keys x 8 heads - or 16 = "any NS <= 1024, bound
// checked at run time": the generic form holds 16 registers per token and runs 16 exponentials where 10 are live (same values, same order
// of additions: the padding entries contribute exp(-inf) = 0, so both forms round alike).
// EE: experts per gate as a compile-time constant (4 / 8), 0 = run-time a.E (<= 16).
template <int PP, bool SC, int RT_TPW, int KPL = 16, int EE = 0>     // PP: tokens laid side by side in a wave in phase B: 4 when 2E+2 <= 16, 2 when <= 32
__device__ __forceinline__ void router_tokens(const RouterDev& a, const int n0, const int N, const float* sc_row0, const int sc_ld,
                                              const float* rt_ws) {
    const Planes cq = a.cq; const float* __restrict__ Wg = a.Wg;
    const int T = a.T, D = a.D, E = EE ? EE : a.E, NS = a.NS, Hh = a.Hh;
    constexpr int PE = EE ? EE : 16;      // partial-logit registers per token
    (void)D; (void)cq; (void)rt_ws; (void)NS; (void)Hh; (void)sc_row0; (void)sc_ld; (void)T; (void)Wg;
    // gate weights staged once per block (every wave re-reading E*D floats per token through L1/L2 was the kernel's
    // whole cost); a wave then walks RT_TPW tokens
    const int lane = threadIdx.x & 63;
    // phase A: the RT_TPW tokens' feature loads are issued together (token features = MoE cross-attention output, bf16
    // planes; Wg/bg already contain out_proj folded in), then E partial dot products per lane and token
    float parts[RT_TPW][PE];
    if constexpr (SC) {
        // lane owns head (lane % Hh) and the keys lane/Hh + (64/Hh) i: columns lane + 64 i (coalesced)
        const int kpl = KPL == 16 ? (NS >> 6) : KPL;   // columns per lane (<= 16)
        float sv[RT_TPW][KPL];
#pragma unroll
        for (int tok = 0; tok < RT_TPW; ++tok) {
            const float* srow = sc_row0 + (int64_t)(min(n0 + tok, N - 1) - n0) * sc_ld + lane;
#pragma unroll
            for (int i = 0; i < KPL; ++i) sv[tok][i] = i < kpl ? srow[64 * i] : -INFINITY;
        }
#pragma unroll
        for (int tok = 0; tok < RT_TPW; ++tok) {
            const int bclip = min(n0 + tok, N - 1) / T;
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < KPL; ++i) m = fmaxf(m, sv[tok][i]);
            for (int o = Hh; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            float l = 0.f;
#pragma unroll
            for (int i = 0; i < KPL; ++i) { sv[tok][i] = i < kpl ? __expf(sv[tok][i] - m) : 0.f; l += sv[tok][i]; }
            for (int o = Hh; o < 64; o <<= 1) l += __shfl_xor(l, o, 64);
            const float inv = 1.f / l;
            const float* vwb = Wg + ((int64_t)bclip * NS + lane) * E;
#pragma unroll
            for (int e = 0; e < PE; ++e) parts[tok][e] = 0.f;
            if (E == 4) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < KPL; ++i)
                    if (i < kpl) {
                        const float4 w = *reinterpret_cast<const float4*>(vwb + (int64_t)64 * i * 4);
                        acc.x += sv[tok][i] * w.x; acc.y += sv[tok][i] * w.y; acc.z += sv[tok][i] * w.z; acc.w += sv[tok][i] * w.w;
                    }
                parts[tok][0] = acc.x * inv; parts[tok][1] = acc.y * inv; parts[tok][2] = acc.z * inv; parts[tok][3] = acc.w * inv;
            } else if (PE >= 8 && E == 8) {
                // two 16-byte loads per (lane, key column) instead of eight 4-byte loads 32 B apart (8 experts, 48128 tokens: this loop
                // was 469 us of a block evaluation, profiles/r02_c3_kernel_stats.csv)
                float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
                for (int i = 0; i < KPL; ++i)
                    if (i < kpl) {
                        const float4 w0 = *reinterpret_cast<const float4*>(vwb + (int64_t)64 * i * 8);
                        const float4 w1 = *reinterpret_cast<const float4*>(vwb + (int64_t)64 * i * 8 + 4);
                        a0.x += sv[tok][i] * w0.x; a0.y += sv[tok][i] * w0.y; a0.z += sv[tok][i] * w0.z; a0.w += sv[tok][i] * w0.w;
                        a1.x += sv[tok][i] * w1.x; a1.y += sv[tok][i] * w1.y; a1.z += sv[tok][i] * w1.z; a1.w += sv[tok][i] * w1.w;
                    }
                parts[tok][0] = a0.x * inv; parts[tok][1] = a0.y * inv; parts[tok][2] = a0.z * inv; parts[tok][3] = a0.w * inv;
                if constexpr (PE >= 8) { parts[tok][4] = a1.x * inv; parts[tok][5] = a1.y * inv; parts[tok][6] = a1.z * inv; parts[tok][7] = a1.w * inv; }
            } else {
#pragma unroll
                for (int e = 0; e < PE; ++e) {
                    float acc = 0.f;
                    if (e < E) {
#pragma unroll
                        for (int i = 0; i < KPL; ++i)
                            if (i < kpl) acc += sv[tok][i] * vwb[(int64_t)64 * i * E + e];
                    }
                    parts[tok][e] = acc * inv;
                }
            }
        }
    } else {
        float xv[RT_TPW][12];      // D <= 768: 3 x 4 values per lane
#pragma unroll
        for (int tok = 0; tok < RT_TPW; ++tok) {
            const bf16_t* xh = cq.p + (int64_t)min(n0 + tok, N - 1) * D;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int k = lane * 4 + i * 256;
                if (k < D) {
                    const bf16x4 hv = *reinterpret_cast<const bf16x4*>(xh + k);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[tok][i * 4 + j] = bf2f(hv[j]);
                    if (cq.np == 2) {
                        const bf16x4 lv = *reinterpret_cast<const bf16x4*>(xh + cq.plane + k);
#pragma unroll
                        for (int j = 0; j < 4; ++j) xv[tok][i * 4 + j] += bf2f(lv[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[tok][i * 4 + j] = 0.f;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < PE; ++e) {
            float acc[RT_TPW];
#pragma unroll
            for (int tok = 0; tok < RT_TPW; ++tok) acc[tok] = 0.f;
            if (e < E) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int k = lane * 4 + i * 256;
                    if (k < D) {
                        const float4 wv = *reinterpret_cast<const float4*>(rt_ws + e * D + k);
#pragma unroll
                        for (int tok = 0; tok < RT_TPW; ++tok)
                            acc[tok] += xv[tok][i * 4] * wv.x + xv[tok][i * 4 + 1] * wv.y + xv[tok][i * 4 + 2] * wv.z + xv[tok][i * 4 + 3] * wv.w;
                    }
                }
            }
#pragma unroll
            for (int tok = 0; tok < RT_TPW; ++tok) parts[tok][e] = acc[tok];
        }
    }
    // phase B (router_phase_b): the E partial dot products per lane are reduced together: after exchanging with lane^32 a lane keeps
    // half of the experts, after lane^16 a quarter, ... then the remaining value is summed over the rest of the wave (7 shuffles for
    // E = 4 instead of 24); lane e then holds logit e, from where the token's own lanes fetch it.
    auto logit_of = [&](int t0, int tokq, int sl) __attribute__((always_inline)) {
        float mylogit = 0.f;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const float (&part)[PE] = parts[t0 + j];
            float logit_e = 0.f;      // valid on lanes [0,E)
            if (E == 4) { if constexpr (PE >= 4) logit_e = reduce_logits<4, PE>(part, lane); }
            else if (E == 8) { if constexpr (PE >= 8) logit_e = reduce_logits<8, PE>(part, lane); }
            else {
#pragma unroll
                for (int e = 0; e < PE; ++e) {
                    if (e < E) {
                        const float r = wave_sum(part[e]);
                        if (lane == e) logit_e = r;
                    }
                }
            }
            const float v = __shfl(logit_e, sl < E ? sl : 0, 64);
            if (tokq == j) mylogit = v;
        }
        return mylogit;
    };
    router_phase_b<PP, RT_TPW>(a, n0, N, logit_of);
}
