// Flash attention for the DiT blocks (gfx950, head_dim 96, bf16 planes, fp32 softmax).
//
//   out = softmax(q k^T * s) v            (self, T keys)
//       + w[h] * softmax(q ky^T * s) vy   (cross, L keys; two SEPARATE softmaxes,
//                                           flag_large_dit_moe.py:381-402)
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32
// query rows.  K tiles (64 keys) and V^T tiles live in LDS and are shared by the 4
// waves.  S is computed TRANSPOSED (S^T = K Q^T, mfma(A=K, B=Q)) so a lane owns one
// query column: the row max / row sum are 32 in-register values + one exchange with
// lane^32.  For P V the MFMA k-slot order is free as long as A and B agree, so each
// lane feeds its own 8 P registers as the B operand and reads the matching keys of
// V^T (stored d-major by the QKV GEMM epilogue) with two 8-byte LDS reads - no
// cross-lane shuffle of P at all.
// SPLIT = bf16x3 parity mode (hi*hi + lo*hi + hi*lo for both Q K^T and P V).
// (Round 3, measured and dropped: the cross keys visited twice - statistics pass in front of the self pass, accumulate pass behind
//  it - removes the 48-KB stash (LDS 74 -> 26 KB), but at two waves per SIMD the kernel runs 64.4 us against 61.7 with the stash, and
//  three waves per SIMD need <= 168 registers where the kernel holds 254 (o 48 + s 32 + q 24 + K/V prefetch 24 + fragments): forced,
//  the compiler spills 41.  Also measured and dropped: the K tile's rows permuted like the V^T tile's so that a lane's 8 probabilities
//  of one P.V MFMA are 8 consecutive keys and the V^T fragment is ONE 16-byte LDS read (144-byte rows) instead of two 8-byte ones:
//  58.0 us against 55.3 on like boxes - and a whole-uint4 LDS store of the prefetch register array sent that array through scratch
//  memory (108 us) until it was written component-wise.)
#include <stdlib.h>

#include "kernels.h"

#define HD 96
#define KT 64                 // keys per tile
#define KPITCH 104            // bf16 elements per K row in LDS (208 B: 13 x 16 B -> conflict-free b128 reads)
#define VPITCH 68             // bf16 elements per V^T row in LDS (136 B -> conflict-free b64 reads)

// source row of LDS row r of a 32-row MFMA operand tile in the "16 consecutive columns per lane" layout (see gemm_bf16.hip: P16)
__device__ __forceinline__ int p16_row(int r) { const int c = r & 31; return (r & ~31) + ((c >> 2) & 1) * 16 + (c >> 3) * 4 + (c & 3); }

struct AttnDev {
    const bf16_t* q; int64_t q_plane;
    const bf16_t* k; int64_t k_plane;
    const bf16_t* vt; int64_t vt_plane;
    const bf16_t* ky; int64_t ky_plane;
    const bf16_t* vyt; int64_t vyt_plane;
    const float* cross_w;
    bf16_t* out; int64_t out_plane; int out_np;
    int B, T, Tpad, L, Lpad, H, D;
    int has_self, has_cross, kv_batch_mod, nq;
    float scale_log2e, defer_thr;
};

// ABL (tuning only): 1 = no K/V tile traffic in the loop, 2 = no softmax math, 3 = no P.V MFMAs, 4 = no Q.K^T MFMAs
template <bool SPLIT, int ABL = 0>
__global__ void __launch_bounds__(256) attn_kernel(const AttnDev p) {
    constexpr int NP = SPLIT ? 2 : 1;
    __shared__ __attribute__((aligned(16))) bf16_t Kl[NP][KT * KPITCH];
    __shared__ __attribute__((aligned(16))) bf16_t Vl[NP][HD * VPITCH];
    __shared__ float stash[4][48][64];     // the cross-attention result waits here while self-attention runs

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, ql = lane & 31;
    // XCD-aware order: the q-tiles of one (batch, head) are consecutive blocks of the same XCD so its K/V stay in that L2
    const int Lb = blockIdx.x, jx = Lb >> 3;
    const int bh = (jx / p.nq) * 8 + (Lb & 7);
    if (bh >= p.B * p.H) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = (jx % p.nq) * 128 + wave * 32;
    const int qrow = q0 + ql;
    const int qrow_c = qrow < p.T ? qrow : p.T - 1;

    // ---- Q fragments (B operand): Q[q = ql][d = ks*16 + g*8 .. +8]
    bf16x8 qf[6], qlf[6];
    {
        const bf16_t* qp = p.q + ((int64_t)b * p.T + qrow_c) * p.D + h * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
            if constexpr (SPLIT) qlf[ks] = *reinterpret_cast<const bf16x8*>(qp + p.q_plane + ks * 16);
        }
    }

    f32x16 o[3];
    float res_l = 1.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

    auto kv_pass = [&](const bf16_t* Kg, int64_t kplane, const bf16_t* Vg, int64_t vplane, int nkeys, int vpad, int kb_batch,
                       f32x16 (&acc)[3], float& l_out) __attribute__((always_inline)) {
        float m_run = -1e30f, l_run = 0.f;
        const int ntiles = (nkeys + KT - 1) / KT;
        // register prefetch (bf16 mode): the global loads of tile kt+1 are issued before tile kt is multiplied and
        // written to LDS after it, so their latency hides behind the MFMAs (split mode has no registers to spare)
        uint4 kreg[3], vreg[3];
        auto tile_load = [&](int kt, uint4 (&kr)[3], uint4 (&vr)[3], int pl) {
            const int key0 = kt * KT;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int id = tid + i * 256;
                const int key = id / 12, c = id - key * 12;
                const bool ok = (key0 + key) < nkeys;
                const bf16_t* src = Kg + ((int64_t)kb_batch * nkeys + (ok ? key0 + key : 0)) * p.D + h * HD + c * 8;
                kr[i] = ok ? *reinterpret_cast<const uint4*>(src + pl * kplane) : make_uint4(0, 0, 0, 0);
                const int d = id >> 3, c2 = id & 7;
                // (LDS row d of the V^T tile holds V^T row pi(d): the O accumulator then carries 16 CONSECUTIVE head-dim columns per
                //  lane and the result leaves as 16-byte stores - see the epilogue)
                const bf16_t* vs = Vg + ((int64_t)(kb_batch * p.H + h) * HD + p16_row(d)) * vpad + key0 + c2 * 8;
                vr[i] = *reinterpret_cast<const uint4*>(vs + pl * vplane);
            }
        };
        auto tile_store = [&](const uint4 (&kr)[3], const uint4 (&vr)[3], int pl) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int id = tid + i * 256;
                const int key = id / 12, c = id - key * 12;
                *reinterpret_cast<uint4*>(&Kl[pl][key * KPITCH + c * 8]) = kr[i];
                const int d = id >> 3, c2 = id & 7;
                uint2* dst = reinterpret_cast<uint2*>(&Vl[pl][d * VPITCH + c2 * 8]);
                dst[0] = make_uint2(vr[i].x, vr[i].y);
                dst[1] = make_uint2(vr[i].z, vr[i].w);
            }
        };
        if constexpr (!SPLIT) tile_load(0, kreg, vreg, 0);
        for (int kt = 0; kt < ntiles; ++kt) {
            const int key0 = kt * KT;
            __syncthreads();
            if constexpr (!SPLIT) {
                if (ABL != 1 || kt == 0) tile_store(kreg, vreg, 0);
            } else {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    tile_load(kt, kreg, vreg, pl);
                    tile_store(kreg, vreg, pl);
                }
            }
            __syncthreads();
            if constexpr (!SPLIT) {
                if (ABL != 1 && kt + 1 < ntiles) tile_load(kt + 1, kreg, vreg, 0);
            }
            // ---- S^T = K Q^T  (two 32-key sub-tiles, two independent accumulator chains; the K fragments of k-step ks+1 are requested
            // before the MFMAs of ks are queued - the plain loop waited out an LDS round trip in front of every pair of MFMAs)
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            if constexpr (!SPLIT && ABL != 4) {
                bf16x8 kfr[2][2];
                auto krd = [&](int ks, int slot) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
                        kfr[slot][kb] = *reinterpret_cast<const bf16x8*>(&Kl[0][(kb * 32 + ql) * KPITCH + ks * 16 + g * 8]);
                };
                krd(0, 0);
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) {
                    if (ks + 1 < 6) krd(ks + 1, (ks + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks & 1][kb], qf[ks], s[kb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) {
                    const int off = (kb * 32 + ql) * KPITCH + ks * 16 + g * 8;
                    bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Kl[0][off]);
                    if constexpr (ABL == 4) s[kb][ks] += (float)kf[0];
                    else s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
                    if constexpr (SPLIT) {
                        bf16x8 klo = *reinterpret_cast<const bf16x8*>(&Kl[1][off]);
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(klo, qf[ks], s[kb], 0, 0, 0);
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qlf[ks], s[kb], 0, 0, 0);
                    }
                }
            }
            }
            if constexpr (ABL != 2) {
            // ---- online softmax (base-2 domain).  Lean VALU path: the key mask is applied only on a partial tile,
            // the scale is folded into one fma per element feeding the raw v_exp_f32, and the accumulator rescale is
            // skipped (wave-uniformly) on tiles that do not raise any row maximum.
            const bool partial = key0 + KT > nkeys;
            if (partial) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                        if (key >= nkeys) s[kb][r] = -INFINITY;
                    }
            }
            float tmax = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[kb][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax * p.scale_log2e);     // every tile holds >= 1 valid key: finite
            // deferred rescale: the running maximum (and with it the 48 accumulator registers) is only moved when some row of the wave
            // outgrew it by more than 2^thr - until then P = exp2(s - m_run) <= 2^thr, harmless in fp32 sums and relative in bf16.
            // thr = 0 is the exact running maximum.  The decision depends on this wave's 32 query rows only (not on the batch).
            if (!__all(m_new <= m_run + p.defer_thr)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[kb][r], p.scale_log2e, -m_run));    // exp2(-inf) = 0 for masked keys
                    s[kb][r] = e;
                    psum += e;
                }
            l_run += psum;
            }
            // ---- O^T += V^T P^T   (k-slot e of lane group g <-> key base + 8*(e>>2) + 4g + (e&3))
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    bf16x8 ph, pl_;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float pv = s[kb][hs * 8 + e];
                        ph[e] = f2bf(pv);
                        if constexpr (SPLIT) pl_[e] = f2bf(pv - bf2f(ph[e]));
                    }
                    const int kbase = kb * 32 + hs * 16 + 4 * g;
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) {
                        const int voff = (dt * 32 + ql) * VPITCH + kbase;
                        bf16x4 v0 = *reinterpret_cast<const bf16x4*>(&Vl[0][voff]);
                        bf16x4 v1 = *reinterpret_cast<const bf16x4*>(&Vl[0][voff + 8]);
                        bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                        if constexpr (ABL == 3) acc[dt][0] += (float)vf[0] * (float)ph[0];
                        else acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, ph, acc[dt], 0, 0, 0);
                        if constexpr (SPLIT) {
                            bf16x4 w0 = *reinterpret_cast<const bf16x4*>(&Vl[1][voff]);
                            bf16x4 w1 = *reinterpret_cast<const bf16x4*>(&Vl[1][voff + 8]);
                            bf16x8 vlo = __builtin_shufflevector(w0, w1, 0, 1, 2, 3, 4, 5, 6, 7);
                            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vlo, ph, acc[dt], 0, 0, 0);
                            acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pl_, acc[dt], 0, 0, 0);
                        }
                    }
                }
        }
        l_out = l_run + __shfl_xor(l_run, 32, 64);
    };

    // cross attention first (2 short tiles); its weighted, normalised result is parked in LDS so the long
    // self-attention pass runs with a single accumulator set (keeps the kernel at 2 waves per SIMD)
    if (p.has_cross) {
        float lc;
        const int kb_batch = p.kv_batch_mod > 0 ? (b % p.kv_batch_mod) : b;
        kv_pass(p.ky, p.ky_plane, p.vyt, p.vyt_plane, p.L, p.Lpad, kb_batch, o, lc);
        const float w = (p.cross_w ? p.cross_w[h] : 1.f) / lc;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= w;
    }
    if (p.has_self) {
        if (p.has_cross) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { stash[wave][i * 16 + r][lane] = o[i][r]; o[i][r] = 0.f; }
        }
        kv_pass(p.k, p.k_plane, p.vt, p.vt_plane, p.T, p.Tpad, b, o, res_l);
        const float inv = 1.f / res_l;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = o[i][r] * inv + (p.has_cross ? stash[wave][i * 16 + r][lane] : 0.f);
    }
    if (qrow < p.T) {
        const int64_t base = ((int64_t)b * p.T + qrow) * p.D + h * HD;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int d0 = dt * 32 + 16 * g + 8 * hf;           // accumulator register r <-> head-dim column dt*32 + 16 g + r
                bf16x8 hi;
#pragma unroll
                for (int i = 0; i < 8; ++i) hi[i] = f2bf(o[dt][hf * 8 + i]);
                *reinterpret_cast<bf16x8*>(p.out + base + d0) = hi;
                if (p.out_np == 2) {
                    bf16x8 lo;
#pragma unroll
                    for (int i = 0; i < 8; ++i) lo[i] = f2bf(o[dt][hf * 8 + i] - bf2f(hi[i]));
                    *reinterpret_cast<bf16x8*>(p.out + p.out_plane + base + d0) = lo;
                }
            }
    }
}

int launch_attention(const AttnArgs& a, hipStream_t st) {
    if (a.hd != HD) VB_FAIL(VB_E_INVALID, "attention: head_dim %d unsupported (built for %d)", a.hd, HD);
    if (a.has_self && (a.Tpad % KT || a.Tpad < a.T)) VB_FAIL(VB_E_INVALID, "attention: Tpad=%d must be a multiple of %d >= T", a.Tpad, KT);
    if (a.has_cross && (a.Lpad % KT || a.Lpad < a.L)) VB_FAIL(VB_E_INVALID, "attention: Lpad=%d must be a multiple of %d >= L", a.Lpad, KT);
    if (!a.has_self && !a.has_cross) VB_FAIL(VB_E_INVALID, "attention: nothing to do");
    AttnDev d;
    d.q = a.q.p; d.q_plane = a.q.plane; d.k = a.k.p; d.k_plane = a.k.plane; d.vt = a.vt.p; d.vt_plane = a.vt.plane;
    d.ky = a.ky.p; d.ky_plane = a.ky.plane; d.vyt = a.vyt.p; d.vyt_plane = a.vyt.plane; d.cross_w = a.cross_w;
    d.out = a.out.p; d.out_plane = a.out.plane; d.out_np = a.out.np;
    d.B = a.B; d.T = a.T; d.Tpad = a.Tpad; d.L = a.L; d.Lpad = a.Lpad; d.H = a.H; d.D = a.H * a.hd;
    d.has_self = a.has_self; d.has_cross = a.has_cross; d.kv_batch_mod = a.kv_batch_mod;
    d.scale_log2e = a.scale * 1.4426950408889634f;
    d.defer_thr = vb_tune().attn_defer_thr;
    const double ae = (double)a.B * a.H * a.hd * 2.0 * a.q.np;      // bytes per token (or key) row of one q/k/v/out tensor
    ProfScope prof(1, 4.0 * a.B * a.H * a.T * a.hd * ((a.has_self ? a.T : 0) + (a.has_cross ? a.L : 0)),
                   ae * (2.0 * a.T + (a.has_self ? 2.0 * a.T : 0) + (a.has_cross ? 2.0 * a.L : 0)), st);
    d.nq = cdiv(a.T, 128);
    dim3 grid(d.nq * ((a.B * a.H + 7) / 8 * 8));
    if (a.q.np == 2) hipLaunchKernelGGL((attn_kernel<true, 0>), grid, dim3(256), 0, st, d);
    else {
#ifdef VB_EXPERIMENTS      // ablation instances exist in the experiments build only (tools/attn_bench.py)
        const int abl = vb_tune().attn_ablate;
        if (abl == 1) hipLaunchKernelGGL((attn_kernel<false, 1>), grid, dim3(256), 0, st, d);
        else if (abl == 2) hipLaunchKernelGGL((attn_kernel<false, 2>), grid, dim3(256), 0, st, d);
        else if (abl == 3) hipLaunchKernelGGL((attn_kernel<false, 3>), grid, dim3(256), 0, st, d);
        else if (abl == 4) hipLaunchKernelGGL((attn_kernel<false, 4>), grid, dim3(256), 0, st, d);
        else
#endif
        hipLaunchKernelGGL((attn_kernel<false, 0>), grid, dim3(256), 0, st, d);
    }
    VB_CHECK_LAUNCH();
    return VB_OK;
}
