// bf16 MFMA GEMM for the DiT (gfx950):  C[m][n] = sum_k A[m][k] * B[n][k]
//
//  * 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave 64x64 as 2x2
//    v_mfma_f32_32x32x16_bf16 tiles, fp32 accumulation.
//  * Both operands are K-contiguous; tiles are register-staged into an XOR-swizzled
//    LDS image (16-B chunk c of row r lives at chunk c ^ ((r>>1)&7)) so the
//    ds_read_b128 fragment reads of 16 consecutive rows hit 16 distinct 16-B slots.
//  * LDS is double buffered: global loads of tile t+1 are in flight while the MFMAs
//    of tile t run; one barrier per K tile.
//  * The MFMA is issued "swapped" (weights as the A operand) so a lane owns one
//    output ROW m and 4 consecutive output COLUMNS per accumulator quad: pairwise
//    epilogues (RoPE, SwiGLU) are lane-local and stores are 8-16 B per lane.
//  * nseg == 3 is the split-precision mode: the K loop walks (A_hi,B_hi), (A_lo,B_hi),
//    (A_hi,B_lo) - three bf16 passes give fp32-class products.
//  * Grouped problems (Band-MoE experts): the grid's y index walks the m-tiles of all
//    groups; group row ranges come from a device array written by the bucket kernel.
#include <stdlib.h>

#include <string.h>

#include <type_traits>

#include "kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256
#define P8_MIN_TILES 96     // selection threshold of the 8-wave 256 x 256 kernel (tiles of the launch)

struct GemmDev {
    const bf16_t* A; int64_t a_plane; int lda; const int* a_rows; int a_koff_group;
    const bf16_t* B; int64_t b_plane; int ldb; int64_t b_group_stride;
    int M, N, K, nseg, ngroups; const int* group_off; int c_noff_group;
    const float* bias; int64_t bias_group_stride;
    bf16_t* out; int64_t out_plane; int out_np; int ldc;
    float* out32; int ldc32;
    const float* gate; int gate_ld; int T;
    const int* rows_out; const float* row_scale; const float* y32_in; int n_tiles;
    const float* row_scale2; int scale_split;
    int grp_rows, grp_tiles, grp_xcd;   // uniform groups (per-clip operands): rows per group, row tiles per group (0 = off); XCD-affine tile order
    const float* add32; int dup_rows;   // EPI_F32: + add32[m][n]; second copy of the row at m + dup_rows
    int conv_ci, conv_ktap, conv_dil, conv_agrp, conv_arow0; int64_t conv_btap; const float* res32;   // conv-as-GEMM mode (EPI_F32_CT), see GemmArgs
    int ncc, rpx;                   // 128x128 kernel, wide N: column tiles are visited in chunks of ncc (0 = off) over the rpx row tiles of an XCD
    bf16_t* q; int64_t q_plane; bf16_t* k; int64_t k_plane; bf16_t* vt; int64_t vt_plane; int qkv_np;
    const float* rope_cos; const float* rope_sin; int H, hd, Tpad, D;
    float rT, rhd, rD;              // reciprocals for fdiv(): the epilogues decompose row -> (clip, t) and column -> (head, d)
    int no_vt16;                    // QKV P16: keep the V third on the row-per-lane layout (VB_QKV_VT16_OFF: bit-identity switch)
    int epi_old;                    // gated-residual staged epilogue: request the residual after the staging, as rounds 1-3a did (VB_BAND_EPI_OLD)
    unsigned long long* trace;      // tuning only (vbdbg_gemm_trace): per block {t_start, t_loop_end, t_end, hw ids}
    int abl;                        // tuning only (VB_GEMM_ABLATE): 6 = QKV without the V^T stores, 7 = without the q/k stores, 8 = without the RoPE table loads
};

// x / d for 0 <= x < 2^21 without the ~40-instruction integer division: (x + 0.5) / d is at least 0.5/d away
// from every integer, far more than the fp32 rounding of the product.  rinv = 1.0f / d.
__device__ __forceinline__ int fdiv(int x, float rinv) { return (int)(((float)x + 0.5f) * rinv); }

__device__ __forceinline__ void store4p(bf16_t* base, int64_t plane, int np, int64_t idx, const float v[4]) {
    bf16x4 hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) hi[i] = f2bf(v[i]);
    *reinterpret_cast<bf16x4*>(base + idx) = hi;
    if (np == 2) {
        bf16x4 lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) lo[i] = f2bf(v[i] - bf2f(hi[i]));
        *reinterpret_cast<bf16x4*>(base + plane + idx) = lo;
    }
}
__device__ __forceinline__ void store1p(bf16_t* base, int64_t plane, int np, int64_t idx, float v) {
    bf16_t hi = f2bf(v);
    base[idx] = hi;
    if (np == 2) base[plane + idx] = f2bf(v - bf2f(hi));
}

// "P16" column layout: the MFMA accumulator of a 32 x 32 tile gives lane (row, fk) the columns q*8 + fk*4 + e (four separate quads).  When
// LDS row c' of the weight tile is filled with weight row pi(c') = ((c'>>2)&1)*16 + (c'>>3)*4 + (c'&3) instead of c' (a permutation of
// the SOURCE rows of the tile DMA: nothing else moves, the fragment reads stay conflict-free), the same accumulator holds the 16
// CONSECUTIVE output columns fk*16 .. fk*16+15 of the lane's row: bf16 results leave as 16-byte stores (two per tile instead of four
// 8-byte ones), RoPE table entries arrive as 16-byte loads, and every row gets 64 contiguous bytes per tile.  Same values, same
// arithmetic - only which lane holds what.
__device__ __forceinline__ int p16_src_row(int r) { const int c = r & 31; return (r & ~31) + ((c >> 2) & 1) * 16 + (c >> 3) * 4 + (c & 3); }
__device__ __forceinline__ void store8p(bf16_t* base, int64_t plane, int np, int64_t idx, const float v[8]) {
    bf16x8 hi;
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = f2bf(v[i]);
    *reinterpret_cast<bf16x8*>(base + idx) = hi;
    if (np == 2) {
        bf16x8 lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) lo[i] = f2bf(v[i] - bf2f(hi[i]));
        *reinterpret_cast<bf16x8*>(base + plane + idx) = lo;
    }
}
// QKV + RoPE epilogue of one wave in the P16 layout (same arithmetic as epi_store<EPI_QKV_ROPE>, element for element)
template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue_qkv_p16(const struct GemmDev& p, f32x16 (&acc)[TM][TN], int row_base, int rows_end, int n_base,
                                                      int frow, int fk);

// Epilogue in two halves.  epi_load<EPI>() issues every global LOAD an output quad needs (bias, residual + gate,
// RoPE table entries, partial expert sum); epi_store<EPI>() does the math and the stores.  The kernels call epi_load
// for all quads of a 32-row slab first and only then epi_store: vmcnt retires loads and stores in order, so a load
// issued behind a store also waits for that store's round trip - interleaved they serialise one memory round trip per
// quad (measured 1.4x on the residual GEMMs).
struct EpiPre { float4 a, b; };

template <int EPI>
__device__ __forceinline__ void epi_load(const GemmDev& p, int g, int m, int tok, int n, EpiPre& e) {
    e.a = make_float4(0.f, 0.f, 0.f, 0.f); e.b = e.a;
    if constexpr (EPI == EPI_PLANES || EPI == EPI_F32 || EPI == EPI_GELU_PLANES || EPI == EPI_HEADS_T || EPI == EPI_F32_CT) {
        if (p.bias) e.a = *reinterpret_cast<const float4*>(p.bias + g * p.bias_group_stride + n);
        if constexpr (EPI == EPI_F32) {
            if (p.add32) e.b = *reinterpret_cast<const float4*>(p.add32 + (int64_t)m * p.ldc32 + g * p.c_noff_group + n);
        }
        if constexpr (EPI == EPI_F32_CT) {
            if (p.res32) {
                const int b = fdiv(m, p.rT), t = m - b * p.T;
                const float* rp = p.res32 + ((int64_t)b * p.N + n) * p.T + t;
                e.b = make_float4(rp[0], rp[p.T], rp[2 * (int64_t)p.T], rp[3 * (int64_t)p.T]);
            }
        }
    } else if constexpr (EPI == EPI_RESID_GATE) {
        const int col = g * p.c_noff_group + n;
        e.a = *reinterpret_cast<const float4*>(p.out32 + (int64_t)m * p.ldc32 + col);
        e.b = *reinterpret_cast<const float4*>(p.gate + (int64_t)fdiv(m, p.rT) * p.gate_ld + col);
    } else if constexpr (EPI == EPI_SCATTER_ADD_PLANES) {
        e.a = *reinterpret_cast<const float4*>(p.y32_in + (int64_t)tok * p.ldc32 + n);
    } else if constexpr (EPI == EPI_QKV_ROPE) {
        if (n < 2 * p.D && p.abl != 8) {
            const int nn = n - fdiv(n, p.rD) * p.D, t = m - fdiv(m, p.rT) * p.T;
            const int jd = (nn - fdiv(nn, p.rhd) * p.hd) >> 1;
            const float2 cs = *reinterpret_cast<const float2*>(p.rope_cos + (int64_t)t * (p.hd / 2) + jd);
            const float2 sn = *reinterpret_cast<const float2*>(p.rope_sin + (int64_t)t * (p.hd / 2) + jd);
            e.a = make_float4(cs.x, cs.y, sn.x, sn.y);
        }
    }
}

template <int EPI>
__device__ __forceinline__ void epi_store(const GemmDev& p, int g, int m, int tok, float scale, int n, float v[4], const EpiPre& e) {
    // m: global row (slot) index, n: column within the group's [0,N), 4 consecutive columns, all < N
    // The arithmetic is pinned (no implicit contraction, explicit fmaf): the launcher picks the tile configuration from the
    // problem size, and a clip's result must not depend on the batch it rides in - every kernel variant has to round alike.
#pragma clang fp contract(off)
    if constexpr (EPI == EPI_PLANES || EPI == EPI_F32 || EPI == EPI_GELU_PLANES || EPI == EPI_HEADS_T || EPI == EPI_F32_CT) {
        v[0] += e.a.x; v[1] += e.a.y; v[2] += e.a.z; v[3] += e.a.w;
    }
    if constexpr (EPI == EPI_PLANES) {
        store4p(p.out, p.out_plane, p.out_np, (int64_t)m * p.ldc + g * p.c_noff_group + n, v);
    } else if constexpr (EPI == EPI_GELU_PLANES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = 0.5f * v[i] * (1.0f + erff(v[i] * 0.70710678118654752440f));
        store4p(p.out, p.out_plane, p.out_np, (int64_t)m * p.ldc + g * p.c_noff_group + n, v);
    } else if constexpr (EPI == EPI_F32) {
        if (p.add32) { v[0] += e.b.x; v[1] += e.b.y; v[2] += e.b.z; v[3] += e.b.w; }
        *reinterpret_cast<float4*>(p.out32 + (int64_t)m * p.ldc32 + g * p.c_noff_group + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (p.dup_rows > 0) *reinterpret_cast<float4*>(p.out32 + (int64_t)(m + p.dup_rows) * p.ldc32 + g * p.c_noff_group + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (EPI == EPI_F32_CT) {
        const int b = fdiv(m, p.rT), t = m - b * p.T;
        if (p.res32) { v[0] += e.b.x; v[1] += e.b.y; v[2] += e.b.z; v[3] += e.b.w; }
#pragma unroll
        for (int i = 0; i < 4; ++i) p.out32[((int64_t)b * p.N + n + i) * p.T + t] = v[i];
    } else if constexpr (EPI == EPI_RESID_GATE) {
        const int col = g * p.c_noff_group + n;
        float4 h;
        h.x = fmaf(e.b.x, v[0], e.a.x); h.y = fmaf(e.b.y, v[1], e.a.y); h.z = fmaf(e.b.z, v[2], e.a.z); h.w = fmaf(e.b.w, v[3], e.a.w);
        *reinterpret_cast<float4*>(p.out32 + (int64_t)m * p.ldc32 + col) = h;
    } else if constexpr (EPI == EPI_SWIGLU) {
        float o0 = silu_f(v[0]) * v[1], o1 = silu_f(v[2]) * v[3];
        if (p.row_scale2) { o0 *= scale; o1 *= scale; }      // routed gate weight folded into the hidden row (same two roundings as the P16 path)
        int64_t idx = (int64_t)m * p.ldc + g * p.c_noff_group + (n >> 1);
        bf16_t h0 = f2bf(o0), h1 = f2bf(o1);
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 hv; hv[0] = h0; hv[1] = h1;
        *reinterpret_cast<bf16x2*>(p.out + idx) = hv;
        if (p.out_np == 2) {
            bf16x2 lv; lv[0] = f2bf(o0 - bf2f(h0)); lv[1] = f2bf(o1 - bf2f(h1));
            *reinterpret_cast<bf16x2*>(p.out + p.out_plane + idx) = lv;
        }
    } else if constexpr (EPI == EPI_GEGLU) {
        // T5DenseGatedActDense: gelu_new(wi_0 x) * (wi_1 x), NewGELUActivation = 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
        float o[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float x = v[2 * i];
            const float inner = 0.7978845608028654f * (x + 0.044715f * (x * x * x));
            o[i] = (0.5f * x * (1.0f + tanhf(inner))) * v[2 * i + 1];
        }
        int64_t idx = (int64_t)m * p.ldc + g * p.c_noff_group + (n >> 1);
        bf16_t h0 = f2bf(o[0]), h1 = f2bf(o[1]);
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 hv; hv[0] = h0; hv[1] = h1;
        *reinterpret_cast<bf16x2*>(p.out + idx) = hv;
        if (p.out_np == 2) {
            bf16x2 lv; lv[0] = f2bf(o[0] - bf2f(h0)); lv[1] = f2bf(o[1] - bf2f(h1));
            *reinterpret_cast<bf16x2*>(p.out + p.out_plane + idx) = lv;
        }
    } else if constexpr (EPI == EPI_SCATTER_F32) {
        *reinterpret_cast<float4*>(p.out32 + (int64_t)tok * p.ldc32 + n) = make_float4(scale * v[0], scale * v[1], scale * v[2], scale * v[3]);
    } else if constexpr (EPI == EPI_SCATTER_ADD_PLANES) {
        float o[4] = {fmaf(scale, v[0], e.a.x), fmaf(scale, v[1], e.a.y), fmaf(scale, v[2], e.a.z), fmaf(scale, v[3], e.a.w)};
        store4p(p.out, p.out_plane, p.out_np, (int64_t)tok * p.ldc + n, o);
    } else if constexpr (EPI == EPI_HEADS_T) {
        int b = fdiv(m, p.rT), t = m - b * p.T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int nn = n + i;
            int h = fdiv(nn, p.rhd), d = nn - h * p.hd;
            store1p(p.out, p.out_plane, p.out_np, ((int64_t)(b * p.H + h) * p.hd + d) * p.Tpad + t, v[i]);
        }
    } else if constexpr (EPI == EPI_QKV_ROPE) {
        int sec = fdiv(n, p.rD);         // uniform over the 4 columns (D % 4 == 0)
        int nn = n - sec * p.D;
        int b = fdiv(m, p.rT), t = m - b * p.T;
        if (sec < 2) {
            const float c0 = e.a.x, c1 = e.a.y, s0 = e.a.z, s1 = e.a.w;
            float o[4] = {fmaf(v[0], c0, -(v[1] * s0)), fmaf(v[0], s0, v[1] * c0), fmaf(v[2], c1, -(v[3] * s1)), fmaf(v[2], s1, v[3] * c1)};
            if (p.abl == 7) return;
            if (sec == 0) store4p(p.q, p.q_plane, p.qkv_np, (int64_t)m * p.D + nn, o);
            else store4p(p.k, p.k_plane, p.qkv_np, (int64_t)m * p.D + nn, o);
        } else {
            if (p.abl == 6) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int c = nn + i;
                int h = fdiv(c, p.rhd), d = c - h * p.hd;
                store1p(p.vt, p.vt_plane, p.qkv_np, ((int64_t)(b * p.H + h) * p.hd + d) * p.Tpad + t, v[i]);
            }
        }
    }
}

// the whole epilogue of one wave: rows slab by slab (i), loads of a slab first, then math + stores
template <int EPI, int TM = 2, int TN = 2>
__device__ __forceinline__ void wave_epilogue(const GemmDev& p, int g, f32x16 (&acc)[TM][TN], int row_base, int rows_end, int n_base,
                                              int frow, int fk) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int slot = row_base + i * 32 + frow;
        if (slot >= rows_end) continue;
        int tok = slot; float scale = 1.f;
        if constexpr (EPI == EPI_SCATTER_F32 || EPI == EPI_SCATTER_ADD_PLANES) {
            tok = p.rows_out[slot];
            scale = p.row_scale[tok];
        }
        if constexpr (EPI == EPI_SWIGLU) {
            if (p.row_scale2) scale = (slot < p.scale_split ? p.row_scale : p.row_scale2)[p.a_rows[slot]];
        }
        EpiPre pre[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n_base + j * 32 + q * 8 + fk * 4;
                if (n < p.N) epi_load<EPI>(p, g, slot, tok, n, pre[j][q]);
            }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n_base + j * 32 + q * 8 + fk * 4;
                if (n >= p.N) continue;     // N % 4 == 0 is required
                float v[4] = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                epi_store<EPI>(p, g, slot, tok, scale, n, v, pre[j][q]);
            }
    }
}

template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue_qkv_p16(const GemmDev& p, f32x16 (&acc)[TM][TN], int row_base, int rows_end, int n_base,
                                                      int frow, int fk) {
#pragma clang fp contract(off)
    const int hd2 = p.hd >> 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = row_base + i * 32 + frow;
        if (m >= rows_end) continue;
        const int b = fdiv(m, p.rT), t = m - b * p.T;
        // loads of the slab first (RoPE table: 8 pairs = two 16-byte loads each for cos and sin), then math + stores
        float4 cs[TN][2], sn[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n_base + j * 32 + fk * 16;
            if (n < 2 * p.D) {
                const int nn = n - fdiv(n, p.rD) * p.D;
                const int jd = (nn - fdiv(nn, p.rhd) * p.hd) >> 1;
                const float* cp = p.rope_cos + (int64_t)t * hd2 + jd;
                const float* sp = p.rope_sin + (int64_t)t * hd2 + jd;
                cs[j][0] = *reinterpret_cast<const float4*>(cp); cs[j][1] = *reinterpret_cast<const float4*>(cp + 4);
                sn[j][0] = *reinterpret_cast<const float4*>(sp); sn[j][1] = *reinterpret_cast<const float4*>(sp + 4);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n_base + j * 32 + fk * 16;
            if (n >= p.N) continue;                          // N % 16 == 0 is required on this path
            const int sec = fdiv(n, p.rD);
            const int nn = n - sec * p.D;
            if (sec < 2) {
                const float c8[8] = {cs[j][0].x, cs[j][0].y, cs[j][0].z, cs[j][0].w, cs[j][1].x, cs[j][1].y, cs[j][1].z, cs[j][1].w};
                const float s8[8] = {sn[j][0].x, sn[j][0].y, sn[j][0].z, sn[j][0].w, sn[j][1].x, sn[j][1].y, sn[j][1].z, sn[j][1].w};
                float o[16];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v0 = acc[i][j][2 * e], v1 = acc[i][j][2 * e + 1];
                    o[2 * e] = fmaf(v0, c8[e], -(v1 * s8[e]));
                    o[2 * e + 1] = fmaf(v0, s8[e], v1 * c8[e]);
                }
                bf16_t* dst = sec == 0 ? p.q : p.k;
                const int64_t pl = sec == 0 ? p.q_plane : p.k_plane;
                store8p(dst, pl, p.qkv_np, (int64_t)m * p.D + nn, o);
                store8p(dst, pl, p.qkv_np, (int64_t)m * p.D + nn + 8, o + 8);
            } else {
                const int h = fdiv(nn, p.rhd), d0 = nn - h * p.hd;          // 16 | hd: the 16 columns stay inside one head
                const int64_t base = ((int64_t)(b * p.H + h) * p.hd + d0) * p.Tpad + t;
#pragma unroll
                for (int e = 0; e < 16; ++e) store1p(p.vt, p.vt_plane, p.qkv_np, base + (int64_t)e * p.Tpad, acc[i][j][e]);
            }
        }
    }
}

// V third of the QKV projection with the MFMA operands' roles exchanged (the workgroup reads the weight tile as its "row" operand and the
// token tile - source rows permuted - as its "column" operand): a lane then owns ONE head-dim column d and 16 CONSECUTIVE tokens per
// 32 x 32 tile, i.e. 32 contiguous bytes of the per-head V^T image [d][t] the attention kernel reads - two 16-byte stores where the
// row-per-lane layout needed sixteen 2-byte ones.  Same products, same k order: bit-identical.  Needs T % 16 == 0.
template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue_vt_p16(const GemmDev& p, f32x16 (&acc)[TM][TN], int d_base, int tok_base, int rows_end, int frow, int fk) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int n = d_base + i * 32 + frow;             // output column of the projection (this lane's weight row)
        if (n >= p.N) continue;
        const int nn = n - 2 * p.D;
        const int h = fdiv(nn, p.rhd), d = nn - h * p.hd;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int m0 = tok_base + j * 32 + 16 * fk;
            if (m0 >= rows_end) continue;
            const int b = fdiv(m0, p.rT), t = m0 - b * p.T;
            const int64_t base = ((int64_t)(b * p.H + h) * p.hd + d) * p.Tpad + t;
            float o[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = acc[i][j][e];
            if (m0 + 15 < rows_end) {
                store8p(p.vt, p.vt_plane, p.qkv_np, base, o);
                store8p(p.vt, p.vt_plane, p.qkv_np, base + 8, o + 8);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (m0 + e < rows_end) store1p(p.vt, p.vt_plane, p.qkv_np, base + e, o[e]);
            }
        }
    }
}

// SwiGLU epilogue of one wave in the P16 layout: 16 consecutive (w1, w3)-interleaved columns = 8 hidden values = one 16-byte store
template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue_swiglu_p16(const GemmDev& p, int g, f32x16 (&acc)[TM][TN], int row_base, int rows_end, int n_base,
                                                         int frow, int fk) {
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = row_base + i * 32 + frow;
        if (m >= rows_end) continue;
        float gs = 1.f;
        if (p.row_scale2) gs = (m < p.scale_split ? p.row_scale : p.row_scale2)[p.a_rows[m]];     // routed gate weight of this slot's token
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n_base + j * 32 + fk * 16;
            if (n >= p.N) continue;                          // N % 16 == 0 is required on this path
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = silu_f(acc[i][j][2 * e]) * acc[i][j][2 * e + 1];
                if (p.row_scale2) o[e] *= gs;
            }
            store8p(p.out, p.out_plane, p.out_np, (int64_t)m * p.ldc + g * p.c_noff_group + (n >> 1), o);
        }
    }
}

// Block-wide epilogue staged through LDS (which is free once the k-loop is over).  The MFMA accumulator layout gives a lane
// one output ROW and 4 consecutive columns, so direct stores put 8-16 B pieces of 32 different rows in every instruction.
// Here a slab of 64 rows x BN columns goes to LDS as fp32 and is read back row-major: a lane still owns 4 consecutive
// columns of one row (the epi_load / epi_store contract) but a wave now covers whole 128-B lines - residual / bias loads and
// all stores are full-line transactions.  The V third of the QKV projection is staged TRANSPOSED instead and written as
// 4 consecutive tokens of one (head, d) row: the per-head V^T image the attention kernel reads, in 8-B pieces of 128-B runs.
// Needs 64 * (BN + 4) * 4 bytes of LDS (BN * 68 * 4 for the transposed variant).
// (NWC wave columns x 2 wave rows, NT threads: 2 x 2 / 256 for the 4-wave kernels, 4 x 2 / 512 for the 8-wave kernel)
// HOIST (gated-residual epilogue of the fused band-expert kernel): the residual and gate values of a whole slab are requested BEFORE the
// slab goes through LDS, not after the second barrier in two passes: one exposed HBM round trip per slab, overlapped with the staging,
// instead of two behind it (same loads, same arithmetic, other issue order): band experts 63.2 -> 55.8 us at 12032 tokens (same box).
template <int EPI, int TM, int TN, int NWC = 2, int NT = NTHREADS, bool HOIST = false>
__device__ __forceinline__ void staged_epilogue(const GemmDev& p, int g, f32x16 (&acc)[TM][TN], float* stg, int row0, int rows_end,
                                                int n0, int tid, int wr, int wc, int frow, int fk) {
    static_assert(NT == 128 * NWC, "two wave rows of NWC waves");
    static_assert(!HOIST || EPI == EPI_RESID_GATE, "hoisted loads: gated-residual epilogue only");
    constexpr int BNB = NWC * 32 * TN;
    constexpr int PITCH = BNB + 4;               // floats; +4 keeps the 16-B column writes of 8 consecutive rows on distinct banks
    constexpr int QPR = BNB / 4;                 // quads per row
    constexpr int QPT = 64 * QPR / NT;           // quads per thread per slab
    bool vsec = false;
    if constexpr (EPI == EPI_QKV_ROPE) vsec = n0 >= 2 * p.D && (p.D % BNB) == 0 && (p.T & 3) == 0 && (p.Tpad & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        EpiPre preH[HOIST ? QPT : 1];
        if constexpr (HOIST) {
#pragma unroll
            for (int k = 0; k < QPT; ++k) {
                const int idx = tid + k * NT;
                const int lr = idx / QPR, cq = idx - lr * QPR;
                int slot = row0 + (lr >> 5) * 32 * TM + i * 32 + (lr & 31);
                int n = n0 + cq * 4;
                if (slot >= rows_end || n >= p.N) { slot = row0; n = n0; }      // (clamped, not branched: the stores below skip it)
                epi_load<EPI>(p, g, slot, slot, n, preH[k]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): own LDS reads done (loop fragments / previous slab)
        __builtin_amdgcn_s_barrier();
        if constexpr (EPI == EPI_QKV_ROPE) {
            if (vsec) {
                constexpr int PT = 64 + 4;       // transposed image [col][row]
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = wc * 32 * TN + j * 32 + (r >> 2) * 8 + fk * 4 + (r & 3);
                        stg[col * PT + wr * 32 + frow] = acc[i][j][r];
                    }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
                constexpr int IPT = BNB * 16 / NT;
#pragma unroll
                for (int k = 0; k < IPT; ++k) {
                    const int idx = tid + k * NT;
                    const int rq = idx & 15, c = idx >> 4;
                    const int lr = rq * 4;
                    const int slot = row0 + (lr >> 5) * 32 * TM + i * 32 + (lr & 31);
                    const int n = n0 + c;
                    if (slot >= rows_end || n >= p.N) continue;
                    const float4 vv = *reinterpret_cast<const float4*>(stg + c * PT + lr);
                    float v[4] = {vv.x, vv.y, vv.z, vv.w};
                    const int nn = n - 2 * p.D;
                    const int h = fdiv(nn, p.rhd), d = nn - h * p.hd;
                    const int b = fdiv(slot, p.rT), t = slot - b * p.T;
                    const int64_t base = ((int64_t)(b * p.H + h) * p.hd + d) * p.Tpad;
                    if (slot + 3 < rows_end && t + 3 < p.T) {
                        store4p(p.vt, p.vt_plane, p.qkv_np, base + t, v);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int m = slot + e;
                            if (m >= rows_end) break;
                            const int bb = fdiv(m, p.rT), tt = m - bb * p.T;
                            store1p(p.vt, p.vt_plane, p.qkv_np, ((int64_t)(bb * p.H + h) * p.hd + d) * p.Tpad + tt, v[e]);
                        }
                    }
                }
                continue;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wc * 32 * TN + j * 32 + q * 8 + fk * 4;
                *reinterpret_cast<float4*>(stg + (wr * 32 + frow) * PITCH + col) =
                    make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        // (in passes of at most 8 quads per thread: loads of a pass are issued before its stores; bounds the live registers)
        constexpr int QC = QPT > 8 ? QPT / 2 : QPT;
        static_assert(QPT % QC == 0, "quads per thread must split evenly");
#pragma unroll
        for (int kb = 0; kb < QPT; kb += QC) {
        EpiPre pre[QC];
        int slot_[QC], tok_[QC]; float scale_[QC];
#pragma unroll
        for (int kk = 0; kk < QC; ++kk) {
            const int k = kk, idx = tid + (kb + kk) * NT;
            const int lr = idx / QPR, cq = idx - lr * QPR;
            const int slot = row0 + (lr >> 5) * 32 * TM + i * 32 + (lr & 31);
            const int n = n0 + cq * 4;
            slot_[k] = (slot < rows_end && n < p.N) ? slot : -1;
            tok_[k] = slot; scale_[k] = 1.f;
            if (slot_[k] >= 0) {
                if constexpr (EPI == EPI_SCATTER_F32 || EPI == EPI_SCATTER_ADD_PLANES) {
                    tok_[k] = p.rows_out[slot];
                    scale_[k] = p.row_scale[tok_[k]];
                }
                if constexpr (EPI == EPI_SWIGLU) {
                    if (p.row_scale2) scale_[k] = (slot < p.scale_split ? p.row_scale : p.row_scale2)[p.a_rows[slot]];
                }
                if constexpr (HOIST) pre[k] = preH[kb + kk];
                else epi_load<EPI>(p, g, slot, tok_[k], n, pre[k]);
            }
        }
#pragma unroll
        for (int kk = 0; kk < QC; ++kk) {
            const int k = kk;
            if (slot_[k] < 0) continue;
            const int idx = tid + (kb + kk) * NT;
            const int lr = idx / QPR, cq = idx - lr * QPR;
            const float4 vv = *reinterpret_cast<const float4*>(stg + lr * PITCH + cq * 4);
            float v[4] = {vv.x, vv.y, vv.z, vv.w};
            epi_store<EPI>(p, g, slot_[k], tok_[k], scale_[k], n0 + cq * 4, v, pre[k]);
        }
        }
    }
}

__device__ __forceinline__ int lds_off(int row, int c) {   // byte offset inside a [128][64] bf16 tile
    return row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
}

template <int EPI>
__global__ void __launch_bounds__(NTHREADS) gemm_bf16_kernel(const GemmDev p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][BM * BK * 2];   // [buf][A/B][16 KB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- which (group, m-tile) is this block -------------------------------------
    int g = 0, row0, rows_end, tile_n;
    {
        // XCD-aware tile order: block L runs on XCD L%8 (8 private L2s).  All N-tiles of one M-tile are
        // consecutive blocks of the SAME XCD, so an A tile is fetched into one L2 once instead of once per N-tile.
        const int L = blockIdx.x, nN = p.n_tiles;
        const int jx = L >> 3;
        tile_n = jx % nN;
        int tmg = (jx / nN) * 8 + (L & 7);
        if (p.group_off) {
            bool found = false;
            for (int gi = 0; gi < p.ngroups; ++gi) {
                int lo = p.group_off[gi], hi = p.group_off[gi + 1];
                int nt = (hi - lo + BM - 1) / BM;
                if (tmg < nt) { g = gi; row0 = lo + tmg * BM; rows_end = hi; found = true; break; }
                tmg -= nt;
            }
            if (!found) return;
        } else {
            g = blockIdx.z;              // groups that share the row range (band experts)
            row0 = tmg * BM; rows_end = p.M;
            if (row0 >= rows_end) return;
        }
    }
    const int n0 = tile_n * BN;
    const int K = p.K;
    const int KT = (K + BK - 1) / BK;
    const int total = KT * p.nseg;

    // ---- per-thread global load slots: 4 chunks of A, 4 chunks of B ----------------
    const bf16_t* aptr[4]; const bf16_t* bptr[4]; bool aval[4], bval[4]; int lds_w[4];
    const int cch = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = (tid >> 3) + i * 32;
        int slot = row0 + r;
        aval[i] = slot < rows_end;
        int arow = aval[i] ? (p.a_rows ? p.a_rows[slot] : slot) : 0;
        aptr[i] = p.A + (int64_t)arow * p.lda + g * p.a_koff_group + cch * 8;
        int nrow = n0 + r;
        bval[i] = nrow < p.N;
        bptr[i] = p.B + g * p.b_group_stride + (int64_t)(bval[i] ? nrow : 0) * p.ldb + cch * 8;
        lds_w[i] = lds_off(r, cch);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[4], rb[4];
    auto gload = [&](int t) {
        int seg = t / KT;
        int k0 = (t - seg * KT) * BK;
        int64_t ao = (seg == 1) ? p.a_plane : 0;
        int64_t bo = (seg == 2) ? p.b_plane : 0;
        bool kin = (k0 + cch * 8) < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = (aval[i] && kin) ? *reinterpret_cast<const uint4*>(aptr[i] + ao + k0) : make_uint4(0, 0, 0, 0);
            rb[i] = (bval[i] && kin) ? *reinterpret_cast<const uint4*>(bptr[i] + bo + k0) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(&lds[buf][0][lds_w[i]]) = ra[i];
            *reinterpret_cast<uint4*>(&lds[buf][1][lds_w[i]]) = rb[i];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();

    const int frow = lane & 31, fk = lane >> 5;
    for (int t = 0; t < total; ++t) {
        const int buf = t & 1;
        if (t + 1 < total) gload(t + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bf[2];
            int c = ks * 2 + fk;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(&lds[buf][0][lds_off(wr * 64 + i * 32 + frow, c)]);
                bf[i] = *reinterpret_cast<const bf16x8*>(&lds[buf][1][lds_off(wc * 64 + i * 32 + frow, c)]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < total) lstore(buf ^ 1);
        __syncthreads();
    }

    wave_epilogue<EPI>(p, g, acc, row0 + wr * 64, rows_end, n0 + wc * 64, frow, fk);
}

// ---- variant 2: tiles DMA'd straight into an LDS ring (global_load_lds, 16 B / lane, no VGPR staging, no ds_write) ----
// The LDS image must be lane-linear (wave-uniform base + lane*16), so the XOR swizzle is applied to the per-lane
// SOURCE chunk instead: LDS slot (row, c') receives global chunk c = c' ^ swz(row); the fragment reads use the same
// involution.  NST stages: tiles t+1 .. t+NST-1 are in flight while tile t is multiplied; the wait is a COUNTED
// s_waitcnt vmcnt((NST-2)*loads_per_tile) + a raw s_barrier, so the DMA queue is never drained inside the loop.
// Needs K % BKT == 0 (no zero fill on this path); out-of-range rows read a clamped valid row and are never stored.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int BKT> __device__ __forceinline__ int lds_off_t(int row, int c) {
    if constexpr (BKT == 64) return row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    else return row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    // s_waitcnt simm16 (gfx9): vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; only vmcnt is counted here
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}

// which epilogues of the 128x128 kernel go through LDS (staged_epilogue) instead of storing from the MFMA layout
#ifndef VB_STAGED_MASK
#define VB_STAGED_MASK 0x91       /* measured per epilogue: PLANES, SWIGLU, GELU_PLANES gain; QKV_ROPE and the fp32 ones do not */
#endif
#define STAGED_EPI(E) (((VB_STAGED_MASK) >> (E)) & 1)

// ABL (tuning only): 1 = no tile DMA in the loop, 2 = no MFMA, 3 = no LDS fragment reads
template <int EPI, int BKT, int NST, int ABL = 0, bool P16 = false>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(BKT * NST == 96 ? 3 : 1, BKT * NST == 96 ? 3 : 8)))
gemm_bf16_glds_kernel(const GemmDev p) {
    static_assert(!P16 || EPI == EPI_QKV_ROPE || EPI == EPI_SWIGLU, "the P16 column layout is wired for the QKV + RoPE and SwiGLU epilogues");
    constexpr int CH = BKT / 8;              // 16-B chunks per tile row
    constexpr int RS = 64 / CH;              // tile rows covered by one wave-wide DMA (1 KB)
    constexpr int SPW = CH / 2;              // DMA pieces per wave per operand per tile
    constexpr int LPT = 2 * SPW;             // loads per wave per tile
    constexpr int OPB = BM * BKT * 2;        // bytes of one operand tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * 2 * OPB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    unsigned long long t_start = 0;
    if constexpr (EPI == EPI_F32) { if (p.trace) t_start = __builtin_amdgcn_s_memtime(); }

    int g = 0, row0, rows_end, tile_n;
    {
        const int L = blockIdx.x, nN = p.n_tiles;
        const int jx = L >> 3;
        int rt;
        if (p.ncc > 0) {
            // wide N (QKV: 18 column tiles = 3.5 MB of weights against a 4 MB L2 per XCD): an XCD walks ALL its row tiles for one
            // chunk of ncc column tiles before moving to the next chunk, so the live weight set is ncc/nN of the matrix
            const int per = p.rpx * p.ncc;
            const int ch = jx / per, rem = jx - ch * per;
            rt = rem / p.ncc;
            tile_n = ch * p.ncc + (rem - rt * p.ncc);
        } else {
            tile_n = jx % nN;
            rt = jx / nN;
        }
        int tmg = rt * 8 + (L & 7);
        if (p.grp_rows > 0) {
            int lt;       // row tile inside the group
            if (p.grp_xcd) {
                // per-group B operands (caption-gate scores: one folded key matrix per clip) and a multiple of 8 groups: XCD x = L & 7 serves
                // the groups x, x + 8, ... - all row tiles of a group on one XCD, its B operand in one L2 (it was fetched by all eight)
                g = (L & 7) + 8 * (rt / p.grp_tiles);
                lt = rt % p.grp_tiles;
            } else {
                // shared B operand (conv-as-GEMM) or a group count the XCDs do not divide: plain enumeration of (group, row tile)
                g = tmg / p.grp_tiles;
                lt = tmg - g * p.grp_tiles;
            }
            if (g >= p.ngroups) return;
            row0 = g * p.grp_rows + lt * BM; rows_end = (g + 1) * p.grp_rows;
            if (row0 >= rows_end) return;
        } else if (p.group_off) {
            bool found = false;
            for (int gi = 0; gi < p.ngroups; ++gi) {
                int lo = p.group_off[gi], hi = p.group_off[gi + 1];
                int nt = (hi - lo + BM - 1) / BM;
                if (tmg < nt) { g = gi; row0 = lo + tmg * BM; rows_end = hi; found = true; break; }
                tmg -= nt;
            }
            if (!found) return;
        } else {
            g = blockIdx.z;
            row0 = tmg * BM; rows_end = p.M;
            if (row0 >= rows_end) return;
        }
    }
    const int n0 = tile_n * BN;
    const int KT = (ABL == 5) ? 0 : p.K / BKT;
    const int total = KT * p.nseg;
    // QKV, P16: a tile of the V third exchanges the operands' roles (see wave_epilogue_vt_p16) - the permutation then belongs to the TOKEN rows
    bool vsec = false;
    if constexpr (P16 && EPI == EPI_QKV_ROPE) vsec = n0 >= 2 * p.D && (p.T & 15) == 0 && (p.Tpad & 7) == 0 && !p.no_vt16;       // (16-byte V^T stores: T and the padded pitch)

    const bf16_t* asrc[SPW]; const bf16_t* bsrc[SPW];
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
        const int s = wave * SPW + i;
        const int r = RS * s + lane / CH;
        const int cs = lane % CH;
        const int c = (BKT == 64) ? (cs ^ ((r >> 1) & 7)) : (cs ^ ((r >> 2) & 3));
        int slot = row0 + (vsec ? p16_src_row(r) : r);
        if (slot >= rows_end) slot = row0;
        int arow = p.a_rows ? p.a_rows[slot] : slot;
        if constexpr (EPI == EPI_F32_CT) {
            if (p.conv_ktap > 0) arow = g * p.conv_agrp + p.conv_arow0 + (slot - g * p.grp_rows);      // clip g's padded plane image, row of tap 0
        }
        asrc[i] = p.A + (int64_t)arow * p.lda + g * p.a_koff_group + c * 8;
        int nrow = n0 + ((P16 && !vsec) ? p16_src_row(r) : r);
        if (nrow >= p.N) nrow = 0;
        bsrc[i] = p.B + g * p.b_group_stride + (int64_t)nrow * p.ldb + c * 8;
    }
    auto issue = [&](int t) {
        const int st = t % NST;
        const int seg = t / KT;
        const int k0 = (t - seg * KT) * BKT;
        int64_t ao = (seg == 1 ? p.a_plane : 0) + k0;
        int64_t bo = (seg == 2 ? p.b_plane : 0) + k0;
        if constexpr (EPI == EPI_F32_CT && BKT == 64) {
            if (p.conv_ktap > 0) {       // conv mode: k-tile -> (tap, channel chunk); tap j reads the rows j * dil below tap 0's
                const int kt = t - seg * KT, tap = kt / p.conv_ktap, c0 = (kt - tap * p.conv_ktap) * BKT;
                ao = (seg == 1 ? p.a_plane : 0) + (int64_t)tap * p.conv_dil * p.lda + c0;
                bo = (seg == 2 ? p.b_plane : 0) + (int64_t)tap * p.conv_btap + c0;
            }
        }
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int s = wave * SPW + i;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[i] + ao), (lds_ptr_t)(&lds[(st * 2 + 0) * OPB + s * 1024]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(bsrc[i] + bo), (lds_ptr_t)(&lds[(st * 2 + 1) * OPB + s * 1024]), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < total) issue(t);
    const int frow = lane & 31, fk = lane >> 5;
    for (int t = 0; t < total; ++t) {
        const int st = t % NST;
        // tiles t+1 .. min(total-1, t+NST-2) may stay in flight
        const int ahead = min(total - 1, t + NST - 2) - t;
        if (NST >= 4 && ahead >= 2) wait_vmcnt<2 * LPT>();
        else if (NST >= 3 && ahead >= 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();           // tile t landed everywhere; everyone finished reading stage (t-1)%NST
        if (ABL != 1 && t + NST - 1 < total) issue(t + NST - 1);
        // (V third of QKV: the weight tile plays the token tile's part and vice versa - a uniform pointer exchange, same code)
        const unsigned char* As = &lds[(st * 2 + (vsec ? 1 : 0)) * OPB];
        const unsigned char* Bs = &lds[(st * 2 + (vsec ? 0 : 1)) * OPB];
        // fragment reads are software-pipelined in registers: the ds_reads of k-step ks+1 are issued before the MFMAs of
        // k-step ks (the compiler otherwise reuses one register set and serialises read -> wait -> 4 MFMA per k-step)
        bf16x8 af[2][2], bf[2][2];
        auto fload = [&](int ks, int slot) {
            const int c = ks * 2 + fk;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (ABL == 3) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { af[slot][i][e] = (bf16_t)(float)(t + e); bf[slot][i][e] = (bf16_t)(float)(ks + e); }
                } else {
                    af[slot][i] = *reinterpret_cast<const bf16x8*>(As + lds_off_t<BKT>(wr * 64 + i * 32 + frow, c));
                    bf[slot][i] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<BKT>(wc * 64 + i * 32 + frow, c));
                }
            }
        };
        fload(0, 0);
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < BKT / 16) fload(ks + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (ABL == 2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[i][j][e] += (float)af[cur][i][e] * (float)bf[cur][j][e];
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    unsigned long long t_loop = 0;
    if constexpr (EPI == EPI_F32) { if (p.trace) t_loop = __builtin_amdgcn_s_memtime(); }
    if constexpr (ABL == 4) {
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) sink += acc[i][j][0] + acc[i][j][9];
        if (sink == 12345.678f) p.out32[0] = sink;
    } else {
        // (conv-as-GEMM, EPI_F32_CT: a channel-major epilogue staged through LDS - 16-byte residual loads and stores - measured the same
        //  as the direct 4-byte one, 234 vs 230 us per VAE layer: these launches are mainloop-bound at K = 1920 .. 7680; not kept)
        if constexpr (P16 && EPI == EPI_QKV_ROPE) {
            if (vsec) wave_epilogue_vt_p16<2, 2>(p, acc, n0 + wr * 64, row0 + wc * 64, rows_end, frow, fk);
            else wave_epilogue_qkv_p16<2, 2>(p, acc, row0 + wr * 64, rows_end, n0 + wc * 64, frow, fk);
        }
        else if constexpr (P16 && EPI == EPI_SWIGLU) wave_epilogue_swiglu_p16<2, 2>(p, g, acc, row0 + wr * 64, rows_end, n0 + wc * 64, frow, fk);
        else if constexpr (STAGED_EPI(EPI)) staged_epilogue<EPI, 2, 2>(p, g, acc, reinterpret_cast<float*>(lds), row0, rows_end, n0, tid, wr, wc, frow, fk);
        else wave_epilogue<EPI>(p, g, acc, row0 + wr * 64, rows_end, n0 + wc * 64, frow, fk);
    }
    if constexpr (EPI == EPI_F32) {
        if (p.trace && tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* tr = p.trace + (size_t)blockIdx.x * 4;
            tr[0] = t_start; tr[1] = t_loop; tr[2] = __builtin_amdgcn_s_memtime(); tr[3] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
}


// ---- routed experts, second product, ONE launch (vocal2music_moe.py:154-167: y = m_c FFN^c(u) + m_a FFN^a(u)) ------------------
// The two w2 GEMMs (caption group: scatter m_c * H_c W2c^T as fp32; acoustic group: read it back, add m_a * H_a W2a^T, write bf16
// planes) round-tripped a [N][768] fp32 partial sum through HBM: 37 MB written + 37 MB re-read per block evaluation at 8 clips, for
// two launches at 10-13 % of the MFMA peak.  Here the tokens are bucketed by their (caption expert, acoustic expert) PAIR (E*E groups,
// bucket_place_kernel: the caption slots ARE the pair slots) and ONE grouped launch walks K = 2H as a plain GEMM: first half A = the
// tile's own rows of the routed hidden tensor (caption slots, contiguous) against W2c[c], second half A = the tokens' acoustic-slot
// rows (gathered) against W2a[a].  The per-token gate weights m_c / m_a ride in the hidden rows (folded in by the SwiGLU epilogue
// before the bf16 rounding - the same relative rounding error as scaling the fp32 product afterwards), so one accumulator serves.
// Tile 128 x (64 TN): 128 x 192 when that brings the launch from two rounds of the CUs down to one (8 clips: 110 row tiles x 4
// = 440 workgroups at two per CU), 128 x 128 otherwise.  Ring / swizzle / P16 column layout as in gemm_bf16_glds_kernel<*, 64, 2>.
struct PairDev {
    const bf16_t* Hs; int ldh;                               // routed hidden [2N][H] bf16, slot order (caption slots, then acoustic slots)
    const bf16_t* W2; int64_t w_stride; int ldw;             // [2E][D][H]
    const int* pair_off; const int* perm; const int* pair_pa;   // pair slot p: caption row = p, acoustic row = pair_pa[p], token = perm[p]
    bf16_t* out; int ldc;                                    // y planes [N][D] (one plane: bf16 production mode)
    int N, D, H, E, n_tiles;
};
template <int TN>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) moe_w2_pair_kernel(const PairDev p) {
    constexpr int BKT = 64, NST = 2, CH = 8, RS = 8;
    constexpr int BNP = 64 * TN;                             // columns per tile
    constexpr int SPA = 4, SPB = BNP / 32;                   // 1-KB DMA pieces per wave: A (128 rows), B (BNP rows)
    constexpr int ABYTES = BM * BKT * 2, BBYTES = BNP * BKT * 2, STAGE = ABYTES + BBYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    int g = 0, row0 = 0, rows_end = 0, tile_n;
    {
        const int L = blockIdx.x, nN = p.n_tiles;
        const int jx = L >> 3;
        tile_n = jx % nN;
        int tmg = (jx / nN) * 8 + (L & 7);
        bool found = false;
        const int G = p.E * p.E;
        for (int gi = 0; gi < G; ++gi) {
            const int lo = p.pair_off[gi], hi = p.pair_off[gi + 1];
            const int nt = (hi - lo + BM - 1) / BM;
            if (tmg < nt) { g = gi; row0 = lo + tmg * BM; rows_end = hi; found = true; break; }
            tmg -= nt;
        }
        if (!found) return;
    }
    const int ec = g / p.E, ea = g - ec * p.E;
    const int n0 = tile_n * BNP;
    const int KT = p.H / BKT;
    const int total = 2 * KT;

    const bf16_t* asrc[2][SPA]; const bf16_t* bsrc[SPB];     // B: one pointer per piece, the expert half is a uniform offset
    const int64_t boff1 = (int64_t)(p.E + ea - ec) * p.w_stride;
#pragma unroll
    for (int i = 0; i < SPA; ++i) {
        const int r = RS * (wave * SPA + i) + lane / CH;
        const int c = (lane % CH) ^ ((r >> 1) & 7);
        int slot = row0 + r;
        if (slot >= rows_end) slot = row0;
        asrc[0][i] = p.Hs + (int64_t)slot * p.ldh + c * 8;                    // caption half: the pair slots ARE the caption slots
        asrc[1][i] = p.Hs + (int64_t)p.pair_pa[slot] * p.ldh + c * 8;         // acoustic half: gathered
    }
#pragma unroll
    for (int i = 0; i < SPB; ++i) {
        const int r = RS * (wave * SPB + i) + lane / CH;
        const int c = (lane % CH) ^ ((r >> 1) & 7);
        int nrow = n0 + p16_src_row(r);                   // P16 column layout: a lane ends up with 16 consecutive output columns
        if (nrow >= p.D) nrow = 0;
        bsrc[i] = p.W2 + (int64_t)ec * p.w_stride + (int64_t)nrow * p.ldw + c * 8;
    }
    auto issue = [&](int t) {
        const int st = t % NST;
        const int half = t >= KT ? 1 : 0;
        const int k0 = (t - half * KT) * BKT;
        unsigned char* sa = lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < SPA; ++i) {
            const bf16_t* ap = half ? asrc[1][i] : asrc[0][i];
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(ap + k0), (lds_ptr_t)(sa + (wave * SPA + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < SPB; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(bsrc[i] + (half ? boff1 : 0) + k0), (lds_ptr_t)(sa + ABYTES + (wave * SPB + i) * 1024), 16, 0, 0);
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0);
    const int frow = lane & 31, fk = lane >> 5;
    for (int t = 0; t < total; ++t) {
        const int st = t % NST;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();           // tile t landed everywhere; everyone finished reading stage (t-1)%NST
        if (t + 1 < total) issue(t + 1);
        const unsigned char* As = lds + st * STAGE;
        const unsigned char* Bs = As + ABYTES;
        bf16x8 af[2][2], bf[2][TN];
        auto fload = [&](int ks, int slot) {
            const int c = ks * 2 + fk;
#pragma unroll
            for (int i = 0; i < 2; ++i) af[slot][i] = *reinterpret_cast<const bf16x8*>(As + lds_off_t<BKT>(wr * 64 + i * 32 + frow, c));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<BKT>(wc * 32 * TN + j * 32 + frow, c));
        };
        fload(0, 0);
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < BKT / 16) fload(ks + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // epilogue: a lane owns one token row and 16 consecutive columns per 32 x 32 tile (P16 layout): 16-byte stores
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = row0 + wr * 64 + i * 32 + frow;
        if (slot >= rows_end) continue;
        const int tok = p.perm[slot];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * 32 * TN + j * 32 + fk * 16;
            if (n >= p.D) continue;                                       // D % 16 == 0
            float o[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = acc[i][j][e];
            store8p(p.out, 0, 1, (int64_t)tok * p.ldc + n, o);
            store8p(p.out, 0, 1, (int64_t)tok * p.ldc + n + 8, o + 8);
        }
    }
}
int launch_moe_w2_pair(const MoeW2PairArgs& a, hipStream_t st) {
    if (a.H % 64 || a.D % 16 || a.E < 1 || a.E * a.E > 16) VB_FAIL(VB_E_INVALID, "moe_w2_pair: H=%d D=%d E=%d unsupported", a.H, a.D, a.E);
    PairDev d;
    d.Hs = a.Hs; d.ldh = a.H; d.W2 = a.W2; d.w_stride = (int64_t)a.D * a.H; d.ldw = a.H;
    d.pair_off = a.pair_off; d.perm = a.perm; d.pair_pa = a.pair_pa;
    d.out = a.out; d.ldc = a.D; d.N = a.N; d.D = a.D; d.H = a.H; d.E = a.E;
    const int mt = cdiv(a.N, BM) + a.E * a.E;                    // upper bound of the row tiles over all pair groups
    ProfScope prof(0, 2.0 * a.N * a.D * 2.0 * a.H, 2.0 * a.N * a.H * 2.0 + 2.0 * a.E * a.D * a.H * 2.0 + (double)a.N * a.D * 2.0, st);
    // 128 x 192 tiles when 128 x 128 would need a second round of the 512 workgroup slots (two per CU) and 192-wide tiles do not
    const int wide = vb_tune().w2_pair == 3 || (vb_tune().w2_pair == 1 && a.D % 192 == 0 && mt * cdiv(a.D, 128) > 512 && mt * (a.D / 192) <= 512);
    if (wide) {
        d.n_tiles = a.D / 192;
        hipLaunchKernelGGL(moe_w2_pair_kernel<3>, dim3(d.n_tiles * ((mt + 7) / 8 * 8)), dim3(NTHREADS), 0, st, d);
    } else {
        d.n_tiles = cdiv(a.D, BN);
        hipLaunchKernelGGL(moe_w2_pair_kernel<2>, dim3(d.n_tiles * ((mt + 7) / 8 * 8)), dim3(NTHREADS), 0, st, d);
    }
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// ---- 128 x 192 tiles, two workgroups per CU, gated-residual epilogue in the P16 layout (attention out-proj at >= 8 clips) ----------
// 12032 x 768 makes 564 tiles of 128 x 128 (1.1 rounds of the 512 two-per-CU slots, priced as 2) and 252 of 192 x 192 on the one-per-CU
// kernel, whose tiles all reach their read-modify-write epilogue together (39.8 us, 12.4 % MFMA-busy).  128 x 192 tiles make 376
// workgroups = ONE round at two per CU (80 KB of LDS each): the co-resident workgroup overlaps the other's epilogue, and the P16 layout
// gives a lane 16 consecutive fp32 columns (four 16-byte loads of h, four of the gate row, four stores).  Same k order and epilogue
// arithmetic as every other tile configuration (test_gemm_tile_configurations_round_alike).
// (Round 3, built, measured and removed - verdict item 1c: the Band-MoE input RMSNorm fused into this launch by "last arriver": every
//  workgroup publishes its stores with an agent-scope release, bumps its row panel's counter, and the workgroup that lands the panel's
//  last column tile normalises the 128 rows.  Bit-identical to the separate launch, no spinning, no placement assumed - and the
//  agent-scope release is a buffer_wbl2 per workgroup: 36 -> 202 us per launch (1272 -> 1155 mel-s/s).  Dropping the release would
//  lean on all column tiles of a panel sharing one XCD's L2, which the programming model does not promise.  The tail's registers also
//  slowed the un-fused path of the same kernel, 36 -> 55 us.)
template <int TN>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm_bf16_wide_resid_kernel(const GemmDev p) {
    constexpr int BKT = 64, NST = 2, CH = 8, RS = 8;
    constexpr int BNP = 64 * TN;
    constexpr int SPA = 4, SPB = BNP / 32;
    constexpr int ABYTES = BM * BKT * 2, BBYTES = BNP * BKT * 2, STAGE = ABYTES + BBYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int L = blockIdx.x, nN = p.n_tiles;
    const int jx = L >> 3;
    const int tile_n = jx % nN;
    const int row0 = ((jx / nN) * 8 + (L & 7)) * BM, rows_end = p.M;
    if (row0 >= rows_end) return;
    const int n0 = tile_n * BNP;
    const int KT = p.K / BKT;
    const int total = KT * p.nseg;

    const bf16_t* asrc[SPA]; const bf16_t* bsrc[SPB];
#pragma unroll
    for (int i = 0; i < SPA; ++i) {
        const int r = RS * (wave * SPA + i) + lane / CH;
        const int c = (lane % CH) ^ ((r >> 1) & 7);
        int slot = row0 + r;
        if (slot >= rows_end) slot = row0;
        asrc[i] = p.A + (int64_t)slot * p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < SPB; ++i) {
        const int r = RS * (wave * SPB + i) + lane / CH;
        const int c = (lane % CH) ^ ((r >> 1) & 7);
        int nrow = n0 + p16_src_row(r);
        if (nrow >= p.N) nrow = 0;
        bsrc[i] = p.B + (int64_t)nrow * p.ldb + c * 8;
    }
    auto issue = [&](int t) {
        const int st = t % NST;
        const int seg = t / KT;
        const int k0 = (t - seg * KT) * BKT;
        const int64_t ao = (seg == 1 ? p.a_plane : 0) + k0;
        const int64_t bo = (seg == 2 ? p.b_plane : 0) + k0;
        unsigned char* sa = lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < SPA; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[i] + ao), (lds_ptr_t)(sa + (wave * SPA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < SPB; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(bsrc[i] + bo), (lds_ptr_t)(sa + ABYTES + (wave * SPB + i) * 1024), 16, 0, 0);
    };
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    issue(0);
    const int frow = lane & 31, fk = lane >> 5;
    for (int t = 0; t < total; ++t) {
        const int st = t % NST;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (t + 1 < total) issue(t + 1);
        const unsigned char* As = lds + st * STAGE;
        const unsigned char* Bs = As + ABYTES;
        bf16x8 af[2][2], bf[2][TN];
        auto fload = [&](int ks, int slot) {
            const int c = ks * 2 + fk;
#pragma unroll
            for (int i = 0; i < 2; ++i) af[slot][i] = *reinterpret_cast<const bf16x8*>(As + lds_off_t<BKT>(wr * 64 + i * 32 + frow, c));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<BKT>(wc * 32 * TN + j * 32 + frow, c));
        };
        fload(0, 0);
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < BKT / 16) fload(ks + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // epilogue (EPI_RESID_GATE): out32[m][n..n+15] = fmaf(gate[clip(m)][n..], v, out32[m][n..]); loads of a row slab first, then the stores
    // (round 3, measured and dropped: both row slabs' residual values requested together - 248 VGPRs, 38.4 against 36.6 us)
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = row0 + wr * 64 + i * 32 + frow;
            if (m >= rows_end) continue;
            float* hrow = p.out32 + (int64_t)m * p.ldc32;
            const float* grow = p.gate + (int64_t)fdiv(m, p.rT) * p.gate_ld;
            float4 hv[TN][4], gv[TN][4];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wc * 32 * TN + j * 32 + fk * 16;
                if (n < p.N) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        hv[j][q] = *reinterpret_cast<const float4*>(hrow + n + 4 * q);
                        gv[j][q] = *reinterpret_cast<const float4*>(grow + n + 4 * q);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wc * 32 * TN + j * 32 + fk * 16;
                if (n >= p.N) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 o;
                    o.x = fmaf(gv[j][q].x, acc[i][j][4 * q + 0], hv[j][q].x); o.y = fmaf(gv[j][q].y, acc[i][j][4 * q + 1], hv[j][q].y);
                    o.z = fmaf(gv[j][q].z, acc[i][j][4 * q + 2], hv[j][q].z); o.w = fmaf(gv[j][q].w, acc[i][j][4 * q + 3], hv[j][q].w);
                    *reinterpret_cast<float4*>(hrow + n + 4 * q) = o;
                }
            }
        }
    }
}

// ---- variant 3: big block tiles, one workgroup per CU --------------------------------------------------------------
// Per-block traces of the 128x128 kernel (tools/gemm_trace.py) show a k-iteration costs ~1500 cycles against 640 cycles
// of MFMA: every iteration moves 32 KB through the CU's 64 B/clk vector-memory path (512 cycles at best) and waits for
// the tile issued one iteration earlier, and a launch of M=12032 x N=768 leaves 2.2 tiles per CU (priced as 3).  This
// kernel raises the flops per byte DMA'd and fits the tile grid to the 256 CUs instead:
//   * block tile (64 TM) x (64 TN), 4 waves as 2x2, a wave owns (32 TM) x (32 TN) = TM x TN MFMA 32x32 tiles;
//     192x192 (TM = TN = 3) moves 48 KB per 1440 MFMA cycles and makes 63 x 4 = 252 tiles of 12032 x 768;
//   * BK = 64, NSTB-stage LDS ring filled by global_load_lds (same lane-linear image + source-side XOR swizzle as
//     above), counted vmcnt waits: NSTB-1 tiles in flight;
//   * fragment reads double-buffered in registers across the 4 k-steps of a tile;
//   * epilogue staged through LDS (free after the loop): a slab of 64 rows x BN goes to LDS as fp32, is read back
//     row-major so a lane owns 4 consecutive columns AND a wave covers whole 128-B lines - residual/bias loads and all
//     stores become full-line transactions (the MFMA layout stores 32-B pieces of 32 different rows per instruction).
// ABL (tuning only): 1 = no tile DMA inside the loop, 2 = no MFMA, 3 = no fragment reads
template <int EPI, int TM, int TN, int NSTB, int ABL = 0>
__global__ void __launch_bounds__(NTHREADS) gemm_bf16_big_kernel(const GemmDev p) {
    constexpr int BMB = 64 * TM, BNB = 64 * TN;
    constexpr int PA = BMB / 32, PB = BNB / 32;        // 1-KB DMA pieces (8 tile rows) per wave per tile
    constexpr int LPT = PA + PB;
    constexpr int ABYTES = BMB * 128, BBYTES = BNB * 128, STAGE = ABYTES + BBYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    unsigned long long t_start = 0, t_pro = 0, t_loop = 0;
    if constexpr (EPI == EPI_F32) { if (p.trace) t_start = __builtin_amdgcn_s_memtime(); }

    int g = 0, row0, rows_end, tile_n;
    {
        const int L = blockIdx.x, nN = p.n_tiles;
        const int jx = L >> 3;
        tile_n = jx % nN;
        int tmg = (jx / nN) * 8 + (L & 7);
        if (p.grp_rows > 0) {
            // uniform groups (conv-as-GEMM: one clip per group, shared weights): plain enumeration of (group, row tile)
            g = tmg / p.grp_tiles;
            if (g >= p.ngroups) return;
            row0 = g * p.grp_rows + (tmg - g * p.grp_tiles) * BMB; rows_end = (g + 1) * p.grp_rows;
            if (row0 >= rows_end) return;
        } else if (p.group_off) {
            bool found = false;
            for (int gi = 0; gi < p.ngroups; ++gi) {
                int lo = p.group_off[gi], hi = p.group_off[gi + 1];
                int nt = (hi - lo + BMB - 1) / BMB;
                if (tmg < nt) { g = gi; row0 = lo + tmg * BMB; rows_end = hi; found = true; break; }
                tmg -= nt;
            }
            if (!found) return;
        } else {
            g = blockIdx.z;
            row0 = tmg * BMB; rows_end = p.M;
            if (row0 >= rows_end) return;
        }
    }
    const int n0 = tile_n * BNB;
    const int KT = p.K / 64;
    const int total = KT * p.nseg;

    const bf16_t* asrc[PA]; const bf16_t* bsrc[PB];
    {
        const int r8 = lane >> 3, cs = lane & 7;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = 8 * (wave * PA + i) + r8;
            const int c = cs ^ ((r >> 1) & 7);
            int slot = row0 + r;
            if (slot >= rows_end) slot = row0;
            int arow = p.a_rows ? p.a_rows[slot] : slot;
            if constexpr (EPI == EPI_F32_CT) {
                if (p.conv_ktap > 0) arow = g * p.conv_agrp + p.conv_arow0 + (slot - g * p.grp_rows);
            }
            asrc[i] = p.A + (int64_t)arow * p.lda + g * p.a_koff_group + c * 8;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int r = 8 * (wave * PB + i) + r8;
            const int c = cs ^ ((r >> 1) & 7);
            int nrow = n0 + r;
            if (nrow >= p.N) nrow = 0;
            bsrc[i] = p.B + g * p.b_group_stride + (int64_t)nrow * p.ldb + c * 8;
        }
    }
    auto issue = [&](int t) {
        const int st = t % NSTB;
        const int seg = t / KT;
        const int k0 = (t - seg * KT) * 64;
        int64_t ao = (seg == 1 ? p.a_plane : 0) + k0;
        int64_t bo = (seg == 2 ? p.b_plane : 0) + k0;
        if constexpr (EPI == EPI_F32_CT) {
            if (p.conv_ktap > 0) {       // conv mode: k-tile -> (tap, channel chunk), see gemm_bf16_glds_kernel
                const int kt = t - seg * KT, tap = kt / p.conv_ktap, c0 = (kt - tap * p.conv_ktap) * 64;
                ao = (seg == 1 ? p.a_plane : 0) + (int64_t)tap * p.conv_dil * p.lda + c0;
                bo = (seg == 2 ? p.b_plane : 0) + (int64_t)tap * p.conv_btap + c0;
            }
        }
        unsigned char* sa = ldsb + st * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[i] + ao), (lds_ptr_t)(sa + (wave * PA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PB; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(bsrc[i] + bo), (lds_ptr_t)(sa + ABYTES + (wave * PB + i) * 1024), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Software pipeline across tiles (NSTB = 3 stages, 2 tiles of DMA in flight):
    //   a tile's LAST fragment set (k-step 3) is read into registers during k-step 2, so its stage is dead at the
    //   "boundary" inside iteration t: [lgkmcnt(0)] [tile t+1 landed] [barrier] -> read k-step 0 of tile t+1, queue the
    //   k-step-3 MFMAs of tile t, and refill stage t%3 with tile t+3 while they run.  LDS latency, the barrier and the DMA
    //   issue of a tile boundary all sit in the shadow of 9 MFMAs instead of in front of them.
    static_assert(NSTB == 3, "pipeline written for a 3-stage ring");
#pragma unroll
    for (int t = 0; t < NSTB; ++t)
        if (t < total) issue(t);
    const int frow = lane & 31, fk = lane >> 5;
    bf16x8 af[2][TM], bf[2][TN];
    auto fload = [&](int st, int ks, int slot) {
        const unsigned char* As = ldsb + st * STAGE;
        const unsigned char* Bs = As + ABYTES;
        const int c = ks * 2 + fk;
        if constexpr (ABL == 3) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) af[slot][i][e] = (bf16_t)(float)(st + e + i);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) bf[slot][j][e] = (bf16_t)(float)(ks + e + j);
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const bf16x8*>(As + lds_off_t<64>(wr * 32 * TM + i * 32 + frow, c));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<64>(wc * 32 * TN + j * 32 + frow, c));
    };
    auto mfmas = [&](int slot) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (ABL == 2) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) acc[i][j][e] += (float)af[slot][i][e] * (float)bf[slot][j][e];
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[slot][j], af[slot][i], acc[i][j], 0, 0, 0);
                }
            }
    };
    if (total > 2) wait_vmcnt<2 * LPT>();
    else if (total > 1) wait_vmcnt<LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (total > 0) fload(0, 0, 0);
    if constexpr (EPI == EPI_F32) { if (p.trace) t_pro = __builtin_amdgcn_s_memtime(); }
    int st = 0;
    for (int t = 0; t < total; ++t) {
        // k-steps 0..2 (fragments of k-step ks+1 are requested before the MFMAs of ks are queued)
        fload(st, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        fload(st, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        fload(st, 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        // boundary
        const int nst = st == NSTB - 1 ? 0 : st + 1;
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): k-step 3 fragments are in registers -> stage st is dead
        if (t + 1 < total) {
            if (t + 2 < total) wait_vmcnt<LPT>(); else wait_vmcnt<0>();     // tile t+1 landed (tile t+2 may stay in flight)
        }
        __builtin_amdgcn_s_barrier();
        if (t + 1 < total) fload(nst, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 1 && t + NSTB < total) issue(t + NSTB);               // refills stage st
        st = nst;
    }

    if constexpr (EPI == EPI_F32) { if (p.trace) t_loop = __builtin_amdgcn_s_memtime(); }
    // gated residual on the 192 x 192 and 64 x 64 tiles (out-proj / unfused band w2 at 32 clips, long form, one or two clips): residual loads
    // requested before the staging, like the fused band-expert kernel (see staged_epilogue HOIST)
    if constexpr (EPI == EPI_RESID_GATE && ((TM == 3 && TN == 3) || (TM == 1 && TN == 1))) {
        if (!p.epi_old) staged_epilogue<EPI, TM, TN, 2, NTHREADS, true>(p, g, acc, reinterpret_cast<float*>(ldsb), row0, rows_end, n0, tid, wr, wc, frow, fk);
        else staged_epilogue<EPI, TM, TN>(p, g, acc, reinterpret_cast<float*>(ldsb), row0, rows_end, n0, tid, wr, wc, frow, fk);
    } else
    staged_epilogue<EPI, TM, TN>(p, g, acc, reinterpret_cast<float*>(ldsb), row0, rows_end, n0, tid, wr, wc, frow, fk);
    if constexpr (EPI == EPI_F32) {
        if (p.trace && tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* tr = p.trace + (size_t)blockIdx.x * 4;
            tr[0] = t_start; tr[1] = t_loop; tr[2] = __builtin_amdgcn_s_memtime();
            tr[3] = ((unsigned long long)(t_pro - t_start) << 36) | ((unsigned long long)xcc << 32) | hwid;
        }
    }
}

// ---- variant 4: 256 x 256 tiles, 8 waves.  Round 2 measured it slower inside the DiT (quad-layout epilogues, docs/history.md, round 2) and
// kept it in the experiments build; with the P16 epilogues of round 3 it wins on the two wide projections (round 4, same box:
// QKV + RoPE 65.3 -> 60.7 us, whole two-stream run +3.2 %, two clips 56.8 -> 51.8 ms) and the product library instantiates exactly those
// two forms (software-pipelined schedule, P16 layout; ring shape: launch_p8_product).  The other schedules / ring shapes /
// ablations below compile only with -DVB_EXPERIMENTS (`VB_BUILD_EXPERIMENTS=1 python -m versband_amd.build`) ----
// 256 x 256 tiles, 8 waves in two staggered groups ("ping-pong") ------------------------------------
// What bounds the 4-wave kernels above is the L2 -> LDS feed (52-60 GB/s per CU whatever the ring, section 5 of DESIGN.md): at
// 64 flop per byte DMA'd (128^2) or 96 (192^2) the MFMA pipe idles half the time, and with one wave per SIMD every barrier, DMA
// issue and LDS read of a wave is dead time of its SIMD's matrix pipe.  This kernel
//   * moves 128 flop per byte DMA'd: block tile 256 x 256 (accumulators = half of the CU's register file), 8 waves as 2 (M) x 4
//     (N), a wave owns 128 x 64 = 4 x 2 MFMA 32x32 tiles;
//   * keeps (NST-1) ring stages of [256 + 256 rows] x BKT in flight (BKT = 32: five 32-KB stages = all 160 KB of LDS, 128 KB in
//     flight) with counted vmcnt - never a drain inside the loop;
//   * runs the two waves of every SIMD in OPPOSITE phases: the wave rows (wr = 0 / 1) are staggered by one s_barrier, so while
//     one wave issues its 16 MFMAs of a 32-deep sub-stage (512 matrix-pipe cycles, s_setprio 1) its partner reads the 12
//     fragments of its next sub-stage from LDS, issues its DMA pieces and waits for them - load latency, DMA issue and the
//     barriers sit beside the partner's MFMAs instead of in front of one's own (MI355X_MICROARCH.md, "Two waves per SIMD").
// Barrier algebra (k-th s_barrier of group 0 pairs with the k-th of group 1; group 1 executes one extra barrier first):
//   group 0:  L(0) X M(0) Y L(1) X M(1) Y ...          group 1:  E L(0) X M(0) Y L(1) ...
//   -> g0's X(u) = g1's Y(u-1), g0's Y(u) = g1's X(u).  Stage t is read by g0 in L_g0(t) and by g1 one half-period later; its
//   slot is refilled in the L phase of stage t+1 (both groups are past their reads by then), and "stage t+1 has landed
//   everywhere" is each wave's counted vmcnt in its last L phase of stage t followed by the barrier both groups pass before
//   either reads it.  Past the end the ring keeps issuing dummy reloads into dead slots so the counts stay uniform.
// Accumulation order per output element is the same as in every other variant (k ascending in 16-deep MFMA steps): bit-identical.
// PP = 1: the ping-pong schedule above.  PP = 0: every wave is software-pipelined on its own instead - the fragments of k-step
// ks+1 are requested before the 8 MFMAs of k-step ks are queued, the DMA pieces of the stage NST-1 ahead are issued between those
// MFMAs, ONE barrier per ring stage whose latency (and the next stage's first fragment reads) hides behind the last 8 MFMAs of
// the stage; the two waves of a SIMD are not synchronised against each other and simply fill each other's issue gaps.
// ABL (tuning only, PP = 0): 1 = no DMA inside the loop, 2 = no fragment reads, 3 = no MFMA, 4 = no barrier, 5 = no epilogue
template <int EPI, int BKT, int NST, int PP, int ABL = 0, int STG = 1>
__global__ void __launch_bounds__(512) gemm_bf16_p8_kernel(const GemmDev p) {
    constexpr int BMP = 256, BNP = 256;
    constexpr int NSUB = BKT / 32;                 // 32-deep sub-stages per ring stage
    constexpr int OPB = BMP * BKT * 2;             // bytes of one operand of a stage
    constexpr int STAGE = 2 * OPB;
    constexpr int PPW = OPB / 1024 / 8;            // 1-KB DMA pieces per wave per operand per stage
    constexpr int LPT = 2 * PPW;                   // DMA instructions per wave per stage
    constexpr int CH = BKT / 8;                    // 16-B chunks per tile row
    constexpr int RPP = 64 / CH;                   // tile rows per piece
    static_assert(BKT == 32 || BKT == 64, "BKT");
    static_assert(NST >= 2 && (NST - 2) * LPT < 64, "ring depth vs vmcnt range");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsp[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    int g = 0, row0, rows_end, tile_n;
    {
        const int L = blockIdx.x, nN = p.n_tiles;
        const int jx = L >> 3;
        tile_n = jx % nN;
        int tmg = (jx / nN) * 8 + (L & 7);
        if (p.group_off) {
            bool found = false;
            for (int gi = 0; gi < p.ngroups; ++gi) {
                int lo = p.group_off[gi], hi = p.group_off[gi + 1];
                int nt = (hi - lo + BMP - 1) / BMP;
                if (tmg < nt) { g = gi; row0 = lo + tmg * BMP; rows_end = hi; found = true; break; }
                tmg -= nt;
            }
            if (!found) return;
        } else {
            g = blockIdx.z;
            row0 = tmg * BMP; rows_end = p.M;
            if (row0 >= rows_end) return;
        }
    }
    const int n0 = tile_n * BNP;
    const int KT = p.K / BKT;
    const int total = KT * p.nseg;

    const bf16_t* asrc[PPW]; const bf16_t* bsrc[PPW];
    {
        const int rr = lane / CH, cs = lane % CH;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int r = RPP * (wave * PPW + i) + rr;
            const int c = (BKT == 64) ? (cs ^ ((r >> 1) & 7)) : (cs ^ ((r >> 2) & 3));
            int slot = row0 + r;
            if (slot >= rows_end) slot = row0;
            const int arow = p.a_rows ? p.a_rows[slot] : slot;
            asrc[i] = p.A + (int64_t)arow * p.lda + g * p.a_koff_group + c * 8;
            int nrow = n0 + (STG == 2 ? p16_src_row(r) : r);       // STG = 2: P16 column layout (16 consecutive output columns per lane)
            if (nrow >= p.N) nrow = 0;
            bsrc[i] = p.B + g * p.b_group_stride + (int64_t)nrow * p.ldb + c * 8;
        }
    }
    // issue state: stage counter, its ring slot and its (segment, k) position - advanced incrementally (scalar adds / selects, no
    // division in the loop); past the last stage the position stays on the last one: dummy reloads into dead slots
    int iss_t = 0, iss_slot = 0, iss_kt = 0, iss_seg = 0;
    auto issue_pieces = [&](int q0, int q1) {           // DMA pieces [q0, q1) of the current issue stage
        const int64_t ao = (iss_seg == 1 ? p.a_plane : 0) + iss_kt * BKT;
        const int64_t bo = (iss_seg == 2 ? p.b_plane : 0) + iss_kt * BKT;
        unsigned char* sa = ldsp + iss_slot * STAGE;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (i < q0 || i >= q1) continue;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[i] + ao), (lds_ptr_t)(sa + (wave * PPW + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(bsrc[i] + bo), (lds_ptr_t)(sa + OPB + (wave * PPW + i) * 1024), 16, 0, 0);
        }
    };
    auto issue_advance = [&]() {
        iss_t += 1;
        iss_slot = iss_slot == NST - 1 ? 0 : iss_slot + 1;
        if (iss_t < total) {
            iss_kt += 1;
            if (iss_kt == KT) { iss_kt = 0; iss_seg += 1; }
        }
    };
    auto issue = [&]() { issue_pieces(0, PPW); issue_advance(); };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fk = lane >> 5;
    bf16x8 af[2][4], bf[2][2];
    auto fload = [&](int st, int sub) {
        const unsigned char* As = ldsp + st * STAGE;
        const unsigned char* Bs = As + OPB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = (sub * 2 + ks) * 2 + fk;
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[ks][j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<BKT>(wc * 64 + j * 32 + frow, c));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(As + lds_off_t<BKT>(wr * 128 + i * 32 + frow, c));
        }
    };

#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue();
    wait_vmcnt<(NST - 2) * LPT>();                 // this wave's pieces of stage 0 have landed
    __builtin_amdgcn_s_barrier();
    if constexpr (PP == 0) {
        constexpr int KS = BKT / 16;               // 16-deep k-steps per ring stage
        bf16x8 fa[2][4], fb[2][2];
        auto fload1 = [&](int st_, int ks, int slot) {
            if constexpr (ABL == 2) { if (st_ >= 0) return; }
            const unsigned char* As = ldsp + st_ * STAGE;
            const unsigned char* Bs = As + OPB;
            const int c = ks * 2 + fk;
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[slot][j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<BKT>(wc * 64 + j * 32 + frow, c));
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[slot][i] = *reinterpret_cast<const bf16x8*>(As + lds_off_t<BKT>(wr * 128 + i * 32 + frow, c));
        };
        auto mfma2 = [&](int slot, int i) {
            if constexpr (ABL == 3) {
                acc[i][0][0] += (float)fb[slot][0][0] * (float)fa[slot][i][1];
                acc[i][1][0] += (float)fb[slot][1][2] * (float)fa[slot][i][3];
                return;
            }
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[slot][0], fa[slot][i], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[slot][1], fa[slot][i], acc[i][1], 0, 0, 0);
        };
        if constexpr (ABL == 2) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) fb[sl][j][e] = (bf16_t)(float)(lane + e + j);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) fa[sl][i][e] = (bf16_t)(float)(lane - e + i);
            }
        }
        if (total > 0) fload1(0, 0, 0);
        int st = 0;
        for (int t = 0; t < total; ++t) {
            const int nst = st == NST - 1 ? 0 : st + 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cur = ks & 1;
                if (ks == KS - 1) {
                    // stage boundary: my reads of stage t are in registers; after the barrier stage t+1 is here for everyone and
                    // the slot of stage t may be refilled
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    if constexpr (ABL != 1) wait_vmcnt<(NST - 2) * LPT>();
                    if constexpr (ABL != 4) __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // 8 MFMAs of k-step ks.  The NEXT k-step's fragment reads are issued two MFMAs INTO the batch (not in front of
                // it): the compiler waits lgkmcnt(0) before a batch's first MFMA, and this way nothing younger than the batch's
                // own fragments is outstanding at that point - the reads fly under the remaining six MFMAs.  The DMA pieces of
                // the stage NST-1 ahead go between the MFMA pairs of the stage's first k-step.
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    mfma2(cur, i);
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == 0) {
                        if (ks < KS - 1) fload1(st, ks + 1, cur ^ 1);
                        else fload1(nst, 0, cur ^ 1);        // (past the last stage: a landed dummy reload)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (ABL != 1 && ks == 0 && i < PPW) {
                        issue_pieces(i, i + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (ks == 0) issue_advance();
            }
            st = nst;
        }
        wait_vmcnt<0>();
        if constexpr (ABL == 5) {
            float sink = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) sink += acc[i][0][0] + acc[i][1][9];
            if (sink == 12345.678f) p.out32[0] = sink;
            return;
        }
        if constexpr (STG == 2) {
            // round 3 (prepared, NOT yet measured): the P16 epilogues of the 4-wave kernels on the 256 x 256 tile - the quad-layout epilogue
            // was what this kernel lost on in round 2 (docs/history.md, round 3 open items); the V third keeps the row-per-lane stores
            static_assert(EPI == EPI_QKV_ROPE || EPI == EPI_SWIGLU, "P16 epilogues exist for QKV + RoPE and SwiGLU");
            if constexpr (EPI == EPI_QKV_ROPE) wave_epilogue_qkv_p16<4, 2>(p, acc, row0 + wr * 128, rows_end, n0 + wc * 64, frow, fk);
            else wave_epilogue_swiglu_p16<4, 2>(p, g, acc, row0 + wr * 128, rows_end, n0 + wc * 64, frow, fk);
        }
        else if constexpr (STG) staged_epilogue<EPI, 4, 2, 4, 512>(p, g, acc, reinterpret_cast<float*>(ldsp), row0, rows_end, n0, tid, wr, wc, frow, fk);
        else wave_epilogue<EPI, 4, 2>(p, g, acc, row0 + wr * 128, rows_end, n0 + wc * 64, frow, fk);
        return;
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();     // stagger: group 1 runs one half-period behind group 0
    int st = 0;
    for (int t = 0; t < total; ++t) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            // ---- L: refill the slot both groups have left, fetch this sub-stage's fragments, make sure stage t+1 is here
            if (sub == 0) issue();
            fload(st, sub);
            if (sub == NSUB - 1) wait_vmcnt<(NST - 2) * LPT>();
            __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): fragments in registers
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();          // X
            __builtin_amdgcn_sched_barrier(0);
            // ---- M: 16 MFMAs, the partner wave of this SIMD is in its L phase
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();          // Y
            __builtin_amdgcn_sched_barrier(0);
        }
        st = st == NST - 1 ? 0 : st + 1;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();     // re-align the two groups
    wait_vmcnt<0>();                               // the dummy tail reloads must have landed before LDS becomes the staging slab
    staged_epilogue<EPI, 4, 2, 4, 512>(p, g, acc, reinterpret_cast<float*>(ldsp), row0, rows_end, n0, tid, wr, wc, frow, fk);
}

// p8 variants (VB_GEMM_P8): ping-pong 1 = BK 32 x 5 stages, 2 = BK 32 x 4, 3 = BK 64 x 2; software-pipelined 4 / 5 / 6 = the same rings
template <int EPI, int BKT, int NST, int PP, int ABL = 0, int STG = 1>
static void launch_p8_v(const GemmDev& d, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)NST * 2 * 256 * BKT * 2;
    static_assert(lds >= (size_t)256 * 68 * 4 && lds <= 160 * 1024, "LDS budget");
    static OnceFlags attr;
    vb_set_max_lds_once(attr, reinterpret_cast<const void*>(gemm_bf16_p8_kernel<EPI, BKT, NST, PP, ABL, STG>), (int)lds);
    hipLaunchKernelGGL((gemm_bf16_p8_kernel<EPI, BKT, NST, PP, ABL, STG>), grid, dim3(512), lds, st, d);
}
// product form (tile configuration 89): software-pipelined schedule, BK 32 x 5 stages, P16 column layout - QKV + RoPE and SwiGLU only
template <int EPI>
static void launch_p8_product(const GemmDev& d, dim3 grid, hipStream_t st) {
    // Ring shape (round 5): 64-deep stages x 2 (128 KB) instead of 32-deep x 5 (160 KB).  A DMA instruction of a 32-deep stage covers 16
    // rows x HALF a 128-B line, and both halves of every line cross the CU's 64 B/clk fill path: tools/probe/tile_feed_probe measures 66 GB/s
    // per CU for that pattern against 110 for full lines - less than a 256 x 256 tile's MFMAs consume at full rate (77).  Two 64-deep stages
    // have one stage of lead instead of four, and still win: same box, whole network evaluation 1821 -> 1786 us at 8 clips, 1237 -> 1218 at 4
    // (QKV + RoPE 64.3 -> 62.0 us, routed SwiGLU unchanged; profiles/r05_p8_ring_ab.txt).  Bit-identical (same k order).
    if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_SWIGLU) {
#ifdef VB_EXPERIMENTS
        if (vb_tune().gemm_p8_ring == 325 && d.K % 32 == 0) { launch_p8_v<EPI, 32, 5, 0, 0, 2>(d, grid, st); return; }       // the round-4 ring (A/B)
        if (vb_tune().gemm_p8_ring == 324 && d.K % 32 == 0) { launch_p8_v<EPI, 32, 4, 0, 0, 2>(d, grid, st); return; }
#endif
        if (d.K % 64 == 0) launch_p8_v<EPI, 64, 2, 0, 0, 2>(d, grid, st);
        else launch_p8_v<EPI, 32, 5, 0, 0, 2>(d, grid, st);
    }
}
#ifdef VB_EXPERIMENTS
#ifndef P8_DEFAULT_BKT
#define P8_DEFAULT_BKT 32
#define P8_DEFAULT_NST 5
#define P8_DEFAULT_PP 0
#endif
template <int EPI>
static void launch_p8(const GemmDev& d, dim3 grid, hipStream_t st, int variant) {
    if constexpr (EPI == EPI_F32) {      // the alternative schedules / ring shapes exist for the micro-benchmark only (tools/gemm_p8_bench.py)
        if (variant == 1) { launch_p8_v<EPI, 32, 5, 1>(d, grid, st); return; }
        if (variant == 2) { launch_p8_v<EPI, 32, 4, 1>(d, grid, st); return; }
        if (variant == 3 && d.K % 64 == 0) { launch_p8_v<EPI, 64, 2, 1>(d, grid, st); return; }
        if (variant == 4) { launch_p8_v<EPI, 32, 5, 0>(d, grid, st); return; }
        if (variant == 5) { launch_p8_v<EPI, 32, 4, 0>(d, grid, st); return; }
        if (variant == 6 && d.K % 64 == 0) { launch_p8_v<EPI, 64, 2, 0>(d, grid, st); return; }
        if (variant == 11) { launch_p8_v<EPI, 32, 5, 0, 1>(d, grid, st); return; }       // ablations of the pipelined 32 x 5 ring
        if (variant == 12) { launch_p8_v<EPI, 32, 5, 0, 2>(d, grid, st); return; }
        if (variant == 13) { launch_p8_v<EPI, 32, 5, 0, 3>(d, grid, st); return; }
        if (variant == 14) { launch_p8_v<EPI, 32, 5, 0, 4>(d, grid, st); return; }
        if (variant == 15) { launch_p8_v<EPI, 32, 5, 0, 5>(d, grid, st); return; }
    }
    if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_SWIGLU) {
        // VB_GEMM_P8_P16: P16 column layout (needs the software-pipelined schedule, K % 64 == 0 and 16-byte aligned output rows, like the 4-wave form)
        if ((vb_tune().gemm_p8_p16 & (1 << EPI)) && P8_DEFAULT_PP == 0 && d.K % 64 == 0 && d.N % 16 == 0 &&
            (EPI == EPI_QKV_ROPE || (d.ldc % 8 == 0 && d.c_noff_group % 8 == 0))) {
            launch_p8_v<EPI, P8_DEFAULT_BKT, P8_DEFAULT_NST, 0, 0, 2>(d, grid, st);
            return;
        }
        if (vb_tune().gemm_p8_direct & (1 << EPI)) { launch_p8_v<EPI, P8_DEFAULT_BKT, P8_DEFAULT_NST, P8_DEFAULT_PP, 0, 0>(d, grid, st); return; }
    }
    launch_p8_v<EPI, P8_DEFAULT_BKT, P8_DEFAULT_NST, P8_DEFAULT_PP>(d, grid, st);
}
template <int EPI> struct P8Epi { static constexpr bool ok = EPI == EPI_PLANES || EPI == EPI_F32 || EPI == EPI_QKV_ROPE || EPI == EPI_RESID_GATE ||
                                                            EPI == EPI_SWIGLU || EPI == EPI_SCATTER_F32 || EPI == EPI_SCATTER_ADD_PLANES; };

#endif  // VB_EXPERIMENTS

template <int EPI, int TM, int TN, int NSTB>
static void launch_big(const GemmDev& d, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)NSTB * (64 * TM + 64 * TN) * 128;
#ifdef VB_EXPERIMENTS
    if constexpr (EPI == EPI_F32 && TM == 3 && TN == 3) {
        const int abl = vb_tune().gemm_ablate;
        if (abl >= 1 && abl <= 3) {
            auto k = abl == 1 ? gemm_bf16_big_kernel<EPI, TM, TN, NSTB, 1> : (abl == 2 ? gemm_bf16_big_kernel<EPI, TM, TN, NSTB, 2> : gemm_bf16_big_kernel<EPI, TM, TN, NSTB, 3>);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, grid, dim3(NTHREADS), lds, st, d);
            return;
        }
    }
#endif
    static_assert(lds >= (size_t)64 * (64 * TN + 4) * 4, "epilogue staging slab must fit in the ring");
    static OnceFlags attr;
    vb_set_max_lds_once(attr, reinterpret_cast<const void*>(gemm_bf16_big_kernel<EPI, TM, TN, NSTB>), (int)lds);
    hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, TM, TN, NSTB>), grid, dim3(NTHREADS), lds, st, d);
}

#ifdef VB_EXPERIMENTS
// ---- variant 5 (round 5): PERSISTENT 8-wave kernel - one workgroup per CU walks a balanced share of the launch ------------------------
// What the 256 x 256 kernel above (variant 4) loses, by its own ablations (profiles/r02_gemm_p8_microbench.txt): a third of every
// launch is the epilogue of a tile with nothing beside it (one workgroup per CU: 61.0 -> 39.7 us without it at 12032 x 2304 x 768),
// the tile grid is quantised (423 tiles on 256 CUs = two rounds for 1.65 rounds of work), every tile starts with a cold ring, and a
// 32-deep stage makes every DMA instruction touch 16 HALF lines (64 of a row's 128 bytes): each 128-B line crosses the CU's 64 B/clk
// L1 fill path twice, 1.57 MB per 100-MFLOP tile = 24.6k cycles - as long as the tile's MFMAs.  This kernel
//   * is launched ONCE per CU.  The rows are cut into 64-row units; XCD x owns the units [U x/8, U (x+1)/8) for all column tiles, its
//     workgroups split that (column tile, unit) range - column-major - into equal contiguous shares (+-1 unit).  A share is walked as
//     tiles of 1..4 units x 256 columns (IM = units: a wave row owns IM 32-row MFMA blocks), cut at column-tile and row-group ends;
//   * keeps ONE DMA ring running across tile boundaries: stages are 64 deep (a DMA instruction = 8 rows x one full 128-B line), the
//     ring has five 32-KB slots holding the items A0 B0 A1 B1 .. (item q -> slot q mod 5); while stage s is multiplied, B(s+1) and
//     A(s+2) are issued into the two slots stage s-1 left: 1.5 stages (96 KB) in flight, the next tile's first stages arrive under
//     the current tile's last MFMAs, and the stage-end wait is the counted vmcnt(4) + one barrier;
//   * stores a tile straight from the accumulators in the P16 layout (no LDS staging - the ring never stops) and goes on: the stores
//     drain under the next tile's MFMAs.  vmcnt counts loads AND stores in issue order per kind; stores between the DMA items only
//     make a counted wait more conservative, never less (loads retire in order among themselves);
//   * writes the V third of QKV with the MFMA's operands exchanged (tokens as the row operand, P16 permutation on the TOKEN rows): a
//     lane owns one head-dim column and 16 consecutive tokens = the V^T image's 32 contiguous bytes, the same instruction operand
//     roles as the 4-wave kernel's V tiles.
// Per output element the products are accumulated in the same order as in every other variant (k ascending, 16 per MFMA): bit-identical.
struct PkTile { int g, row0, rows_end, n0, im; bool valid; };

template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue_vt_pk(const GemmDev& p, f32x16 (&acc)[TM][TN], int tok_base, int rows_end, int n_base, int frow, int fk) {
    // acc[i][j][e]: token tok_base + i*32 + 16*fk + e, projection column n_base + j*32 + frow (see the note above)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_base + j * 32 + frow;
        if (n >= p.N) continue;
        const int nn = n - 2 * p.D;
        const int h = fdiv(nn, p.rhd), d = nn - h * p.hd;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m0 = tok_base + i * 32 + 16 * fk;
            if (m0 >= rows_end) continue;
            const int b = fdiv(m0, p.rT), t = m0 - b * p.T;
            const int64_t base = ((int64_t)(b * p.H + h) * p.hd + d) * p.Tpad + t;
            float o[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = acc[i][j][e];
            if (m0 + 15 < rows_end) {
                store8p(p.vt, p.vt_plane, p.qkv_np, base, o);
                store8p(p.vt, p.vt_plane, p.qkv_np, base + 8, o + 8);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (m0 + e < rows_end) store1p(p.vt, p.vt_plane, p.qkv_np, base + e, o[e]);
            }
        }
    }
}

// q / k thirds of QKV + RoPE in the P16 layout: wave_epilogue_qkv_p16 without its V branch (the persistent kernel writes V tiles with
// wave_epilogue_vt_pk; the launcher takes it only where that form applies) - the 16 two-byte V^T stores of that branch cost 32 address
// registers the tile loop's own state has no room for.  Same arithmetic, element for element.
template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue_qk_p16(const GemmDev& p, f32x16 (&acc)[TM][TN], int row_base, int rows_end, int n_base, int frow, int fk) {
#pragma clang fp contract(off)
    const int hd2 = p.hd >> 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = row_base + i * 32 + frow;
        if (m < rows_end) {
            const int t = m - fdiv(m, p.rT) * p.T;
            float4 cs[TN][2], sn[TN][2];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n_base + j * 32 + fk * 16;
                const int nn = n - fdiv(n, p.rD) * p.D;
                const int jd = (nn - fdiv(nn, p.rhd) * p.hd) >> 1;
                const float* cp = p.rope_cos + (int64_t)t * hd2 + jd;
                const float* sp = p.rope_sin + (int64_t)t * hd2 + jd;
                cs[j][0] = *reinterpret_cast<const float4*>(cp); cs[j][1] = *reinterpret_cast<const float4*>(cp + 4);
                sn[j][0] = *reinterpret_cast<const float4*>(sp); sn[j][1] = *reinterpret_cast<const float4*>(sp + 4);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n_base + j * 32 + fk * 16;
                const int sec = fdiv(n, p.rD);
                const int nn = n - sec * p.D;
                const float c8[8] = {cs[j][0].x, cs[j][0].y, cs[j][0].z, cs[j][0].w, cs[j][1].x, cs[j][1].y, cs[j][1].z, cs[j][1].w};
                const float s8[8] = {sn[j][0].x, sn[j][0].y, sn[j][0].z, sn[j][0].w, sn[j][1].x, sn[j][1].y, sn[j][1].z, sn[j][1].w};
                float o[16];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v0 = acc[i][j][2 * e], v1 = acc[i][j][2 * e + 1];
                    o[2 * e] = fmaf(v0, c8[e], -(v1 * s8[e]));
                    o[2 * e + 1] = fmaf(v0, s8[e], v1 * c8[e]);
                }
                bf16_t* dst = sec == 0 ? p.q : p.k;
                const int64_t pl = sec == 0 ? p.q_plane : p.k_plane;
                store8p(dst, pl, p.qkv_np, (int64_t)m * p.D + nn, o);
                store8p(dst, pl, p.qkv_np, (int64_t)m * p.D + nn + 8, o + 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);        // (a slab's table loads are not hoisted over the previous slab's stores: 32 registers per slab)
    }
}

// gated residual in the P16 layout (the arithmetic of epi_store<EPI_RESID_GATE> / gemm_bf16_wide_resid_kernel): loads of a row slab, then its stores
template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue_resid_p16(const GemmDev& p, int g, f32x16 (&acc)[TM][TN], int row_base, int rows_end, int n_base,
                                                        int frow, int fk) {
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = row_base + i * 32 + frow;
        if (m >= rows_end) continue;
        float* hrow = p.out32 + (int64_t)m * p.ldc32 + g * p.c_noff_group;
        const float* grow = p.gate + (int64_t)fdiv(m, p.rT) * p.gate_ld + g * p.c_noff_group;
        float4 hv[TN][4], gv[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n_base + j * 32 + fk * 16;
            if (n < p.N) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    hv[j][q] = *reinterpret_cast<const float4*>(hrow + n + 4 * q);
                    gv[j][q] = *reinterpret_cast<const float4*>(grow + n + 4 * q);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n_base + j * 32 + fk * 16;
            if (n >= p.N) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 o;
                o.x = fmaf(gv[j][q].x, acc[i][j][4 * q + 0], hv[j][q].x); o.y = fmaf(gv[j][q].y, acc[i][j][4 * q + 1], hv[j][q].y);
                o.z = fmaf(gv[j][q].z, acc[i][j][4 * q + 2], hv[j][q].z); o.w = fmaf(gv[j][q].w, acc[i][j][4 * q + 3], hv[j][q].w);
                *reinterpret_cast<float4*>(hrow + n + 4 * q) = o;
            }
        }
    }
}

template <int V> struct PkInt { static constexpr int value = V; };

#define PK_MAX_GROUPS 16
// MAXG: capacity of the row-group table held in scalar registers (0 = one group, the rows [0, M))
// ABL (experiments build, timing only): 1 = no DMA inside the loop, 2 = no fragment reads, 3 = no MFMA, 5 = no epilogue, 6 = 1 + 2, 7 = 2 + 3;
// TRACE: wave 0 stamps s_memtime at every stage end (arrival, release) and tile end into p.trace (64 words per workgroup)
template <int EPI, int MAXG, int ABL = 0, bool TRACE = false>
__global__ void __launch_bounds__(512) gemm_bf16_pk_kernel(const GemmDev p) {
    constexpr int SLOT = 256 * 128;               // one operand of a 64-deep stage: [256 rows][128 B], 16-B chunks XOR-swizzled by row
    constexpr bool P16L = EPI == EPI_QKV_ROPE || EPI == EPI_SWIGLU || EPI == EPI_RESID_GATE;      // 16 consecutive output columns per lane
    static_assert(EPI == EPI_QKV_ROPE || EPI == EPI_SWIGLU || EPI == EPI_RESID_GATE || EPI == EPI_F32, "epilogues wired for the persistent kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsp[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 31, fk = lane >> 5;
    const int KT = p.K >> 6;
    const int TS = KT * p.nseg;                   // stages per tile

    // ---- the row groups in 64-row units (prefix sums, all wave-uniform).  Index min(g, ng) reads the last offset for the unused entries: 0 units.
    int ucum[MAXG + 1], goff[MAXG + 1];
    if constexpr (MAXG > 0) {
        const int ng = p.ngroups;
#pragma unroll
        for (int g = 0; g <= MAXG; ++g) goff[g] = p.group_off[g < ng ? g : ng];
        ucum[0] = 0;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) ucum[g + 1] = ucum[g] + ((goff[g + 1] - goff[g] + 63) >> 6);
    } else {
        goff[0] = 0; ucum[0] = (p.M + 63) >> 6;
    }
    const int U = ucum[MAXG];
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, PX = gridDim.x >> 3;
    const int ua = U * xcd / 8;
    const int nu = U * (xcd + 1) / 8 - ua;
    if (nu <= 0) return;
    const int nN = p.n_tiles;
    const int Wx = nu * nN;
    const int w0 = (int)((int64_t)Wx * jx / PX), wend = (int)((int64_t)Wx * (jx + 1) / PX);
    if (w0 >= wend) return;
    const float rnu = 1.0f / (float)nu;

    // next tile of the share [w, wend): at most 4 units, inside one column tile and one row group; runs of 5..7 units are halved
    auto next = [&](int& w, PkTile& t) {
        if (w >= wend) { t.valid = false; return; }
        const int n = fdiv(w, rnu);
        const int ul = w - n * nu;
        const int ug = ua + ul;
        int g = 0, lo = 0, hi = p.M, ub = 0, ue = U;
        if constexpr (MAXG > 0) {
#pragma unroll
            for (int i = 1; i < MAXG; ++i) g += ug >= ucum[i] ? 1 : 0;
            lo = goff[0]; hi = goff[1]; ue = ucum[1];
#pragma unroll
            for (int i = 1; i < MAXG; ++i)
                if (g == i) { lo = goff[i]; hi = goff[i + 1]; ub = ucum[i]; ue = ucum[i + 1]; }
        }
        int run = ue - ug;
        if (nu - ul < run) run = nu - ul;
        if (wend - w < run) run = wend - w;
        const int units = run <= 4 ? run : (run >= 8 ? 4 : (run + 1) >> 1);
        t.g = g; t.row0 = lo + 64 * (ug - ub);
        t.rows_end = hi < t.row0 + 64 * units ? hi : t.row0 + 64 * units;
        t.n0 = n * 256; t.im = units; t.valid = true;
        w += units;
    };
    // (QKV: the launcher guarantees T % 16 == 0, Tpad % 8 == 0 and 2 D % 256 == 0 - V tiles are whole tiles and their 16-token runs stay inside a clip)
    auto is_vt = [&](const PkTile& t) { return EPI == EPI_QKV_ROPE && t.n0 >= 2 * p.D; };

    // ---- DMA cursors: A runs two stages, B one stage ahead of the multiplication; each walks the same tile sequence with its own iterator
    PkTile aT, bT, cT;
    int wa = w0, wb = w0, wcu = w0;
    next(wa, aT); next(wb, bT); next(wcu, cT);
    int asrc[4], bsrc[4];                         // element offsets from p.A / p.B (< 2^31: checked by the launcher)
    // (the lane id goes through an empty asm wherever per-tile code starts from it: what is derived from it there - source rows,
    //  swizzled chunks, output columns - would otherwise be hoisted out of the tile loop and held in registers across the mainloop)
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto set_a = [&](const PkTile& t) {
        const int lo_ = opaque_lane();
        const int rr = lo_ >> 3, cs = lo_ & 7;
        const bool vt = is_vt(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * (wave * 4 + i) + rr;
            const int c = cs ^ ((r >> 1) & 7);
            int slot = t.row0 + (vt ? p16_src_row(r) : r);
            if (slot >= t.rows_end) slot = t.row0;
            const int arow = p.a_rows ? p.a_rows[slot] : slot;
            asrc[i] = arow * p.lda + t.g * p.a_koff_group + c * 8;
        }
    };
    auto set_b = [&](const PkTile& t) {
        const int lo_ = opaque_lane();
        const int rr = lo_ >> 3, cs = lo_ & 7;
        const bool vt = is_vt(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * (wave * 4 + i) + rr;
            const int c = cs ^ ((r >> 1) & 7);
            int nrow = t.n0 + ((P16L && !vt) ? p16_src_row(r) : r);
            if (nrow >= p.N) nrow = 0;
            bsrc[i] = t.g * (int)p.b_group_stride + nrow * p.ldb + c * 8;
        }
    };
    set_a(aT); set_b(bT);
    int a_kt = 0, a_seg = 0, a_slot = 0, b_kt = 0, b_seg = 0, b_slot = 1;
    bool in_loop = false;                         // (ablations only)
    auto issue_a = [&](int i) {
        if constexpr (ABL == 1 || ABL == 6) { if (in_loop) return; }       // (prologue issues only)
        const bf16_t* ab = p.A + (a_seg == 1 ? p.a_plane : 0) + a_kt * 64;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(ab + (unsigned)asrc[i]), (lds_ptr_t)(ldsp + a_slot * SLOT + (wave * 4 + i) * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int i) {
        if constexpr (ABL == 1 || ABL == 6) { if (in_loop) return; }
        const bf16_t* bb = p.B + (b_seg == 2 ? p.b_plane : 0) + b_kt * 64;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(bb + (unsigned)bsrc[i]), (lds_ptr_t)(ldsp + b_slot * SLOT + (wave * 4 + i) * 1024), 16, 0, 0);
    };
    // (past the last tile the cursors keep re-issuing their last tile's stages into dead slots: the wait counts stay uniform)
    auto adv_a = [&]() {
        a_slot = a_slot >= 3 ? a_slot - 3 : a_slot + 2;
        if (++a_kt == KT) {
            a_kt = 0;
            if (++a_seg == p.nseg) {
                a_seg = 0;
                if (aT.valid) { next(wa, aT); if (aT.valid) set_a(aT); }
            }
        }
    };
    auto adv_b = [&]() {
        b_slot = b_slot >= 3 ? b_slot - 3 : b_slot + 2;
        if (++b_kt == KT) {
            b_kt = 0;
            if (++b_seg == p.nseg) {
                b_seg = 0;
                if (bT.valid) { next(wb, bT); if (bT.valid) set_b(bT); }
            }
        }
    };

    f32x16 acc[4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    // prologue: A0 B0 A1, stage 0 landed everywhere
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a(i);
    adv_a();
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_b(i);
    adv_b();
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_a(i);
    adv_a();
    wait_vmcnt<4>();
    __builtin_amdgcn_s_barrier();
    in_loop = true;
    unsigned long long* trp = nullptr;            // (TRACE) this workgroup's 64 stamps: [0] start, then per stage (arrive, release), tile ends interleaved as they come
    int trn = 0;
    auto stamp = [&]() {
        if constexpr (TRACE) {
            if (trp && trn < 64 && wave == 0) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) trp[trn] = t;
                ++trn;
            }
        }
    };
    if constexpr (TRACE) { if (p.trace) trp = p.trace + (size_t)blockIdx.x * 64; }
    stamp();

    bf16x8 fa[2][4], fb[2][2];
    int cA = 0, cB = 1;                           // ring slots of the stage being multiplied
    int loff[4];                                  // lane part of a fragment's LDS address per 16-deep k-step: row frow, chunk (2 ks + fk) ^ swizzle(frow)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) loff[ks] = frow * 128 + (((ks * 2 + fk) ^ ((frow >> 1) & 7)) << 4);
    // one tile: im 32-row MFMA blocks per wave row (wave-uniform branches around the blocks a short tile does not have),
    // VT = exchanged MFMA operand roles (V third of QKV)
    auto run_tile = [&](auto VTc, const int im) {
        constexpr bool VT = decltype(VTc)::value != 0;
        // fragment addresses = (slot base + the wave's row block: uniform) + the lane's swizzled offset of the k-step (loff) + i * 4096 (immediate)
        const int abase = wr * 32 * im * 128, bbase = wc * 64 * 128;
        auto fload = [&](int sA, int sB, int ks, int buf) {
            if constexpr (ABL == 2 || ABL == 6 || ABL == 7) { if (in_loop) return; }
            const unsigned char* Ab = ldsp + (sA * SLOT + abase) + loff[ks];
            const unsigned char* Bb = ldsp + (sB * SLOT + bbase) + loff[ks];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[buf][j] = *reinterpret_cast<const bf16x8*>(Bb + j * 4096);
            // (all four row blocks whatever im: an unconditional definition keeps the fragment registers dead across the epilogue; the
            //  addresses of blocks a short tile does not have stay inside the slot)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(Ab + i * 4096);
        };
        auto mfma2 = [&](int buf, int i) {
            if constexpr (ABL == 3 || ABL == 7) {
                acc[i][0][0] += (float)fb[buf][0][0] * (float)fa[buf][i][1];
                acc[i][1][0] += (float)fb[buf][1][2] * (float)fa[buf][i][3];
                return;
            }
            if constexpr (VT) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i], fb[buf][0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i], fb[buf][1], acc[i][1], 0, 0, 0);
            } else {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][0], fa[buf][i], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][1], fa[buf][i], acc[i][1], 0, 0, 0);
            }
        };
        fload(cA, cB, 0, 0);
        for (int t = 0; t < TS; ++t) {
            const int nA = cA >= 3 ? cA - 3 : cA + 2, nB = cB >= 3 ? cB - 3 : cB + 2;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1;
                if (ks == 3) {
                    // stage boundary: my reads of this stage are in registers; after the barrier the next stage is here for everyone
                    // and this stage's two slots may be refilled (the DMA issues of the next stage's first two k-steps)
                    stamp();
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    wait_vmcnt<4>();
                    __builtin_amdgcn_s_barrier();
                    stamp();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < im) {
                        mfma2(cur, i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (i == 0) {
                        // next k-step's fragments, requested two MFMAs into this k-step's batch (they fly under the rest of it); the
                        // first k-step of the next TILE is requested after the epilogue instead (its registers are the epilogue's)
                        if (ks < 3) fload(cA, cB, ks + 1, cur ^ 1);
                        else if (t + 1 < TS) fload(nA, nB, 0, cur ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (ks == 0) { issue_b(i); __builtin_amdgcn_sched_barrier(0); }
                    if (ks == 1) { issue_a(i); __builtin_amdgcn_sched_barrier(0); }
                }
                if (ks == 0) adv_b();
                if (ks == 1) adv_a();
            }
            cA = nA; cB = nB;
        }
    };

    while (cT.valid) {
        const bool vt = is_vt(cT);
        if constexpr (EPI == EPI_QKV_ROPE) {
            if (vt) run_tile(PkInt<1>(), cT.im);
            else run_tile(PkInt<0>(), cT.im);
        } else {
            run_tile(PkInt<0>(), cT.im);
        }
        // ---- epilogue straight from the accumulators; rows of the blocks a wave row does not own (IM < 4) are masked by its row end
        const int row_base = cT.row0 + wr * 32 * cT.im;
        int rows_end_w = row_base + 32 * cT.im;
        if (cT.rows_end < rows_end_w) rows_end_w = cT.rows_end;
        const int n_base = cT.n0 + wc * 64;
        if constexpr (ABL == 5) {
            float sink = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) sink += acc[i][0][0] + acc[i][1][9];
            if (sink == 12345.678f) p.out32[0] = sink;
            zero_acc();
            next(wcu, cT);
            continue;
        }
        const int le_ = opaque_lane();
        const int frow_e = le_ & 31, fk_e = le_ >> 5;
        if constexpr (EPI == EPI_QKV_ROPE) {
            if (vt) wave_epilogue_vt_pk<4, 2>(p, acc, row_base, rows_end_w, n_base, frow_e, fk_e);
            else wave_epilogue_qk_p16<4, 2>(p, acc, row_base, rows_end_w, n_base, frow_e, fk_e);
        } else if constexpr (EPI == EPI_SWIGLU) {
            wave_epilogue_swiglu_p16<4, 2>(p, cT.g, acc, row_base, rows_end_w, n_base, frow_e, fk_e);
        } else if constexpr (EPI == EPI_RESID_GATE) {
            wave_epilogue_resid_p16<4, 2>(p, cT.g, acc, row_base, rows_end_w, n_base, frow_e, fk_e);
        } else {
            wave_epilogue<EPI, 4, 2>(p, cT.g, acc, row_base, rows_end_w, n_base, frow_e, fk_e);
        }
        stamp();
        zero_acc();
        next(wcu, cT);
    }
    wait_vmcnt<0>();                              // the cursors' dummy tail loads must have landed before the workgroup's LDS is handed on
}

static int vb_num_cus() {
    static std::atomic<int> cached[64] = {};
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 63;
    int n = cached[d].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
        cached[d].store(n, std::memory_order_relaxed);
    }
    return n;
}
template <int EPI, int MAXG>
static void launch_pk_g(const GemmDev& d, hipStream_t st) {
    static OnceFlags attr;
    vb_set_max_lds_once(attr, reinterpret_cast<const void*>(gemm_bf16_pk_kernel<EPI, MAXG>), 5 * 256 * 128);
    const int grid = vb_num_cus() / 8 * 8;        // one workgroup per CU (160 KB of LDS each)
    hipLaunchKernelGGL((gemm_bf16_pk_kernel<EPI, MAXG>), dim3(grid), dim3(512), 5 * 256 * 128, st, d);
}
template <int ABL, bool TRACE>
static void launch_pk_x(const GemmDev& d, hipStream_t st) {
    static OnceFlags attr;
    vb_set_max_lds_once(attr, reinterpret_cast<const void*>(gemm_bf16_pk_kernel<EPI_F32, 0, ABL, TRACE>), 5 * 256 * 128);
    hipLaunchKernelGGL((gemm_bf16_pk_kernel<EPI_F32, 0, ABL, TRACE>), dim3(vb_num_cus() / 8 * 8), dim3(512), 5 * 256 * 128, st, d);
}
// plain fp32-output GEMM on the persistent kernel (tools/gemm_pk_bench.py): VB_GEMM_PK_F32 = 1 + ablation code, 100 = traced
static void launch_pk_f32(const GemmDev& d0, hipStream_t st, int code) {
    GemmDev d = d0;
    d.n_tiles = cdiv(d.N, 256);
    switch (code) {
        case 100: launch_pk_x<0, true>(d, st); break;
        case 2: launch_pk_x<1, false>(d, st); break;
        case 3: launch_pk_x<2, false>(d, st); break;
        case 4: launch_pk_x<3, false>(d, st); break;
        case 6: launch_pk_x<5, false>(d, st); break;
        case 7: launch_pk_x<6, false>(d, st); break;
        case 8: launch_pk_x<7, false>(d, st); break;
        default: launch_pk_x<0, false>(d, st); break;
    }
}
template <int EPI>
static void launch_pk(const GemmDev& d0, hipStream_t st) {
    GemmDev d = d0;
    d.n_tiles = cdiv(d.N, 256);
    if (!d.group_off) launch_pk_g<EPI, 0>(d, st);
    else if constexpr (EPI != EPI_QKV_ROPE) {       // (QKV has no row groups)
        if (d.ngroups <= 8) launch_pk_g<EPI, 8>(d, st);
        else launch_pk_g<EPI, PK_MAX_GROUPS>(d, st);
    }
}

#endif  // VB_EXPERIMENTS (persistent kernel)

template <int EPI>
static void launch_t(const GemmDev& d, dim3 grid, hipStream_t st) {
    // 128 x 128 tiles, two workgroups per CU: tile DMA (global_load_lds) with BK = 64 x 2 stages; K % 64 = 32 (96-channel bands at 8
    // experts) takes BK = 32 x 4 stages; anything else the register-staged kernel.  (tools/gemm_bench.py compared further ring shapes:
    // VB_GEMM_VARIANT / VB_GEMM_ABLATE exist in the experiments build only.)
#ifdef VB_EXPERIMENTS
    const int variant = vb_tune().gemm_variant;
    const int abl = vb_tune().gemm_ablate;
    if constexpr (EPI == EPI_F32) {
        if (abl == 1 && d.K % 64 == 0) { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 2, 1>), grid, dim3(NTHREADS), 0, st, d); return; }
        if (abl == 2 && d.K % 64 == 0) { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 2, 2>), grid, dim3(NTHREADS), 0, st, d); return; }
        if (abl == 3 && d.K % 64 == 0) { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 2, 3>), grid, dim3(NTHREADS), 0, st, d); return; }
        if (abl == 4 && d.K % 64 == 0) { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 2, 4>), grid, dim3(NTHREADS), 0, st, d); return; }
        if (abl == 5 && d.K % 64 == 0) { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 2, 5>), grid, dim3(NTHREADS), 0, st, d); return; }
    }
    if (variant == 0) { hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, grid, dim3(NTHREADS), 0, st, d); return; }
    if (variant == 2 && d.K % 32 == 0) { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 32, 4>), grid, dim3(NTHREADS), 0, st, d); return; }
    if (variant == 3 && d.K % 32 == 0) { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 32, 3>), grid, dim3(NTHREADS), 0, st, d); return; }
    if ((variant == 4 || variant == 5) && d.K % 64 == 0) {
        // deeper rings of the 128 x 128 tile (3 stages = 96 KB, 4 = 128 KB: one workgroup per CU) - the one-clip A/B of round 5: SLOWER, 41.6 ms per
        // clip with either against 40.8 with the two-stage ring at two workgroups per CU (same box, two runs each, parity ok): DESIGN 5.0
        constexpr bool p16able = EPI == EPI_QKV_ROPE || EPI == EPI_SWIGLU;
        bool p16 = false;
        if constexpr (EPI == EPI_QKV_ROPE) p16 = d.hd % 16 == 0 && d.D % 16 == 0 && d.N % 16 == 0 && !d.group_off && d.ngroups <= 1 && !vb_tune().qkv_p16_off;
        if constexpr (EPI == EPI_SWIGLU) p16 = d.N % 16 == 0 && d.ldc % 8 == 0 && d.c_noff_group % 8 == 0 && !vb_tune().qkv_p16_off;
        if constexpr (p16able) {
            if (p16) {
                if (variant == 4) { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 3, 0, true>), grid, dim3(NTHREADS), 0, st, d); }
                else { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 4, 0, true>), grid, dim3(NTHREADS), 0, st, d); }
                return;
            }
        }
        if (variant == 4) { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 3>), grid, dim3(NTHREADS), 0, st, d); }
        else { hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 4>), grid, dim3(NTHREADS), 0, st, d); }
        return;
    }
#endif
    if constexpr (EPI == EPI_QKV_ROPE) {
        // QKV + RoPE: 16 consecutive columns per lane (P16 layout) whenever heads and sections are 16-aligned and no grouping is involved
        if (d.K % 64 == 0 && d.hd % 16 == 0 && d.D % 16 == 0 && d.N % 16 == 0 && !d.group_off && d.ngroups <= 1 && !vb_tune().qkv_p16_off) {
            hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 2, 0, true>), grid, dim3(NTHREADS), 0, st, d);
            return;
        }
    }
    if constexpr (EPI == EPI_SWIGLU) {
        // SwiGLU (routed w1/w3, gathered rows, grouped): P16 layout, 16-byte hidden stores straight from the accumulators instead of
        // the LDS-staged slab (no epilogue barriers); ldc and the group's column offset must keep the stores 16-byte aligned
        if (d.K % 64 == 0 && d.N % 16 == 0 && d.ldc % 8 == 0 && d.c_noff_group % 8 == 0 && !vb_tune().qkv_p16_off) {
            hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 2, 0, true>), grid, dim3(NTHREADS), 0, st, d);
            return;
        }
    }
    if (d.K % 64 == 0) hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 64, 2>), grid, dim3(NTHREADS), 0, st, d);
    else if (d.K % 32 == 0) hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, 32, 4>), grid, dim3(NTHREADS), 0, st, d);   // K = 96 bands (8 experts)
    else hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, grid, dim3(NTHREADS), 0, st, d);
}


// ---- band-expert FFN, fused ------------------------------------------------------------------------------------------
// Band-MoE "frequency" experts (vocal2music_moe.py:171-178; FeedForward flag_large_dit_moe.py:480-485): expert e sees only
// the 192-channel band e of y and produces only band e:   z[:, band e] = W2_e . ( silu(W1_e y_e) * (W3_e y_e) ).
// As two grouped GEMMs this wrote and re-read the [N][E*512] hidden tensor (98 MB per block evaluation) and streamed every
// weight byte through L2->LDS at 64 flop/B (the feed that bounds these K = 192 / 512 GEMMs).  Here one
// workgroup owns 192 tokens x one band (12032 tokens x 4 bands = 252 workgroups = one round of the 256 CUs):
//   * the token tile's band y_e [192 x 192] is DMA'd into LDS once and stays;
//   * the hidden dimension is walked in chunks of 64: w1/w3 rows of the chunk (interleaved, [128 x 192]) stream through a
//     2-stage ring as three K-slabs -> acc1 [192 x 128] -> SwiGLU lane-locally -> bf16 chunk [192 x 64] into LDS as the A
//     operand of the second product -> one slab of w2 [192 x 64] -> acc2 [192 x 192] += ...   (171 flop per byte DMA'd);
//   * the gated residual epilogue of the unfused w2 GEMM is reused as is (staged_epilogue<EPI_RESID_GATE>).
// bf16 (np = 1) only: the split-precision parity mode keeps the two-GEMM path.
static unsigned long long* g_gemm_trace = nullptr;
extern "C" void vbdbg_gemm_trace(void* buf) { g_gemm_trace = static_cast<unsigned long long*>(buf); }   // tuning tool hook, not ABI

struct BandDev {
    GemmDev ep;                    // epilogue view: out32/ldc32, gate/gate_ld, T/rT, c_noff_group = band, N = band
    const bf16_t* Y; int ldy;      // [M][ldy], band e at column e * 192
    const bf16_t* W13; const bf16_t* W2;   // [E][2H][192] (w1/w3 rows interleaved), [E][192][H]
    int M, H, E;
};
#define BF_BM 192
#define BF_BAND 192
template <bool HOIST>      // HOIST: see staged_epilogue (VB_BAND_EPI_OLD=1 selects the two-pass form)
__global__ void __launch_bounds__(NTHREADS) band_ffn_kernel(const BandDev p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bl[];
    constexpr int HCH = BF_BM * 128;              // bytes of the [192 x 64] bf16 hidden chunk
    constexpr int NSLOT = 8;                      // ring slots of 16 KB
    unsigned char* Hs = bl;
    unsigned char* ring = bl + HCH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 31, fk = lane >> 5;
    // XCD x handles band x % E (the band's 590 KB of weights stay in that XCD's L2), two XCDs per band at E = 4
    const int L = blockIdx.x;
    const int e = (L & 7) % p.E;
    const int rt = (L >> 3) * (8 / p.E) + (L & 7) / p.E;
    const int row0 = rt * BF_BM;
    if (row0 >= p.M) return;
    const int rows_end = p.M;
    const int r8 = lane >> 3, cs = lane & 7;
    unsigned long long tq0 = 0, tq1 = 0, tq2 = 0;
    if (p.ep.trace) tq0 = __builtin_amdgcn_s_memtime();

    // The token tile's band y_e [192 x 192] is this workgroup's A operand for the whole first product: every wave keeps the
    // fragments of its 96 rows in registers (3 row tiles x 12 k-steps x 16 B per lane = 144 VGPRs, loaded once), which leaves
    // the LDS to the weight stream.
    bf16x8 ay[3][12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int row = row0 + wr * 96 + i * 32 + frow;
        if (row >= rows_end) row = row0;
        const bf16_t* src = p.Y + (int64_t)row * p.ldy + e * BF_BAND + fk * 8;
#pragma unroll
        for (int kk = 0; kk < 12; ++kk) ay[i][kk] = *reinterpret_cast<const bf16x8*>(src + kk * 16);
    }
    const bf16_t* w13 = p.W13 + (int64_t)e * 2 * p.H * BF_BAND;
    const bf16_t* w2 = p.W2 + (int64_t)e * BF_BAND * p.H;
    // weight stream: per hidden chunk 5 loads - three K-slabs of w13 [128 x 64] (16 KB, 4 DMA pieces per wave) and two K-halves
    // of the w2 slab [192 x 32] (12 KB, 3 pieces per wave) - through a ring of eight 16-KB slots, SEVEN loads ahead of the one
    // being multiplied.  A load is 18-24 MFMAs of work per wave (0.4 us) against a ~2.3 us L2 round trip: the stream rate is
    // (bytes in flight) / latency, so one-ahead double buffering ran at 69 us per launch and three-ahead at 59; the waits are
    // counted vmcnt, never a drain.
    int a_off[4], b_off[3];          // element offsets of this lane's DMA pieces inside a w13 slab / a w2 half-slab
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * (wave * 4 + i) + r8;
        a_off[i] = r * BF_BAND + ((cs ^ ((r >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = 16 * (wave * 3 + i) + (lane >> 2);
        b_off[i] = r * p.H + (((lane & 3) ^ ((r >> 2) & 3)) << 3);
    }
    const int nchunk = p.H / 64;
    const int nload = nchunk * 5;
    auto issue = [&](int q) {
        unsigned char* dst = ring + (q & (NSLOT - 1)) * 16384;
        while (q >= nload) q -= 5;                // past the end: reload the same-typed piece of the last chunk into a dead slot, so
                                                  // every step sees the piece counts its counted vmcnt assumes
        const int hc = q / 5, t = q - hc * 5;
        if (t < 3) {
            const bf16_t* src = w13 + (int64_t)hc * 128 * BF_BAND + t * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + a_off[i]), (lds_ptr_t)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
        } else {
            const bf16_t* src = w2 + hc * 64 + (t - 3) * 32;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + b_off[i]), (lds_ptr_t)(dst + (wave * 3 + i) * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int q = 0; q < NSLOT - 1; ++q) issue(q);

    f32x16 acc1[3][2], acc2[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    }
    // step q multiplies load q; loads q+1 .. q+6 (already issued) may stay in flight: AHEAD = their DMA pieces per wave
    auto step_begin = [&](int q, auto ahead) -> const unsigned char* {
        wait_vmcnt<decltype(ahead)::value>();
        __builtin_amdgcn_s_waitcnt(0xc07f);       // own LDS writes (SwiGLU chunk) done before the barrier publishes them
        __builtin_amdgcn_s_barrier();             // load q landed everywhere; everyone is done with load q-1's slot
        issue(q + NSLOT - 1);                     // -> slot (q-1) % NSLOT
        return ring + (q & (NSLOT - 1)) * 16384;
    };
    // fragment reads run one k-step ahead of the MFMAs that use them (one wave per SIMD: an LDS round trip in front of every batch of
    // six MFMAs was as long as the batch)
    auto phase_a = [&](int kc, const unsigned char* Bs) {
        bf16x8 bf[2][2];
        auto rd = [&](int ks, int slot) {
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<64>(wc * 64 + j * 32 + frow, ks * 2 + fk));
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) rd(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks & 1][j], ay[i][kc * 4 + ks], acc1[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto phase_b = [&](const unsigned char* Bs, int khalf) {
        bf16x8 af[2][3], bf[2][3];
        auto rd = [&](int ks, int slot) {
#pragma unroll
            for (int i = 0; i < 3; ++i) af[slot][i] = *reinterpret_cast<const bf16x8*>(Hs + lds_off_t<64>(wr * 96 + i * 32 + frow, (khalf * 2 + ks) * 2 + fk));
#pragma unroll
            for (int j = 0; j < 3; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<32>(wc * 96 + j * 32 + frow, ks * 2 + fk));
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks + 1 < 2) rd(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks & 1][j], af[ks & 1][i], acc2[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using std::integral_constant;
#pragma unroll 1
    for (int hc = 0; hc < nchunk; ++hc) {
        const int q0 = hc * 5;
        // piece counts per wave of the six loads behind the consumed one (pattern A4 A4 A4 B3 B3, cyclic)
        const unsigned char* b0 = step_begin(q0 + 0, integral_constant<int, 22>());
        if (hc == 0 && p.ep.trace) tq1 = __builtin_amdgcn_s_memtime();
        phase_a(0, b0);
        phase_a(1, step_begin(q0 + 1, integral_constant<int, 22>()));
        phase_a(2, step_begin(q0 + 2, integral_constant<int, 21>()));
        {
            // SwiGLU on the interleaved (w1, w3) column pairs -> this chunk's hidden values, bf16, as the next A operand
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = wr * 96 + i * 32 + frow;
                        const int h0 = wc * 32 + j * 16 + q * 4 + fk * 2;
                        bf16x2 hv;
                        hv[0] = f2bf(silu_f(acc1[i][j][q * 4 + 0]) * acc1[i][j][q * 4 + 1]);
                        hv[1] = f2bf(silu_f(acc1[i][j][q * 4 + 2]) * acc1[i][j][q * 4 + 3]);
                        *reinterpret_cast<bf16x2*>(Hs + lds_off_t<64>(row, h0 >> 3) + (h0 & 7) * 2) = hv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc1[i][j][q * 4 + r] = 0.f;
                    }
        }
        phase_b(step_begin(q0 + 3, integral_constant<int, 21>()), 0);
        phase_b(step_begin(q0 + 4, integral_constant<int, 22>()), 1);
    }
    wait_vmcnt<0>();                              // the dummy tail loads
    if (p.ep.trace) tq2 = __builtin_amdgcn_s_memtime();
    // gated residual: h[:, band e] += gate * z   (same epilogue as the unfused w2 GEMM; LDS is free now)
    // (round 3: the same epilogue straight from the accumulators in the P16 column layout - no LDS slab, no epilogue barriers - measured
    //  62.9 against 63.5 us: the 74 MB read-modify-write of all 252 workgroups at once is an HBM burst, not an instruction-issue problem)
    staged_epilogue<EPI_RESID_GATE, 3, 3, 2, NTHREADS, HOIST>(p.ep, e, acc2, reinterpret_cast<float*>(bl), row0, rows_end, 0, tid, wr, wc, frow, fk);
    if (p.ep.trace && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* tr = p.ep.trace + (size_t)blockIdx.x * 4;
        tr[0] = tq0; tr[1] = tq1; tr[2] = tq2; tr[3] = __builtin_amdgcn_s_memtime();
    }
}

// ---- band-expert FFN, fused, 96-channel bands (8 experts per group, BASELINE configs[2]) ------------------------------------
// Same idea as band_ffn_kernel for band = 96: at E = 8 the hidden tensor of the band experts is [N][8 x 512] bf16 - 394 MB written by
// the w1/w3 GEMM and read back by the w2 GEMM per block evaluation at 32 clips (306 + 166 us, profiles/r02_final_c3_kernel_stats.csv).
// One workgroup owns 256 tokens x one band; 4 waves, wave w owns rows [64 w, 64 w + 64) in BOTH products (2 row tiles; 4 column tiles
// of the 128-column w1/w3 chunk, 3 column tiles of the 96 outputs), so a wave reads back only the hidden values it wrote itself:
//   * y_e [64 x 96] per wave lives in registers (2 x 6 fragments);
//   * per 64-wide hidden chunk 4 loads through a ring of eight 12-KB slots, seven ahead: three K-slabs of w13 [128 rows x 32] (8 KB,
//     2 DMA pieces per wave) and the w2 slab [96 rows x 64] (12 KB, 3 pieces per wave); counted vmcnt (pattern 2 2 2 3);
//   * acc1 [64 x 128] -> SwiGLU lane-locally -> bf16 hidden chunk [64 x 64] in LDS -> acc2 [64 x 96] += hidden . w2 slab;
//   * gated residual epilogue straight from the MFMA layout (wave_epilogue<EPI_RESID_GATE>): a lane owns one row and 4 consecutive
//     columns, 16-byte loads / stores.
// k runs ascending in both products, as in the grouped GEMMs: bit-identical to the unfused path.
#define B96_BM 256
#define B96_BAND 96
#define B96_SLOT 12288
__global__ void __launch_bounds__(NTHREADS) band_ffn96_kernel(const BandDev p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bl96[];
    constexpr int HCH = B96_BM * 128;             // bytes of the [256 x 64] bf16 hidden chunk
    constexpr int NSLOT = 8;
    unsigned char* Hs = bl96;
    unsigned char* ring = bl96 + HCH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fk = lane >> 5;
    const int L = blockIdx.x;
    const int e = (L & 7) % p.E;
    const int rt = (L >> 3) * (8 / p.E) + (L & 7) / p.E;
    const int row0 = rt * B96_BM;
    if (row0 >= p.M) return;
    const int rows_end = p.M;

    bf16x8 ay[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row = row0 + wave * 64 + i * 32 + frow;
        if (row >= rows_end) row = row0;
        const bf16_t* src = p.Y + (int64_t)row * p.ldy + e * B96_BAND + fk * 8;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) ay[i][kk] = *reinterpret_cast<const bf16x8*>(src + kk * 16);
    }
    const bf16_t* w13 = p.W13 + (int64_t)e * 2 * p.H * B96_BAND;
    const bf16_t* w2 = p.W2 + (int64_t)e * B96_BAND * p.H;
    int a_off[2], b_off[3];          // element offsets of this lane's DMA pieces inside a w13 K-slab / the w2 slab
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * (wave * 2 + i) + (lane >> 2);                  // 16 rows x 64 B per piece
        a_off[i] = r * B96_BAND + (((lane & 3) ^ ((r >> 2) & 3)) << 3);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = 8 * (wave * 3 + i) + (lane >> 3);                   // 8 rows x 128 B per piece
        b_off[i] = r * p.H + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
    }
    const int nchunk = p.H / 64;
    const int nload = nchunk * 4;
    auto issue = [&](int q) {
        unsigned char* dst = ring + (q & (NSLOT - 1)) * B96_SLOT;
        while (q >= nload) q -= 4;                // past the end: same-typed dummy reload into a dead slot (keeps the counted vmcnt pattern)
        const int hc = q >> 2, t = q & 3;
        if (t < 3) {
            const bf16_t* src = w13 + (int64_t)hc * 128 * B96_BAND + t * 32;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + a_off[i]), (lds_ptr_t)(dst + (wave * 2 + i) * 1024), 16, 0, 0);
        } else {
            const bf16_t* src = w2 + hc * 64;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + b_off[i]), (lds_ptr_t)(dst + (wave * 3 + i) * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int q = 0; q < NSLOT - 1; ++q) issue(q);

    f32x16 acc1[2][4], acc2[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    }
    // step q multiplies load q; loads q+1 .. q+6 (already issued) may stay in flight: AHEAD = their DMA pieces per wave
    auto step_begin = [&](int q, auto ahead) -> const unsigned char* {
        wait_vmcnt<decltype(ahead)::value>();
        __builtin_amdgcn_s_waitcnt(0xc07f);       // own LDS traffic done before the barrier
        __builtin_amdgcn_s_barrier();             // load q landed everywhere; everyone is done with load q-1's slot
        issue(q + NSLOT - 1);                     // -> slot (q-1) % NSLOT
        return ring + (q & (NSLOT - 1)) * B96_SLOT;
    };
    auto phase_a = [&](int t, const unsigned char* Bs) {
        bf16x8 bf[2][4];
        auto rd = [&](int ks, int slot) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<32>(j * 32 + frow, ks * 2 + fk));
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks + 1 < 2) rd(ks + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][j], ay[i][t * 2 + ks], acc1[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto phase_b = [&](const unsigned char* Bs) {
        bf16x8 af[2][2], bf[2][3];
        auto rd = [&](int ks, int slot) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[slot][i] = *reinterpret_cast<const bf16x8*>(Hs + lds_off_t<64>(wave * 64 + i * 32 + frow, ks * 2 + fk));
#pragma unroll
            for (int j = 0; j < 3; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off_t<64>(j * 32 + frow, ks * 2 + fk));
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) rd(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks & 1][j], af[ks & 1][i], acc2[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using std::integral_constant;
#pragma unroll 1
    for (int hc = 0; hc < nchunk; ++hc) {
        const int q0 = hc * 4;
        // pieces per wave of the six loads behind the consumed one (pattern A2 A2 A2 B3, cyclic)
        phase_a(0, step_begin(q0 + 0, integral_constant<int, 13>()));
        phase_a(1, step_begin(q0 + 1, integral_constant<int, 14>()));
        phase_a(2, step_begin(q0 + 2, integral_constant<int, 14>()));
        {
            // SwiGLU on the interleaved (w1, w3) column pairs -> this chunk's hidden values (rows of this wave only), bf16
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = wave * 64 + i * 32 + frow;
                        const int h0 = j * 16 + q * 4 + fk * 2;
                        bf16x2 hv;
                        hv[0] = f2bf(silu_f(acc1[i][j][q * 4 + 0]) * acc1[i][j][q * 4 + 1]);
                        hv[1] = f2bf(silu_f(acc1[i][j][q * 4 + 2]) * acc1[i][j][q * 4 + 3]);
                        *reinterpret_cast<bf16x2*>(Hs + lds_off_t<64>(row, h0 >> 3) + (h0 & 7) * 2) = hv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc1[i][j][q * 4 + r] = 0.f;
                    }
        }
        phase_b(step_begin(q0 + 3, integral_constant<int, 13>()));
    }
    wait_vmcnt<0>();                              // the dummy tail loads
    // gated residual: h[:, band e] += gate * z   (same arithmetic as the unfused w2 GEMM's epilogue)
    wave_epilogue<EPI_RESID_GATE, 2, 3>(p.ep, e, acc2, row0 + wave * 64, rows_end, 0, frow, fk);
}

int launch_band_ffn(const BandFfnArgs& a, hipStream_t st) {
    if ((a.band != BF_BAND && a.band != B96_BAND) || a.H % 64 || (8 % a.E) || a.E > 8) VB_FAIL(VB_E_INVALID, "band_ffn: band=%d H=%d E=%d unsupported", a.band, a.H, a.E);
    BandDev d;
    memset(&d, 0, sizeof(d));
    d.Y = a.y; d.ldy = a.ldy; d.W13 = a.w13; d.W2 = a.w2; d.M = a.M; d.H = a.H; d.E = a.E;
    d.ep.out32 = a.out32; d.ep.ldc32 = a.ldc32; d.ep.gate = a.gate; d.ep.gate_ld = a.gate_ld; d.ep.T = a.T > 0 ? a.T : 1;
    d.ep.rT = 1.0f / (float)d.ep.T; d.ep.rhd = 1.f; d.ep.rD = 1.f; d.ep.hd = 1; d.ep.D = 1;
    d.ep.c_noff_group = a.band; d.ep.N = a.band; d.ep.M = a.M; d.ep.trace = g_gemm_trace;
    if (a.M >= (1 << 21)) VB_FAIL(VB_E_INVALID, "band_ffn: M exceeds fdiv()");
    if (a.band == B96_BAND) {
        const int tiles96 = cdiv(a.M, B96_BM);
        const int nblk96 = cdiv(tiles96, 8 / a.E) * 8;
        constexpr size_t lds96 = (size_t)B96_BM * 128 + 8 * B96_SLOT;      // hidden chunk (32 KB) + 8 ring slots of 12 KB = 128 KB
        static OnceFlags attr96;
        vb_set_max_lds_once(attr96, reinterpret_cast<const void*>(band_ffn96_kernel), (int)lds96);
        ProfScope prof96(0, 2.0 * a.M * a.E * ((double)2 * a.H * a.band + (double)a.band * a.H),
                         (double)a.M * a.E * a.band * (2.0 + 8.0) + (double)a.E * 3.0 * a.H * a.band * 2.0, st);
        hipLaunchKernelGGL(band_ffn96_kernel, dim3(nblk96), dim3(NTHREADS), lds96, st, d);
        VB_CHECK_LAUNCH();
        return VB_OK;
    }
    const int row_tiles = cdiv(a.M, BF_BM);
    const int per8 = 8 / a.E;                         // row tiles per group of 8 consecutive blocks
    const int nblk = cdiv(row_tiles, per8) * 8;
    constexpr size_t lds = (size_t)BF_BM * 128 + 8 * 16384;   // hidden chunk (24 KB) + 8 ring slots of 16 KB = 152 KB
    static OnceFlags attr;
    static OnceFlags attr_old;
    const bool hoist = !vb_tune().band_epi_old;
    if (hoist) vb_set_max_lds_once(attr, reinterpret_cast<const void*>(band_ffn_kernel<true>), (int)lds);
    else vb_set_max_lds_once(attr_old, reinterpret_cast<const void*>(band_ffn_kernel<false>), (int)lds);
    ProfScope prof(0, 2.0 * a.M * a.E * ((double)2 * a.H * a.band + (double)a.band * a.H),
                   (double)a.M * a.E * a.band * (2.0 + 8.0) + (double)a.E * 3.0 * a.H * a.band * 2.0, st);
    if (hoist) hipLaunchKernelGGL(band_ffn_kernel<true>, dim3(nblk), dim3(NTHREADS), lds, st, d);
    else hipLaunchKernelGGL(band_ffn_kernel<false>, dim3(nblk), dim3(NTHREADS), lds, st, d);
    VB_CHECK_LAUNCH();
    return VB_OK;
}


int launch_gemm(const GemmArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0) return VB_OK;
    if (a.N % 4 || a.K % 8 || a.lda % 8 || a.ldb % 8) VB_FAIL(VB_E_INVALID, "gemm: N%%4, K%%8, lda%%8, ldb%%8 must be 0 (N=%d K=%d)", a.N, a.K);
    if (a.nseg != 1 && a.nseg != 3) VB_FAIL(VB_E_INVALID, "gemm: nseg must be 1 or 3");
    // "row-range groups": the groups partition the rows (device array of offsets, or uniform groups of group_rows rows each); otherwise
    // ngroups > 1 means the same rows against several operands (grid z)
    const bool row_groups = a.group_off != nullptr || a.group_rows > 0;
    GemmDev d;
    d.A = a.A; d.a_plane = a.a_plane; d.lda = a.lda; d.a_rows = a.a_rows; d.a_koff_group = a.a_koff_group;
    d.B = a.B; d.b_plane = a.b_plane; d.ldb = a.ldb; d.b_group_stride = a.b_group_stride;
    d.M = a.M; d.N = a.N; d.K = a.K; d.nseg = a.nseg; d.ngroups = a.ngroups; d.group_off = a.group_off;
    d.c_noff_group = a.c_noff_group; d.bias = a.bias; d.bias_group_stride = a.bias_group_stride;
    d.out = a.out.p; d.out_plane = a.out.plane; d.out_np = a.out.np; d.ldc = a.ldc;
    d.out32 = a.out32; d.ldc32 = a.ldc32; d.gate = a.gate; d.gate_ld = a.gate_ld; d.T = a.T > 0 ? a.T : 1;
    d.rows_out = a.rows_out; d.row_scale = a.row_scale; d.y32_in = a.y32_in; d.row_scale2 = a.row_scale2; d.scale_split = a.scale_split;
    d.add32 = a.add32; d.dup_rows = a.dup_rows;
    d.conv_ci = a.conv_ci; d.conv_ktap = 0; d.conv_dil = a.conv_dil; d.conv_agrp = a.conv_agrp; d.conv_arow0 = a.conv_arow0; d.conv_btap = a.conv_btap;
    d.res32 = a.res32;
    if (a.conv_ci > 0) {
        if (a.epi != EPI_F32_CT || a.group_rows <= 0 || a.conv_ci % 64 || a.K % a.conv_ci || a.a_rows || a.group_rows != (a.T > 0 ? a.T : 1))
            VB_FAIL(VB_E_INVALID, "gemm: conv mode needs EPI_F32_CT, uniform groups of T rows, Ci %% 64 == 0");
        d.conv_ktap = a.conv_ci / 64;
    }
    d.q = a.q.p; d.q_plane = a.q.plane; d.k = a.k.p; d.k_plane = a.k.plane; d.vt = a.vt.p; d.vt_plane = a.vt.plane;
    d.qkv_np = a.q.np; d.rope_cos = a.rope_cos; d.rope_sin = a.rope_sin; d.H = a.H; d.hd = a.hd > 0 ? a.hd : 1;
    d.Tpad = a.Tpad; d.D = a.D > 0 ? a.D : 1;
    d.rT = 1.0f / (float)d.T; d.rhd = 1.0f / (float)d.hd; d.rD = 1.0f / (float)d.D;
    if (a.M >= (1 << 21) || (int64_t)a.N * (a.ngroups > 0 ? a.ngroups : 1) >= (1 << 21)) VB_FAIL(VB_E_INVALID, "gemm: index ranges exceed fdiv()");
    d.trace = g_gemm_trace; d.abl = vb_tune().gemm_ablate; d.no_vt16 = vb_tune().qkv_vt16_off ? 1 : 0; d.epi_old = vb_tune().band_epi_old ? 1 : 0;
    const double gz_ = (row_groups || a.ngroups <= 1) ? 1.0 : (double)a.ngroups;        // groups that share the row range multiply the work
    const double npl_ = a.nseg == 3 ? 2.0 : 1.0, MN_ = (double)a.M * a.N * gz_;
    double ob_;                                                                         // result (+ read-modify) bytes of the epilogue
    switch (a.epi) {
        case EPI_F32: case EPI_F32_CT: case EPI_SCATTER_F32: ob_ = 4.0 * MN_; break;
        case EPI_RESID_GATE: ob_ = 8.0 * MN_; break;
        case EPI_SWIGLU: case EPI_GEGLU: ob_ = MN_ * a.out.np; break;
        case EPI_SCATTER_ADD_PLANES: ob_ = MN_ * (4.0 + 2.0 * a.out.np); break;
        case EPI_QKV_ROPE: ob_ = 2.0 * MN_ * a.q.np; break;
        default: ob_ = 2.0 * MN_ * a.out.np; break;
    }
    ProfScope prof(a.prof_class, 2.0 * MN_ * a.K,
                   2.0 * npl_ * ((double)a.M * a.K * (a.a_koff_group ? gz_ : 1.0) + (double)a.N * a.K * (a.ngroups > 1 ? a.ngroups : 1)) + ob_, st);
    // tile configuration: 0 = 128x128 (two workgroups per CU), else (TM, TN) of the big-tile kernel (one per CU).  The
    // big tiles are taken when K allows the DMA ring; among them the one that wastes the fewest tile-slots of the last
    // round of 256 CUs and of partial edge tiles wins.  VB_GEMM_TILE=22|33|11|21|23 overrides (the bit-identity
    // test; the 128 x 256 / 256 x 128 forms 24 / 42 spilled - 576-640 B of scratch at 254-256 VGPRs - and were removed in round 4).
    int cfg = 0;
    if (a.K % 64 == 0) {
        const int forced = vb_tune().gemm_tile;
        if (forced >= 0) {
            if (forced != 22 && forced != 33 && forced != 11 && forced != 21 && forced != 23) VB_FAIL(VB_E_INVALID, "gemm: VB_GEMM_TILE=%d is not a tile configuration (22, 33, 11, 21, 23)", forced);
            cfg = forced == 22 ? 0 : forced;
            if (cfg == 23 && !(a.epi == EPI_RESID_GATE && !row_groups && a.ngroups <= 1 && a.N % 192 == 0)) cfg = 0;    // 23 serves the plain gated-residual GEMM
        } else {
            // measured (tools/gemm_tilecfg.py): the 192x192 kernel wins when its tiles fit one round of the 256 CUs
            // (12032 x 768: 252 tiles); with several rounds per CU the 128x128 kernel (two co-resident workgroups
            // overlapping each other's epilogue) is as fast or faster.
            const int gz = row_groups ? 1 : (a.ngroups > 0 ? a.ngroups : 1);
            const int64_t rt = row_groups ? (cdiv(a.M, 192) + a.ngroups) : cdiv(a.M, 192);
            const int64_t t33 = rt * cdiv(a.N, 192) * gz;
            const int64_t t22 = (row_groups ? (cdiv(a.M, BM) + a.ngroups) : (int64_t)cdiv(a.M, BM)) * cdiv(a.N, BN) * gz;
            const int64_t t23 = (int64_t)cdiv(a.M, BM) * (a.N / 192);
            if (a.epi == EPI_RESID_GATE && !row_groups && gz == 1 && a.N % 192 == 0 && t22 > 512 && t23 <= 512 && vb_tune().wide_resid) cfg = 23;
            // (K >= 384 since round 3: at one clip the band experts' K = 192 first product took the 192 x 192 one-per-CU kernel - three
            //  k-iterations under a 24-KB-per-stage ring, 18 us - where 128 x 128 tiles run ~8: one 20 s clip 40.2 -> 38.2 ms)
            else if (t33 <= 256 + 16 && t22 > 320 && !a.rows_out && a.K >= vb_tune().big_tile_min_k) cfg = 33;      // (row-scatter epilogues measured slower on it)
            // small problems (one or two clips): 128x128 tiles leave most CUs idle and a tile's 12 k-iterations are pure DMA latency;
            // 64x64 tiles (three workgroups per CU) make 4x the tiles.  VB_GEMM_SMALL=0 keeps the 128x128 kernel, 21 takes 128x64.
            // (threshold 200 tiles since round 3: at 4 clips x 2 branches - one sub-batch of the two-stream configuration, 282 tiles - the
            //  128 x 128 kernel is ahead when another stream shares the GPU: 1287 vs 1280 mel-s/s over three interleaved runs; per-clip
            //  grouped launches with >= 8 clips keep the 128 x 128 kernel for its XCD-affine tile order)
            else if (vb_tune().gemm_small && t22 < vb_tune().gemm_small_tiles && !(a.group_rows > 0 && a.ngroups >= 8 && a.ngroups % 8 == 0 && !vb_tune().no_xcd_groups))
                cfg = vb_tune().gemm_small;
        }
    }
    // 8-wave 256 x 256 kernel with the P16 epilogues (round 4) on the two wide projections - QKV + RoPE and the routed SwiGLU - whenever the
    // launch makes at least P8_MIN_TILES of its tiles (two clips and up; one clip keeps the small tiles) and the P16 layout applies (the
    // conditions of the 4-wave P16 forms).  Same k order, same epilogue arithmetic: bit-identical to the 4-wave kernels
    // (VB_GEMM_P8_OFF=1 keeps those; test_p8_p16_projections_are_bit_identical), so a clip's bits still do not depend on its batch.
    if ((a.epi == EPI_QKV_ROPE || a.epi == EPI_SWIGLU) && a.K % 64 == 0 && a.N % 16 == 0 && vb_tune().gemm_tile < 0 && !vb_tune().gemm_p8_off &&
        !vb_tune().qkv_p16_off && a.conv_ci == 0 && a.group_rows == 0) {
        const int gz = row_groups ? 1 : (a.ngroups > 0 ? a.ngroups : 1);
        const int64_t t88 = (int64_t)(row_groups ? (cdiv(a.M, 256) + a.ngroups) : cdiv(a.M, 256)) * cdiv(a.N, 256) * gz;
        const bool lay = a.epi == EPI_QKV_ROPE ? (d.hd % 16 == 0 && d.D % 16 == 0 && !a.group_off && a.ngroups <= 1)
                                               : (a.ldc % 8 == 0 && a.c_noff_group % 8 == 0);
        if (lay && t88 >= P8_MIN_TILES) cfg = 89;
#ifdef VB_EXPERIMENTS
        // round 5: the persistent form of the same tile (gemm_bf16_pk_kernel, bit-identical again) - MEASURED NOT FASTER (profiles/r05_gemm_pk_ab.txt,
        // DESIGN 5.0): an experiments-build opt-in (VB_GEMM_PK=1), the product keeps the per-tile launch
        const bool pk_ok = a.epi == EPI_SWIGLU ? (!a.group_off || a.ngroups <= PK_MAX_GROUPS)
                                               : ((d.T & 15) == 0 && (d.Tpad & 7) == 0 && !d.no_vt16 && ((2 * d.D) & 255) == 0 && a.q.np == a.vt.np);
        if (cfg == 89 && vb_tune().gemm_pk > 0 && gz == 1 && pk_ok && (int64_t)a.M * a.lda < (1ll << 31) &&
            (int64_t)(a.ngroups > 0 ? a.ngroups : 1) * a.b_group_stride + (int64_t)a.N * a.ldb < (1ll << 31)) cfg = 90;
#endif
    }
    // the 192 x 192 tile with the RoPE / V-transpose epilogue needs more registers than a wave has (640 B of scratch at 256 VGPRs): QKV takes the
    // 128 x 128 tiles wherever the rules above (or VB_GEMM_TILE) say 33 - M in (2176, 2560] rows - bit-identical like every tile choice
    if (cfg == 33 && a.epi == EPI_QKV_ROPE) cfg = 0;
#ifdef VB_EXPERIMENTS
    if (a.epi == EPI_F32 && vb_tune().gemm_pk_f32 > 0 && a.K % 64 == 0 && !row_groups && a.ngroups <= 1 && a.conv_ci == 0 && !a.add32) {
        launch_pk_f32(d, st, vb_tune().gemm_pk_f32);
        VB_CHECK_LAUNCH();
        return VB_OK;
    }
    // 8-wave 256 x 256 kernel (variant 4): taken when the problem makes enough of its tiles to occupy a good part of the chip - it
    // moves half the bytes per flop through the L2 -> LDS feed, so it wins even at ~55 % of the CUs busy (12032 x 768: 141 tiles);
    // small problems (one 20 s clip: 18 tiles) stay on the 128 x 128 kernel.  VB_GEMM_P8 = 0 off / 1..3 force a ring shape.
    {
        const int p8 = vb_tune().gemm_p8;
        const bool epi_ok = a.epi == EPI_PLANES || a.epi == EPI_F32 || a.epi == EPI_QKV_ROPE || a.epi == EPI_RESID_GATE ||
                            a.epi == EPI_SWIGLU || a.epi == EPI_SCATTER_F32 || a.epi == EPI_SCATTER_ADD_PLANES;
        if (epi_ok && a.K % 32 == 0 && p8 != 0 && vb_tune().gemm_tile < 0) {
            const int gz = row_groups ? 1 : (a.ngroups > 0 ? a.ngroups : 1);
            const int64_t t88 = (int64_t)(row_groups ? (cdiv(a.M, 256) + a.ngroups) : cdiv(a.M, 256)) * cdiv(a.N, 256) * gz;
            // measured (tools/gemm_p8_bench.py, profiles/r02_gemm_p8_microbench.txt): with one workgroup per CU nothing overlaps a
            // tile's epilogue, so the 8-wave kernel only wins where the mainloop outweighs the output traffic - the wide
            // projections (N >= 1024: QKV, routed w1/w3); the N = 768 / 640 launches stay on the two-per-CU 128 x 128 kernel
            // and inside the DiT even those lose (A/B in the pipeline, profiles/r02_p8_pipeline_ab.txt: GEMM class 74 -> 86 ms per pass):
            // QKV's RoPE / V-transpose epilogue and the gathered grouped SwiGLU cost more on a 256-row tile than the mainloop gains.
            // The kernel therefore stays an opt-in (VB_GEMM_P8 >= 1) - kept for the micro-benchmark and as the record of the experiment.
            if (p8 > 0) cfg = 88;
            else if (p8 < 0 && (vb_tune().gemm_p8_mask & (1 << a.epi)) && t88 >= P8_MIN_TILES) cfg = 88;     // per-epilogue opt-in (A/B tool)
        }
    }
#endif
    if (a.conv_ci > 0) {
        // conv-as-GEMM: 128 x 128 tiles, or 64 x 64 (three workgroups per CU) when 128 x 128 would make fewer than 200 workgroups - one or two
        // clips; same k order in both, so a clip's bits do not depend on the batch
        const int64_t t22c = (int64_t)a.ngroups * cdiv(a.group_rows, BM) * cdiv(a.N, BN);
        cfg = (vb_tune().gemm_small == 11 && t22c < vb_tune().gemm_small_tiles) ? 11 : 0;
    }
    if (cfg == 23) {     // 128 x 192, two per CU, gated-residual epilogue (gemm_bf16_wide_resid_kernel)
        if (a.epi != EPI_RESID_GATE || row_groups || a.N % 192 || a.K % 64) VB_FAIL(VB_E_INVALID, "gemm: tile 23 serves the plain gated-residual GEMM only");
        d.n_tiles = a.N / 192;
        hipLaunchKernelGGL(gemm_bf16_wide_resid_kernel<3>, dim3(d.n_tiles * ((cdiv(a.M, BM) + 7) / 8 * 8)), dim3(NTHREADS), 0, st, d);
        VB_CHECK_LAUNCH();
        return VB_OK;
    }
    const int cfgt = cfg == 90 ? 89 : cfg;           // (the persistent kernel walks the 8-wave kernel's 256 x 256 tiles; it sizes its own grid)
    const int bm = cfgt ? 64 * (cfgt / 10 > 4 ? 4 : cfgt / 10) : BM, bn = cfgt ? 64 * (cfgt % 10 > 4 ? 4 : cfgt % 10) : BN;
    int mt = row_groups ? (cdiv(a.M, bm) + a.ngroups) : cdiv(a.M, bm);
    d.n_tiles = cdiv(a.N, bn);
    d.grp_rows = 0; d.grp_tiles = 0; d.grp_xcd = 0;
    if (a.group_rows > 0 && !a.group_off && !(!cfg || (cfg == 11 && a.conv_ci > 0)))
        VB_FAIL(VB_E_INVALID, "gemm: uniform groups without an offset array run on the 128 x 128 kernel (or 64 x 64 in conv mode) only");
    if (a.conv_ci > 0 && ((cfg != 0 && cfg != 11) || a.K % 64)) VB_FAIL(VB_E_INVALID, "gemm: conv mode runs on the 128 x 128 / 64 x 64 DMA kernels (K %% 64 == 0)");
    if ((!cfg || (cfg == 11 && a.conv_ci > 0)) && a.group_rows > 0 && a.K % 32 == 0) {      // uniform groups: 128 x 128 kernel; 64 x 64 in conv mode
        d.grp_rows = a.group_rows; d.grp_tiles = cdiv(a.group_rows, bm);
        d.grp_xcd = (a.conv_ci == 0 && a.ngroups % 8 == 0 && !vb_tune().no_xcd_groups) ? 1 : 0;
        mt = d.grp_xcd ? a.ngroups * d.grp_tiles : (a.ngroups * d.grp_tiles + 7) / 8 * 8;
    }
    //  uniform groups described only by group_rows must have been taken up above: on any other route (K % 32 != 0 -> the
    // register-staged kernel) every row would silently run against group 0's operand
    if (a.group_rows > 0 && !a.group_off && d.grp_rows == 0) VB_FAIL(VB_E_INVALID, "gemm: uniform row groups need K %% 32 == 0 (K=%d)", a.K);
    dim3 grid(d.n_tiles * ((mt + 7) / 8 * 8), 1, row_groups ? 1 : (a.ngroups > 0 ? a.ngroups : 1));
    d.ncc = 0; d.rpx = (mt + 7) / 8;
    if (!cfg && !row_groups) {       // VB_GEMM_NCHUNK=c (tuning, default off until measured in the pipeline): column chunking for wide N
        const int c = vb_tune().gemm_nchunk;
        if (c > 0 && d.n_tiles > c && d.n_tiles % c == 0) d.ncc = c;
    }
#ifdef VB_EXPERIMENTS
#define VB_P8_LAUNCH(E) if constexpr (P8Epi<E>::ok) launch_p8<E>(d, grid, st, vb_tune().gemm_p8);
#define VB_PK_LAUNCH(E) if constexpr (E == EPI_QKV_ROPE || E == EPI_SWIGLU) launch_pk<E>(d, st);
#else
#define VB_P8_LAUNCH(E)
#define VB_PK_LAUNCH(E)
#endif
#define VB_GEMM_CASE(E) \
        case E: \
            if (cfg == 88) { VB_P8_LAUNCH(E) } \
            else if (cfg == 90) { VB_PK_LAUNCH(E) } \
            else if (cfg == 89) launch_p8_product<E>(d, grid, st); \
            else if (cfg == 33) { if constexpr (E != EPI_QKV_ROPE) launch_big<E, 3, 3, 3>(d, grid, st); } \
            else if (cfg == 11) launch_big<E, 1, 1, 3>(d, grid, st); \
            else if (cfg == 21) launch_big<E, 2, 1, 3>(d, grid, st); \
            else launch_t<E>(d, grid, st); \
            break;
    switch (a.epi) {
        VB_GEMM_CASE(EPI_PLANES)
        VB_GEMM_CASE(EPI_F32)
        VB_GEMM_CASE(EPI_QKV_ROPE)
        VB_GEMM_CASE(EPI_RESID_GATE)
        VB_GEMM_CASE(EPI_SWIGLU)
        VB_GEMM_CASE(EPI_SCATTER_F32)
        VB_GEMM_CASE(EPI_SCATTER_ADD_PLANES)
        VB_GEMM_CASE(EPI_GELU_PLANES)
        VB_GEMM_CASE(EPI_HEADS_T)
        VB_GEMM_CASE(EPI_F32_CT)
        VB_GEMM_CASE(EPI_GEGLU)
        default: VB_FAIL(VB_E_INVALID, "gemm: bad epilogue %d", a.epi);
    }
#undef VB_GEMM_CASE
#undef VB_P8_LAUNCH
#undef VB_PK_LAUNCH
    VB_CHECK_LAUNCH();
    return VB_OK;
}
