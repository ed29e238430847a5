// fp32 "few rows x wide weight" linears (gfx950): the per-item conditioning GEMVs (t-embedder MLP, the adaLN
// modulations of ALL blocks + final layer in one launch, high-level gate logits) and the DiT FinalLayer.
//
// Layout of the work: a block owns 16 input rows (staged once in LDS, activation / LayerNorm+modulate applied
// while staging) and a slab of outputs.  Inside a wave, lane = (row r = lane & 15, output slot o = lane >> 4):
// each lane runs a private K-long dot product for ONE (row, output) pair - the 16 lanes of an output share the
// weight address (one broadcast 16-B fetch), the 16 rows sit in distinct LDS banks (row pitch K+4 floats) - so
// there is no cross-lane reduction at all (the v1 kernel spent its time in 96 shuffles per output).
#include <type_traits>

#include "kernels.h"

#define RL_R 16

template <int MODE>   // 0: out[r][n] = b[n] + W[n] . act(x[r] (+ x2[r % mod]))      1: FinalLayer, out[(b*C+n)*T + t]
__global__ void __launch_bounds__(256) rowlin_kernel(const float* __restrict__ x, int x_ld, const int64_t* x_row_idx,
                                                    const float* __restrict__ x2, int x2_ld, int x2_mod,
                                                    const float* shift, const float* scale, int mod_ld, int T, float eps,
                                                    const float* __restrict__ W, const float* __restrict__ bias, int R, int N, int K,
                                                    int act_in, int n_per_block, float* out, int out_ld, int sin_rows) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [RL_R][K + 4]
    const int KP = K + 4;
    const int r0 = blockIdx.y * RL_R;
    const int nr = min(RL_R, R - r0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (MODE == 0) {
        for (int id = threadIdx.x; id < RL_R * K; id += 256) {
            int r = id / K, k = id - r * K;
            float v = 0.f;
            if (r < nr) {
                int64_t xr = x_row_idx ? x_row_idx[r0 + r] : (int64_t)(r0 + r);
                if (sin_rows > 0 && (xr < 0 || xr >= sin_rows)) {
                    // x is the tabulated timestep sinusoid ([cos | sin] of t * exp(-ln(1e4) i / half), flag_large_dit_moe.py:110-128):
                    // an index outside the table (stochastic_encode-style callers pass t up to num_timesteps) is computed, never read
                    const int half = K >> 1, i = k < half ? k : k - half;
                    const float a = (float)xr * expf(-9.210340371976184f * (float)i / (float)half);
                    v = k < half ? cosf(a) : sinf(a);
                } else {
                    v = x[xr * x_ld + k];
                }
                if (x2) v += x2[(int64_t)((r0 + r) % x2_mod) * x2_ld + k];
                if (act_in == 1) v = v / (1.f + expf(-v));
            }
            xs[r * KP + k] = v;
        }
    } else {
        // LayerNorm (no affine) + modulate, one wave per row (4 rows per wave)
        for (int r = wave; r < RL_R; r += 4) {
            if (r < nr) {
                const float* xr = x + (int64_t)(r0 + r) * x_ld;
                float s = 0.f;
                for (int k = lane; k < K; k += 64) s += xr[k];
                const float mean = wave_sum(s) / (float)K;
                float vs = 0.f;
                for (int k = lane; k < K; k += 64) { float d = xr[k] - mean; vs += d * d; }
                const float rs = rsqrtf(wave_sum(vs) / (float)K + eps);
                const int b = (r0 + r) / T;
                for (int k = lane; k < K; k += 64) {
                    float y = (xr[k] - mean) * rs;
                    xs[r * KP + k] = y * (1.f + scale[(int64_t)b * mod_ld + k]) + shift[(int64_t)b * mod_ld + k];
                }
            } else {
                for (int k = lane; k < K; k += 64) xs[r * KP + k] = 0.f;
            }
        }
    }
    __syncthreads();
    const int r = lane & 15, o = lane >> 4;
    const float* xrow = xs + r * KP;
    const int n_begin = blockIdx.x * n_per_block;
    const int n_end = min(N, n_begin + n_per_block);
    for (int nb = n_begin + wave * 4; nb < n_end; nb += 16) {
        const int n = nb + o;
        const bool ok = n < n_end;
        const float* wrow = W + (int64_t)(ok ? n : n_begin) * K;
        float a0 = 0.f, a1 = 0.f;
        for (int k = 0; k < K; k += 8) {
            const float4 w0 = *reinterpret_cast<const float4*>(wrow + k);
            const float4 w1 = *reinterpret_cast<const float4*>(wrow + k + 4);
            const float4 x0 = *reinterpret_cast<const float4*>(xrow + k);
            const float4 x1 = *reinterpret_cast<const float4*>(xrow + k + 4);
            a0 += w0.x * x0.x + w0.y * x0.y + w0.z * x0.z + w0.w * x0.w;
            a1 += w1.x * x1.x + w1.y * x1.y + w1.z * x1.z + w1.w * x1.w;
        }
        if (ok && r < nr) {
            const float v = a0 + a1 + (bias ? bias[n] : 0.f);
            if constexpr (MODE == 0) {
                out[(int64_t)(r0 + r) * out_ld + n] = v;
            } else {
                const int row = r0 + r, b = row / T, t = row - b * T;
                out[((int64_t)b * N + n) * T + t] = v;
            }
        }
    }
}

template <int MODE>
static int launch_rowlin(const float* x, int x_ld, const int64_t* idx, const float* x2, int x2_ld, int x2_mod, const float* shift,
                         const float* scale, int mod_ld, int T, float eps, const float* W, const float* bias, int R, int N, int K,
                         int act_in, float* out, int out_ld, hipStream_t st, int sin_rows = 0) {
    if (K % 8 || K > 4096) VB_FAIL(VB_E_INVALID, "rowlin: K=%d must be %%8 and <= 4096", K);
    // slab of outputs per block: enough blocks to fill the chip, at least 16 outputs (one pass of the 4 waves)
    const int row_groups = cdiv(R, RL_R);
    int npb = 64;
    while (npb > 16 && (int64_t)cdiv(N, npb) * row_groups < 512) npb >>= 1;
    dim3 grid(cdiv(N, npb), row_groups);
    size_t sh = (size_t)RL_R * (K + 4) * sizeof(float);
    static OnceFlags attr_set[2];
    vb_set_max_lds_once(attr_set[MODE], reinterpret_cast<const void*>(rowlin_kernel<MODE>), 160 * 1024 - 4096);
    hipLaunchKernelGGL(rowlin_kernel<MODE>, grid, dim3(256), sh, st, x, x_ld, idx, x2, x2_ld, x2_mod > 0 ? x2_mod : 1, shift, scale, mod_ld,
                       T > 0 ? T : 1, eps, W, bias, R, N, K, act_in, npb, out, out_ld, sin_rows);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

int launch_gemv_rows_idx(const float* x, int x_ld, const int64_t* idx, const float* x2, int x2_ld, int x2_mod, const float* W,
                         const float* bias, int R, int N, int K, int act_in, float* out, int out_ld, hipStream_t st, int sin_rows) {
    return launch_rowlin<0>(x, x_ld, idx, x2, x2_ld, x2_mod, nullptr, nullptr, 0, 1, 0.f, W, bias, R, N, K, act_in, out, out_ld, st, sin_rows);
}
int launch_gemv_rows(const float* x, int x_ld, const float* x2, int x2_ld, int x2_mod, const float* W, const float* bias, int R,
                     int N, int K, int act_in, float* out, int out_ld, hipStream_t st) {
    return launch_rowlin<0>(x, x_ld, nullptr, x2, x2_ld, x2_mod, nullptr, nullptr, 0, 1, 0.f, W, bias, R, N, K, act_in, out, out_ld, st);
}
// FinalLayer, dedicated kernel (round 3): LN(no affine) -> modulate -> Linear(D -> C <= 64) -> out[b][c][t], one WAVE per token row.
// The projection is 768 x 20: as an MFMA GEMM it fills 16 % of a 128-wide tile and needed its input as split planes first (LN + modulate
// kernel 16 us + GEMM 19 us per evaluation); here a row's 12 values per lane stay in registers from the LayerNorm on, the weight matrix
// sits in LDS (61 KB, read as conflict-free float4), and every output is one wave reduction - exact fp32 arithmetic, memory-bound on
// reading h once.
// wave-wide sum WITHOUT the LDS crossbar: row-level butterflies and broadcasts as DPP modifiers of VALU adds (quad_perm, row_ror 4 / 8,
// row_bcast 15 / 31), total in lane 63, handed to every lane through an SGPR.  (__shfl_xor is ds_bpermute on gfx9: 6 LDS-pipe
// instructions per sum - 20 sums per token row made the first version of the kernel below LDS-issue-bound, 52 us at 12 032 rows.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto mv = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += mv(v, std::integral_constant<int, 0xb1>());     // quad_perm [1,0,3,2]
    v += mv(v, std::integral_constant<int, 0x4e>());     // quad_perm [2,3,0,1]
    v += mv(v, std::integral_constant<int, 0x124>());    // row_ror 4
    v += mv(v, std::integral_constant<int, 0x128>());    // row_ror 8
    v += mv(v, std::integral_constant<int, 0x142>());    // row_bcast 15
    v += mv(v, std::integral_constant<int, 0x143>());    // row_bcast 31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// EUL (round 5): the CFG combination and the Euler update in the same launch (cfm1_audio.py:154-160: v = v_u + s (v_c - v_u); x += dt v) -
// a wave takes TWO tokens of the conditional half and the same two of the unconditional half (rows m and m + rows / 2), so it holds
// both branches' velocities of its tokens and updates x [B][C][T] in place with the arithmetic of euler_cfg_kernel (two fused
// multiply-adds); the velocity tensor is never written.  Block 0 also advances the sampler's device-side step counter and the timestep
// indices for the NEXT step (step_advance_kernel's work): nothing in this launch reads them.  Saves two launches per Euler step.
struct FinalEuler { float* x; float cfg_scale; const float* dt_table; int k; int* step; int64_t* t_idx_cur; const int64_t* t_table; int n_steps, Beff; };
template <int NQ, bool EUL = false>      // D = 256 * NQ
__global__ void __launch_bounds__(256) final_layer_kernel(const float* __restrict__ h, const float* __restrict__ shift,
                                                         const float* __restrict__ scale, int mod_ld, const float* __restrict__ W,
                                                         const float* __restrict__ bias, int rows, int T, int C, float eps, float* out,
                                                         const FinalEuler fe) {
    constexpr int D = 256 * NQ;
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [C][D]
    for (int id = threadIdx.x * 4; id < C * D; id += 1024) *reinterpret_cast<float4*>(wl + id) = *reinterpret_cast<const float4*>(W + id);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a wave walks FOUR rows at a time: every weight quad read from LDS serves all of them (the LDS reads, not the 240 FMAs per row, set
    // the pace) and their reduction chains interleave
    constexpr int RW = 4;
    const int half = rows >> 1;
    // row r of the group starting at row0: consecutive rows, or (EUL) tokens row0, row0 + 1 of the conditional half and of the unconditional half
    auto row_of = [&](int row0, int r) { return EUL ? min(row0 + (r & 1), half - 1) + (r >> 1) * half : min(row0 + r, rows - 1); };
    const int row_end = EUL ? half : rows, row_step = EUL ? 2 : RW;
    for (int row0 = (blockIdx.x * 4 + wave) * row_step; row0 < row_end; row0 += gridDim.x * 4 * row_step) {
        float4 x[RW][NQ];
        int bb[RW], tt[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int row = row_of(row0, r);
            const float* xr = h + (int64_t)row * D;
#pragma unroll
            for (int i = 0; i < NQ; ++i) x[r][i] = *reinterpret_cast<const float4*>(xr + lane * 4 + 256 * i);
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int row = row_of(row0, r);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NQ; ++i) s += (x[r][i].x + x[r][i].y) + (x[r][i].z + x[r][i].w);
            const float mean = wave_sum_dpp(s) / (float)D;
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                x[r][i].x -= mean; x[r][i].y -= mean; x[r][i].z -= mean; x[r][i].w -= mean;
                vs += (x[r][i].x * x[r][i].x + x[r][i].y * x[r][i].y) + (x[r][i].z * x[r][i].z + x[r][i].w * x[r][i].w);
            }
            const float rs = rsqrtf(wave_sum_dpp(vs) / (float)D + eps);
            bb[r] = row / T; tt[r] = row - bb[r] * T;
            const float* sc = scale + (int64_t)bb[r] * mod_ld; const float* sh = shift + (int64_t)bb[r] * mod_ld;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const float4 a = *reinterpret_cast<const float4*>(sc + lane * 4 + 256 * i), c = *reinterpret_cast<const float4*>(sh + lane * 4 + 256 * i);
                x[r][i].x = x[r][i].x * rs * (1.f + a.x) + c.x; x[r][i].y = x[r][i].y * rs * (1.f + a.y) + c.y;
                x[r][i].z = x[r][i].z * rs * (1.f + a.z) + c.z; x[r][i].w = x[r][i].w * rs * (1.f + a.w) + c.w;
            }
        }
        float res[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) res[r] = 0.f;
        for (int n = 0; n < C; ++n) {
            const float* wr = wl + n * D + lane * 4;
            float acc[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = 0.f;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const float4 w = *reinterpret_cast<const float4*>(wr + 256 * i);
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r] += (x[r][i].x * w.x + x[r][i].y * w.y) + (x[r][i].z * w.z + x[r][i].w * w.w);
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const float t = wave_sum_dpp(acc[r]);
                if (lane == n) res[r] = t;
            }
        }
        if (lane < C) {
            const float bv = bias ? bias[lane] : 0.f;
            if constexpr (EUL) {
                const float dt = fe.dt_table[fe.k];
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (row0 + j < half) {
                        const float vc = res[j] + bv, vu = res[2 + j] + bv;
                        const float e = fmaf(fe.cfg_scale, vc - vu, vu);
                        float* xp = fe.x + ((int64_t)bb[j] * C + lane) * T + tt[j];
                        *xp = fmaf(dt, e, *xp);
                    }
            } else {
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    if (row0 + r < rows) out[((int64_t)bb[r] * C + lane) * T + tt[r]] = res[r] + bv;
            }
        }
    }
    if constexpr (EUL) {
        if (blockIdx.x == 0 && fe.step) {
            const int kn = fe.k + 1;
            if (threadIdx.x == 0) *fe.step = kn;
            const int ki = kn < fe.n_steps ? kn : fe.n_steps - 1;
            for (int i = threadIdx.x; i < fe.Beff; i += 256) fe.t_idx_cur[i] = fe.t_table[ki];
        }
    }
}
bool final_layer_fused_ok(int D, int C) { return (D == 256 || D == 512 || D == 768 || D == 1024) && C <= 64 && (size_t)C * D * 4 <= 96 * 1024; }
int launch_final_layer_fused(const float* h, const float* shift, const float* scale, int mod_ld, const float* W, const float* bias,
                             int rows, int D, int T, int C, float eps, float* out, hipStream_t st) {
    if (!final_layer_fused_ok(D, C)) VB_FAIL(VB_E_INVALID, "final_layer_fused: D=%d C=%d unsupported", D, C);
    const size_t sh = (size_t)C * D * sizeof(float);
    const int grid = min(cdiv(rows, 16), 512);
    static OnceFlags attr[4];
    auto go = [&](auto nq) {
        constexpr int NQ = decltype(nq)::value;
        vb_set_max_lds_once(attr[NQ - 1], reinterpret_cast<const void*>(final_layer_kernel<NQ>), 96 * 1024);
        hipLaunchKernelGGL(final_layer_kernel<NQ>, dim3(grid), dim3(256), sh, st, h, shift, scale, mod_ld, W, bias, rows, T > 0 ? T : 1, C, eps, out, FinalEuler{});
    };
    if (D == 256) go(std::integral_constant<int, 1>()); else if (D == 512) go(std::integral_constant<int, 2>());
    else if (D == 768) go(std::integral_constant<int, 3>()); else go(std::integral_constant<int, 4>());
    VB_CHECK_LAUNCH();
    return VB_OK;
}
// FinalLayer + CFG + Euler update + step advance in one launch (see FinalEuler): rows = 2 x (B T) token rows, conditional half first
int launch_final_layer_euler(const float* h, const float* shift, const float* scale, int mod_ld, const float* W, const float* bias,
                             int rows, int D, int T, int C, float eps, float* x, float cfg_scale, const float* dt_table, int k, int* step,
                             int64_t* t_idx_cur, const int64_t* t_table, int n_steps, int Beff, hipStream_t st) {
    if (!final_layer_fused_ok(D, C) || (rows & 1) || !x || !dt_table) VB_FAIL(VB_E_INVALID, "final_layer_euler: D=%d C=%d rows=%d unsupported", D, C, rows);
    const size_t sh = (size_t)C * D * sizeof(float);
    const int grid = min(cdiv(rows / 2, 8), 512);
    FinalEuler fe{x, cfg_scale, dt_table, k, step, t_idx_cur, t_table, n_steps, Beff};
    static OnceFlags attr[4];
    auto go = [&](auto nq) {
        constexpr int NQ = decltype(nq)::value;
        vb_set_max_lds_once(attr[NQ - 1], reinterpret_cast<const void*>(final_layer_kernel<NQ, true>), 96 * 1024);
        hipLaunchKernelGGL((final_layer_kernel<NQ, true>), dim3(grid), dim3(256), sh, st, h, shift, scale, mod_ld, W, bias, rows, T > 0 ? T : 1, C, eps,
                           (float*)nullptr, fe);
    };
    if (D == 256) go(std::integral_constant<int, 1>()); else if (D == 512) go(std::integral_constant<int, 2>());
    else if (D == 768) go(std::integral_constant<int, 3>()); else go(std::integral_constant<int, 4>());
    VB_CHECK_LAUNCH();
    return VB_OK;
}
// FinalLayer (vocal2music_moe.py:287-291): LN(no affine, eps) -> modulate -> Linear(D->C) -> out[b][c][t]
int launch_final_layer(const float* h, const float* shift, const float* scale, int mod_ld, const float* W, const float* bias,
                       int rows, int D, int T, int C, float eps, float* out, hipStream_t st) {
    return launch_rowlin<1>(h, D, nullptr, nullptr, 0, 1, shift, scale, mod_ld, T, eps, W, bias, rows, C, D, 0, out, 0, st);
}
