// Memory-bound and tiny kernels of the path (gfx950): norms, casts, GEMVs for the
// per-item conditioning vectors, the Band-MoE router + bucketing, Euler/CFG update,
// stem helpers, GroupNorm statistics and the VAE attention softmax.
// All are wave64-shaped: one wave per row with float4 lanes where rows are 768 wide.
#include "kernels.h"
#include "router_dev.h"


// ---------------------------------------------------------------------------
// RMSNorm * w, then adaLN modulate, written as bf16 planes (feeds the MFMA GEMMs)
//   RMSNorm  flag_large_dit_moe.py:52-77 ; modulate :80-81
// ---------------------------------------------------------------------------
// (round 3: the row stays in registers between the two passes - the second pass re-read it from L2 - and a wave walks TWO rows with all
//  their loads issued up front; same arithmetic in the same order, bit-identical)
template <int NQ>     // D = 256 * NQ (NQ float4 per lane and row); NQ = 0: generic D % 4 == 0 (re-reads the row)
__global__ void __launch_bounds__(256) rmsnorm_mod_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                         const float* shift, const float* scale, int mod_ld, int rows, int D,
                                                         int T, float eps, Planes out) {
#pragma clang fp contract(off)      // both forms must round alike: no implicit fma
    const int lane = threadIdx.x & 63;
    if constexpr (NQ == 0) {
        const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (row >= rows) return;
        const float* x = h + (int64_t)row * D;
        float ss = 0.f;
        for (int k = lane * 4; k < D; k += 256) {
            float4 v = *reinterpret_cast<const float4*>(x + k);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        ss = wave_sum(ss);
        const float r = rsqrtf(ss / (float)D + eps);
        const int b = row / T;
        for (int k = lane * 4; k < D; k += 256) {
            float4 v = *reinterpret_cast<const float4*>(x + k);
            float4 ww = *reinterpret_cast<const float4*>(w + k);
            float o[4] = {v.x * r * ww.x, v.y * r * ww.y, v.z * r * ww.z, v.w * r * ww.w};
            if (scale) {
                float4 sc = *reinterpret_cast<const float4*>(scale + (int64_t)b * mod_ld + k);
                float4 sh = *reinterpret_cast<const float4*>(shift + (int64_t)b * mod_ld + k);
                o[0] = o[0] * (1.f + sc.x) + sh.x; o[1] = o[1] * (1.f + sc.y) + sh.y;
                o[2] = o[2] * (1.f + sc.z) + sh.z; o[3] = o[3] * (1.f + sc.w) + sh.w;
            }
            bf16x4 hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) hi[i] = f2bf(o[i]);
            *reinterpret_cast<bf16x4*>(out.p + (int64_t)row * D + k) = hi;
            if (out.np == 2) {
                bf16x4 lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) lo[i] = f2bf(o[i] - bf2f(hi[i]));
                *reinterpret_cast<bf16x4*>(out.p + out.plane + (int64_t)row * D + k) = lo;
            }
        }
    } else {
        const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
        if (row0 >= rows) return;
        float4 v[2][NQ];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const float* x = h + (int64_t)min(row0 + rr, rows - 1) * D;
#pragma unroll
            for (int i = 0; i < NQ; ++i) v[rr][i] = *reinterpret_cast<const float4*>(x + lane * 4 + 256 * i);
        }
        float4 ww[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) ww[i] = *reinterpret_cast<const float4*>(w + lane * 4 + 256 * i);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = row0 + rr;
            if (row >= rows) break;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NQ; ++i) ss += v[rr][i].x * v[rr][i].x + v[rr][i].y * v[rr][i].y + v[rr][i].z * v[rr][i].z + v[rr][i].w * v[rr][i].w;
            ss = wave_sum(ss);
            const float r = rsqrtf(ss / (float)D + eps);
            const int b = row / T;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int k = lane * 4 + 256 * i;
                float o[4] = {v[rr][i].x * r * ww[i].x, v[rr][i].y * r * ww[i].y, v[rr][i].z * r * ww[i].z, v[rr][i].w * r * ww[i].w};
                if (scale) {
                    float4 sc = *reinterpret_cast<const float4*>(scale + (int64_t)b * mod_ld + k);
                    float4 sh = *reinterpret_cast<const float4*>(shift + (int64_t)b * mod_ld + k);
                    o[0] = o[0] * (1.f + sc.x) + sh.x; o[1] = o[1] * (1.f + sc.y) + sh.y;
                    o[2] = o[2] * (1.f + sc.z) + sh.z; o[3] = o[3] * (1.f + sc.w) + sh.w;
                }
                bf16x4 hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) hi[e] = f2bf(o[e]);
                *reinterpret_cast<bf16x4*>(out.p + (int64_t)row * D + k) = hi;
                if (out.np == 2) {
                    bf16x4 lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) lo[e] = f2bf(o[e] - bf2f(hi[e]));
                    *reinterpret_cast<bf16x4*>(out.p + out.plane + (int64_t)row * D + k) = lo;
                }
            }
        }
    }
}
int launch_rmsnorm_mod(const float* h, const float* w, const float* shift, const float* scale, int mod_ld, int rows, int D,
                       int T, float eps, Planes out, hipStream_t st) {
    if (D % 4) VB_FAIL(VB_E_INVALID, "rmsnorm: D%%4");
    const int Tq = T > 0 ? T : 1;
    if (D == 768 && !vb_tune().rmsnorm_generic) hipLaunchKernelGGL(rmsnorm_mod_kernel<3>, dim3(cdiv(rows, 8)), dim3(256), 0, st, h, w, shift, scale, mod_ld, rows, D, Tq, eps, out);
    else if (D == 1024 && !vb_tune().rmsnorm_generic) hipLaunchKernelGGL(rmsnorm_mod_kernel<4>, dim3(cdiv(rows, 8)), dim3(256), 0, st, h, w, shift, scale, mod_ld, rows, D, Tq, eps, out);
    else hipLaunchKernelGGL(rmsnorm_mod_kernel<0>, dim3(cdiv(rows, 4)), dim3(256), 0, st, h, w, shift, scale, mod_ld, rows, D, Tq, eps, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// ---------------------------------------------------------------------------
// LayerNorm (optional affine) -> fp32 and/or planes
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* w, const float* bvec, int rows,
                                                       int D, float eps, float* out32, Planes outp) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += xr[k];
    const float mean = wave_sum(s) / (float)D;
    float vs = 0.f;
    for (int k = lane; k < D; k += 64) { float d = xr[k] - mean; vs += d * d; }
    const float r = rsqrtf(wave_sum(vs) / (float)D + eps);
    for (int k = lane; k < D; k += 64) {
        float o = (xr[k] - mean) * r;
        if (w) o = o * w[k] + bvec[k];
        if (out32) out32[(int64_t)row * D + k] = o;
        if (outp.p) {
            bf16_t hi = f2bf(o);
            outp.p[(int64_t)row * D + k] = hi;
            if (outp.np == 2) outp.p[outp.plane + (int64_t)row * D + k] = f2bf(o - bf2f(hi));
        }
    }
}
int launch_layernorm(const float* x, const float* w, const float* b, int rows, int D, float eps, float* out32, Planes outp,
                     hipStream_t st) {
    hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, w, b, rows, D, eps, out32, outp);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// ---------------------------------------------------------------------------
// casts
// ---------------------------------------------------------------------------
__global__ void cast_planes_kernel(const float* __restrict__ x, int64_t n, Planes out) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            float4 v = *reinterpret_cast<const float4*>(x + i);
            float o[4] = {v.x, v.y, v.z, v.w};
            bf16x4 hi, lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) { hi[k] = f2bf(o[k]); lo[k] = f2bf(o[k] - bf2f(hi[k])); }
            *reinterpret_cast<bf16x4*>(out.p + i) = hi;
            if (out.np == 2) *reinterpret_cast<bf16x4*>(out.p + out.plane + i) = lo;
        } else {
            for (int64_t j = i; j < n; ++j) {
                bf16_t hi = f2bf(x[j]);
                out.p[j] = hi;
                if (out.np == 2) out.p[out.plane + j] = f2bf(x[j] - bf2f(hi));
            }
        }
    }
}
int launch_cast_planes(const float* x, int64_t n, Planes out, hipStream_t st) {
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_planes_kernel, dim3(blocks), dim3(256), 0, st, x, n, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
__global__ void split_rows_kernel(const float* __restrict__ x, int64_t rows, int cols, int cpad, bf16_t* out, int64_t plane) {
    const int q = cpad >> 2;
    const int64_t total = rows * q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / q;
        const int c = (int)(i - r * q) * 4;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (c + k < cols) ? x[r * cols + c + k] : 0.f;
        bf16x4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) { hi[k] = f2bf(v[k]); lo[k] = f2bf(v[k] - bf2f(hi[k])); }
        *reinterpret_cast<bf16x4*>(out + r * cpad + c) = hi;
        *reinterpret_cast<bf16x4*>(out + plane + r * cpad + c) = lo;
    }
}
int launch_split_rows(const float* x, int64_t rows, int cols, int cpad, bf16_t* out, int64_t plane, hipStream_t st) {
    if (cpad % 4 || cpad < cols) VB_FAIL(VB_E_INVALID, "split_rows: cpad=%d cols=%d", cpad, cols);
    int64_t blocks = (rows * (cpad / 4) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(split_rows_kernel, dim3((int)blocks), dim3(256), 0, st, x, rows, cols, cpad, out, plane);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
__global__ void planes_to_f32_kernel(Planes in, int64_t n, float* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = bf2f(in.p[i]);
        if (in.np == 2) v += bf2f(in.p[in.plane + i]);
        out[i] = v;
    }
}
int launch_planes_to_f32(Planes in, int64_t n, float* out, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3(blocks), dim3(256), 0, st, in, n, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
__global__ void fill_f32_kernel(float* p, int64_t n, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
int launch_fill_f32(float* p, int64_t n, float v, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, st, p, n, v);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// mean over L rows: [B][L][D] -> [B][D]   (pooled caption, vocal2music_moe.py:410-412)
__global__ void mean_rows_kernel(const float* __restrict__ x, int B, int L, int D, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    int b = i / D, d = i - b * D;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += x[((int64_t)b * L + l) * D + d];
    out[i] = s / (float)L;
}
int launch_mean_rows(const float* x, int B, int L, int D, float* out, hipStream_t st) {
    hipLaunchKernelGGL(mean_rows_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, st, x, B, L, D, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// ---------------------------------------------------------------------------
// stem helpers (vocal2music_moe.py:388-393)
// ---------------------------------------------------------------------------
// out[b][d][t] = table[idx[b][t]][d]
__global__ void embed_t_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table, int B, int T, int D, int vocab, float* out) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, t0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 64 x 4
    for (int i = ty; i < 64; i += 4) {
        int t = t0 + i;
        if (t < T && d0 + tx < D) {
            //  the host range-checks index tensors once per identity; the read itself is clamped so that a buffer refilled behind
            // that cache (raw pointer writes, DLPack aliases) can never index outside the table
            int64_t ix = idx[(int64_t)b * T + t];
            ix = ix < 0 ? 0 : (ix >= vocab ? vocab - 1 : ix);
            tile[i][tx] = table[ix * D + d0 + tx];
        }
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int d = d0 + i, t = t0 + tx;
        if (d < D && t < T) out[((int64_t)b * D + d) * T + t] = tile[tx][i];
    }
}
int launch_embed_t(const int64_t* idx, const float* table, int B, int T, int D, int vocab, float* out, hipStream_t st) {
    hipLaunchKernelGGL(embed_t_kernel, dim3(cdiv(T, 64), cdiv(D, 64), B), dim3(256), 0, st, idx, table, B, T, D, vocab, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
// out[b][c][t] = (a[2t]+a[2t+1])/2 + (b[2t]+b[2t+1])/2    (AvgPool1d(2) of both branches, then sum)
__global__ void pool_add_kernel(const float* __restrict__ a, const float* __restrict__ bb, int64_t rows, int T_in, float* out) {
    const int To = T_in / 2;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * To) return;
    int64_t r = i / To; int t = (int)(i - r * To);
    const float* pa = a + r * T_in + 2 * t;
    const float* pb = bb + r * T_in + 2 * t;
    out[i] = (pa[0] + pa[1]) * 0.5f + (pb[0] + pb[1]) * 0.5f;
}
int launch_pool_add(const float* a, const float* b, int B, int C, int T_in, float* out, hipStream_t st) {
    int64_t n = (int64_t)B * C * (T_in / 2);
    hipLaunchKernelGGL(pool_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, (int64_t)B * C, T_in, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
// [B][C][T_in] -> [B][T_out][C]; rows t >= T_in repeat the last input row (length fix-up :397-401)
__global__ void transpose_bct_btc_kernel(const float* __restrict__ in, int C, int T_in, int T_out, float* out) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        int c = c0 + i, t = t0 + tx;
        int ts = t < T_in ? t : T_in - 1;
        if (c < C && t < T_out) tile[i][tx] = in[((int64_t)b * C + c) * T_in + ts];
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int t = t0 + i, c = c0 + tx;
        if (t < T_out && c < C) out[((int64_t)b * T_out + t) * C + c] = tile[tx][i];
    }
}
int launch_transpose_bct_btc(const float* in, int B, int C, int T_in, int T_out, float* out, hipStream_t st) {
    hipLaunchKernelGGL(transpose_bct_btc_kernel, dim3(cdiv(T_out, 64), cdiv(C, 64), B), dim3(256), 0, st, in, C, T_in, T_out, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}


// ---------------------------------------------------------------------------
// CFG + Euler:  x[b] += dt * (v_u + s*(v_c - v_u))     (cfm1_audio.py:160 + fixed-step Euler)
// v holds the cond rows [0,B) then the uncond rows [B,2B)
// ---------------------------------------------------------------------------
__global__ void euler_cfg_kernel(float* x, const float* __restrict__ v, int64_t n, float cfg_scale, const float* dt_table,
                                 const int* step, float dt_val, int has_uncond) {
    const float dt = dt_table ? dt_table[step ? *step : 0] : dt_val;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float e = v[i];
    if (has_uncond) {
        float eu = v[n + i];
        e = eu + cfg_scale * (e - eu);
    }
    x[i] = x[i] + dt * e;
}
int launch_euler_cfg(float* x, const float* v, int B, int64_t per, float cfg_scale, const float* dt_table, const int* step,
                     float dt_val, int has_uncond, hipStream_t st) {
    int64_t n = (int64_t)B * per;
    hipLaunchKernelGGL(euler_cfg_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, v, n, cfg_scale, dt_table, step,
                       dt_val, has_uncond);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
// step bookkeeping for graph replay: (reset) step=0 or step+=1; t_idx_cur[:] = t_table[step]
__global__ void step_advance_kernel(int* step, int64_t* t_idx_cur, const int64_t* t_table, int n_steps, int Beff, int reset) {
    __shared__ int s;
    if (threadIdx.x == 0) {
        s = reset ? 0 : (*step + 1);
        *step = s;
    }
    __syncthreads();
    int k = s < n_steps ? s : n_steps - 1;
    for (int i = threadIdx.x; i < Beff; i += blockDim.x) t_idx_cur[i] = t_table[k];
}
// noise key of a sampler call -> the parameter block behind the step counter: step[4..9] = {seed, clip_base, nfe_base} as 3 x int64
__global__ void sampler_params_kernel(int* step, unsigned long long seed, long long clip_base, int nfe_base) {
    long long* prm = reinterpret_cast<long long*>(step + 4);
    prm[0] = (long long)seed; prm[1] = clip_base; prm[2] = nfe_base;
}
int launch_sampler_params(int* step, uint64_t seed, int64_t clip_base, int nfe_base, hipStream_t st) {
    hipLaunchKernelGGL(sampler_params_kernel, dim3(1), dim3(1), 0, st, step, (unsigned long long)seed, (long long)clip_base, nfe_base);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
int launch_step_ctl(int* step, int64_t* t_idx_cur, const int64_t* t_table, int n_steps, int Beff, int reset, hipStream_t st) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, st, step, t_idx_cur, t_table, n_steps, Beff, reset);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

template <int PP, bool SC, int RT_TPW = RT_TPW_MAX, int KPL = 16, int EE = 0>
__global__ void __launch_bounds__(256) router_kernel(const RouterDev a) {
    // gate weights staged once per block (every wave re-reading E*D floats per token through L1/L2 was the kernel's
    // whole cost); a wave then walks RT_TPW tokens
    extern __shared__ float rt_ws[];
    if constexpr (!SC) {
        for (int i = threadIdx.x * 4; i < a.E * a.D; i += 256 * 4) *reinterpret_cast<float4*>(rt_ws + i) = *reinterpret_cast<const float4*>(a.Wg + i);
        __syncthreads();
    }
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RT_TPW;
    if (n0 >= a.N) return;
    router_tokens<PP, SC, RT_TPW, KPL, EE>(a, n0, a.N, SC ? a.sc + (int64_t)n0 * a.NS : nullptr, a.NS, rt_ws);
}
template <int PP, bool SC, int TPW, int KPL = 16, int EE = 0>
static void launch_router_v(dim3 grid, size_t lds, hipStream_t st, const RouterDev& a) {
    hipLaunchKernelGGL((router_kernel<PP, SC, TPW, KPL, EE>), grid, dim3(256), lds, st, a);
}
int launch_router(Planes cq, const float* Wg, const float* bg, const float* la, int la_mod_rows, const float* hl, int hl_ld,
                  const float* g1, const float* g2, const float* g3, int N, int T, int D, int E, int* ic, int* ia, float* mc,
                  float* ma, float* lc_out, int B, uint64_t seed, int64_t clip_base, int nfe_base, const int* step, int block,
                  hipStream_t st, const float* sc, int NS, int Hh, int* cnt, int cnt_G, int cnt_pairs) {
    const int Bq = B > 0 ? B : 1;
    // tokens per wave: TWO (round 3; rounds 1-2: four).  Two tokens side by side amortise the noise generator and the arg-max, keep the
    // wave's registers at half of the four-token form and put twice the waves on a SIMD: same box, 12032 tokens 23.1 -> 20.0 us, whole
    // runs +1.2 % (8 clips, two streams), +2.9 % (E = 8, 32 clips), +1.8 % (4 x 120 s).  A launch that would not even put one workgroup on
    // every CU that way (one or two clips) takes one token per wave instead - the kernel is pure latency there (29 us at 1504 tokens).
    // The shape of configs/vocal2music.yaml (80 caption keys x 8 heads = 10 score columns per lane, E = 4) runs with both as compile-time
    // constants: 10 exponentials per token instead of 16 and - what matters - loads the compiler can hoist: under the run-time bound
    // `i < NS / 64` every one of the token's ten 16-byte gate-weight loads sat behind its own branch, ten L2 latencies in a row
    // (19.7 -> 15.2 us at 12032 tokens; VB_ROUTER_GENERIC: the run-time-bound form, same bits).  E = 8 keeps the run-time form: with
    // 20 weight registers per score column the hoisted form needs 220 VGPRs = two waves per SIMD and measured 0.7 % behind it (32 clips).
    const int forced = vb_tune().router_tpw;                  // VB_ROUTER_TPW=1|2|4 (tuning)
    const bool small = forced ? forced == 1 : cdiv(N, 4 * RT_TPW_MAX) < 256;
    const bool two = forced ? forced == 2 && 2 * E + 2 <= 32 : 2 * E + 2 <= 32;
    const dim3 grid(cdiv(N, 4 * (small ? 1 : (two ? 2 : RT_TPW_MAX))));
    const int pp = 2 * E + 2 <= 16 ? 4 : (2 * E + 2 <= 32 ? 2 : 1);
    RouterDev a;
    a.cq = cq; a.Wg = Wg; a.bg = bg; a.la = la; a.la_rows = la_mod_rows; a.hl = hl; a.hl_ld = hl_ld; a.g1 = g1; a.g2 = g2; a.g3 = g3;
    a.N = N; a.T = T; a.D = D; a.E = E; a.ic = ic; a.ia = ia; a.mc = mc; a.ma = ma; a.lc_out = lc_out; a.B = Bq; a.seed = seed;
    a.clip_base = clip_base; a.nfe_base = nfe_base; a.step = step; a.block = block; a.sc = sc; a.NS = sc ? NS : 0; a.Hh = sc ? Hh : 1;
    a.cnt = cnt; a.cnt_G = cnt_G; a.cnt_pairs = cnt_pairs;
    if (sc) {
        // folded caption gate: logits from attention scores + per-clip VW (see router_tokens)
        if (NS % 64 || NS > 1024 || Hh < 1 || Hh > 64 || (Hh & (Hh - 1))) VB_FAIL(VB_E_INVALID, "router: NS=%d heads=%d unsupported", NS, Hh);
        const bool fixed = NS == 640 && !vb_tune().router_generic;
        if (fixed && E == 4) {
            if (small) launch_router_v<1, true, 1, 10, 4>(grid, 0, st, a);
            else if (two) launch_router_v<2, true, 2, 10, 4>(grid, 0, st, a);
            else launch_router_v<4, true, 4, 10, 4>(grid, 0, st, a);
        } else if (small) launch_router_v<1, true, 1>(grid, 0, st, a);
        else if (two) launch_router_v<2, true, 2>(grid, 0, st, a);
        else if (pp == 4) launch_router_v<4, true, 4>(grid, 0, st, a);
        else if (pp == 2) launch_router_v<2, true, 4>(grid, 0, st, a);
        else launch_router_v<1, true, 4>(grid, 0, st, a);
        VB_CHECK_LAUNCH();
        return VB_OK;
    }
    if ((E * D) % 4 != 0 || (size_t)E * D * sizeof(float) > 64 * 1024) VB_FAIL(VB_E_INVALID, "router: E*D=%d unsupported", E * D);
    const size_t lds = (size_t)E * D * sizeof(float);
    if (small) launch_router_v<1, false, 1>(grid, lds, st, a);
    else if (two) launch_router_v<2, false, 2>(grid, lds, st, a);
    else if (pp == 4) launch_router_v<4, false, 4>(grid, lds, st, a);
    else if (pp == 2) launch_router_v<2, false, 4>(grid, lds, st, a);
    else launch_router_v<1, false, 4>(grid, lds, st, a);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// Per-clip constants of the folded caption gate (once per clip and block):
//   cbias[b][j*Hh + h] = sum_d bq_s[h*hd + d] * Kc[b][j][h*hd + d]                      (q-bias part of the scores)
//   VW[b][j*Hh + h][e] = sum_d Vc^T[b][h][d][j] * Wcg[e][h*hd + d]                      (values pre-contracted with the gate rows)
__global__ void __launch_bounds__(256) gate_fold_kernel(Planes kc, Planes vct, const float* __restrict__ bq_s, const float* __restrict__ wcg,
                                                       int Beff, int L, int Lpad, int Hh, int hd, int E, float* cbias, float* vw) {
    const int D = Hh * hd;
    const int total = Beff * L * Hh;
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int h = id % Hh, bj = id / Hh;
        const int j = bj % L, b = bj / L;
        const bf16_t* kr = kc.p + ((int64_t)(b * L + j)) * D + h * hd;
        float cb = 0.f;
        for (int d = 0; d < hd; ++d) {
            float kv = bf2f(kr[d]);
            if (kc.np == 2) kv += bf2f(kr[kc.plane + d]);
            cb += bq_s[h * hd + d] * kv;
        }
        cbias[id] = cb;
        const bf16_t* vr = vct.p + ((int64_t)(b * Hh + h) * hd) * Lpad + j;
        for (int e = 0; e < E; ++e) {
            float acc = 0.f;
            for (int d = 0; d < hd; ++d) {
                float vv = bf2f(vr[(int64_t)d * Lpad]);
                if (vct.np == 2) vv += bf2f(vr[vct.plane + (int64_t)d * Lpad]);
                acc += vv * wcg[(int64_t)e * D + h * hd + d];
            }
            vw[(int64_t)id * E + e] = acc;
        }
    }
}
int launch_gate_fold(Planes kc, Planes vct, const float* bq_s, const float* wcg, int Beff, int L, int Lpad, int Hh, int hd, int E,
                     float* cbias, float* vw, hipStream_t st) {
    hipLaunchKernelGGL(gate_fold_kernel, dim3(cdiv(Beff * L * Hh, 256)), dim3(256), 0, st, kc, vct, bq_s, wcg, Beff, L, Lpad, Hh, hd, E, cbias, vw);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
__global__ void iota_mul_kernel(int* out, int n, int mul) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = i * mul;
}
int launch_iota_mul(int* out, int n, int mul, hipStream_t st) {
    hipLaunchKernelGGL(iota_mul_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, out, n, mul);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// idx[n] = first argmax_e (logits[n][e] + gumbel[n][e])   (hard Gumbel-softmax, :81-93)
__global__ void router_top1_kernel(const float* __restrict__ logits, const float* __restrict__ gum, int N, int E, int* idx) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float best = -INFINITY; int bi = 0;
    for (int e = 0; e < E; ++e) {
        float z = logits[(int64_t)n * E + e] + gum[(int64_t)n * E + e];
        if (z > best) { best = z; bi = e; }
    }
    idx[n] = bi;
}
int launch_router_top1(const float* logits, const float* gumbel, int N, int E, int* idx, hipStream_t st) {
    hipLaunchKernelGGL(router_top1_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, logits, gumbel, N, E, idx);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// ---------------------------------------------------------------------------
// Stable bucketing of tokens by routed expert: slots [0,N) caption groups, [N,2N) acoustic groups;
// perm[slot] = token, group_off[2E+1].  Two multi-block kernels (deterministic, ascending token order inside a group):
//   bucket_count : per 256-token block, per group counts (wave ballots)       -> counts[nblk][2E]
//   bucket_place : every block re-derives its bases from the small counts table, ranks its tokens with ballots and
//                  writes perm; block 0 also writes group_off.
// `counts` lives right behind perm's 2N entries (perm buffers are sized 2N + nblk*2E + 64 by the engine).
// ---------------------------------------------------------------------------
#define BK_T 256
#define BK_G 32          // groups a launch can rank: 2E expert groups, or E*E (caption, acoustic) PAIR groups when E*E <= 16
// Pair mode (pair_off != null, E*E <= 16): the tokens are ranked ONCE, by their (caption expert c, acoustic expert a) pair.  Both
// expert-group orders fall out of the same E*E counts: the caption slots are the pair slots in c-major order (a caption group = E
// consecutive pair buckets), the acoustic slots the same buckets laid out a-major behind them - the order of the rows INSIDE an
// expert group is free (every row of a grouped GEMM is independent), so one rank per token serves perm (both halves), group_off and
// the single-launch w2 product (moe_w2_pair_kernel), whose caption-half rows are then simply its own row range:
//   perm[p] = token of pair slot p = caption slot p;  perm[pair_pa[p]] = the same token's acoustic slot;  pair_off[E*E + 1].
template <bool PAIRS>
__global__ void __launch_bounds__(BK_T) bucket_count_kernel(const int* __restrict__ ic, const int* __restrict__ ia, int N, int E, int G,
                                                           int* counts) {
    __shared__ int wc[4][BK_G];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x * BK_T + tid;
    const int gc = n < N ? ic[n] : -1, ga = n < N ? E + ia[n] : -1;
    const int gp = n < N ? gc * E + (ga - E) : -1;
    for (int g = 0; g < G; ++g) {
        const unsigned long long m = __ballot(PAIRS ? (gp == g) : (g < E ? (gc == g) : (ga == g)));
        if (lane == 0) wc[wave][g] = __popcll(m);
    }
    __syncthreads();
    if (tid < G) counts[blockIdx.x * G + tid] = wc[0][tid] + wc[1][tid] + wc[2][tid] + wc[3][tid];
}
template <bool PAIRS>
__global__ void __launch_bounds__(BK_T) bucket_place_kernel(const int* __restrict__ ic, const int* __restrict__ ia, int N, int E, int G,
                                                           const int* __restrict__ counts, int nblk, int* group_off, int* perm,
                                                           int* pair_off, int* pair_pa, int* counts_clear) {
    // (round 5) this block's row of the OTHER count table: the next router launch adds into it
    if (counts_clear && threadIdx.x < G) counts_clear[blockIdx.x * G + threadIdx.x] = 0;
    __shared__ int base[BK_G];        // slot of this block's first token of every group (pair mode: caption slot)
    __shared__ int base2[BK_G];       // pair mode: acoustic slot of this block's first token of every pair
    __shared__ int wc[4][BK_G];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // group start = sum of all earlier groups' totals; + this group's tokens in earlier blocks.  The counts table is
    // summed by the whole block (thread = (row-of-counts, group)), not by G serial threads.
    __shared__ int tot[BK_G], bef[BK_G];
    if (tid < BK_G) { tot[tid] = 0; bef[tid] = 0; }
    __syncthreads();
    {
        const int g = tid % G;
        int t = 0, bf = 0;
        for (int b = tid / G; b < nblk; b += BK_T / G) {
            const int cnt = counts[b * G + g];
            t += cnt;
            if (b < (int)blockIdx.x) bf += cnt;
        }
        if (tid < (BK_T / G) * G) { atomicAdd(&tot[g], t); atomicAdd(&bef[g], bf); }
    }
    __syncthreads();
    if (tid < G) {
        int before_groups = 0;
        for (int g = 0; g < tid; ++g) before_groups += tot[g];
        base[tid] = before_groups + bef[tid];
        if constexpr (PAIRS) {
            // a-major order of the same buckets: everything with a smaller acoustic expert, then the same a with a smaller caption expert
            const int c = tid / E, a = tid - c * E;
            int beforeT = 0;
            for (int g = 0; g < G; ++g) {
                const int c2 = g / E, a2 = g - c2 * E;
                if (a2 < a || (a2 == a && c2 < c)) beforeT += tot[g];
            }
            base2[tid] = N + beforeT + bef[tid];
            if (blockIdx.x == 0) {
                pair_off[tid] = before_groups;
                if (tid == G - 1) pair_off[G] = before_groups + tot[tid];
                if (a == 0) group_off[c] = before_groups;                 // caption group c starts at its first pair bucket
                if (c == 0) group_off[E + a] = N + beforeT;               // acoustic group a starts at pair (0, a) in a-major order
                if (tid == 0) { group_off[E] = N; group_off[2 * E] = 2 * N; }
            }
        } else if (blockIdx.x == 0) {
            group_off[tid] = before_groups;
            if (tid == G - 1) group_off[G] = before_groups + tot[tid];
        }
    }
    const int n = blockIdx.x * BK_T + tid;
    const int gc = n < N ? ic[n] : -1, ga = n < N ? E + ia[n] : -1;
    const int gp = n < N ? gc * E + (ga - E) : -1;
    int rank_c = 0, rank_a = 0;
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int g = 0; g < G; ++g) {
        const unsigned long long m = __ballot(PAIRS ? (gp == g) : (g < E ? (gc == g) : (ga == g)));
        if (lane == 0) wc[wave][g] = __popcll(m);
        if constexpr (PAIRS) {
            if (g == gp) rank_c = __popcll(m & lower);
        } else {
            if (g == gc) rank_c = __popcll(m & lower);
            if (g == ga) rank_a = __popcll(m & lower);
        }
    }
    __syncthreads();
    if (n < N) {
        if constexpr (PAIRS) {
            int r = rank_c;
            for (int w = 0; w < wave; ++w) r += wc[w][gp];
            const int pc = base[gp] + r, pa = base2[gp] + r;
            perm[pc] = n;
            perm[pa] = n;
            pair_pa[pc] = pa;
        } else {
            int pc = base[gc] + rank_c, pa = base[ga] + rank_a;
            for (int w = 0; w < wave; ++w) { pc += wc[w][gc]; pa += wc[w][ga]; }
            perm[pc] = n;
            perm[pa] = n;
        }
    }
}
// (Round 3, measured and removed: count + place as ONE launch at up to 64 blocks, every block re-deriving all chunks' counts itself
//  instead of reading the table of a first launch: correct, bit-identical - and 62.7 us against 4.7 + 6.3, because the walk over the 47
//  chunks is 47 dependent L2 round trips per block.)
// Small token counts (one or two clips: the reference's serving shape, scripts/test_final.py:357): count + place in ONE launch of one
// 1024-thread workgroup - at 1504 tokens the two multi-block kernels above are two ~4.6-us launch floors for 6 blocks of work.
// Same result, bit for bit (stable: ascending token order inside a group).
#define BKS_T 1024
#define BKS_CH 4            // chunks of 1024 tokens: N <= 4096
template <bool PAIRS>
__global__ void __launch_bounds__(BKS_T) bucket_small_kernel(const int* __restrict__ ic, const int* __restrict__ ia, int N, int E,
                                                            int* group_off, int* perm, int* pair_off, int* pair_pa) {
    __shared__ int cnt[BKS_CH * 16][BK_G];      // [chunk * 16 + wave][group]: count, then exclusive prefix inside the group
    __shared__ int gbase[BK_G + 1];             // group start (pair mode: caption-major start of the pair bucket)
    __shared__ int gbase2[BK_G];                // pair mode: acoustic-major start of the pair bucket (slots [N, 2N))
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = PAIRS ? E * E : 2 * E;
    const int nch = (N + BKS_T - 1) / BKS_T;
    int gc[BKS_CH], ga[BKS_CH], rc[BKS_CH], ra[BKS_CH];
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int ch = 0; ch < BKS_CH; ++ch) {
        gc[ch] = -1; ga[ch] = -1; rc[ch] = 0; ra[ch] = 0;
        if (ch < nch) {
            const int n = ch * BKS_T + tid;
            if (n < N) {
                if (PAIRS) { gc[ch] = ic[n] * E + ia[n]; } else { gc[ch] = ic[n]; ga[ch] = E + ia[n]; }
            }
            for (int g = 0; g < G; ++g) {
                const unsigned long long m = __ballot(PAIRS ? (gc[ch] == g) : (g < E ? (gc[ch] == g) : (ga[ch] == g)));
                if (lane == 0) cnt[ch * 16 + wave][g] = __popcll(m);
                if (g == gc[ch]) rc[ch] = __popcll(m & lower);
                if (!PAIRS && g == ga[ch]) ra[ch] = __popcll(m & lower);
            }
        }
    }
    __syncthreads();
    __shared__ int tot[BK_G];
    if (tid < G) {
        int run = 0;
        for (int i = 0; i < nch * 16; ++i) { const int c = cnt[i][tid]; cnt[i][tid] = run; run += c; }
        tot[tid] = run;
    }
    __syncthreads();
    if (tid < G) {
        // every group works out its own start from the totals (no serial section): caption-major prefix, and in pair mode the
        // acoustic-major start of the same bucket (everything with a smaller acoustic expert, then the same a with a smaller c)
        int before = 0;
        for (int g = 0; g < tid; ++g) before += tot[g];
        gbase[tid] = before;
        if (tid == G - 1) gbase[G] = before + tot[tid];
        if (PAIRS) {
            const int c = tid / E, a = tid - c * E;
            int b2 = 0;
            for (int g2 = 0; g2 < G; ++g2) {
                const int c2 = g2 / E, a2 = g2 - c2 * E;
                if (a2 < a || (a2 == a && c2 < c)) b2 += tot[g2];
            }
            gbase2[tid] = N + b2;
        }
    }
    __syncthreads();
    if (PAIRS) {
        if (tid <= G) pair_off[tid] = gbase[tid];
        if (tid < E) { group_off[tid] = gbase[tid * E]; group_off[E + tid] = gbase2[tid]; }      // caption group c = pair (c, 0); acoustic a = pair (0, a)
        if (tid == 0) { group_off[E] = N; group_off[2 * E] = 2 * N; }
    } else {
        if (tid <= G) group_off[tid] = gbase[tid];
    }
#pragma unroll
    for (int ch = 0; ch < BKS_CH; ++ch) {
        const int n = ch * BKS_T + tid;
        if (ch < nch && n < N) {
            if (PAIRS) {
                const int r = cnt[ch * 16 + wave][gc[ch]] + rc[ch];
                const int pc = gbase[gc[ch]] + r, pa = gbase2[gc[ch]] + r;
                perm[pc] = n;
                perm[pa] = n;
                pair_pa[pc] = pa;
            } else {
                perm[gbase[gc[ch]] + cnt[ch * 16 + wave][gc[ch]] + rc[ch]] = n;
                perm[gbase[ga[ch]] + cnt[ch * 16 + wave][ga[ch]] + ra[ch]] = n;
            }
        }
    }
}
static_assert(RT_CNT_BLOCK == BK_T, "the router counts per bucket block");
bool bucket_router_counts_ok(int N) { return N > BKS_T * BKS_CH; }
int bucket_counts_ints(int N) { return 2 * (cdiv(N, BK_T) * BK_G + 32); }
int* bucket_counts(int* perm, int N, int which) { return perm + 2 * (size_t)N + (size_t)which * (cdiv(N, BK_T) * BK_G + 32); }
int launch_bucket(const int* ic, const int* ia, int N, int E, int* group_off, int* perm, hipStream_t st, int* pair_off, int* pair_pa,
                  const int* counts_ready, int* counts_clear) {
    if (E > 16) VB_FAIL(VB_E_INVALID, "bucket: E=%d > 16", E);
    const bool pairs = pair_off != nullptr;
    if (pairs && E * E > 16) VB_FAIL(VB_E_INVALID, "bucket: pair mode needs E*E <= 16 (E=%d)", E);
    if (N <= BKS_T * BKS_CH) {
        if (counts_ready) VB_FAIL(VB_E_INVALID, "bucket: router-side counts belong to the two-kernel form (N > %d)", BKS_T * BKS_CH);
        if (pairs) hipLaunchKernelGGL(bucket_small_kernel<true>, dim3(1), dim3(BKS_T), 0, st, ic, ia, N, E, group_off, perm, pair_off, pair_pa);
        else hipLaunchKernelGGL(bucket_small_kernel<false>, dim3(1), dim3(BKS_T), 0, st, ic, ia, N, E, group_off, perm, nullptr, nullptr);
        VB_CHECK_LAUNCH();
        return VB_OK;
    }
    const int G = pairs ? E * E : 2 * E;
    const int nblk = cdiv(N, BK_T);
    int* counts = bucket_counts(perm, N, 0);     // scratch tail of the perm buffer (see bucket_scratch_ints)
    if (pairs) {
        if (!counts_ready) hipLaunchKernelGGL(bucket_count_kernel<true>, dim3(nblk), dim3(BK_T), 0, st, ic, ia, N, E, G, counts);
        hipLaunchKernelGGL(bucket_place_kernel<true>, dim3(nblk), dim3(BK_T), 0, st, ic, ia, N, E, G, counts_ready ? counts_ready : counts, nblk, group_off,
                           perm, pair_off, pair_pa, counts_clear);
    } else {
        if (!counts_ready) hipLaunchKernelGGL(bucket_count_kernel<false>, dim3(nblk), dim3(BK_T), 0, st, ic, ia, N, E, G, counts);
        hipLaunchKernelGGL(bucket_place_kernel<false>, dim3(nblk), dim3(BK_T), 0, st, ic, ia, N, E, G, counts_ready ? counts_ready : counts, nblk, group_off,
                           perm, nullptr, nullptr, counts_clear);
    }
    VB_CHECK_LAUNCH();
    return VB_OK;
}
int bucket_scratch_ints(int N, int E) { (void)E; return bucket_counts_ints(N) + 64; }

// ---------------------------------------------------------------------------
// proj_in (Conv1d C -> D, k taps; vocal2music_moe.py:395) as a GEMM: the latent window of every token as one K-contiguous row
//   A[plane][m = b*T + t][k = tap*32 + ci] = split-bf16( x[b][ci][t + tap - pad] )   (zero outside the clip, for ci >= C and for k >= taps*32)
// and the split conv weights [2][taps][D][32] re-laid as B[plane][D][KP].  One thread = one 16-byte group of 8 consecutive k.
// ---------------------------------------------------------------------------
__global__ void im2col_latent_kernel(const float* __restrict__ x, int B, int C, int T, int taps, int pad, int KP, bf16_t* __restrict__ out,
                                     int64_t plane) {
    const int gpr = KP >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * T * gpr) return;
    const int m = (int)(idx / gpr), gi = (int)(idx - (int64_t)m * gpr);
    const int b = m / T, t = m - b * T;
    const int k0 = gi * 8, tap = k0 >> 5, ci0 = k0 & 31;
    const int ts = t + tap - pad;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = ci0 + e;
        const float v = (tap < taps && ci < C && ts >= 0 && ts < T) ? x[((int64_t)b * C + ci) * T + ts] : 0.f;
        hi[e] = f2bf(v);
        lo[e] = f2bf(v - bf2f(hi[e]));
    }
    *reinterpret_cast<bf16x8*>(out + (int64_t)m * KP + k0) = hi;
    *reinterpret_cast<bf16x8*>(out + plane + (int64_t)m * KP + k0) = lo;
}
int launch_im2col_latent(const float* x, int B, int C, int T, int taps, int pad, int KP, bf16_t* out, int64_t plane, hipStream_t st) {
    if (KP % 8 || C > 32 || taps * 32 > KP) VB_FAIL(VB_E_INVALID, "im2col_latent: C=%d taps=%d KP=%d", C, taps, KP);
    const int64_t n = (int64_t)B * T * (KP / 8);
    hipLaunchKernelGGL(im2col_latent_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, B, C, T, taps, pad, KP, out, plane);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
__global__ void conv_w_to_gemm_kernel(const bf16_t* __restrict__ w3, int64_t w3_plane, int taps, int D, int KP, bf16_t* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (plane, co, k)
    if (idx >= (int64_t)2 * D * KP) return;
    const int k = (int)(idx % KP), co = (int)((idx / KP) % D), pl = (int)(idx / ((int64_t)KP * D));
    const int tap = k >> 5, ci = k & 31;
    out[idx] = tap < taps ? w3[pl * w3_plane + ((int64_t)tap * D + co) * 32 + ci] : f2bf(0.f);
}
int launch_conv_w_to_gemm(const bf16_t* w3, int64_t w3_plane, int taps, int D, int KP, bf16_t* out, hipStream_t st) {
    const int64_t n = (int64_t)2 * D * KP;
    hipLaunchKernelGGL(conv_w_to_gemm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, w3, w3_plane, taps, D, KP, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// element (row n = (branch*B + b)*T + t, e) of stream (seed, clip_base + b, nfe, branch, block, gate)
__global__ void fill_gumbel_kernel(float* out, int B, int n_branch, int T, int width, uint64_t seed, int64_t clip_base,
                                   int nfe_base, const int* step, int block, int gate) {
    const int64_t n_el = (int64_t)n_branch * B * T * width;
    const int nfe = nfe_base + (step ? *step : 0);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_el; i += stride) {
        int64_t row = i / width; int e = (int)(i - row * width);
        int bb = (int)(row / T), t = (int)(row - (int64_t)bb * T);
        int branch = bb / B, b = bb - branch * B;
        out[i] = gumbel_draw(seed, clip_base + b, nfe, branch, block, gate, t, width, e);
    }
}
int launch_fill_gumbel(float* out, int B, int n_branch, int T, int width, uint64_t seed, int64_t clip_base, int nfe_base,
                       const int* step, int block, int gate, hipStream_t st) {
    int64_t n = (int64_t)n_branch * B * T * width;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_gumbel_kernel, dim3(blocks), dim3(256), 0, st, out, B, n_branch, T, width, seed, clip_base, nfe_base, step,
                       block, gate);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// out[n][e] = x[n] . W[e] + b[e]   (acoustic gate logits, precompute)
__global__ void __launch_bounds__(256) rows_dot_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ bias, int N, int D, int E, float* out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    for (int e = 0; e < E; ++e) {
        float acc = 0.f;
        for (int k = lane * 4; k < D; k += 256) {
            float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)n * D + k);
            float4 wv = *reinterpret_cast<const float4*>(W + (int64_t)e * D + k);
            acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
        acc = wave_sum(acc);
        if (lane == 0) out[(int64_t)n * E + e] = acc + bias[e];
    }
}
int launch_rows_dot(const float* x, const float* W, const float* bias, int N, int D, int E, float* out, hipStream_t st) {
    hipLaunchKernelGGL(rows_dot_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, x, W, bias, N, D, E, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// ---------------------------------------------------------------------------
// GroupNorm statistics (autoencoder1d.py:165-166): one block per (b, group); the group's
// channels are contiguous in [B][C][T].  Two-pass mean / biased variance.
// ---------------------------------------------------------------------------
#define GS_T 1024
__global__ void __launch_bounds__(GS_T) gn_stats_kernel(const float* __restrict__ x, int C, int T, int groups, float eps, float* mean,
                                                       float* rstd) {
    __shared__ float red[GS_T / 64];
    __shared__ float s_mean;
    const int bg = blockIdx.x;
    const int64_t n = (int64_t)(C / groups) * T;
    const float* p = x + (int64_t)bg * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    const int64_t n4 = vec ? n / 4 : 0;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n4; i += GS_T) {
        const float4 v = reinterpret_cast<const float4*>(p)[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += GS_T) s += p[i];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < GS_T / 64; ++w) t += red[w];
        s_mean = t / (float)n;
    }
    __syncthreads();
    const float m = s_mean;
    float v = 0.f;
    for (int64_t i = threadIdx.x; i < n4; i += GS_T) {
        const float4 q = reinterpret_cast<const float4*>(p)[i];
        const float d0 = q.x - m, d1 = q.y - m, d2 = q.z - m, d3 = q.w - m;
        v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += GS_T) { float d = p[i] - m; v += d * d; }
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < GS_T / 64; ++w) t += red[w];
        mean[bg] = m;
        rstd[bg] = rsqrtf(t / (float)n + eps);
    }
}
int launch_gn_stats(const float* x, int B, int C, int T, int groups, float eps, float* mean, float* rstd, hipStream_t st) {
    if (C % groups) VB_FAIL(VB_E_INVALID, "gn_stats: C%%groups");
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * groups), dim3(GS_T), 0, st, x, C, T, groups, eps, mean, rstd);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// FinalLayer input: LayerNorm without affine (eps) then adaLN modulate, as split-bf16 planes; one wave per row, the row
// (D <= 1024) lives in registers between the two reductions
__global__ void __launch_bounds__(256) layernorm_mod_planes_kernel(const float* __restrict__ h, const float* __restrict__ shift,
                                                                  const float* __restrict__ scale, int mod_ld, int rows, int D, int T,
                                                                  float eps, Planes out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = h + (int64_t)row * D;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + i * 256;
        v[i] = k < D ? *reinterpret_cast<const float4*>(x + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + i * 256;
        if (k < D) {
            const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    const int b = row / T;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + i * 256;
        if (k >= D) continue;
        const float4 sc = *reinterpret_cast<const float4*>(scale + (int64_t)b * mod_ld + k);
        const float4 sh = *reinterpret_cast<const float4*>(shift + (int64_t)b * mod_ld + k);
        const float o[4] = {(v[i].x - mean) * rs * (1.f + sc.x) + sh.x, (v[i].y - mean) * rs * (1.f + sc.y) + sh.y,
                            (v[i].z - mean) * rs * (1.f + sc.z) + sh.z, (v[i].w - mean) * rs * (1.f + sc.w) + sh.w};
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = f2bf(o[e]); lo[e] = f2bf(o[e] - bf2f(hi[e])); }
        *reinterpret_cast<bf16x4*>(out.p + (int64_t)row * D + k) = hi;
        if (out.np == 2) *reinterpret_cast<bf16x4*>(out.p + out.plane + (int64_t)row * D + k) = lo;
    }
}
int launch_layernorm_mod_planes(const float* h, const float* shift, const float* scale, int mod_ld, int rows, int D, int T, float eps,
                                Planes out, hipStream_t st) {
    if (D % 4 || D > 1024) VB_FAIL(VB_E_INVALID, "layernorm_mod_planes: D=%d", D);
    hipLaunchKernelGGL(layernorm_mod_planes_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, h, shift, scale, mod_ld, rows, D, T > 0 ? T : 1, eps, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// Pre-pass for wide conv layers: x f32 [B][C][T] -> activated, split-bf16, TRANSPOSED planes [2][B][Tp][C] (C contiguous), with
// XT_HEAD zero rows in front and zero rows behind (Tp = T_eff + XT_HEAD + XT_TAIL), so the conv kernel can DMA its input window
// straight into LDS: the pointwise transform (GroupNorm affine, swish, LeakyReLU), the hi/lo split, the transpose and the zero
// padding happen ONCE here instead of once per output-channel tile of the convolution (12 tiles on the 1536-channel layers).
__global__ void __launch_bounds__(256) xt_planes_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int groups, int act,
                                                       float slope, int upsample2, int C, int T_in, int Tp, bf16_t* out, int64_t plane) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, r0 = blockIdx.x * 64;          // r = row of the padded image
    const int T_eff = upsample2 ? 2 * T_in : T_in;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int cpg = groups > 0 ? C / groups : 1;
    {
        const int t = r0 + tx - XT_HEAD;
        const bool tin = t >= 0 && t < T_eff;
        const int ts = upsample2 ? (t >> 1) : t;
        for (int cc = ty; cc < 64; cc += 4) {
            const int c = c0 + cc;
            float v = 0.f;
            if (tin && c < C) {
                v = x[((int64_t)b * C + c) * T_in + ts];
                if (act == ACT_GN || act == ACT_GN_SWISH) {
                    const int grp = c / cpg;
                    const float rs = rstd[b * groups + grp] * gamma[c];
                    v = v * rs + (beta[c] - mean[b * groups + grp] * rs);
                    if (act == ACT_GN_SWISH) v = v / (1.f + __expf(-v));
                } else if (act == ACT_LRELU) {
                    v = v > 0.f ? v : v * slope;
                }
            }
            tile[tx][cc] = v;
        }
    }
    __syncthreads();
    const int tr = threadIdx.x >> 2, cg = (threadIdx.x & 3) * 16;
    const int r = r0 + tr;
    if (r < Tp && c0 + cg < C) {
        bf16x8 hi[2], lo[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = tile[tr][cg + e];
            const bf16_t h = f2bf(v);
            hi[e >> 3][e & 7] = h;
            lo[e >> 3][e & 7] = f2bf(v - bf2f(h));
        }
        bf16_t* dst = out + ((int64_t)b * Tp + r) * C + c0 + cg;
        *reinterpret_cast<bf16x8*>(dst) = hi[0];
        *reinterpret_cast<bf16x8*>(dst + 8) = hi[1];
        *reinterpret_cast<bf16x8*>(dst + plane) = lo[0];
        *reinterpret_cast<bf16x8*>(dst + plane + 8) = lo[1];
    }
}
int launch_xt_planes(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int groups, int act,
                     float slope, int upsample2, int B, int C, int T_in, bf16_t* out, hipStream_t st) {
    if (C % 16) VB_FAIL(VB_E_INVALID, "xt_planes: C %% 16");
    const int Tp = xt_rows(upsample2 ? 2 * T_in : T_in);
    hipLaunchKernelGGL(xt_planes_kernel, dim3(cdiv(Tp, 64), cdiv(C, 64), B), dim3(256), 0, st, x, mean, rstd, gamma, beta, groups, act, slope,
                       upsample2, C, T_in, Tp, out, (int64_t)B * Tp * C);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// BigVGAN anti-aliased periodic activation (alias_free_torch Activation1d, ratio 2, 12-tap Kaiser-sinc filter f):
//   up[v]  = 2 * sum_i xp[i] f[v + 15 - 2 i]          xp = x replicate-padded by 5          (UpSample1d, resample.py:10-32)
//   s[v]   = up[v] + inv_beta * sin^2(alpha * up[v])                                         (Snake / SnakeBeta, activations.py)
//   out[t] = sum_k f[k] s[clamp(2 t + k - 5, 0, 2T-1)]                                       (DownSample1d / LowPassFilter1d)
// One workgroup = 256 outputs of one (batch, channel) row: the 523 intermediate samples are computed once into LDS.
#define AA_TT 256
__global__ void __launch_bounds__(256) aa_act_kernel(const float* __restrict__ x, const float* __restrict__ alpha, const float* __restrict__ inv_beta,
                                                    const float* __restrict__ filt, int C, int T, float* out) {
    __shared__ float xs[AA_TT + 16];
    __shared__ float ss[2 * AA_TT + 16];
    __shared__ float f[12];
    const int row = blockIdx.y, c = row % C;
    const int t0 = blockIdx.x * AA_TT;
    const float* xr = x + (int64_t)row * T;
    const int tid = threadIdx.x;
    if (tid < 12) f[tid] = filt[tid];
    for (int j = tid; j < AA_TT + 16; j += 256) {
        int pos = t0 - 6 + j;
        pos = pos < 0 ? 0 : (pos > T - 1 ? T - 1 : pos);
        xs[j] = xr[pos];
    }
    __syncthreads();
    const float a = alpha[c], ib = inv_beta[c];
    for (int q = tid; q < 2 * AA_TT + 11; q += 256) {
        int v = 2 * t0 - 5 + q;
        v = v < 0 ? 0 : (v > 2 * T - 1 ? 2 * T - 1 : v);
        const int i_lo = (v + 5) >> 1;                      // ceil((v + 4) / 2)
        float up = 0.f;
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int i = i_lo + m;
            const int tap = v + 15 - 2 * i;                 // 11 - (v+5)%2 ... >= 0 by construction for m < 6
            if (tap >= 0 && tap < 12) up += xs[i - t0 + 1] * f[tap];
        }
        up *= 2.f;
        const float sn = sinf(up * a);
        ss[q] = up + ib * (sn * sn);
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t < T) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) acc += f[k] * ss[2 * tid + k];
        out[(int64_t)row * T + t] = acc;
    }
}
int launch_aa_act(const float* x, const float* alpha, const float* inv_beta, const float* filt, int B, int C, int T, float* out, hipStream_t st) {
    hipLaunchKernelGGL(aa_act_kernel, dim3(cdiv(T, AA_TT), B * C), dim3(256), 0, st, x, alpha, inv_beta, filt, C, T, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// A operand of the adaLN tabulation GEMM: row (step, sample) = silu(temb[step] + cemb[sample]) as split-bf16 planes
__global__ void __launch_bounds__(256) silu_sum_planes_kernel(const float* __restrict__ temb, const float* __restrict__ cemb, int rows,
                                                             int D, int nsample, bf16_t* out, int64_t plane) {
    const int64_t total = (int64_t)rows * (D / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / (D / 4)), k = (int)(i - (int64_t)r * (D / 4)) * 4;
        const int step = r / nsample, smp = r - step * nsample;
        const float4 a = *reinterpret_cast<const float4*>(temb + (int64_t)step * D + k);
        const float4 b = *reinterpret_cast<const float4*>(cemb + (int64_t)smp * D + k);
        float v[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sv = v[e] / (1.f + expf(-v[e]));          // same expression as the row-linear kernel's input activation
            hi[e] = f2bf(sv);
            lo[e] = f2bf(sv - bf2f(hi[e]));
        }
        *reinterpret_cast<bf16x4*>(out + (int64_t)r * D + k) = hi;
        *reinterpret_cast<bf16x4*>(out + plane + (int64_t)r * D + k) = lo;
    }
}
int launch_silu_sum_planes(const float* temb, const float* cemb, int rows, int D, int nsample, bf16_t* out, int64_t plane, hipStream_t st) {
    if (D % 4) VB_FAIL(VB_E_INVALID, "silu_sum_planes: D %% 4");
    int64_t blocks = ((int64_t)rows * (D / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(silu_sum_planes_kernel, dim3((int)blocks), dim3(256), 0, st, temb, cemb, rows, D, nsample, out, plane);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// GroupNorm affine (+ swish) applied once, for the wide VAE layers: the conv kernels can fuse it into their staging, but a
// layer with Co/128 output-channel tiles would then redo the exp/div of every input element Co/128 times
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int C, int T, int groups, int swish, float* out) {
    const int row = blockIdx.y;                 // b * C + c
    const int b = row / C, c = row - b * C;
    const int grp = c / (C / groups);
    const float rs = rstd[b * groups + grp] * gamma[c];
    const float sh = beta[c] - mean[b * groups + grp] * rs;
    const float* xr = x + (int64_t)row * T;
    float* orow = out + (int64_t)row * T;
    for (int t = (blockIdx.x * 256 + threadIdx.x) * 4; t < T; t += gridDim.x * 1024) {
        if (t + 3 < T && (T & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(xr + t);
            float o[4] = {v.x * rs + sh, v.y * rs + sh, v.z * rs + sh, v.w * rs + sh};
            if (swish) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = o[k] / (1.f + __expf(-o[k]));
            }
            *reinterpret_cast<float4*>(orow + t) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            for (int k = t; k < min(t + 4, T); ++k) {
                float o = xr[k] * rs + sh;
                if (swish) o = o / (1.f + __expf(-o));
                orow[k] = o;
            }
        }
    }
}
int launch_gn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int B, int C, int T,
                    int groups, int swish, float* out, hipStream_t st) {
    if (C % groups) VB_FAIL(VB_E_INVALID, "gn_apply: C %% groups");
    hipLaunchKernelGGL(gn_apply_kernel, dim3(cdiv(T, 1024), B * C), dim3(256), 0, st, x, mean, rstd, gamma, beta, C, T, groups, swish, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// softmax over the last dim of s[B][R][Cc], written transposed: out_t[b][c][r]
__global__ void __launch_bounds__(256) softmax_rows_t_kernel(const float* __restrict__ s, int R, int Cc, float* out_t) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* p = s + ((int64_t)b * R + row) * Cc;
    float m = -INFINITY;
    for (int c = lane; c < Cc; c += 64) m = fmaxf(m, p[c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < Cc; c += 64) sum += expf(p[c] - m);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int c = lane; c < Cc; c += 64) out_t[((int64_t)b * Cc + c) * R + row] = expf(p[c] - m) * inv;
}
int launch_softmax_rows_t(const float* s, int B, int R, int Ccols, float* out_t, hipStream_t st) {
    hipLaunchKernelGGL(softmax_rows_t_kernel, dim3(cdiv(R, 4), B), dim3(256), 0, st, s, R, Ccols, out_t);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

// out[i] = i / div   (row -> step index of the tabulated conditioning vectors)
__global__ void iota_div_kernel(int64_t* out, int n, int div) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i / div;
}
int launch_iota_div(int64_t* out, int n, int div, hipStream_t st) {
    hipLaunchKernelGGL(iota_div_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, out, n, div);
    VB_CHECK_LAUNCH();
    return VB_OK;
}


// ---------------------------------------------------------------------------
// long-form generation (BASELINE configs[4], build-defined: versband_amd/longform.py): cross-fade of the window results
//   out[b][c][t] = sum_w wgt_w(t) * parts[w*B + b][c][t - s_w] / sum_w wgt_w(t)
// wgt = 1 inside a window, linear ramps (k+1)/(ov+1) over the overlap with the previous window and 1 - (k+1)/(ov+1) over the overlap
// with the next one (the interior points of linspace(0, 1, ov + 2)), the minimum of the two where both apply - window order and
// arithmetic of longform.crossfade_windows (the torch restatement the oracle fixture was generated with).
// ---------------------------------------------------------------------------
struct XfadeStarts { int s[64]; };
__global__ void __launch_bounds__(256) crossfade_windows_kernel(const float* __restrict__ parts, XfadeStarts st, int nw, int B, int C, int n, int T,
                                                                float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int t = (int)(i % T);
    const int64_t bc = i / T;
    const int b = (int)(bc / C), c = (int)(bc - (int64_t)b * C);
    float acc = 0.f, wsum = 0.f;
    for (int w = 0; w < nw; ++w) {
        const int s = st.s[w], u = t - s;
        if (u < 0 || u >= n) continue;
        float wt = 1.f;
        if (w > 0) {
            const int ov = st.s[w - 1] + n - s;
            if (ov > 0 && u < ov) wt = (float)(u + 1) / (float)(ov + 1);
        }
        if (w + 1 < nw) {
            const int ov = s + n - st.s[w + 1];
            if (ov > 0 && u >= n - ov) wt = fminf(wt, 1.f - (float)(u - (n - ov) + 1) / (float)(ov + 1));
        }
        acc += parts[(((int64_t)w * B + b) * C + c) * n + u] * wt;
        wsum += wt;
    }
    out[i] = acc / wsum;
}
int launch_crossfade_windows(const float* parts, const int* starts, int nw, int B, int C, int n, int T, float* out, hipStream_t st) {
    if (nw < 1 || nw > 64) VB_FAIL(VB_E_INVALID, "crossfade: %d windows (1..64)", nw);
    XfadeStarts xs;
    for (int w = 0; w < 64; ++w) xs.s[w] = w < nw ? starts[w] : 0;
    for (int w = 0; w < nw; ++w)
        if (xs.s[w] < 0 || xs.s[w] + n > T || (w > 0 && (xs.s[w] <= xs.s[w - 1] || xs.s[w] > xs.s[w - 1] + n)))
            VB_FAIL(VB_E_INVALID, "crossfade: window %d at %d (length %d) does not continue the cover of [0, %d)", w, xs.s[w], n, T);
    if (xs.s[0] != 0 || xs.s[nw - 1] + n != T) VB_FAIL(VB_E_INVALID, "crossfade: the windows do not cover [0, %d)", T);
    const int64_t tot = (int64_t)B * C * T;
    hipLaunchKernelGGL(crossfade_windows_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, parts, xs, nw, B, C, n, T, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
