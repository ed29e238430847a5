// T5 encoder helpers (gfx950): token-embedding gather and the self-attention of the text encoder.
// The encoder sees 80 tokens per caption: its projections are small GEMMs on the shared bf16x3 MFMA kernel; attention is
// 80 x 80 x 64 per (caption, head) - 0.4 MFLOP - so it runs as a plain fp32 kernel, one workgroup per (caption, head), with
// q, k, v of the head in LDS.  HF semantics (transformers T5Attention): scores = q k^T + position_bias, NO 1/sqrt(d) scaling,
// softmax in fp32, no mask (the reference passes none, ldm/modules/encoders/modules.py:221).
#include "kernels.h"

__global__ void gather_rows_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table, int rows, int D, int vocab, float* out) {
    const int r = blockIdx.x;
    int64_t id = idx[r];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const float* src = table + id * D;
    for (int k = threadIdx.x * 4; k < D; k += blockDim.x * 4) *reinterpret_cast<float4*>(out + (int64_t)r * D + k) = *reinterpret_cast<const float4*>(src + k);
}
int launch_gather_rows(const int64_t* idx, const float* table, int rows, int D, int vocab, float* out, hipStream_t st) {
    if (D % 4) VB_FAIL(VB_E_INVALID, "gather_rows: D %% 4");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, st, idx, table, rows, D, vocab, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}

#define T5_LMAX 128
#define T5_DKV 64
// qkv planes [B*L][3*heads*dkv] (q | k | v), out planes [B*L][heads*dkv]; block = (head, caption), 256 threads
__global__ void __launch_bounds__(256) t5_attn_kernel(Planes qkv, const float* __restrict__ pos_bias, int pos_len, int L, int heads,
                                                     Planes out) {
    __shared__ float qs[T5_LMAX][T5_DKV + 1], ks[T5_LMAX][T5_DKV + 1], vs[T5_LMAX][T5_DKV + 1];
    __shared__ float ps[4][T5_LMAX];
    const int h = blockIdx.x, b = blockIdx.y;
    const int inner = heads * T5_DKV, ld = 3 * inner;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int id = tid; id < L * T5_DKV; id += 256) {
        const int t = id / T5_DKV, d = id - t * T5_DKV;
        const int64_t base = ((int64_t)b * L + t) * ld + h * T5_DKV + d;
        float q = bf2f(qkv.p[base]), k = bf2f(qkv.p[base + inner]), v = bf2f(qkv.p[base + 2 * inner]);
        if (qkv.np == 2) { q += bf2f(qkv.p[qkv.plane + base]); k += bf2f(qkv.p[qkv.plane + base + inner]); v += bf2f(qkv.p[qkv.plane + base + 2 * inner]); }
        qs[t][d] = q; ks[t][d] = k; vs[t][d] = v;
    }
    __syncthreads();
    // one wave per query row (rows wave, wave+4, ...): lanes over keys (2 per lane up to 128), then over d for P.V
    for (int i = wave; i < L; i += 4) {
        float s[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = lane + 64 * jj;
            float acc = -INFINITY;
            if (j < L) {
                acc = 0.f;
                for (int d = 0; d < T5_DKV; ++d) acc += qs[i][d] * ks[j][d];
                acc += pos_bias[((int64_t)h * pos_len + i) * pos_len + j];
            }
            s[jj] = acc;
        }
        float m = fmaxf(s[0], s[1]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float p0 = (lane < L) ? expf(s[0] - m) : 0.f, p1 = (lane + 64 < L) ? expf(s[1] - m) : 0.f;
        float l = p0 + p1;
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
        const float inv = 1.f / l;
        ps[wave][lane] = p0 * inv;
        ps[wave][lane + 64] = p1 * inv;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;                               // lane = d
        for (int j = 0; j < L; ++j) o += ps[wave][j] * vs[j][lane];
        const int64_t oi = ((int64_t)b * L + i) * inner + h * T5_DKV + lane;
        const bf16_t hi = f2bf(o);
        out.p[oi] = hi;
        if (out.np == 2) out.p[out.plane + oi] = f2bf(o - bf2f(hi));
        __builtin_amdgcn_wave_barrier();
    }
}
int launch_t5_attention(Planes qkv, const float* pos_bias, int pos_len, int B, int L, int heads, int dkv, Planes out, hipStream_t st) {
    if (dkv != T5_DKV || L > T5_LMAX || L > pos_len || L < 1) VB_FAIL(VB_E_INVALID, "t5_attention: d_kv=%d L=%d (built for d_kv 64, L <= 128)", dkv, L);
    hipLaunchKernelGGL(t5_attn_kernel, dim3(heads, B), dim3(256), 0, st, qkv, pos_bias, pos_len, L, heads, out);
    VB_CHECK_LAUNCH();
    return VB_OK;
}
