// Caption-gate scores + Band-MoE router in ONE launch (gfx950, bf16 production mode).
//
// The folded caption gate (engine.hip, vocal2music_moe.py:119-151) was two launches: a grouped GEMM writing every token's attention
// scores against its clip's caption keys for all heads ([N][NS] fp32, NS = L * heads = 640: 30.8 MB per block evaluation at 8 clips)
// and the router kernel reading them back (softmax per head, contraction with the per-clip VW, Gumbel arg-max).  Here a workgroup
// owns 64 tokens of one clip and ALL NS score columns, so the scores never leave the CU:
//   * mainloop: [64 x K] x [NS x K]^T on the bf16 MFMA pipe, 4 waves, wave w owns all 64 rows x columns [w NS/4, (w+1) NS/4)
//     (2 x NJ accumulator tiles of 32 x 32); BK = 32, 3-stage LDS ring filled by global_load_lds (one stage = 64 + NS rows of 64 B =
//     45 KB at NS = 640), counted vmcnt, same lane-linear image + source-side XOR swizzle and the same ascending order of 16-deep
//     k-steps as gemm_bf16_glds_kernel - the scores are bit-identical to the two-launch path;
//   * epilogue: scores + per-clip bias go to LDS 32 rows at a time (the ring is free after the loop), and every wave runs
//     router_tokens() - the router kernel's own code - on 8 of those rows, reading the score rows from LDS instead of HBM.
// Grid: clips x ceil(T / 64) workgroups, the tiles of a clip on one XCD (its 983 KB of folded keys stay in that L2).
#include <type_traits>

#include "kernels.h"
#include "router_dev.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int N> __device__ __forceinline__ void sr_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}
// byte offset of 16-B chunk c of tile row `row` in a BK = 32 operand image (64 B per row; chunk XOR-swizzled by the row)
__device__ __forceinline__ int sr_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

#define SR_BM 64
#define SR_NST 3

struct ScoreRouterDev {
    const bf16_t* A; int lda;                               // token features [N][lda] (bf16)
    const bf16_t* Bm; int64_t b_clip_stride; int ldb;       // folded keys per clip: [Beff][NS][ldb]
    const float* bias; int bias_clip_stride;                // [Beff][NS]
    int K, Beff, tiles_per_clip;
    unsigned long long* trace;                              // tuning only (vbdbg_sr_trace): per block {start, loop end, end}
    RouterDev r;
};
static unsigned long long* g_sr_trace = nullptr;
extern "C" void vbdbg_sr_trace(void* buf) { g_sr_trace = static_cast<unsigned long long*>(buf); }   // tuning tool hook, not ABI

template <int NJ, int PP>
__global__ void __launch_bounds__(256) score_router_kernel(const ScoreRouterDev p) {
    constexpr int NS = NJ * 128;
    constexpr int ROWS = SR_BM + NS;                  // tile rows of one stage: 64 token rows, then NS key rows
    constexpr int STAGE = ROWS * 64;                  // bytes (32 bf16 per row)
    constexpr int PPW = ROWS / 16 / 4;                // 1-KB DMA pieces (16 rows) per wave per stage
    static_assert(ROWS % 64 == 0, "pieces must divide over 4 waves");
    constexpr int PITCH = NS + 4;                     // floats per staged score row
    static_assert(32 * PITCH * 4 <= SR_NST * STAGE, "score staging must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char sr_lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Lb = blockIdx.x, jx = Lb >> 3;
    const int clip = (jx / p.tiles_per_clip) * 8 + (Lb & 7);
    if (clip >= p.Beff) return;
    const int tile = jx % p.tiles_per_clip;
    const int T = p.r.T;
    const int row0 = clip * T + tile * SR_BM;
    const int rows_end = (clip + 1) * T;

    unsigned long long tq0 = 0, tq1 = 0;
    if (p.trace) tq0 = __builtin_amdgcn_s_memtime();
    const bf16_t* src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int s = wave * PPW + i;
        const int r = 16 * s + (lane >> 2);           // stage row
        const int cs = lane & 3;
        if (r < SR_BM) {
            const int c = cs ^ ((r >> 2) & 3);
            int slot = row0 + r;
            if (slot >= rows_end) slot = row0;
            src[i] = p.A + (int64_t)slot * p.lda + c * 8;
        } else {
            const int n = r - SR_BM;
            const int c = cs ^ ((n >> 2) & 3);
            src[i] = p.Bm + (int64_t)clip * p.b_clip_stride + (int64_t)n * p.ldb + c * 8;
        }
    }
    auto issue = [&](int t) {
        unsigned char* dst = sr_lds + (t % SR_NST) * STAGE + wave * PPW * 1024;
        const int k0 = t * 32;
#pragma unroll
        for (int i = 0; i < PPW; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[i] + k0), (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
    };

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = p.K / 32;
#pragma unroll
    for (int t = 0; t < SR_NST - 1; ++t)
        if (t < KT) issue(t);
    const int frow = lane & 31, fk = lane >> 5;
    for (int t = 0; t < KT; ++t) {
        if (t + 1 < KT) sr_wait_vmcnt<PPW>();         // tile t landed, tile t + 1 may stay in flight
        else sr_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                 // ... everywhere; everyone is done reading stage (t - 1) % NST
        if (t + SR_NST - 1 < KT) issue(t + SR_NST - 1);
        const unsigned char* As = sr_lds + (t % SR_NST) * STAGE;
        const unsigned char* Bs = As + SR_BM * 64;
        bf16x8 af[2][2], bf[2][NJ];
        auto fload = [&](int ks, int slot) {
            const int c = ks * 2 + fk;
#pragma unroll
            for (int i = 0; i < 2; ++i) af[slot][i] = *reinterpret_cast<const bf16x8*>(As + sr_off(i * 32 + frow, c));
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(Bs + sr_off(wave * 32 * NJ + j * 32 + frow, c));
        };
        fload(0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks + 1 < 2) fload(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks & 1][j], af[ks & 1][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (p.trace) tq1 = __builtin_amdgcn_s_memtime();
    // ---- scores (+ bias) -> LDS, 32 token rows at a time; then the router on those rows
    float* stg = reinterpret_cast<float*>(sr_lds);
    const float* bias = p.bias + (int64_t)clip * p.bias_clip_stride;
    // (two explicit instances: a loop over `half` is not unrolled around the inlined router and would index acc[] dynamically -> scratch)
    auto do_half = [&](auto half_c) __attribute__((always_inline)) {
        constexpr int half = decltype(half_c)::value;
        __syncthreads();                              // ring reads (half 0) / the previous half's router reads are done
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wave * 32 * NJ + j * 32 + q * 8 + fk * 4;
                const float4 b4 = *reinterpret_cast<const float4*>(bias + col);
                float4 v;
                v.x = acc[half][j][q * 4 + 0] + b4.x; v.y = acc[half][j][q * 4 + 1] + b4.y;
                v.z = acc[half][j][q * 4 + 2] + b4.z; v.w = acc[half][j][q * 4 + 3] + b4.w;
                *reinterpret_cast<float4*>(stg + frow * PITCH + col) = v;
            }
        __syncthreads();
#pragma unroll 1
        for (int gi = 0; gi < 2; ++gi) {
            const int lr = wave * 8 + gi * 4;         // first of this wave's 4 rows in the staged half
            const int n0 = row0 + half * 32 + lr;
            if (n0 < rows_end) router_tokens<PP, true, 4>(p.r, n0, rows_end, stg + lr * PITCH, PITCH, nullptr);
        }
    };
    do_half(std::integral_constant<int, 0>());
    do_half(std::integral_constant<int, 1>());
    if (p.trace && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* tr = p.trace + (size_t)blockIdx.x * 4;
        tr[0] = tq0; tr[1] = tq1; tr[2] = __builtin_amdgcn_s_memtime(); tr[3] = 1;
    }
}

bool score_router_supported(int NS, int K, int E, int Hh) {
    const int nj = NS / 128;
    return NS % 128 == 0 && nj >= 4 && nj <= 6 && K % 32 == 0 && K >= 64 && E >= 2 && 2 * E + 2 <= 32 && Hh >= 1 && Hh <= 64 && (Hh & (Hh - 1)) == 0;
}

template <int NJ, int PP>
static void launch_sr(const ScoreRouterDev& d, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)SR_NST * (SR_BM + NJ * 128) * 64;
    static OnceFlags attr;
    vb_set_max_lds_once(attr, reinterpret_cast<const void*>(score_router_kernel<NJ, PP>), (int)lds);
    hipLaunchKernelGGL((score_router_kernel<NJ, PP>), grid, dim3(256), lds, st, d);
}

int launch_score_router(const ScoreRouterArgs& a, hipStream_t st) {
    if (!score_router_supported(a.NS, a.K, a.E, a.Hh)) VB_FAIL(VB_E_INVALID, "score_router: NS=%d K=%d E=%d heads=%d unsupported", a.NS, a.K, a.E, a.Hh);
    if (a.lda % 8 || a.ldb % 8 || a.T < 1 || a.Beff < 1) VB_FAIL(VB_E_INVALID, "score_router: lda/ldb must be multiples of 8, T and clips positive");
    ScoreRouterDev d;
    d.A = a.A; d.lda = a.lda; d.Bm = a.Bm; d.b_clip_stride = (int64_t)a.NS * a.ldb; d.ldb = a.ldb; d.bias = a.bias; d.bias_clip_stride = a.NS;
    d.K = a.K; d.Beff = a.Beff; d.tiles_per_clip = cdiv(a.T, SR_BM); d.trace = g_sr_trace;
    RouterDev& r = d.r;
    r.cq = Planes{nullptr, 0, 1}; r.Wg = a.vw; r.bg = a.bg; r.la = a.la; r.la_rows = a.la_rows; r.hl = a.hl; r.hl_ld = a.hl_ld;
    r.g1 = a.g1; r.g2 = a.g2; r.g3 = a.g3; r.N = a.Beff * a.T; r.T = a.T; r.D = a.K; r.E = a.E; r.ic = a.ic; r.ia = a.ia; r.mc = a.mc; r.ma = a.ma;
    r.lc_out = nullptr; r.B = a.B > 0 ? a.B : 1; r.seed = a.seed; r.clip_base = a.clip_base; r.nfe_base = a.nfe_base; r.step = a.step;
    r.block = a.block; r.sc = nullptr; r.NS = a.NS; r.Hh = a.Hh; r.cnt = nullptr; r.cnt_G = 0; r.cnt_pairs = 0;
    const double n_tok = (double)a.Beff * a.T;
    ProfScope prof(0, 2.0 * n_tok * a.NS * a.K, 2.0 * (n_tok * a.K + (double)a.Beff * a.NS * a.K) + 16.0 * n_tok, st);
    const dim3 grid(d.tiles_per_clip * ((a.Beff + 7) / 8 * 8));
    const int nj = a.NS / 128;
    const bool pp4 = 2 * a.E + 2 <= 16;
#define VB_SR_CASE(NJ) case NJ: if (pp4) launch_sr<NJ, 4>(d, grid, st); else launch_sr<NJ, 2>(d, grid, st); break;
    switch (nj) {
        VB_SR_CASE(4)
        VB_SR_CASE(5)
        VB_SR_CASE(6)
        default: VB_FAIL(VB_E_INVALID, "score_router: NS=%d", a.NS);
    }
#undef VB_SR_CASE
    VB_CHECK_LAUNCH();
    return VB_OK;
}
