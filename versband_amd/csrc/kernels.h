// Internal launch API shared by the engines and the C-ABI wrappers.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------
// bf16 MFMA GEMM   C[m][n] = sum_k A[m][k] * B[n][k]   (both operands K-contiguous)
// ---------------------------------------------------------------------------
enum GemmEpi {
    EPI_PLANES = 0,          // out planes[m][coff+n] = v (+bias)
    EPI_F32 = 1,             // out32[m][coff+n] = v (+bias)
    EPI_QKV_ROPE = 2,        // q,k planes with RoPE; v written transposed per head
    EPI_RESID_GATE = 3,      // out32[m][coff+n] += gate[b(m)][coff+n] * v
    EPI_SWIGLU = 4,          // interleaved (w1,w3) columns -> planes[m][coff + n/2] = silu(v0)*v1
    EPI_SCATTER_F32 = 5,     // out32[rows_out[m]][n] = row_scale[rows_out[m]] * v
    EPI_SCATTER_ADD_PLANES = 6,  // planes[tok][n] = y32_in[tok][n] + row_scale[tok]*v
    EPI_GELU_PLANES = 7,     // planes = gelu_erf(v + bias)
    EPI_HEADS_T = 8,         // planes[((b*H+h)*hd+d)*Tpad + t] = v (+bias),  m = b*T+t, n = h*hd+d
    EPI_GEGLU = 10,          // planes[m][n/2] = gelu_new(v[n]) * v[n+1]  (T5 gated-GELU FFN, rows of wi_0 / wi_1 interleaved)
    EPI_F32_CT = 9,          // out32[(b*N + n)*T + t] = v + bias  (channel-major [B][N][T] output of the FinalLayer), m = b*T+t
    EPI_COUNT = 9
};

struct GemmArgs {
    const bf16_t* A = nullptr; int64_t a_plane = 0; int lda = 0;
    const int* a_rows = nullptr;      // optional row gather (slot -> token)
    int a_koff_group = 0;             // A column offset per group index
    const bf16_t* B = nullptr; int64_t b_plane = 0; int ldb = 0; int64_t b_group_stride = 0;
    int M = 0, N = 0, K = 0;
    int nseg = 1;                     // 1 = plain bf16, 3 = bf16x3 split precision
    int ngroups = 1; const int* group_off = nullptr;   // device [ngroups+1] slot offsets (null: one group [0,M))
    int group_rows = 0;               // > 0 (group_off optional then): every group has exactly this many rows (group g = rows [g R, (g+1) R): per-clip
                                      // operands) - the 128 x 128 kernel then keeps all tiles of a group on ONE XCD (group g -> XCD g % 8), so
                                      // a group's B operand is fetched into one L2 instead of all eight
    int c_noff_group = 0;             // output column offset per group
    int epi = EPI_F32;
    const float* bias = nullptr; int64_t bias_group_stride = 0;
    Planes out = {nullptr, 0, 1}; int ldc = 0;
    float* out32 = nullptr; int ldc32 = 0;
    const float* gate = nullptr; int gate_ld = 0; int T = 1;
    const int* rows_out = nullptr; const float* row_scale = nullptr; const float* y32_in = nullptr;
    // EPI_SWIGLU only: per-row gate weight folded into the hidden value BEFORE its bf16 rounding (rows < scale_split take
    // row_scale[a_rows[m]], the others row_scale2[a_rows[m]]): lets the routed w2 product run as one plain K-concatenated GEMM
    const float* row_scale2 = nullptr; int scale_split = 0;
    // EPI_F32 only: out32[m][n] = v + bias + add32[m][n] (same leading dimension), written to row m AND, when dup_rows > 0, to row
    // m + dup_rows (both CFG branches start from the same embedded latent)
    const float* add32 = nullptr; int dup_rows = 0;
    // Conv1d as a GEMM over pre-activated transposed planes (EPI_F32_CT with group_rows = T_out): K = taps * conv_ci walks tap by tap -
    // the A rows of tap j are the rows of tap 0 shifted by j * conv_dil, its B operand sits conv_btap elements behind tap j-1's; group g
    // (a clip) starts at A row g * conv_agrp + conv_arow0; res32 (same [b][n][t] layout as the output) is added in the epilogue.
    int conv_ci = 0, conv_dil = 1, conv_agrp = 0, conv_arow0 = 0; int64_t conv_btap = 0; const float* res32 = nullptr;
    int prof_class = 0;               // kernel class of the HIP-event profiler this launch is counted under (2 = split-bf16 convolution work)
    Planes q = {nullptr, 0, 1}, k = {nullptr, 0, 1}, vt = {nullptr, 0, 1};
    const float* rope_cos = nullptr; const float* rope_sin = nullptr;
    int H = 1, hd = 1, Tpad = 0, D = 0;
};
int launch_gemm(const GemmArgs& a, hipStream_t st);
// routed experts, second product of BOTH groups in one launch over (caption, acoustic) pair buckets (bf16, E <= 4; moe_w2_pair_kernel):
// out[tok][:] = Hs[caption slot] W2[c]^T + Hs[acoustic slot] W2[E + a]^T  with the gate weights m_c / m_a already folded into the
// hidden rows by the SwiGLU epilogue (GemmArgs::row_scale / row_scale2): one accumulator, K = 2H
struct MoeW2PairArgs {
    const bf16_t* Hs = nullptr; const bf16_t* W2 = nullptr;     // [2N][H] bf16 slot order (gate-scaled); [2E][D][H]
    const int* pair_off = nullptr; const int* perm = nullptr; const int* pair_pa = nullptr;        // launch_bucket pair-mode outputs
    bf16_t* out = nullptr;                                      // out [N][D] bf16
    int N = 0, D = 0, H = 0, E = 0;
};
int launch_moe_w2_pair(const MoeW2PairArgs& a, hipStream_t st);
// fused band-expert FFN (bf16): out32[:, e*band .. ] += gate * ( W2_e . (silu(W1_e y_e) * (W3_e y_e)) ), y_e = y[:, e*band ..]
struct BandFfnArgs {
    const bf16_t* y = nullptr; int ldy = 0;          // [M][ldy] bf16 (one plane)
    const bf16_t* w13 = nullptr; const bf16_t* w2 = nullptr;   // [E][2H][band] interleaved w1/w3 rows, [E][band][H]
    int M = 0, H = 0, E = 0, band = 0;
    float* out32 = nullptr; int ldc32 = 0; const float* gate = nullptr; int gate_ld = 0; int T = 1;
};
int launch_band_ffn(const BandFfnArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// flash attention over bf16 planes (head_dim 96): out = softmax(q k^T s) v  [+ w_h * softmax(q ky^T s) vy]
// ---------------------------------------------------------------------------
struct AttnArgs {
    Planes q;            // [B*T][H*hd]
    Planes k;            // [B*T][H*hd]       (self; may be null when !has_self)
    Planes vt;           // [B][H][hd][Tpad]
    Planes ky;           // [B*L][H*hd]       (cross; may be null)
    Planes vyt;          // [B][H][hd][Lpad]
    const float* cross_w;  // [H] per-head weight of the cross term (null -> 1)
    Planes out;          // [B*T][H*hd]
    int B, T, Tpad, L, Lpad, H, hd;
    int has_self, has_cross;
    int kv_batch_mod;    // cross K/V batch index = b % kv_batch_mod (0: = b)
    float scale;
};
int launch_attention(const AttnArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// fp32 MFMA implicit-GEMM Conv1d   out[b][co][t] over x[b][ci][t]
// ---------------------------------------------------------------------------
enum ConvAct { ACT_NONE = 0, ACT_LRELU = 1, ACT_GN_SWISH = 2, ACT_TANH = 3, ACT_GN = 4 };
struct ConvArgs {
    const float* x = nullptr; int64_t x_bstride = 0; int Ci = 0; int T_in = 0; int x_bmod = 0;
    const float* w = nullptr;        // packed [phase][tap][Ci][Co]
    int64_t w_bstride = 0;           // per-batch weights (VAE attention), 0 = shared
    const float* bias = nullptr;
    int Co = 0, ksize = 1, dil = 1, pad = 0;
    int upsample2 = 0;               // nearest x2 on the input (T_in is the un-upsampled length)
    int in_stride = 1, in_phase = 0; // the convolution reads x[i * in_stride + in_phase] (Downsample1D as two polyphase convs)
    int in_act = ACT_NONE; float in_slope = 0.f;
    const float* gn_mean = nullptr; const float* gn_rstd = nullptr;   // [B][groups]
    const float* gn_gamma = nullptr; const float* gn_beta = nullptr; int gn_groups = 32;
    float* out = nullptr; int64_t out_bstride = 0; int T_out = 0;
    const float* res = nullptr; int64_t res_bstride = 0;   // residual [b][co][t] (same layout as out)
    float alpha = 1.f;               // out = beta*out + alpha*(acc*acc_scale + bias + res)
    float beta = 0.f;
    float acc_scale = 1.f;
    int out_act = ACT_NONE; float out_slope = 0.f;
    int out_transposed = 0;          // out[b][t][co] (+ add[b % add_bmod][t][co])
    const float* add = nullptr; int64_t add_bstride = 0; int add_bmod = 0;
    int B = 1;
    // transposed convolution (polyphase): stride u > 1
    int tr_stride = 1; int tr_pad = 0; int tr_k = 0;
    // optional split-bf16 weights [2][phase][tap][Co][Ci_pad] (Ci_pad % 32 == 0): selects the bf16x3 MFMA kernel
    const bf16_t* wp = nullptr; int64_t wp_plane = 0; int Ci_pad = 0;
    int64_t wp_bstride = 0;          // per-batch split weights (bf16 elements inside a plane), VAE attention
    // optional pre-activated transposed split planes of the input (xt_planes_kernel): [2][B][xt_rows(T_in)][Ci]; x / in_act /
    // GroupNorm fields are then ignored and the window is DMA'd straight into LDS
    const bf16_t* xt = nullptr;
    // optional minimal-filtering weights [P][Ci][Co] fp32 (pack.py:pack_conv_mf): selects conv1d_f32w_kernel (fp32 products, F(2,3)) where it applies
    const float* w_mf = nullptr;
};
int launch_conv1d(const ConvArgs& a, hipStream_t st);
// fused HiFi-GAN ResBlock1 pair (respair_x3.hip): out = beta*out + alpha*(x + b2 + conv2(lrelu(b1 + conv1_dil(lrelu(x)))))
struct RespairArgs {
    const float* x = nullptr; float* out = nullptr; int B = 1, C = 0, T = 0, k = 3, dil = 1;
    const bf16_t* w1 = nullptr; const bf16_t* w2 = nullptr;     // split planes [2][k][C][C]
    const float* b1 = nullptr; const float* b2 = nullptr;
    float slope = 0.1f, alpha = 1.f, beta = 0.f;
};
int launch_respair(const RespairArgs& a, hipStream_t st);
// the same pair in exact fp32 (respair_f32.hip; v_mfma_f32_32x32x2_f32, weights fp32 packed [k][Ci][Co]): C = 32 / 64 / 128
struct RespairF32Args {
    const float* x = nullptr; float* out = nullptr; int B = 1, C = 0, T = 0, k = 3, dil = 1;
    const float* w1 = nullptr; const float* w2 = nullptr; const float* b1 = nullptr; const float* b2 = nullptr;
    float slope = 0.1f, alpha = 1.f, beta = 0.f;
};
bool respair_f32_supported(const RespairF32Args& a);
int launch_respair_f32(const RespairF32Args& a, hipStream_t st);
// the same pair with F(2,3) minimal filtering (respair_f32w.hip): w1 / w2 = pseudo-tap weights [P][C][C] (pack.py:pack_conv_mf); C = 32, k = 3 / 7 / 11
bool respair_f32w_supported(const RespairF32Args& a);
int launch_respair_f32w(const RespairF32Args& a, hipStream_t st);

// ---------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------
int launch_rmsnorm_mod(const float* h, const float* w, const float* shift, const float* scale, int mod_ld,
                       int rows, int D, int T, float eps, Planes out, hipStream_t st);
int launch_layernorm(const float* x, const float* w, const float* b, int rows, int D, float eps, float* out32, Planes outp,
                     hipStream_t st);
int launch_cast_planes(const float* x, int64_t n, Planes out, hipStream_t st);
// f32 [rows][cols] -> split-bf16 planes [2][rows][cpad] (cpad % 4 == 0, columns >= cols zero filled)
int launch_silu_sum_planes(const float* temb, const float* cemb, int rows, int D, int nsample, bf16_t* out, int64_t plane, hipStream_t st);
// LayerNorm (no affine, eps) + adaLN modulate -> split-bf16 planes (FinalLayer input, vocal2music_moe.py:287-291)
int launch_layernorm_mod_planes(const float* h, const float* shift, const float* scale, int mod_ld, int rows, int D, int T, float eps,
                                Planes out, hipStream_t st);
// T5 (t5.hip)
int launch_gather_rows(const int64_t* idx, const float* table, int rows, int D, int vocab, float* out, hipStream_t st);
int launch_t5_attention(Planes qkv, const float* pos_bias, int pos_len, int B, int L, int heads, int dkv, Planes out, hipStream_t st);
// transposed split planes for the DMA-fed conv path (see xt_planes_kernel): rows of the padded image
#define XT_HEAD 64
#define XT_TAIL 384
static inline int xt_rows(int T_eff) { return T_eff + XT_HEAD + XT_TAIL; }
int launch_xt_planes(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int groups, int act,
                     float slope, int upsample2, int B, int C, int T_in, bf16_t* out, hipStream_t st);
// log-mel front-end (melnet.hip)
int launch_stft_frames(const float* wav, int B, int L, int hop, int pad, int pad2, int J, float* X, hipStream_t st);
int launch_mel_tail(const float* spec, int B, int T, int Co4, int nb, int im_off, const float* basisT, int n_mels, float* mel, hipStream_t st);
int launch_aa_act(const float* x, const float* alpha, const float* inv_beta, const float* filt, int B, int C, int T, float* out, hipStream_t st);
int launch_gn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int B, int C, int T,
                    int groups, int swish, float* out, hipStream_t st);
int launch_split_rows(const float* x, int64_t rows, int cols, int cpad, bf16_t* out, int64_t plane, hipStream_t st);
int launch_planes_to_f32(Planes in, int64_t n, float* out, hipStream_t st);
int launch_gemv_rows(const float* x, int x_ld, const float* x2, int x2_ld, int x2_mod, const float* W, const float* bias,
                     int R, int N, int K, int act_in, float* out, int out_ld, hipStream_t st);
int launch_gemv_rows_idx(const float* x, int x_ld, const int64_t* idx, const float* x2, int x2_ld, int x2_mod, const float* W,
                         const float* bias, int R, int N, int K, int act_in, float* out, int out_ld, hipStream_t st, int sin_rows = 0);
// (sin_rows > 0: x is the timestep-sinusoid table with that many rows; indices outside it are evaluated on the fly)
int launch_mean_rows(const float* x, int B, int L, int D, float* out, hipStream_t st);
int launch_embed_t(const int64_t* idx, const float* table, int B, int T, int D, int vocab, float* out, hipStream_t st);   // (indices clamped to [0, vocab))
int launch_pool_add(const float* a, const float* b, int B, int C, int T_in, float* out, hipStream_t st);
int launch_transpose_bct_btc(const float* in, int B, int C, int T_in, int T_out, float* out, hipStream_t st);
int launch_final_layer(const float* h, const float* shift, const float* scale, int mod_ld, const float* W, const float* bias,
                       int rows, int D, int T, int C, float eps, float* out, hipStream_t st);
bool final_layer_fused_ok(int D, int C);
int launch_final_layer_fused(const float* h, const float* shift, const float* scale, int mod_ld, const float* W, const float* bias,
                             int rows, int D, int T, int C, float eps, float* out, hipStream_t st);
int launch_final_layer_euler(const float* h, const float* shift, const float* scale, int mod_ld, const float* W, const float* bias,
                             int rows, int D, int T, int C, float eps, float* x, float cfg_scale, const float* dt_table, int k, int* step,
                             int64_t* t_idx_cur, const int64_t* t_table, int n_steps, int Beff, hipStream_t st);
int launch_euler_cfg(float* x, const float* v, int B, int64_t per, float cfg_scale, const float* dt_table, const int* step,
                     float dt_val, int has_uncond, hipStream_t st);
int launch_router(Planes cq, const float* Wg, const float* bg, const float* la, int la_mod_rows, const float* hl, int hl_ld,
                  const float* g1, const float* g2, const float* g3, int N, int T, int D, int E, int* ic, int* ia, float* mc,
                  float* ma, float* lc_out, int B, uint64_t seed, int64_t clip_base, int nfe_base, const int* step, int block,
                  hipStream_t st, const float* sc = nullptr, int NS = 0, int Hh = 1, int* cnt = nullptr, int cnt_G = 0, int cnt_pairs = 0);
// fused caption-gate scores + router (score_router.hip): bf16 token features x per-clip folded keys -> routing decisions, the
// [N][NS] score matrix stays in LDS.  Bit-identical to launch_gemm(EPI_F32 scores) + launch_router(sc = scores).
struct ScoreRouterArgs {
    const bf16_t* A = nullptr; int lda = 0;          // token features [Beff*T][lda] bf16
    const bf16_t* Bm = nullptr; int ldb = 0;         // folded keys [Beff][NS][ldb] bf16
    const float* bias = nullptr;                     // [Beff][NS]
    const float* vw = nullptr;                       // [Beff][NS][E]
    const float* bg = nullptr; const float* la = nullptr; int la_rows = 0; const float* hl = nullptr; int hl_ld = 0;
    const float* g1 = nullptr; const float* g2 = nullptr; const float* g3 = nullptr;
    int Beff = 0, B = 0, T = 0, K = 0, NS = 0, Hh = 1, E = 0;
    int* ic = nullptr; int* ia = nullptr; float* mc = nullptr; float* ma = nullptr;
    uint64_t seed = 0; int64_t clip_base = 0; int nfe_base = 0; const int* step = nullptr; int block = 0;
};
bool score_router_supported(int NS, int K, int E, int Hh);
int launch_score_router(const ScoreRouterArgs& a, hipStream_t st);
int launch_iota_div(int64_t* out, int n, int div, hipStream_t st);
// (pair_off / pair_pa non-null, E*E <= 16: rank by (caption, acoustic) expert PAIR; both expert-group orders derive from it - see
//  bucket_place_kernel.  pair_off [E*E + 1], pair_pa [N] = acoustic slot of the token in pair / caption slot p)
// counts_ready (round 5): the per-256-token-block group counts were already accumulated by the router (RouterDev::cnt) - the count launch is
// skipped; counts_clear: a second table the place kernel zeroes for the next router launch (ping-pong, see bucket_counts)
int launch_bucket(const int* ic, const int* ia, int N, int E, int* group_off, int* perm, hipStream_t st, int* pair_off = nullptr,
                  int* pair_pa = nullptr, const int* counts_ready = nullptr, int* counts_clear = nullptr);
int bucket_scratch_ints(int N, int E);   // perm buffers must hold 2N + this many ints
bool bucket_router_counts_ok(int N);     // the launch takes the two-kernel (count + place) form whose count the router can provide
int* bucket_counts(int* perm, int N, int which);    // count table `which` (0 / 1) in the scratch tail of the perm buffer
int bucket_counts_ints(int N);           // ints of both tables together (to clear them once per call)
int launch_gate_fold(Planes kc, Planes vct, const float* bq_s, const float* wcg, int Beff, int L, int Lpad, int Hh, int hd, int E,
                     float* cbias, float* vw, hipStream_t st);
int launch_iota_mul(int* out, int n, int mul, hipStream_t st);
// proj_in as a GEMM (elementwise.hip): latent windows as K-contiguous split planes, conv weights re-laid as a GEMM operand
int launch_im2col_latent(const float* x, int B, int C, int T, int taps, int pad, int KP, bf16_t* out, int64_t plane, hipStream_t st);
int launch_conv_w_to_gemm(const bf16_t* w3, int64_t w3_plane, int taps, int D, int KP, bf16_t* out, hipStream_t st);
int launch_router_top1(const float* logits, const float* gumbel, int N, int E, int* idx, hipStream_t st);
int launch_fill_gumbel(float* out, int B, int n_branch, int T, int width, uint64_t seed, int64_t clip_base, int nfe_base,
                       const int* step, int block, int gate, hipStream_t st);
int launch_rows_dot(const float* x, const float* W, const float* bias, int N, int D, int E, float* out, hipStream_t st);
int launch_gn_stats(const float* x, int B, int C, int T, int groups, float eps, float* mean, float* rstd, hipStream_t st);
int launch_softmax_rows_t(const float* s, int B, int R, int Ccols, float* out_t, hipStream_t st);
int launch_sampler_params(int* step, uint64_t seed, int64_t clip_base, int nfe_base, hipStream_t st);   // step[4..9]: noise key of the call
int launch_step_ctl(int* step, int64_t* t_idx_cur, const int64_t* t_table, int n_steps, int Beff, int reset, hipStream_t st);
int launch_fill_f32(float* p, int64_t n, float v, hipStream_t st);
int launch_crossfade_windows(const float* parts, const int* starts, int nw, int B, int C, int n, int T, float* out, hipStream_t st);   // starts: HOST array

// ---------------------------------------------------------------------------
// per-kernel-class HIP-event timing (bench.py roofline): 0 = bf16 GEMM (incl. the fused expert kernels), 1 = attention,
// 2 = conv1d (VAE + vocoder), 3 = fused ResBlock pair.  flops / bytes = ALGORITHMIC work of the launch (2 x MACs; operand +
// result bytes touched once), the figures DESIGN.md section 4 states per kernel.
// ---------------------------------------------------------------------------
void prof_start(int cls, double flops, double bytes, hipStream_t st);
void prof_stop(int cls, hipStream_t st);
struct ProfScope {
    int cls; hipStream_t st;
    ProfScope(int c, double flops, double bytes, hipStream_t s) : cls(c), st(s) { prof_start(c, flops, bytes, s); }
    ~ProfScope() { prof_stop(cls, st); }
};
