/* libversband_hip.so - C ABI of the MI355X-native AccompBand inference path.
 *
 * The reference (AaronZ345/VersBand) is pure Python with no native layer; these entry
 * points are what a maintainer binds (ctypes) underneath the reference's
 * own operator API.  Each entry cites the reference interface it replaces
 * (paths relative to the reference tree).
 *
 * Conventions
 *   - every tensor is a caller-owned, contiguous DEVICE pointer (torch `data_ptr()`);
 *     nothing here allocates or frees device memory: scratch is passed in and its size is
 *     queried with the matching *_workspace_bytes();
 *   - every launch goes to the `stream` argument (a hipStream_t passed as void*);
 *   - return value: 0 = ok, <0 = VB_E_*; vb_last_error() returns a thread-local message;
 *   - no exceptions cross the ABI; one host thread per context / GPU
 *     (scripts/test_final.py:475 spawns one process per GPU).
 *
 * "planes" tensors are bf16 with np planes: plane 0 = round-to-nearest bf16 of the value,
 * plane 1 (np == 2 only) = bf16 of the rounding residual, stored numel elements after
 * plane 0.  np == 2 selects the split-precision ("bf16x3") MFMA path used for parity;
 * np == 1 is the plain bf16 production path.
 */
#ifndef VERSBAND_HIP_H
#define VERSBAND_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VB_OK 0
#define VB_E_INVALID (-1)
#define VB_E_HIP (-2)
#define VB_E_STATE (-3)
#define VB_MAX_DEPTH 16

typedef struct vb_ctx vb_ctx;

int vb_ctx_create(int device, vb_ctx** out);
int vb_ctx_destroy(vb_ctx* ctx);
const char* vb_last_error(void);
int vb_abi_version(void);
/* sha256 (hex) over the library's sources, the public header and the compile flags it was built from; versband_amd/_lib.py compares
 * it with the sources on disk so a stale binary is never run (struct layouts may have moved), and loads a prebuilt library as it is
 * when no sources are shipped beside it */
const char* vb_source_digest(void);
/* 1 when built with -DVB_EXPERIMENTS (ablation instances and the measured-slower kernels of DESIGN.md section 5; never shipped) */
int vb_has_experiments(void);

/* HIP-event timing of one kernel class inside a region (bench.py roofline): bit 0 = bf16 GEMM,
 * bit 1 = attention, bit 2 = conv1d, bit 3 = fused ResBlock pair.  Events are recorded on the launch stream around every launch of
 * the enabled classes; vb_prof_read synchronises the device and sums the elapsed times.
 * Bits 8..15 of class_mask = sampling period n of class 0 (time every n-th launch per host thread; 0/1 = every launch), bits 16..19 = the period of
 * classes 1..3 (0 = the same n), bits 20..23 / 24..27 = which launch of a period is timed (class 0 / the others): walking the phases over as many
 * regions times every launch exactly once per cycle without bracketing neighbouring launches.
 * vb_prof_read: ms_sum, flops and bytes (algorithmic work of those launches) cover the `timed` launches only; `launches` counts all launches of the class. */
/* Tuning / A-B knobs (environment variables, read once per process; a tool or test that changes one at run time calls this
 * afterwards).  Not needed by a product caller.  Kernel selection, results bit-identical: VB_GEMM_TILE (22|33|24|42|11|21),
 * VB_GEMM_SMALL / VB_GEMM_SMALL_TILES (64x64 tiles below that many 128x128 tiles), VB_GEMM_VARIANT, VB_GEMM_NCHUNK, VB_GEMM_P8*,
 * VB_ROUTER_TPW (tokens per wave), VB_BAND_UNFUSED, VB_MOE_UNFUSED, VB_SCORE_FUSED, VB_CONV_CFG, VB_CONV_DIRECT_EPI, VB_NO_GRAPH,
 * VB_BUCKET_COUNT_LAUNCH (bucket counts by their own launch instead of the router's side counts), VB_EULER_LAUNCH (FinalLayer / Euler update /
 * step advance as three launches per step instead of one).
 * Changing the arithmetic (same algorithm, low-order bits differ): VB_ATTN_DEFER (log2 threshold of the deferred softmax rescale,
 * default 8, 0 = exact running maximum), VB_STEM_F32, VB_GATE_UNFOLDED.  Timing-only ablations (results WRONG): VB_GEMM_ABLATE,
 * VB_CONV_ABLATE, VB_ATTN_ABLATE. */
void vb_tune_reload(void);
int vb_prof_enable(int class_mask);
int vb_prof_read(int cls, double* ms_sum, double* flops, double* bytes, int64_t* launches, int64_t* timed);

/* ---------------------------------------------------------------- DiT + Band-MoE ----
 * TxtFlagLargeImprovedDiTV2 (ldm/modules/diffusionmodules/vocal2music_moe.py:293-520),
 * constructor params of configs/vocal2music.yaml:36-43. */
typedef struct {
    int in_channels, hidden, heads, depth, num_experts, ffn_hidden, context_dim, ori_dim, max_len;
    int np;            /* 1 = bf16, 2 = split precision (parity mode) */
    float norm_eps;
} vb_dit_config;

typedef struct {
    /* bf16 planes (np planes each, plane stride = numel) */
    const void* wqkv;   /* [3D][D]  rows wq|wk|wv              flag_large_dit_moe.py:173-181 */
    const void* wo;     /* [D][D]                               :193 */
    const void* wq_m;   /* MoE.cross_attention in_proj rows 0:D vocal2music_moe.py:79 */
    const void* wo_m;   /* MoE.cross_attention.out_proj (kept for reference; folded into wcg/bcg, no GEMM runs on it) */
    const void* w13;    /* [2E][2H][D] routed experts, rows interleaved w1_0,w3_0,w1_1,...  (caption group first) */
    const void* w2;     /* [2E][D][H] */
    const void* w13f;   /* [E][2H][band]  band experts restricted to their channel band (:171-178) */
    const void* w2f;    /* [E][band][H] */
    const void* wky;    /* attention.wk_y [D][ctx]   (precompute only) */
    const void* wvy;    /* attention.wv_y */
    const void* wk_m;   /* MoE.cross_attention in_proj rows D:2D (precompute only) */
    const void* wv_m;   /* rows 2D:3D */
    const void* wqt_s;  /* optional (2 planes): (in_proj rows 0:D)^T * hd^-1/2, [D in][D out] - folds the MoE q-projection into the
                           per-clip caption keys so the gate scores come from ONE grouped GEMM (precompute only) */
    /* fp32 */
    const float* bq_m; const float* bo_m; const float* bk_m; const float* bv_m;
    const float* bq_s;               /* bq_m * hd^-1/2 (with wqt_s) */
    const float* attn_norm_w; const float* ffn_norm_w; const float* y_norm_w;
    const float* cross_w;            /* tanh(attention.gate) [heads]   :401 */
    const float* wcg; const float* bcg;   /* caption_gating_network with cross_attention.out_proj folded in: Wg*Wo [E][D], Wg*bo+bg [E] */
    const float* wag; const float* bag;   /* acoustic_gating_network (precompute only) */
} vb_dit_block_weights;

typedef struct {
    vb_dit_block_weights blocks[VB_MAX_DEPTH];
    const float* t_freq_table;       /* [1000][256] sinusoid table (TimestepEmbedder.timestep_embedding) */
    const float* t_mlp0_w; const float* t_mlp0_b; const float* t_mlp2_w; const float* t_mlp2_b;
    const float* adaln_w; const float* adaln_b;   /* stacked [depth*6D + 2D][D]: blocks' adaLN then final_layer's */
    const void* adaln_wp;                         /* optional: the same matrix as split-bf16 planes [2][rows][D]; lets the sampler tabulate
                                                     all steps' modulations with one bf16x3 MFMA GEMM instead of an fp32 row-linear */
    const float* hl_w; const float* hl_b;         /* stacked high_level_gating_network [depth*2][D] */
    const float* proj_in_w; const float* proj_in_b;  /* conv packed [5][C][D] */
    const void* proj_in_w3;                          /* same weights as split-bf16 planes [2][5][D][32] (bf16x3 conv kernel) */
    const float* final_w; const float* final_b;      /* final_layer.linear [C][D] */
    const void* final_wp;                            /* optional: final_layer.linear as split-bf16 planes [2][C][D] (FinalLayer on the MFMA GEMM) */
    const float* rope_cos; const float* rope_sin;    /* [max_len][hd/2]  precompute_freqs_cis */
    /* precompute only */
    const float* midi_emb; const float* beats_emb;
    const float* midi_conv_w; const float* midi_conv_b;    /* packed [5][D][D] */
    const float* beats_conv_w; const float* beats_conv_b;
    const float* final_proj_w; const float* final_proj_b;  /* packed [1][D][D] */
    const void* midi_conv_w3; const void* beats_conv_w3; const void* final_proj_w3;   /* optional: the three stem convolutions' weights as
                                                              split-bf16 planes [2][taps][D][D] (bf16x3 conv kernel instead of exact fp32) */
    const void* c_emb0; const float* c_emb0_b;             /* planes(2) [D][ori] */
    const void* c_emb2; const float* c_emb2_b;             /* planes(2) [D][D] */
    const float* c_ln_w; const float* c_ln_b;
    const float* cap_ln_w; const float* cap_ln_b;
    const float* cap_lin_w; const float* cap_lin_b;
} vb_dit_weights;

/* replaces: model construction + load_state_dict (scripts/test_final.py:140-151).  The
 * structs are copied; the device tensors they point to stay owned by the caller. */
int vb_dit_load(vb_ctx* ctx, const vb_dit_config* cfg, const vb_dit_weights* w);

size_t vb_dit_cond_bytes(const vb_dit_config* cfg, int B, int n_branch, int T, int L);
size_t vb_dit_workspace_bytes(const vb_dit_config* cfg, int B, int n_branch, int T, int L);

/* Everything of TxtFlagLargeDiT.forward that does not depend on (x,t): acoustic stem
 * (vocal2music_moe.py:388-393), caption embedding + pooled embedding (:407-413), per block
 * context K/V of Attention and MoE.cross_attention and the acoustic gate logits.
 *   t5    f32 [n_branch*B][L][ori_dim]   (cond rows, then uncond rows)
 *   midi, beats  int64 [B][T_mel]        (context['c_concat'], :384-385)
 *   cond  caller buffer of vb_dit_cond_bytes() */
int vb_dit_precompute_cond(vb_ctx* ctx, const float* t5, const int64_t* midi, const int64_t* beats, int B, int n_branch,
                           int T, int T_mel, int L, void* cond, void* ws, void* stream);

/* Injected Gumbel noise (parity path): g = -log(Exp(1)) draws, rows = n_branch*B*T tokens,
 * per block: g1 [rows][2], g2 [rows][E], g3 [rows][E] (order of MoE.forward :134,150,151).
 * Layout [depth][rows][w].  NULL => the engine draws its own (counter based, keyed by
 * (seed, global clip, nfe, block, gate)). */
typedef struct {
    const float* g1; const float* g2; const float* g3;
    uint64_t seed; int64_t clip_base; int nfe;
} vb_noise;

/* One network evaluation for the cond and uncond branches together:
 * replaces the two apply_model() calls of Wrapper_cfg.forward (ldm/models/diffusion/cfm1_audio.py:158-159)
 * -> DiffusionWrapper.forward (ddpm.py:1418-1436) -> TxtFlagLargeDiT.forward.
 *   x      f32 [B][C][T]        (shared by both branches)
 *   t_idx  int64 [n_branch*B]   integer diffusion index (cfm1_audio.py:156)
 *   v_out  f32 [n_branch*B][C][T]
 *   route_out  optional int32 [depth][2][rows] routed expert indices (caption, acoustic) */
int vb_dit_forward(vb_ctx* ctx, const float* x, const int64_t* t_idx, const void* cond, const vb_noise* noise, int B,
                   int n_branch, int T, int L, float* v_out, int32_t* route_out, void* ws, void* stream);

/* x += dt * (v_u + s (v_c - v_u))  - Wrapper_cfg.forward :160 + torchdyn fixed-step Euler */
int vb_euler_cfg_step(float* x, const float* v, int B, int64_t per_item, float cfg_scale, float dt, int has_uncond, void* stream);

/* CFMSampler.sample_cfg (ldm/models/diffusion/cfm1_audio_sampler.py:87-116): n_steps Euler steps.
 *   t_idx_table int64[n_steps], dt_table f32[n_steps] : host OR device arrays (copied with hipMemcpyDefault on the stream; host arrays must outlive that copy)
 *   noise: NULL or per-step injected noise laid out [step][depth][rows][w]
 *   traj  optional f32 [n_steps+1][B][C][T] */
int vb_sample_cfg(vb_ctx* ctx, float* x, const void* cond, int B, int n_branch, int T, int L, int n_steps,
                  const int64_t* t_idx_table, const float* dt_table, float cfg_scale, const vb_noise* noise, float* traj,
                  void* ws, void* stream);
/* The step loop of vb_sample_cfg is captured into a hipGraph the second time a call arrives with the same buffers / shape on a
 * capturable (non-default) stream and replayed from then on (noise key via device memory: any seed / clip base replays).
 * Number of instantiated graphs this context holds (0 = every call so far ran eagerly): */
int vb_sample_graphs(vb_ctx* ctx);

/* ---------------------------------------------------- conv nets (VAE decoder, HiFi-GAN) ----
 * AutoencoderKL.decode (ldm/models/autoencoder1d.py:55-58, Decoder1D :480-512) and
 * HifiGanGenerator.forward (vocoder/hifigan/modules/hifigan.py:126-143) both run as a flat op
 * list over fp32 [B][C][T] buffers; the list is built from the model config by the host side. */
typedef struct { int channels; int tmul; int square; } vb_buf_desc;   /* square: 1 = [T*tmul]^2 floats, 2 = channels x roundup(T*tmul, 32),
                                                                         3 = channels x (T*tmul + 448) (transposed planes with halo) */
/* VB_OP_SPLIT_PLANES: x (f32 [B][rows][cols], rows = Co, cols = Ci, -1 = the buffer's time length) -> out = split-bf16
 * planes [2][B][rows][roundup(cols, 32)]; a later VB_OP_CONV with w_buf = that buffer and ci_pad = -1 uses them as
 * per-batch weights on the bf16x3 kernel (the VAE decoder's single-head attention) */
/* VB_OP_RESPAIR: fused HiFi-GAN ResBlock1 pair (vocoder/hifigan/modules/hifigan.py ResBlock1.forward), Ci = Co = 32 or 64:
 * out = beta*out + alpha*(x + bias2 + conv2_k( lrelu( bias + conv1_{k,dil}( lrelu(x) ) ) )), w_x3 / w2_x3 split planes */
/* VB_OP_GN_APPLY: out = GroupNorm affine of x from `stats` (+ swish when in_act == VB_ACT_GN_SWISH), same layout */
/* VB_OP_AA_ACT: BigVGAN anti-aliased Snake / SnakeBeta (vocoder/bigvgan/alias_free_torch/act.py): out = down2(snake(up2(x))), Ci channels,
 * gn_gamma = alpha (exp'ed when log-scale), gn_beta = 1 / (beta + 1e-9), w = the 12-tap Kaiser-sinc filter */
/* VB_OP_XT_PLANES: x (f32 [B][Ci][T]) -> out = pre-activated (in_act with `stats` / gn_gamma / gn_beta / in_slope), nearest-x2
 * upsampled (upsample2), split-bf16 TRANSPOSED planes with zero halo rows; a VB_OP_CONV with x_planes = 1 reads them by DMA */
enum { VB_OP_CONV = 0, VB_OP_GN_STATS = 1, VB_OP_SOFTMAX_T = 2, VB_OP_SPLIT_PLANES = 3, VB_OP_RESPAIR = 4, VB_OP_GN_APPLY = 5, VB_OP_AA_ACT = 6,
       VB_OP_XT_PLANES = 7 };
enum { VB_ACT_NONE = 0, VB_ACT_LRELU = 1, VB_ACT_GN_SWISH = 2, VB_ACT_TANH = 3, VB_ACT_GN = 4 };
#define VB_BUF_INPUT (-2)
#define VB_BUF_OUTPUT (-3)
typedef struct {
    int kind;
    int x, out, res, stats, w_buf;        /* buffer ids, -1 = none */
    const float* w; const float* bias; const float* gn_gamma; const float* gn_beta;
    int Ci, Co, ksize, dil, pad, upsample2, in_act, out_act, out_transposed, tr_stride, tr_pad, tr_k, gn_groups;
    float in_slope, out_slope, alpha, beta, acc_scale;
    const void* w_x3; int ci_pad;   /* optional split-bf16 weights [2][phase][tap][Co][ci_pad]: selects the bf16x3 MFMA conv kernel */
    const void* w2_x3; const float* bias2;   /* VB_OP_RESPAIR: second convolution (w_x3 == NULL: exact-fp32 pair, w and w2_x3 are the fp32
                                              * packed [k][Ci][Co] weights of the two convolutions; channels 32 / 64 / 128; ci_pad == -2: both are
                                              * minimal-filtering pseudo-tap weights [P][C][C] and the pair runs respair_f32w_kernel, C = 32).
                                              * VB_OP_CONV with w_x3 == NULL: optional fp32 minimal-filtering weights [P][Ci][Co] of the
                                              * same filter (versband_amd/pack.py:pack_conv_mf) - the layer then runs conv1d_f32w_kernel
                                              * (fp32 products, F(2,3): ~1.4-1.5x fewer of them) where its conditions hold */
    int in_stride, in_phase;                 /* VB_OP_CONV: the convolution reads x[i*in_stride + in_phase] (0/1 = plain) */
    int x_planes;                            /* VB_OP_CONV: x is a VB_OP_XT_PLANES buffer (upsample2 then describes how it was made) */
} vb_net_op;

enum { VB_NET_VAE = 0, VB_NET_VOCODER = 1, VB_NET_VAE_ENCODER = 2 };
/* buffer lengths are T * tmul of the base length T passed at run time; in_tmul / out_tmul give the I/O tensors' lengths
 * (decoder: 1 / 2; vocoder: 1 / hop; encoder: 2 / 1 with T = the latent length) */
int vb_net_load(vb_ctx* ctx, int which, const vb_net_op* ops, int n_ops, const vb_buf_desc* bufs, int n_bufs, int in_channels,
                int out_channels, int in_tmul, int out_tmul);
size_t vb_net_workspace_bytes(vb_ctx* ctx, int which, int B, int T);
/* decode_first_stage (ldm/models/diffusion/ddpm_audio.py:379-392): z [B][C][T] -> mel [B][80][2T] */
int vb_vae_decode(vb_ctx* ctx, const float* z, int B, int T, float* mel, void* ws, void* stream);
/* AutoencoderKL.encode (ldm/models/autoencoder1d.py:49-53, Encoder1D :315-409): mel [B][80][2T] -> moments [B][2*embed][T]
 * (mean = first half of the channels, logvar = second half; DiagonalGaussianDistribution is applied by the caller) */
int vb_vae_encode(vb_ctx* ctx, const float* mel, int B, int T, float* moments, void* ws, void* stream);
/* HifiGAN.spec2wav (vocoder/hifigan/hifigan.py:20-30): mel [B][80][T] -> wav [B][T*hop] */
int vb_hifigan_forward(vb_ctx* ctx, const float* mel, int B, int T, float* wav, void* ws, void* stream);

/* Long-form generation (BASELINE configs[4]; build-defined, the reference stops at max_len = 1500 latent tokens: vocal2music_moe.py:421).
 * vb_crossfade_windows: window results parts [nw*B][C][n] (row = w*B + b; windows of equal length n starting at starts[w], a HOST array,
 * covering [0, T) with overlaps) -> out [B][C][T], linear cross-fade over every overlap, weights normalised to one (ramp weights
 * (u + 1) / (ov + 1) in fp32: equal to torch.linspace(0, 1, ov + 2)[1:-1] up to fp32 rounding of the ramp, ~1 ulp, not bit for bit).
 * vb_hifigan_forward_chunked: HifiGAN.spec2wav over a long mel in chunks of `chunk` frames with `halo` frames of context on both sides -
 * identical to whole-clip vocoding when halo >= the generator's receptive field; scratch_in >= B*in_ch*(chunk + 2*halo) floats,
 * scratch_out >= B*out_ch*(chunk + 2*halo)*hop floats, ws = vb_net_workspace_bytes(ctx, VB_NET_VOCODER, B, chunk + 2*halo). */
int vb_crossfade_windows(const float* parts, const int32_t* starts, int nw, int B, int C, int n, int T, float* out, void* stream);
int vb_hifigan_forward_chunked(vb_ctx* ctx, const float* mel, int B, int T, int chunk, int halo, float* wav, void* ws, float* scratch_in,
                               float* scratch_out, void* stream);

/* ------------------------------------------------------------ T5 text encoder ----
 * FrozenTextVocalEmbedder.forward (ldm/modules/encoders/modules.py:216-233): T5EncoderModel(input_ids).last_hidden_state, no
 * attention mask.  HF T5 encoder stack (transformers T5Stack / T5Block / T5Attention / T5DenseGatedActDense / T5LayerNorm):
 * per block  x += O(softmax(Q K^T + rel_pos_bias) V)  on RMS-normed x (no 1/sqrt(d) scaling), x += Wo(gelu_new(Wi0 n) * Wi1 n);
 * final RMS norm.  Runs in split precision (fp32-class) on the same GEMM kernel as the DiT.  The tokenizer stays upstream. */
#define VB_T5_MAX_LAYERS 48
typedef struct { int vocab, d_model, d_kv, heads, d_ff, layers; float eps; } vb_t5_config;
typedef struct {
    const float* ln0;     /* layer.0.layer_norm.weight [d_model] */
    const void* wqkv;     /* planes(2) [3*heads*d_kv][d_model]: q | k | v rows */
    const void* wo;       /* planes(2) [d_model][heads*d_kv] */
    const float* ln1;     /* layer.1.layer_norm.weight */
    const void* wi;       /* planes(2) [2*d_ff][d_model]: rows interleaved wi_0[j], wi_1[j] */
    const void* wo_ff;    /* planes(2) [d_model][d_ff] */
} vb_t5_layer;
typedef struct {
    const float* embed;       /* shared.weight [vocab][d_model] */
    const float* pos_bias;    /* [heads][Lmax][Lmax] relative-position bias of block 0 (shared by all blocks), built by the host */
    int pos_len;              /* Lmax */
    const float* final_ln;    /* encoder.final_layer_norm.weight */
    const float* ones;        /* [d_model] ones (residual "gate") */
    vb_t5_layer layers[VB_T5_MAX_LAYERS];
} vb_t5_weights;
int vb_t5_load(vb_ctx* ctx, const vb_t5_config* cfg, const vb_t5_weights* w);
size_t vb_t5_workspace_bytes(const vb_t5_config* cfg, int B, int L);
/* ids int64 [B][L] -> out f32 [B][L][d_model] */
int vb_t5_encode(vb_ctx* ctx, const int64_t* ids, int B, int L, float* out, void* ws, void* stream);

/* ------------------------------------------------------------ log-mel front-end ----
 * MelNet.forward (preprocess/NAT_mel.py:64-86): clamp to [-1, 1]; reflect-pad (n_fft - hop)/2 samples on both sides (:71; when
 * `center` is set torch.stft reflect-pads the padded signal by n_fft/2 once more); STFT (n_fft, hop; the Hann window - zero-padded to n_fft when win < n_fft - is
 * folded into the basis); sqrt(re^2 + im^2 + 1e-9); mel_basis @ .; log10(max(., 1e-5)).  n_fft must be a multiple of the hop:
 * the windowed DFT of all frames then is one convolution over hop-blocks on the exact-fp32 MFMA convolution kernel.
 *   dft_w   f32 [n_fft/hop][hop][Co4], Co4 = 2*im_off, im_off = roundup(n_fft/2 + 1, 4): column f (< n_fft/2 + 1) = w[n] cos(2 pi f n / n_fft),
 *           column im_off + f = -w[n] sin(2 pi f n / n_fft), n = tap*hop + c; other columns zero
 *   basis_t f32 [n_fft/2 + 1][n_mels]  (the mel filterbank, transposed)
 *   wav f32 [B][L] -> mel f32 [B][n_mels][T] and / or spec f32 [B][T][Co4] (either may be NULL), T = vb_melnet_frames() */
typedef struct { int n_fft, hop, n_mels; } vb_mel_config;
int vb_melnet_load(vb_ctx* ctx, const vb_mel_config* cfg, const float* dft_w, const float* basis_t);
int vb_melnet_frames(const vb_mel_config* cfg, int L, int center);
size_t vb_melnet_workspace_bytes(const vb_mel_config* cfg, int B, int L, int center);
int vb_melnet_forward(vb_ctx* ctx, const float* wav, int B, int L, int center, float* mel, float* spec, void* ws, void* stream);

/* ------------------------------------------------------------------- unit kernels ---- */
/* RMSNorm (flag_large_dit_moe.py:52-77) * w then modulate (:80-81); shift/scale [B][mod_ld] or NULL */
int vb_rmsnorm_modulate(const float* h, const float* w, const float* shift, const float* scale, int mod_ld, int rows, int D,
                        int T, float eps, void* out_planes, int np, void* stream);
/* first argmax of logits + gumbel (hard Gumbel-softmax, vocal2music_moe.py:81-93) */
int vb_router_top1(const float* logits, const float* gumbel, int N, int E, int32_t* idx, void* stream);
/* stable bucketing of tokens by (caption, acoustic) expert: group_off int32[2E+1]; perm int32[2N + scratch] where the
 * first 2N entries are the slot -> token permutation and scratch = vb_route_bucket_scratch_ints(N, E) */
int vb_route_bucket_scratch_ints(int N, int E);
int vb_route_bucket(const int32_t* ic, const int32_t* ia, int N, int E, int32_t* group_off, int32_t* perm, void* stream);
/* the same bucketing ranked by (caption expert, acoustic expert) PAIR (E*E <= 16), what the single-launch routed w2 product reads
 * (vocal2music_moe.py:154-167): pair_off int32[E*E+1] in caption-major order; the caption slots [0,N) ARE the pair slots, the acoustic
 * slots [N,2N) hold the same buckets acoustic-major; pair_pa int32[N] = acoustic slot of the token in pair slot p.  Inside an expert
 * group the rows are ordered by (other expert, token) instead of by token */
int vb_route_bucket_pairs(const int32_t* ic, const int32_t* ia, int N, int E, int32_t* group_off, int32_t* perm, int32_t* pair_off,
                          int32_t* pair_pa, void* stream);
/* generic GEMM  C[M][N] f32 = A[M][K] planes x B[N][K]^T planes (+bias) */
int vb_gemm_bf16(const void* A, const void* Bw, const float* bias, int M, int N, int K, int np, float* C, void* stream);
/* grouped SwiGLU expert FFN (FeedForward, flag_large_dit_moe.py:480-485) over routed buckets:
 *   u planes [Ntok][D]; perm/group_off from vb_route_bucket (G groups); w13 [G][2H][D] interleaved, w2 [G][D][H];
 *   out f32 [Ntok][D] = row_scale[tok] * FFN_g(u[tok]) written at perm order; hidden scratch planes [Nslots][H] */
int vb_grouped_swiglu(const void* u, const int32_t* perm, const int32_t* group_off, int G, int n_slots, const void* w13,
                      const void* w2, const float* row_scale, int D, int H, int np, void* hidden, float* out, void* stream);
/* fused self (+cross) attention, head_dim 96 (Attention.forward, flag_large_dit_moe.py:381-402) */
int vb_attention(const void* q, const void* k, const void* vt, const void* ky, const void* vyt, const float* cross_w, int B, int T,
                 int Tpad, int L, int Lpad, int H, int hd, int np, void* out, void* stream);
/* fp32 Conv1d / ConvTranspose1d on [B][C][T] (weights packed [phase][tap][Ci][Co]); w_x3 != NULL runs the
 * split-bf16 (bf16x3) MFMA kernel on weights packed [2][phase][tap][Co][ci_pad] instead of the exact-f32 MFMA kernel */
int vb_conv1d_f32(const float* x, const float* w, const float* bias, int B, int Ci, int T_in, int Co, int ksize, int dil, int pad,
                  int tr_stride, int tr_pad, int tr_k, int T_out, int in_act, float in_slope, const float* res, float* out,
                  const void* w_x3, int ci_pad, void* stream);
/* Conv1d in fp32 with 1-D minimal filtering (conv1d_f32w.hip; the HiFi-GAN ResBlock convolutions, vocoder/hifigan/modules/hifigan.py:27-64):
 * out = beta*out + alpha*(conv_{k,dil,pad}(act(x)) + bias + res); w = packed [k][Ci][Co] (the fallback when the layer is not eligible),
 * w_mf = the same filter as F(2,3) pseudo-taps [P][Ci][Co] (pack.py:pack_conv_mf).  k = 3 / 5 / 7 / 11, stride 1, Ci % 16 == 0, Co >= 64,
 * T % 4 == 0.  Agrees with vb_conv1d_f32 to fp32 roundoff, not bit for bit. */
int vb_conv1d_f32_mf(const float* x, const float* w, const float* w_mf, const float* bias, int B, int Ci, int T_in, int Co, int ksize, int dil,
                     int pad, int T_out, int in_act, float in_slope, const float* res, float alpha, float beta, float* out, void* stream);
/* HiFi-GAN ResBlock1 pair in exact fp32, one launch (vocoder/hifigan/modules/hifigan.py:27-64; respair_f32.hip):
 * out = beta*out + alpha*(x + b2 + conv2_{k,1}(lrelu(b1 + conv1_{k,dil}(lrelu(x))))), x / out [B][C][T] (distinct buffers), weights fp32
 * packed [k][C ci][C co]; C = 32 / 64 / 128, odd k <= 17, (k-1)*dil <= 60, T % 4 == 0.  Equals two vb_conv1d_f32 launches bit for bit. */
int vb_respair_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, int B, int C, int T, int k, int dil,
                   float slope, float alpha, float beta, float* out, void* stream);
/* The same pair with F(2,3) minimal filtering in both convolutions (respair_f32w.hip): w1_mf / w2_mf = pseudo-tap weights [P][C][C]
 * (pack.py:pack_conv_mf); C = 32, k = 3 / 7 / 11, (k-1)*dil <= 60, T % 4 == 0.  Equals two vb_conv1d_f32_mf launches bit for bit. */
int vb_respair_f32_mf(const float* x, const float* w1_mf, const float* b1, const float* w2_mf, const float* b2, int B, int C, int T, int k, int dil,
                      float slope, float alpha, float beta, float* out, void* stream);
/* counter-based Gumbel draws: out[rows][w], rows = n_branch*B*T */
int vb_fill_gumbel(float* out, int B, int n_branch, int T, int width, uint64_t seed, int64_t clip_base, int nfe, int block, int gate,
                   void* stream);
int vb_cast_planes(const float* x, int64_t n, void* out, int np, void* stream);

#ifdef __cplusplus
}
#endif
#endif
