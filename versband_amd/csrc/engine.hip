// Host-side runtime of libversband_hip.so: context, DiT engine (conditioning precompute,
// one network evaluation, the CFG/Euler sampling loop), the conv-net executor used by the
// VAE decoder and the HiFi-GAN generator, and the extern "C" entry points of
// include/versband_hip.h.  No device allocation happens here: all scratch is carved out of
// caller buffers (sizes from *_bytes()).
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/versband_hip.h"
#include "kernels.h"

thread_local char g_vb_err[512] = "";

// ---- tuning knobs ---------------------------------------------------------------------------
static VbTune g_tune;
static std::atomic<bool> g_tune_loaded{false};
static std::atomic<unsigned> g_tune_gen{0};
static std::mutex g_tune_mu;
static int env_int(const char* k, int dflt) { const char* v = getenv(k); return (v && *v) ? atoi(v) : dflt; }
static void tune_load() {
    VbTune t;
    // product knobs: result-preserving selections the tests flip to compare both forms, plus VB_ATTN_DEFER / VB_NO_GRAPH
    t.router_tpw = env_int("VB_ROUTER_TPW", 0);
    if (const char* v = getenv("VB_ATTN_DEFER")) t.attn_defer_thr = (float)atof(v);     // log2 units; 0 = exact running maximum
    t.gemm_small = env_int("VB_GEMM_SMALL", 11); t.gemm_small_tiles = env_int("VB_GEMM_SMALL_TILES", 200);
    t.gemm_tile = env_int("VB_GEMM_TILE", -1);
    t.conv_direct_epi = getenv("VB_CONV_DIRECT_EPI") != nullptr;
    t.band_unfused = getenv("VB_BAND_UNFUSED") != nullptr;
    t.w2_pair = env_int("VB_W2_PAIR", 1);
    t.qkv_p16_off = getenv("VB_QKV_P16_OFF") != nullptr;
    t.no_xcd_groups = getenv("VB_NO_XCD_GROUPS") != nullptr;
    t.qkv_vt16_off = getenv("VB_QKV_VT16_OFF") != nullptr;
    t.rmsnorm_generic = getenv("VB_RMSNORM_GENERIC") != nullptr;
    t.wide_resid = env_int("VB_WIDE_RESID", 1);
    t.big_tile_min_k = env_int("VB_BIG_TILE_MIN_K", 384);
    t.proj_in_conv = getenv("VB_PROJ_IN_CONV") != nullptr;
    t.conv_gemm_off = getenv("VB_CONV_GEMM_OFF") != nullptr;
    t.final_gemm = getenv("VB_FINAL_GEMM") != nullptr;
    t.router_generic = getenv("VB_ROUTER_GENERIC") != nullptr;
    t.band_epi_old = getenv("VB_BAND_EPI_OLD") != nullptr;
    t.conv_f32_old = getenv("VB_CONV_F32_OLD") != nullptr;
    t.gemm_p8_off = getenv("VB_GEMM_P8_OFF") != nullptr;
    t.bucket_count_launch = getenv("VB_BUCKET_COUNT_LAUNCH") != nullptr;
    t.euler_launch = getenv("VB_EULER_LAUNCH") != nullptr;
    t.conv_f32_rt_taps = getenv("VB_CONV_F32_RT_TAPS") != nullptr;
    t.conv_mf_off = getenv("VB_CONV_MF_OFF") != nullptr; t.conv_mf_occ = env_int("VB_MF_OCC", 2);        // minimal-filtering weights ignored: the direct fp32 kernels (A/B)
    t.no_graph = getenv("VB_NO_GRAPH") != nullptr;
#ifdef VB_EXPERIMENTS
    // experiments build only (VB_BUILD_EXPERIMENTS=1 python -m versband_amd.build): ablations and the measured-slower kernels
    t.gemm_variant = env_int("VB_GEMM_VARIANT", 1); t.gemm_ablate = env_int("VB_GEMM_ABLATE", 0);
    t.gemm_nchunk = env_int("VB_GEMM_NCHUNK", 0); t.gemm_p8 = env_int("VB_GEMM_P8", -1);
    t.gemm_p8_mask = env_int("VB_GEMM_P8_MASK", 0); t.gemm_p8_direct = env_int("VB_GEMM_P8_DIRECT", 0); t.gemm_p8_p16 = env_int("VB_GEMM_P8_P16", 0);
    t.conv_ablate = env_int("VB_CONV_ABLATE", 0);
    t.attn_ablate = env_int("VB_ATTN_ABLATE", 0); t.attn_variant = env_int("VB_ATTN_VARIANT", -1);
    t.score_fused = getenv("VB_SCORE_FUSED") != nullptr;
    t.gemm_pk_f32 = env_int("VB_GEMM_PK_F32", 0); t.gemm_pk = env_int("VB_GEMM_PK", 0); t.gemm_p8_ring = env_int("VB_GEMM_P8_RING", 0);
    // round-2 A/B switches no test flips any more: the caption gate without the fold, the once-per-clip stem convolutions in exact fp32
    t.gate_unfolded = getenv("VB_GATE_UNFOLDED") != nullptr; t.stem_f32 = getenv("VB_STEM_F32") != nullptr;
#endif
    g_tune = t;
    g_tune_gen.fetch_add(1, std::memory_order_relaxed);
    g_tune_loaded.store(true, std::memory_order_release);
}
const VbTune& vb_tune() {
    if (!g_tune_loaded.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        if (!g_tune_loaded.load(std::memory_order_relaxed)) tune_load();
    }
    return g_tune;
}
unsigned vb_tune_generation() { (void)vb_tune(); return g_tune_gen.load(std::memory_order_relaxed); }
// (tools / tests only, single-threaded by contract: no launch may be in flight on another host thread while the knobs change)
extern "C" void vb_tune_reload(void) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    tune_load();
}

// ---- roctx ranges (rocprofv3 --marker-trace): the profiler's marker library is looked up at run time, nothing links against it ----
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
static roctx_push_fn g_roctx_push = nullptr;
static roctx_pop_fn g_roctx_pop = nullptr;
static std::once_flag g_roctx_once;
static void roctx_init() {
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
        g_roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
        if (g_roctx_push && g_roctx_pop) return;
        g_roctx_push = nullptr; g_roctx_pop = nullptr;
    }
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) {
        std::call_once(g_roctx_once, roctx_init);
        on = g_roctx_push != nullptr;
        if (on) (void)g_roctx_push(name);
    }
    ~RoctxRange() { if (on) (void)g_roctx_pop(); }
};

// ---- kernel-class profiling -----------------------------------------------------------------
#define PROF_CLASSES 4
#define PROF_POOL 32768
static int g_prof_mask = 0;
static std::vector<hipEvent_t> g_prof_ev;          // pool of events (pairs)
static size_t g_prof_next = 0;
struct ProfRec { int cls; size_t ev; };
static std::vector<ProfRec> g_prof_recs;
static double g_prof_flops[PROF_CLASSES] = {0, 0, 0, 0};
static double g_prof_bytes[PROF_CLASSES] = {0, 0, 0, 0};
static long long g_prof_launches[PROF_CLASSES] = {0, 0, 0, 0};
static thread_local size_t g_prof_open = (size_t)-1;
static thread_local unsigned g_prof_tick[PROF_CLASSES] = {0, 0, 0, 0};
static thread_local unsigned g_prof_tick_gen = 0;
static std::atomic<unsigned> g_prof_gen{1};         // bumped by vb_prof_enable: every host thread restarts its launch counters
static int g_prof_every[PROF_CLASSES] = {1, 1, 1, 1};   // time every n-th launch of a class (per host thread) ...
static int g_prof_phase[PROF_CLASSES] = {0, 0, 0, 0};   // ... the one with launch index % n == phase
static std::mutex g_prof_mu;                        // several host threads (one per stream) may launch concurrently
void prof_start(int cls, double flops, double bytes, hipStream_t st) {
    g_prof_open = (size_t)-1;
    if (!(g_prof_mask & (1 << cls))) return;
    const unsigned gen = g_prof_gen.load(std::memory_order_relaxed);
    if (g_prof_tick_gen != gen) { g_prof_tick_gen = gen; for (unsigned& t : g_prof_tick) t = 0; }
    const bool sampled = (int)(g_prof_tick[cls]++ % (unsigned)g_prof_every[cls]) == g_prof_phase[cls];
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_launches[cls] += 1;
    if (!sampled || g_prof_next + 2 > g_prof_ev.size()) return;      // counted, not timed
    g_prof_flops[cls] += flops;
    g_prof_bytes[cls] += bytes;
    g_prof_open = g_prof_next;
    g_prof_next += 2;
    (void)hipEventRecord(g_prof_ev[g_prof_open], st);
}
void prof_stop(int cls, hipStream_t st) {
    if (g_prof_open == (size_t)-1) return;
    (void)hipEventRecord(g_prof_ev[g_prof_open + 1], st);
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_recs.push_back(ProfRec{cls, g_prof_open});
    }
    g_prof_open = (size_t)-1;
}

struct NetProgram {
    std::vector<vb_net_op> ops;
    std::vector<vb_buf_desc> bufs;
    int in_ch = 0, out_ch = 0, in_tmul = 1, out_tmul = 1;
    bool loaded = false;
};
// one captured + instantiated step loop of vb_sample_cfg (hipGraph), keyed by everything the launches bake in
struct SampleGraph {
    const void* x = nullptr; const void* cond = nullptr; const void* ws = nullptr;
    int B = 0, nb = 0, T = 0, L = 0, n_steps = 0; float cfg_scale = 0.f; unsigned tune_gen = 0;
    int seen = 0;                         // calls with this key so far (the first runs eagerly, the second captures)
    bool failed = false;                  // capture was refused once (e.g. legacy default stream): stay eager
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    uint64_t last_use = 0;
};
struct vb_ctx {
    int device = 0;
    std::vector<SampleGraph> graphs; uint64_t graph_clock = 0;
    bool dit_loaded = false;
    vb_dit_config cfg;
    vb_dit_weights w;
    NetProgram nets[3];
    bool t5_loaded = false;
    vb_t5_config t5cfg;
    vb_t5_weights t5w;
    bool mel_loaded = false;
    vb_mel_config melcfg;
    const float* mel_dft = nullptr; const float* mel_basis_t = nullptr;
};

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
struct Carver {
    char* base; size_t off = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    template <typename T> T* take(size_t n) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off = align_up(off + n * sizeof(T));
        return p;
    }
};
static inline int pad64(int x) { return (x + 63) / 64 * 64; }
static inline Planes mkp(bf16_t* p, int64_t numel, int np) { return Planes{p, numel, np}; }
static inline Planes wpl(const void* p, int64_t numel, int np) { return Planes{(bf16_t*)p, numel, np}; }

// ------------------------------------------------------------------------------------------
// layouts
// ------------------------------------------------------------------------------------------
#define PIN_KP 192      // K of proj_in as a GEMM: 5 taps x 32 (channels padded) = 160, padded to a multiple of 64
struct CondL {
    float* ac; float* cemb;
    bf16_t* ky[VB_MAX_DEPTH]; bf16_t* vyt[VB_MAX_DEPTH]; bf16_t* kc[VB_MAX_DEPTH]; bf16_t* vct[VB_MAX_DEPTH];
    float* la[VB_MAX_DEPTH];
    // folded caption gate (see router_kernel<.., true>): per clip and block the caption keys with the MoE q-projection folded in
    // (planes [Beff][NS = L*heads][D], row = key*heads + head), the q-bias part of the scores and the gate-contracted values
    bf16_t* mf[VB_MAX_DEPTH]; float* cb[VB_MAX_DEPTH]; float* vw[VB_MAX_DEPTH]; int* clip_off; int NS; bool fold;
    bf16_t* pin_w;       // proj_in weights as a GEMM operand: split planes [2][D][PIN_KP], k = tap * 32 + ci (conv_w_to_gemm_kernel)
    int64_t n_k, n_vt; int Lpad;
    size_t total;
};
static inline bool gate_fold_ok(const vb_dit_config& c, int L) {
    const int NS = L * c.heads;
    return NS % 64 == 0 && NS <= 1024 && NS <= c.hidden && (c.heads & (c.heads - 1)) == 0 && c.heads <= 64 && !vb_tune().gate_unfolded;
}
static CondL carve_cond(void* base, const vb_dit_config& c, int B, int nb, int T, int L) {
    CondL o;
    Carver cv(base);
    const int Beff = B * nb, D = c.hidden, hd = D / c.heads;
    o.Lpad = pad64(L);
    o.n_k = (int64_t)Beff * L * D;
    o.n_vt = (int64_t)Beff * c.heads * hd * o.Lpad;
    o.ac = cv.take<float>((size_t)B * T * D);
    o.cemb = cv.take<float>((size_t)Beff * D);
    for (int i = 0; i < c.depth; ++i) {
        o.ky[i] = cv.take<bf16_t>(o.n_k * c.np);
        o.vyt[i] = cv.take<bf16_t>(o.n_vt * c.np);
        o.kc[i] = cv.take<bf16_t>(o.n_k * c.np);
        o.vct[i] = cv.take<bf16_t>(o.n_vt * c.np);
        o.la[i] = cv.take<float>((size_t)B * T * c.num_experts);
    }
    o.NS = L * c.heads;
    o.fold = gate_fold_ok(c, L);
    o.clip_off = cv.take<int>((size_t)Beff + 1);
    for (int i = 0; i < c.depth; ++i) {
        o.mf[i] = o.fold ? cv.take<bf16_t>((size_t)Beff * o.NS * D * c.np) : nullptr;
        o.cb[i] = o.fold ? cv.take<float>((size_t)Beff * o.NS) : nullptr;
        o.vw[i] = o.fold ? cv.take<float>((size_t)Beff * o.NS * c.num_experts) : nullptr;
    }
    o.pin_w = cv.take<bf16_t>((size_t)2 * D * PIN_KP);
    o.total = cv.off;
    return o;
}

#define T_FREQ_ROWS 1000  // rows of vb_dit_weights.t_freq_table (pack.timestep_table); other indices are computed in the kernel
#define PRE_STEPS 64      // sampler steps whose adaLN / gate vectors are tabulated up front
struct WsL {
    int* step; int64_t* t_idx_cur; int64_t* t_table; float* dt_table;
    float *temb0, *temb, *mod_all, *hl, *h, *cq32, *mc, *ma, *y32, *g1, *g2, *g3, *v;
    bf16_t* modA;                                                   // A operand (planes) of the adaLN tabulation GEMM
    float *temb0_s, *temb_s, *hl_s, *mod_s; int64_t* row_step;     // per-sample tables of the conditioning vectors of every step
    bf16_t *u, *q, *k, *vt, *a, *qm, *cqa, *Hs, *y, *Hf, *pin_a;
    int *ic, *ia, *group_off, *perm, *pair_off, *pair_pa;
    // precompute temporaries
    float *tA, *tB, *tC, *tD, *tE, *cap_pre, *cap32, *pooled, *pooled_ln;
    bf16_t *t5p, *gel, *capp, *yp;
    int64_t n_tok, n_vt; int Tpad, MODW;
    size_t total;
};
static WsL carve_ws(void* base, const vb_dit_config& c, int B, int nb, int T, int L) {
    WsL o;
    Carver cv(base);
    const int Beff = B * nb, D = c.hidden, H = c.ffn_hidden, E = c.num_experts, hd = D / c.heads;
    const int64_t N = (int64_t)Beff * T;
    o.n_tok = N; o.Tpad = pad64(T); o.MODW = c.depth * 6 * D + 2 * D;
    o.n_vt = (int64_t)Beff * c.heads * hd * o.Tpad;
    o.step = cv.take<int>(16);
    o.t_idx_cur = cv.take<int64_t>(Beff);
    o.t_table = cv.take<int64_t>(1024);
    o.dt_table = cv.take<float>(1024);
    o.temb0 = cv.take<float>((size_t)Beff * D);
    o.temb = cv.take<float>((size_t)Beff * D);
    o.mod_all = cv.take<float>((size_t)Beff * o.MODW);
    o.hl = cv.take<float>((size_t)Beff * c.depth * 2);
    o.h = cv.take<float>(N * D);
    o.cq32 = cv.take<float>(N * D);
    o.mc = cv.take<float>(N);
    o.ma = cv.take<float>(N);
    o.y32 = cv.take<float>(N * D);
    o.g1 = cv.take<float>(N * 2);
    o.g2 = cv.take<float>(N * E);
    o.g3 = cv.take<float>(N * E);
    o.v = cv.take<float>((size_t)Beff * c.in_channels * T);
    o.u = cv.take<bf16_t>(N * D * c.np);
    o.q = cv.take<bf16_t>(N * D * c.np);
    o.k = cv.take<bf16_t>(N * D * c.np);
    o.vt = cv.take<bf16_t>(o.n_vt * c.np);
    o.a = cv.take<bf16_t>(N * D * c.np);
    o.qm = cv.take<bf16_t>(N * D * c.np);
    o.cqa = cv.take<bf16_t>(N * D * c.np);
    o.Hs = cv.take<bf16_t>(2 * N * H * c.np);
    o.y = cv.take<bf16_t>(N * D * c.np);
    o.Hf = cv.take<bf16_t>(N * E * H * c.np);
    o.ic = cv.take<int>(N);
    o.ia = cv.take<int>(N);
    o.group_off = cv.take<int>(2 * E + 1);
    o.perm = cv.take<int>(2 * N + bucket_scratch_ints((int)N, E));
    o.pin_a = cv.take<bf16_t>((size_t)2 * B * T * PIN_KP);
    o.pair_off = cv.take<int>(32);
    o.pair_pa = cv.take<int>(N);
    // precompute temporaries
    const int T_mel = 2 * T + 8;
    o.tA = cv.take<float>((size_t)B * D * T_mel);
    o.tB = cv.take<float>((size_t)B * D * T_mel);
    o.tC = cv.take<float>((size_t)B * D * T_mel);
    o.tD = cv.take<float>((size_t)B * D * T_mel);
    o.tE = cv.take<float>((size_t)B * D * T_mel);
    const int64_t NL = (int64_t)Beff * L;
    o.t5p = cv.take<bf16_t>(NL * c.ori_dim * 2);
    o.gel = cv.take<bf16_t>(NL * D * 2);
    o.cap_pre = cv.take<float>(NL * D);
    o.cap32 = cv.take<float>(NL * D);
    o.capp = cv.take<bf16_t>(NL * D * 2);
    o.yp = cv.take<bf16_t>(NL * D * 2);
    o.pooled = cv.take<float>((size_t)Beff * D);
    o.pooled_ln = cv.take<float>((size_t)Beff * D);
    o.temb0_s = cv.take<float>((size_t)PRE_STEPS * D);
    o.temb_s = cv.take<float>((size_t)PRE_STEPS * D);
    o.hl_s = cv.take<float>((size_t)PRE_STEPS * c.depth * 2);
    o.mod_s = cv.take<float>((size_t)PRE_STEPS * Beff * o.MODW);
    o.row_step = cv.take<int64_t>((size_t)PRE_STEPS * Beff);
    o.modA = cv.take<bf16_t>((size_t)2 * PRE_STEPS * Beff * c.hidden);
    o.total = cv.off;
    return o;
}

// ------------------------------------------------------------------------------------------
// DiT: conditioning precompute
// ------------------------------------------------------------------------------------------
static int dit_precompute(vb_ctx* ctx, const float* t5, const int64_t* midi, const int64_t* beats, int B, int nb, int T, int T_mel,
                          int L, void* cond, void* ws, hipStream_t st) {
    const vb_dit_config& c = ctx->cfg;
    const vb_dit_weights& w = ctx->w;
    if (T_mel > 2 * T + 8) VB_FAIL(VB_E_INVALID, "precompute: T_mel=%d too long for T=%d", T_mel, T);
    const int T_ac = T_mel / 2;
    if (abs(T - T_ac) > 2) VB_FAIL(VB_E_INVALID, "precompute: latent length %d vs conditioning length %d differ by more than 2 "
                                   "(vocal2music_moe.py:397 would leave the shapes mismatched)", T, T_ac);
    CondL cd = carve_cond(cond, c, B, nb, T, L);
    WsL s = carve_ws(ws, c, B, nb, T, L);
    const int Beff = B * nb, D = c.hidden, E = c.num_experts, hd = D / c.heads, np = c.np;
    const int64_t NL = (int64_t)Beff * L;

    // ---- acoustic stem (vocal2music_moe.py:388-393)
    ConvArgs cv;
    for (int which = 0; which < 2; ++which) {
        VB_TRY(launch_embed_t(which ? beats : midi, which ? w.beats_emb : w.midi_emb, B, T_mel, D, which ? 3 : 130, s.tA, st));   // Embedding(3) / Embedding(130), vocal2music_moe.py:337-350
        cv = ConvArgs();
        cv.x = s.tA; cv.x_bstride = (int64_t)D * T_mel; cv.Ci = D; cv.T_in = T_mel;
        cv.w = which ? w.beats_conv_w : w.midi_conv_w; cv.bias = which ? w.beats_conv_b : w.midi_conv_b;
        cv.Co = D; cv.ksize = 5; cv.pad = 2; cv.out = which ? s.tC : s.tB; cv.out_bstride = (int64_t)D * T_mel; cv.T_out = T_mel;
        cv.out_act = ACT_LRELU; cv.out_slope = 0.01f; cv.B = B;
        if (const void* w3 = which ? w.beats_conv_w3 : w.midi_conv_w3; w3 && D % 32 == 0 && !vb_tune().stem_f32) {
            cv.wp = (const bf16_t*)w3; cv.Ci_pad = D; cv.wp_plane = (int64_t)5 * D * D;
        }
        VB_TRY(launch_conv1d(cv, st));
    }
    VB_TRY(launch_pool_add(s.tB, s.tC, B, D, T_mel, s.tD, st));
    cv = ConvArgs();
    cv.x = s.tD; cv.x_bstride = (int64_t)D * T_ac; cv.Ci = D; cv.T_in = T_ac; cv.w = w.final_proj_w; cv.bias = w.final_proj_b;
    cv.Co = D; cv.ksize = 1; cv.pad = 0; cv.out = s.tE; cv.out_bstride = (int64_t)D * T_ac; cv.T_out = T_ac; cv.B = B;
    if (w.final_proj_w3 && D % 32 == 0 && !vb_tune().stem_f32) { cv.wp = (const bf16_t*)w.final_proj_w3; cv.Ci_pad = D; cv.wp_plane = (int64_t)D * D; }
    VB_TRY(launch_conv1d(cv, st));
    VB_TRY(launch_transpose_bct_btc(s.tE, B, D, T_ac, T, cd.ac, st));

    // ---- caption embedding (ConditionEmbedder, flag_large_dit_moe.py:149-160) in split precision
    Planes t5p = mkp(s.t5p, NL * c.ori_dim, 2);
    VB_TRY(launch_cast_planes(t5, NL * c.ori_dim, t5p, st));
    GemmArgs g;
    g.A = t5p.p; g.a_plane = t5p.plane; g.lda = c.ori_dim; g.B = (const bf16_t*)w.c_emb0; g.b_plane = (int64_t)D * c.ori_dim;
    g.ldb = c.ori_dim; g.M = (int)NL; g.N = D; g.K = c.ori_dim; g.nseg = 3; g.epi = EPI_GELU_PLANES; g.bias = w.c_emb0_b;
    g.out = mkp(s.gel, NL * D, 2); g.ldc = D;
    VB_TRY(launch_gemm(g, st));
    g = GemmArgs();
    g.A = s.gel; g.a_plane = NL * D; g.lda = D; g.B = (const bf16_t*)w.c_emb2; g.b_plane = (int64_t)D * D; g.ldb = D;
    g.M = (int)NL; g.N = D; g.K = D; g.nseg = 3; g.epi = EPI_F32; g.bias = w.c_emb2_b; g.out32 = s.cap_pre; g.ldc32 = D;
    VB_TRY(launch_gemm(g, st));
    Planes capp = mkp(s.capp, NL * D, 2);
    VB_TRY(launch_layernorm(s.cap_pre, w.c_ln_w, w.c_ln_b, (int)NL, D, 1e-5f, s.cap32, capp, st));
    // pooled caption -> cap_embedder (vocal2music_moe.py:367-370,410-413)
    VB_TRY(launch_mean_rows(s.cap32, Beff, L, D, s.pooled, st));
    VB_TRY(launch_layernorm(s.pooled, w.cap_ln_w, w.cap_ln_b, Beff, D, 1e-5f, s.pooled_ln, Planes{nullptr, 0, 1}, st));
    VB_TRY(launch_gemv_rows(s.pooled_ln, D, nullptr, 0, 1, w.cap_lin_w, w.cap_lin_b, Beff, D, D, 0, cd.cemb, D, st));

    // ---- per block: context K/V for Attention and MoE.cross_attention, acoustic gate logits
    for (int i = 0; i < c.depth; ++i) {
        const vb_dit_block_weights& bw = w.blocks[i];
        Planes yp = mkp(s.yp, NL * D, 2);
        VB_TRY(launch_rmsnorm_mod(s.cap32, bw.y_norm_w, nullptr, nullptr, 0, (int)NL, D, L, c.norm_eps, yp, st));
        VB_HIP(hipMemsetAsync(cd.vyt[i], 0, (size_t)cd.n_vt * np * sizeof(bf16_t), st));
        VB_HIP(hipMemsetAsync(cd.vct[i], 0, (size_t)cd.n_vt * np * sizeof(bf16_t), st));
        for (int which = 0; which < 4; ++which) {
            // 0: ky = y Wk_y^T   1: vy^T   2: kc = cap Wk^T + bk   3: vc^T
            g = GemmArgs();
            const bool from_y = which < 2;
            g.A = from_y ? yp.p : capp.p; g.a_plane = NL * D; g.lda = D;
            const void* W = which == 0 ? bw.wky : which == 1 ? bw.wvy : which == 2 ? bw.wk_m : bw.wv_m;
            g.B = (const bf16_t*)W; g.b_plane = (int64_t)D * D; g.ldb = D; g.M = (int)NL; g.N = D; g.K = D; g.nseg = 3;
            g.bias = which == 2 ? bw.bk_m : which == 3 ? bw.bv_m : nullptr;
            if (which == 0 || which == 2) {
                g.epi = EPI_PLANES; g.out = mkp(which == 0 ? cd.ky[i] : cd.kc[i], cd.n_k, np); g.ldc = D;
            } else {
                g.epi = EPI_HEADS_T; g.out = mkp(which == 1 ? cd.vyt[i] : cd.vct[i], cd.n_vt, np);
                g.T = L; g.H = c.heads; g.hd = hd; g.Tpad = cd.Lpad;
            }
            VB_TRY(launch_gemm(g, st));
        }
        VB_TRY(launch_rows_dot(cd.ac, bw.wag, bw.bag, B * T, D, E, cd.la[i], st));
        if (cd.fold && bw.wqt_s && bw.bq_s) {
            // Mf[b][key*heads + head][:] = hd^-1/2 * sum_d Kc[b][key][head*hd + d] * Wq[head*hd + d][:]   (one launch, head = grid z)
            g = GemmArgs();
            g.A = cd.kc[i]; g.a_plane = cd.n_k; g.lda = D; g.a_koff_group = hd;
            g.B = (const bf16_t*)bw.wqt_s; g.b_plane = (int64_t)D * D; g.ldb = D; g.b_group_stride = hd;
            g.M = (int)NL; g.N = D; g.K = hd; g.nseg = np == 2 ? 3 : 1; g.ngroups = c.heads;
            g.epi = EPI_PLANES; g.out = mkp(cd.mf[i], (int64_t)Beff * cd.NS * D, np); g.ldc = c.heads * D; g.c_noff_group = D;
            VB_TRY(launch_gemm(g, st));
            VB_TRY(launch_gate_fold(mkp(cd.kc[i], cd.n_k, np), mkp(cd.vct[i], cd.n_vt, np), bw.bq_s, bw.wcg, Beff, L, cd.Lpad, c.heads, hd, E,
                                    cd.cb[i], cd.vw[i], st));
        }
    }
    VB_TRY(launch_iota_mul(cd.clip_off, Beff + 1, T, st));
    if (w.proj_in_w3 && c.in_channels <= 32)
        VB_TRY(launch_conv_w_to_gemm((const bf16_t*)w.proj_in_w3, (int64_t)5 * D * 32, 5, D, PIN_KP, cd.pin_w, st));
    return VB_OK;
}

// ------------------------------------------------------------------------------------------
// DiT: one evaluation (both CFG branches batched: rows [0,B) cond, [B,2B) uncond)
// ------------------------------------------------------------------------------------------
// sampler only: FinalLayer + CFG + Euler update + step advance as one launch (launch_final_layer_euler) - x is updated in place, v is not written
struct EulerFuse { float* x; float cfg_scale; const float* dt_table; int k; int* step; int64_t* t_idx_cur; const int64_t* t_table; int n_steps; };
static bool euler_fusable(const vb_ctx* ctx, int n_branch) {
    return n_branch == 2 && ctx->w.final_w && final_layer_fused_ok(ctx->cfg.hidden, ctx->cfg.in_channels) && !vb_tune().final_gemm && !vb_tune().euler_launch;
}
static int dit_forward(vb_ctx* ctx, const float* x, const int64_t* t_idx, const void* cond, const vb_noise* noise, int noise_step,
                       const int* step_ptr, int B, int nb, int T, int L, float* v_out, int32_t* route_out, void* ws, bool zero_vt,
                       const float* pre_mod, const float* pre_hl, hipStream_t st, int evals_before = -1, const EulerFuse* ef = nullptr) {
    const vb_dit_config& c = ctx->cfg;
    const vb_dit_weights& w = ctx->w;
    CondL cd = carve_cond(const_cast<void*>(cond), c, B, nb, T, L);
    WsL s = carve_ws(ws, c, B, nb, T, L);
    const int Beff = B * nb, D = c.hidden, H = c.ffn_hidden, E = c.num_experts, hd = D / c.heads, np = c.np;
    const int N = (int)s.n_tok, MODW = s.MODW, band = D / E;
    const int nseg = np == 2 ? 3 : 1;
    const int64_t ND = (int64_t)N * D;
    if (T > c.max_len) VB_FAIL(VB_E_INVALID, "dit_forward: T=%d exceeds the RoPE table (max_len=%d, vocal2music_moe.py:421)", T, c.max_len);
    if (zero_vt) VB_HIP(hipMemsetAsync(s.vt, 0, (size_t)s.n_vt * np * sizeof(bf16_t), st));
    // bucket counts as a side product of the router (round 5): two count tables in ping-pong - the router of block evaluation e adds into table
    // e & 1, the place kernel that reads it clears table (e + 1) & 1 for the next router.  evals_before < 0: a stand-alone call clears both tables
    // itself; the sampler clears them once per call and passes the number of block evaluations already done.
    const bool rcnt = bucket_router_counts_ok(N) && !vb_tune().bucket_count_launch;
    if (rcnt && evals_before < 0) VB_TRY(launch_fill_f32(reinterpret_cast<float*>(bucket_counts(s.perm, N, 0)), bucket_counts_ints(N), 0.f, st));
    const int eval0 = evals_before < 0 ? 0 : evals_before;

    // ---- timestep embedding + all adaLN modulations + high-level gate logits (depend on (t, caption) only)
    // (the sampler tabulates them for all steps up front and passes this step's rows in pre_mod / pre_hl)
    const float* mod_all = s.mod_all; const float* hl = s.hl; int hl_ld = c.depth * 2;
    if (pre_mod) {
        mod_all = pre_mod; hl = pre_hl; hl_ld = 0;
    } else {
        VB_TRY(launch_gemv_rows_idx(w.t_freq_table, 256, t_idx, nullptr, 0, 1, w.t_mlp0_w, w.t_mlp0_b, Beff, D, 256, 0, s.temb0, D, st, T_FREQ_ROWS));
        VB_TRY(launch_gemv_rows(s.temb0, D, nullptr, 0, 1, w.t_mlp2_w, w.t_mlp2_b, Beff, D, D, 1, s.temb, D, st));
        VB_TRY(launch_gemv_rows(s.temb, D, cd.cemb, D, Beff, w.adaln_w, w.adaln_b, Beff, MODW, D, 1, s.mod_all, MODW, st));
        VB_TRY(launch_gemv_rows(s.temb, D, nullptr, 0, 1, w.hl_w, w.hl_b, Beff, c.depth * 2, D, 0, s.hl, c.depth * 2, st));
    }

    // ---- h = proj_in(x)^T + acoustic   (vocal2music_moe.py:395,415)
    // As a GEMM (round 3): the k = 5 convolution over 20 channels is a K = 100 product per token - as a conv launch it was 1152 workgroups
    // each staging 80 KB of weights for 60 MFMAs per wave (79 us per evaluation, 0.7 TB/s of output).  The latent windows are laid out as
    // K-contiguous split planes (im2col, 6016 x 192 at 8 clips), multiplied with the re-laid conv weights in split precision (fp32-class,
    // as before), + bias + acoustic embedding in the epilogue, and the row is written for BOTH CFG branches (they embed the same x).
    if (w.proj_in_w3 && c.in_channels <= 32 && !vb_tune().proj_in_conv) {
        const int BT = B * T;
        VB_TRY(launch_im2col_latent(x, B, c.in_channels, T, 5, 2, PIN_KP, s.pin_a, (int64_t)BT * PIN_KP, st));
        GemmArgs g;
        g.A = s.pin_a; g.a_plane = (int64_t)BT * PIN_KP; g.lda = PIN_KP; g.B = cd.pin_w; g.b_plane = (int64_t)D * PIN_KP; g.ldb = PIN_KP;
        g.M = BT; g.N = D; g.K = PIN_KP; g.nseg = 3; g.epi = EPI_F32; g.bias = w.proj_in_b; g.out32 = s.h; g.ldc32 = D;
        g.add32 = cd.ac; g.dup_rows = nb == 2 ? BT : 0; g.prof_class = 2;       // split-precision convolution work: counted with the conv class
        VB_TRY(launch_gemm(g, st));
    } else {
        ConvArgs cv;
        cv.x = x; cv.x_bstride = (int64_t)c.in_channels * T; cv.Ci = c.in_channels; cv.T_in = T; cv.x_bmod = B;
        cv.w = w.proj_in_w; cv.bias = w.proj_in_b; cv.Co = D; cv.ksize = 5; cv.pad = 2;
        if (w.proj_in_w3 && c.in_channels <= 32) { cv.wp = (const bf16_t*)w.proj_in_w3; cv.Ci_pad = 32; cv.wp_plane = (int64_t)5 * D * 32; }
        cv.out = s.h; cv.out_bstride = (int64_t)T * D; cv.T_out = T; cv.out_transposed = 1;
        cv.add = cd.ac; cv.add_bstride = (int64_t)T * D; cv.add_bmod = B; cv.B = Beff;
        VB_TRY(launch_conv1d(cv, st));
    }

    Planes u = mkp(s.u, ND, np), q = mkp(s.q, ND, np), k = mkp(s.k, ND, np), vt = mkp(s.vt, s.n_vt, np), a = mkp(s.a, ND, np);
    Planes qm = mkp(s.qm, ND, np), cqa = mkp(s.cqa, ND, np), Hs = mkp(s.Hs, (int64_t)2 * N * H, np), y = mkp(s.y, ND, np);
    Planes Hf = mkp(s.Hf, (int64_t)N * E * H, np);
    const float scale = 1.0f / sqrtf((float)hd);

    for (int i = 0; i < c.depth; ++i) {
        const vb_dit_block_weights& bw = w.blocks[i];
        const float* mod = mod_all + (size_t)i * 6 * D;    // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        // ---- attention (flag_large_dit_moe.py:323-406)
        VB_TRY(launch_rmsnorm_mod(s.h, bw.attn_norm_w, mod, mod + D, MODW, N, D, T, c.norm_eps, u, st));
        GemmArgs g;
        g.A = u.p; g.a_plane = ND; g.lda = D; g.B = (const bf16_t*)bw.wqkv; g.b_plane = (int64_t)3 * D * D; g.ldb = D;
        g.M = N; g.N = 3 * D; g.K = D; g.nseg = nseg; g.epi = EPI_QKV_ROPE; g.q = q; g.k = k; g.vt = vt;
        g.rope_cos = w.rope_cos; g.rope_sin = w.rope_sin; g.H = c.heads; g.hd = hd; g.Tpad = s.Tpad; g.D = D; g.T = T;
        VB_TRY(launch_gemm(g, st));
        AttnArgs at;
        at.q = q; at.k = k; at.vt = vt; at.ky = mkp(cd.ky[i], cd.n_k, np); at.vyt = mkp(cd.vyt[i], cd.n_vt, np);
        at.cross_w = bw.cross_w; at.out = a; at.B = Beff; at.T = T; at.Tpad = s.Tpad; at.L = L; at.Lpad = cd.Lpad; at.H = c.heads;
        at.hd = hd; at.has_self = 1; at.has_cross = 1; at.kv_batch_mod = 0; at.scale = scale;
        VB_TRY(launch_attention(at, st));
        g = GemmArgs();
        g.A = a.p; g.a_plane = ND; g.lda = D; g.B = (const bf16_t*)bw.wo; g.b_plane = (int64_t)D * D; g.ldb = D;
        g.M = N; g.N = D; g.K = D; g.nseg = nseg; g.epi = EPI_RESID_GATE; g.out32 = s.h; g.ldc32 = D; g.gate = mod + 2 * D;
        g.gate_ld = MODW; g.T = T;
        VB_TRY(launch_gemm(g, st));

        // ---- Band-MoE (vocal2music_moe.py:117-185)
        VB_TRY(launch_rmsnorm_mod(s.h, bw.ffn_norm_w, mod + 3 * D, mod + 4 * D, MODW, N, D, T, c.norm_eps, u, st));
        const bool fold = cd.fold && bw.wqt_s && bw.bq_s;
        // routed w2 as ONE launch over (caption, acoustic) pair buckets: bf16 mode, E*E <= 16 groups; otherwise the two grouped w2 launches
        // (at EVERY batch size: the pair form and the two-launch form round differently - the gate weight rides in the bf16 hidden rows -
        //  and a clip's bits must not depend on the batch it rides in)
        const bool w2_pair = np == 1 && E * E <= 16 && H % 64 == 0 && D % 16 == 0 && vb_tune().w2_pair;
        // gates: injected Gumbel arrays (parity path) or counter-based draws generated inside the router kernel
        const float *g1 = nullptr, *g2 = nullptr, *g3 = nullptr;
        if (noise && noise->g1) {
            const size_t so = (size_t)noise_step * c.depth + i;
            g1 = noise->g1 + so * N * 2; g2 = noise->g2 + so * N * E; g3 = noise->g3 + so * N * E;
        }
        // scores + router in one launch (score_router.hip: the [N][NS] score matrix stays in LDS; bit-identical to the two launches
        // below).  MEASURED SLOWER - 91 us against 32 + 22 at 8 clips: its mainloop runs 36 us with one 4-wave workgroup per CU and the
        // router's per-token wave-wide shuffles take 63 us without many waves per SIMD to hide them - so it is an opt-in experiment
        // (VB_SCORE_FUSED=1), kept with its test as the record.
#ifdef VB_EXPERIMENTS
        const bool fused_router = fold && np == 1 && vb_tune().score_fused && score_router_supported(cd.NS, D, E, c.heads) &&
                                  (int64_t)Beff * cdiv(T, 64) >= 96;
#else
        const bool fused_router = false;
#endif
        if (fused_router) {
#ifdef VB_EXPERIMENTS
            ScoreRouterArgs sr;
            sr.A = u.p; sr.lda = D; sr.Bm = cd.mf[i]; sr.ldb = D; sr.bias = cd.cb[i]; sr.vw = cd.vw[i]; sr.bg = bw.bcg; sr.la = cd.la[i];
            sr.la_rows = B * T; sr.hl = hl + i * 2; sr.hl_ld = hl_ld; sr.g1 = g1; sr.g2 = g2; sr.g3 = g3; sr.Beff = Beff; sr.B = B; sr.T = T;
            sr.K = D; sr.NS = cd.NS; sr.Hh = c.heads; sr.E = E; sr.ic = s.ic; sr.ia = s.ia; sr.mc = s.mc; sr.ma = s.ma;
            sr.seed = noise ? noise->seed : 0; sr.clip_base = noise ? noise->clip_base : 0; sr.nfe_base = noise ? noise->nfe : 0;
            sr.step = step_ptr; sr.block = i;
            VB_TRY(launch_score_router(sr, st));
#endif
        } else {
        if (fold) {
            // caption gate, folded: scores of every token against its clip's caption keys for all heads in ONE grouped GEMM
            // (q-projection, q-bias and softmax scale live in the per-clip operand), then softmax + value/gate contraction +
            // routing in the router kernel.  Replaces q-projection GEMM + cross-attention launch + 768-wide gate dot.
            g = GemmArgs();
            g.A = u.p; g.a_plane = ND; g.lda = D; g.B = cd.mf[i]; g.b_plane = (int64_t)Beff * cd.NS * D; g.ldb = D;
            g.b_group_stride = (int64_t)cd.NS * D; g.M = N; g.N = cd.NS; g.K = D; g.nseg = nseg; g.ngroups = Beff;
            g.group_off = cd.clip_off; g.group_rows = T; g.epi = EPI_F32; g.bias = cd.cb[i]; g.bias_group_stride = cd.NS; g.out32 = s.y32; g.ldc32 = cd.NS;
            VB_TRY(launch_gemm(g, st));
        } else {
        g = GemmArgs();
        g.A = u.p; g.a_plane = ND; g.lda = D; g.B = (const bf16_t*)bw.wq_m; g.b_plane = (int64_t)D * D; g.ldb = D;
        g.M = N; g.N = D; g.K = D; g.nseg = nseg; g.epi = EPI_PLANES; g.bias = bw.bq_m; g.out = qm; g.ldc = D;
        VB_TRY(launch_gemm(g, st));
        at = AttnArgs();
        at.q = qm; at.k = Planes{nullptr, 0, np}; at.vt = Planes{nullptr, 0, np}; at.ky = mkp(cd.kc[i], cd.n_k, np);
        at.vyt = mkp(cd.vct[i], cd.n_vt, np); at.cross_w = nullptr; at.out = cqa; at.B = Beff; at.T = T; at.Tpad = s.Tpad; at.L = L;
        at.Lpad = cd.Lpad; at.H = c.heads; at.hd = hd; at.has_self = 0; at.has_cross = 1; at.kv_batch_mod = 0; at.scale = scale;
        VB_TRY(launch_attention(at, st));
        }
        // (MoE.cross_attention.out_proj is folded into the caption gate at pack time: lc = cqa . (Wcg Wo)^T + (Wcg bo + bcg),
        //  so the [N,768]x[768,768] out_proj GEMM never runs - its only consumer is the 768->E gate, vocal2music_moe.py:119-141)
        VB_TRY(launch_router(cqa, fold ? cd.vw[i] : bw.wcg, bw.bcg, cd.la[i], B * T, hl + i * 2, hl_ld, g1, g2, g3, N, T, D, E, s.ic, s.ia, s.mc,
                             s.ma, nullptr, B, noise ? noise->seed : 0, noise ? noise->clip_base : 0, noise ? noise->nfe : 0, step_ptr, i, st,
                             fold ? s.y32 : nullptr, cd.NS, c.heads, rcnt ? bucket_counts(s.perm, N, (eval0 + i) & 1) : nullptr,   // (not on the fused score + router path)
                             w2_pair ? E * E : 2 * E, w2_pair ? 1 : 0));
        }
        const bool rc = rcnt && !fused_router;
        VB_TRY(launch_bucket(s.ic, s.ia, N, E, s.group_off, s.perm, st, w2_pair ? s.pair_off : nullptr, s.pair_pa,
                             rc ? bucket_counts(s.perm, N, (eval0 + i) & 1) : nullptr, rc ? bucket_counts(s.perm, N, (eval0 + i + 1) & 1) : nullptr));
        if (route_out) {
            VB_HIP(hipMemcpyAsync(route_out + ((size_t)i * 2 + 0) * N, s.ic, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, st));
            VB_HIP(hipMemcpyAsync(route_out + ((size_t)i * 2 + 1) * N, s.ia, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, st));
        }
        // routed experts: hidden = silu(u W1^T) * (u W3^T) for both groups in one grouped launch
        g = GemmArgs();
        g.A = u.p; g.a_plane = ND; g.lda = D; g.a_rows = s.perm; g.B = (const bf16_t*)bw.w13; g.b_plane = (int64_t)2 * E * 2 * H * D;
        g.ldb = D; g.b_group_stride = (int64_t)2 * H * D; g.M = 2 * N; g.N = 2 * H; g.K = D; g.nseg = nseg; g.ngroups = 2 * E;
        g.group_off = s.group_off; g.epi = EPI_SWIGLU; g.out = Hs; g.ldc = H;
        if (w2_pair) { g.row_scale = s.mc; g.row_scale2 = s.ma; g.scale_split = N; }      // gate weights ride in the hidden rows
        VB_TRY(launch_gemm(g, st));
        if (w2_pair) {
            MoeW2PairArgs pw;
            pw.Hs = Hs.p; pw.W2 = (const bf16_t*)bw.w2; pw.pair_off = s.pair_off; pw.perm = s.perm; pw.pair_pa = s.pair_pa;
            pw.out = y.p; pw.N = N; pw.D = D; pw.H = H; pw.E = E;
            VB_TRY(launch_moe_w2_pair(pw, st));
        } else {
        // y = m_c * FFN^c(u)  (store), then y += m_a * FFN^a(u) (planes out)
        g = GemmArgs();
        g.A = Hs.p; g.a_plane = Hs.plane; g.lda = H; g.B = (const bf16_t*)bw.w2; g.b_plane = (int64_t)2 * E * D * H; g.ldb = H;
        g.b_group_stride = (int64_t)D * H; g.M = N; g.N = D; g.K = H; g.nseg = nseg; g.ngroups = E; g.group_off = s.group_off;
        g.epi = EPI_SCATTER_F32; g.out32 = s.y32; g.ldc32 = D; g.rows_out = s.perm; g.row_scale = s.mc;
        VB_TRY(launch_gemm(g, st));
        g.B = (const bf16_t*)bw.w2 + (int64_t)E * D * H; g.group_off = s.group_off + E; g.epi = EPI_SCATTER_ADD_PLANES;
        g.y32_in = s.y32; g.row_scale = s.ma; g.out = y; g.ldc = D;
        VB_TRY(launch_gemm(g, st));
        }
        // band experts (frequency-MoE): expert e sees only channel band e and produces only band e
        const bool band_unfused = vb_tune().band_unfused;       // tuning / A-B switch (tests compare both)
        // (a fused workgroup owns 192 tokens x one band for ~50 us whatever the batch: below ~128 workgroups - half a round of the
        //  CUs, i.e. fewer than 4 clips x 2 branches - the two grouped GEMMs, bit-identical, are faster: 55.4 -> 50.4 ms per pass at
        //  one clip, 69.9 -> 65.3 at two, break-even at four, profiles/r02_band_small_batch.txt)
        if (np == 1 && (band == 192 || band == 96) && H % 64 == 0 && E <= 8 && 8 % E == 0 && !band_unfused &&
            (int64_t)cdiv(N, band == 192 ? 192 : 256) * E >= 128) {
            // both products in one launch, hidden kept in LDS (bf16 production mode; independent of the batch size)
            BandFfnArgs bf;
            bf.y = y.p; bf.ldy = D; bf.w13 = (const bf16_t*)bw.w13f; bf.w2 = (const bf16_t*)bw.w2f; bf.M = N; bf.H = H; bf.E = E;
            bf.band = band; bf.out32 = s.h; bf.ldc32 = D; bf.gate = mod + 5 * D; bf.gate_ld = MODW; bf.T = T;
            VB_TRY(launch_band_ffn(bf, st));
            continue;
        }
        g = GemmArgs();
        g.A = y.p; g.a_plane = ND; g.lda = D; g.a_koff_group = band; g.B = (const bf16_t*)bw.w13f; g.b_plane = (int64_t)E * 2 * H * band;
        g.ldb = band; g.b_group_stride = (int64_t)2 * H * band; g.M = N; g.N = 2 * H; g.K = band; g.nseg = nseg; g.ngroups = E;
        g.epi = EPI_SWIGLU; g.out = Hf; g.ldc = E * H; g.c_noff_group = H;
        VB_TRY(launch_gemm(g, st));
        g = GemmArgs();
        g.A = Hf.p; g.a_plane = Hf.plane; g.lda = E * H; g.a_koff_group = H; g.B = (const bf16_t*)bw.w2f; g.b_plane = (int64_t)E * band * H;
        g.ldb = H; g.b_group_stride = (int64_t)band * H; g.M = N; g.N = band; g.K = H; g.nseg = nseg; g.ngroups = E;
        g.epi = EPI_RESID_GATE; g.out32 = s.h; g.ldc32 = D; g.c_noff_group = band; g.gate = mod + 5 * D; g.gate_ld = MODW; g.T = T;
        VB_TRY(launch_gemm(g, st));
    }
    // ---- FinalLayer (vocal2music_moe.py:287-291) -> v [Beff][C][T]
    const float* modf = mod_all + (size_t)c.depth * 6 * D;
    if (ef) {
        VB_TRY(launch_final_layer_euler(s.h, modf, modf + D, MODW, w.final_w, w.final_b, N, D, T, c.in_channels, 1e-6f, ef->x, ef->cfg_scale, ef->dt_table,
                                        ef->k, ef->step, ef->t_idx_cur, ef->t_table, ef->n_steps, Beff, st));
    } else if (w.final_w && final_layer_fused_ok(D, c.in_channels) && !vb_tune().final_gemm) {
        // one wave per token row: LayerNorm + modulate in registers, the 768 x 20 projection against LDS-resident weights (exact fp32)
        VB_TRY(launch_final_layer_fused(s.h, modf, modf + D, MODW, w.final_w, w.final_b, N, D, T, c.in_channels, 1e-6f, v_out, st));
    } else if (w.final_wp && c.in_channels % 4 == 0 && (int64_t)2 * N * H >= ND) {
        // LN + modulate -> split planes (plane 0 in u, plane 1 in the free expert-hidden buffer), then the projection on the MFMA
        // GEMM in split precision (fp32-class in both modes) with a channel-major epilogue: 62 us -> ~35 us per evaluation
        Planes ln{s.u, (int64_t)(s.Hs - s.u), 2};
        VB_TRY(launch_layernorm_mod_planes(s.h, modf, modf + D, MODW, N, D, T, 1e-6f, ln, st));
        GemmArgs g;
        g.A = ln.p; g.a_plane = ln.plane; g.lda = D; g.B = (const bf16_t*)w.final_wp; g.b_plane = (int64_t)c.in_channels * D; g.ldb = D;
        g.M = N; g.N = c.in_channels; g.K = D; g.nseg = 3; g.epi = EPI_F32_CT; g.bias = w.final_b; g.out32 = v_out; g.T = T;
        VB_TRY(launch_gemm(g, st));
    } else {
        VB_TRY(launch_final_layer(s.h, modf, modf + D, MODW, w.final_w, w.final_b, N, D, T, c.in_channels, 1e-6f, v_out, st));
    }
    return VB_OK;
}

// ------------------------------------------------------------------------------------------
// conv-net executor (VAE decoder, HiFi-GAN)
// ------------------------------------------------------------------------------------------
static size_t net_ws_bytes(const NetProgram& n, int B, int T, std::vector<size_t>* offs) {
    size_t off = 0;
    if (offs) offs->clear();
    for (const vb_buf_desc& d : n.bufs) {
        size_t tl = (size_t)T * d.tmul;
        size_t el = d.square == 1 ? tl * tl : (d.square == 2 ? (size_t)d.channels * ((tl + 31) / 32 * 32)
                                  : (d.square == 3 ? (size_t)d.channels * (tl + XT_HEAD + XT_TAIL) : (size_t)d.channels * tl));
        if (offs) offs->push_back(off);
        off = align_up(off + el * B * sizeof(float));
    }
    return off;
}
static int net_run(vb_ctx* ctx, int which, const float* in, int B, int T, float* out, void* ws, hipStream_t st) {
    NetProgram& n = ctx->nets[which];
    if (!n.loaded) VB_FAIL(VB_E_STATE, "net %d not loaded", which);
    VB_HIP(hipSetDevice(ctx->device));
    std::vector<size_t> offs;
    net_ws_bytes(n, B, T, &offs);
    auto ptr = [&](int id) -> float* {
        if (id == VB_BUF_INPUT) return const_cast<float*>(in);
        if (id == VB_BUF_OUTPUT) return out;
        if (id < 0) return nullptr;
        return reinterpret_cast<float*>(static_cast<char*>(ws) + offs[id]);
    };
    auto tlen = [&](int id) -> int {
        if (id == VB_BUF_INPUT) return T * n.in_tmul;
        if (id == VB_BUF_OUTPUT) return T * n.out_tmul;
        return T * n.bufs[id].tmul;
    };
    auto chans = [&](int id) -> int {
        if (id == VB_BUF_INPUT) return n.in_ch;
        if (id == VB_BUF_OUTPUT) return n.out_ch;
        return n.bufs[id].channels;
    };
    auto bstride = [&](int id) -> int64_t {
        if (id >= 0 && n.bufs[id].square == 1) return (int64_t)tlen(id) * tlen(id);
        return (int64_t)chans(id) * tlen(id);
    };
    for (size_t oi = 0; oi < n.ops.size(); ++oi) {
        const vb_net_op& o = n.ops[oi];
        if (o.kind == VB_OP_GN_STATS) {
            float* stp = ptr(o.stats);
            VB_TRY(launch_gn_stats(ptr(o.x), B, o.Ci, tlen(o.x), o.gn_groups, 1e-6f, stp, stp + (size_t)B * o.gn_groups, st));
        } else if (o.kind == VB_OP_SOFTMAX_T) {
            VB_TRY(launch_softmax_rows_t(ptr(o.x), B, tlen(o.x), tlen(o.x), ptr(o.out), st));
        } else if (o.kind == VB_OP_CONV) {
            ConvArgs a;
            a.x = ptr(o.x); a.x_bstride = bstride(o.x); a.T_in = tlen(o.x);
            if (o.x_planes) {
                // input written by VB_OP_XT_PLANES: its buffer's time length is the (upsampled) length the planes were made for
                a.xt = reinterpret_cast<const bf16_t*>(ptr(o.x));
                a.T_in = o.upsample2 ? tlen(o.x) / 2 : tlen(o.x);
            }
            a.Ci = o.Ci > 0 ? o.Ci : tlen(o.x);            // dynamic channel counts: the VAE attention contracts over T
            if (o.w_buf != -1) { a.w = ptr(o.w_buf); a.w_bstride = bstride(o.w_buf); } else { a.w = o.w; }
            a.bias = o.bias; a.Co = o.Co > 0 ? o.Co : tlen(o.out); a.ksize = o.ksize; a.dil = o.dil; a.pad = o.pad; a.upsample2 = o.upsample2;
            a.in_stride = o.in_stride > 1 ? o.in_stride : 1; a.in_phase = o.in_phase;
            a.in_act = o.in_act; a.in_slope = o.in_slope;
            if (o.stats >= 0) {
                float* stp = ptr(o.stats);
                a.gn_mean = stp; a.gn_rstd = stp + (size_t)B * o.gn_groups; a.gn_gamma = o.gn_gamma; a.gn_beta = o.gn_beta;
                a.gn_groups = o.gn_groups;
            }
            a.out = ptr(o.out); a.out_bstride = bstride(o.out); a.T_out = tlen(o.out);
            if (o.res != -1) { a.res = ptr(o.res); a.res_bstride = bstride(o.res); }
            a.alpha = o.alpha; a.beta = o.beta; a.acc_scale = o.acc_scale; a.out_act = o.out_act; a.out_slope = o.out_slope;
            a.out_transposed = o.out_transposed; a.B = B; a.tr_stride = o.tr_stride; a.tr_pad = o.tr_pad; a.tr_k = o.tr_k;
            if (o.w_buf != -1 && o.ci_pad == -1) {
                // per-batch split planes [2][B][Co][Ci_pad] written by an earlier VB_OP_SPLIT_PLANES
                a.Ci_pad = (a.Ci + 31) / 32 * 32;
                a.wp = reinterpret_cast<const bf16_t*>(ptr(o.w_buf)); a.wp_bstride = (int64_t)a.Co * a.Ci_pad;
                a.wp_plane = (int64_t)B * a.wp_bstride; a.w = nullptr;
            } else if (o.w_x3 && o.w_buf == -1) {
                const int phases = o.tr_stride > 1 ? o.tr_stride : 1;
                const int ntaps = o.tr_stride > 1 ? (o.tr_k + o.tr_stride - 1) / o.tr_stride : o.ksize;
                a.wp = (const bf16_t*)o.w_x3; a.Ci_pad = o.ci_pad; a.wp_plane = (int64_t)phases * ntaps * a.Co * o.ci_pad;
            } else if (o.w2_x3 && o.w_buf == -1) {
                a.w_mf = (const float*)o.w2_x3;        // fp32 minimal-filtering weights (VB_OP_CONV, w_x3 == NULL): conv1d_f32w_kernel where it applies
            }
            VB_TRY(launch_conv1d(a, st));
        } else if (o.kind == VB_OP_GN_APPLY) {
            float* stp = ptr(o.stats);
            VB_TRY(launch_gn_apply(ptr(o.x), stp, stp + (size_t)B * o.gn_groups, o.gn_gamma, o.gn_beta, B, o.Ci, tlen(o.x), o.gn_groups,
                                   o.in_act == ACT_GN_SWISH ? 1 : 0, ptr(o.out), st));
        } else if (o.kind == VB_OP_XT_PLANES) {
            const float* stp = o.stats >= 0 ? ptr(o.stats) : nullptr;
            VB_TRY(launch_xt_planes(ptr(o.x), stp, stp ? stp + (size_t)B * o.gn_groups : nullptr, o.gn_gamma, o.gn_beta, o.gn_groups, o.in_act, o.in_slope,
                                    o.upsample2, B, o.Ci, tlen(o.x), reinterpret_cast<bf16_t*>(ptr(o.out)), st));
        } else if (o.kind == VB_OP_AA_ACT) {
            VB_TRY(launch_aa_act(ptr(o.x), o.gn_gamma, o.gn_beta, o.w, B, o.Ci, tlen(o.x), ptr(o.out), st));
        } else if (o.kind == VB_OP_RESPAIR) {
            if (!o.w_x3) {
                // exact-fp32 pair (fp32 vocoder): w / w2_x3 are the fp32 packed [k][Ci][Co] weights of the two convolutions
                RespairF32Args r;
                r.x = ptr(o.x); r.out = ptr(o.out); r.B = B; r.C = o.Ci; r.T = tlen(o.x); r.k = o.ksize; r.dil = o.dil;
                r.w1 = o.w; r.w2 = (const float*)o.w2_x3; r.b1 = o.bias; r.b2 = o.bias2;
                r.slope = o.in_slope; r.alpha = o.alpha; r.beta = o.beta;
                if (!r.w1 || !r.w2 || !r.b1 || !r.b2 || tlen(o.out) != r.T) VB_FAIL(VB_E_INVALID, "net op %zu: incomplete fp32 respair", oi);
                if (o.ci_pad == -2) VB_TRY(launch_respair_f32w(r, st));      // w / w2_x3 are minimal-filtering pseudo-tap weights (fp32mf)
                else VB_TRY(launch_respair_f32(r, st));
                continue;
            }
            RespairArgs r;
            r.x = ptr(o.x); r.out = ptr(o.out); r.B = B; r.C = o.Ci; r.T = tlen(o.x); r.k = o.ksize; r.dil = o.dil;
            r.w1 = (const bf16_t*)o.w_x3; r.w2 = (const bf16_t*)o.w2_x3; r.b1 = o.bias; r.b2 = o.bias2;
            r.slope = o.in_slope; r.alpha = o.alpha; r.beta = o.beta;
            if (!r.w1 || !r.w2 || !r.b1 || !r.b2 || tlen(o.out) != r.T) VB_FAIL(VB_E_INVALID, "net op %zu: incomplete respair", oi);
            VB_TRY(launch_respair(r, st));
        } else if (o.kind == VB_OP_SPLIT_PLANES) {
            const int rows = o.Co > 0 ? o.Co : tlen(o.x), cols = o.Ci > 0 ? o.Ci : tlen(o.x);
            const int cpad = (cols + 31) / 32 * 32;
            VB_TRY(launch_split_rows(ptr(o.x), (int64_t)B * rows, cols, cpad, reinterpret_cast<bf16_t*>(ptr(o.out)), (int64_t)B * rows * cpad, st));
        } else {
            VB_FAIL(VB_E_INVALID, "net op %zu: bad kind %d", oi, o.kind);
        }
    }
    return VB_OK;
}

// ------------------------------------------------------------------------------------------
// extern "C"
// ------------------------------------------------------------------------------------------
extern "C" {

const char* vb_last_error(void) { return g_vb_err; }
int vb_abi_version(void) { return 2; }
// (vb_source_digest() lives in a two-line translation unit versband_amd/build.py generates: csrc/build/vb_digest.cpp)
int vb_has_experiments(void) {
#ifdef VB_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int vb_prof_enable(int class_mask) {
    if (class_mask && g_prof_ev.empty()) {
        g_prof_ev.resize(2 * PROF_POOL);
        for (auto& e : g_prof_ev) VB_HIP(hipEventCreate(&e));
    }
    g_prof_mask = class_mask & 0xff;
    // bits 8..15: sampling period n of class 0 (the GEMM class: ~1250 launches per pass); bits 16..19: period of the other classes (0 = the same n);
    // bits 20..23 / 24..27: the phase of class 0 / of the others - a caller that walks all phases over as many passes times EVERY launch exactly
    // once per cycle without ever bracketing two neighbouring launches (back-to-back event pairs read long kernels twice as long)
    const int e0 = ((class_mask >> 8) & 0xff) > 0 ? ((class_mask >> 8) & 0xff) : 1;
    const int e1 = ((class_mask >> 16) & 0xf) > 0 ? ((class_mask >> 16) & 0xf) : e0;
    for (int i = 0; i < PROF_CLASSES; ++i) {
        g_prof_every[i] = i == 0 ? e0 : e1;
        g_prof_phase[i] = ((class_mask >> (i == 0 ? 20 : 24)) & 0xf) % g_prof_every[i];
    }
    g_prof_gen.fetch_add(1, std::memory_order_relaxed);
    g_prof_next = 0;
    g_prof_recs.clear();
    for (int i = 0; i < PROF_CLASSES; ++i) { g_prof_flops[i] = 0; g_prof_bytes[i] = 0; g_prof_launches[i] = 0; }
    return VB_OK;
}
int vb_prof_read(int cls, double* ms_sum, double* flops, double* bytes, int64_t* launches, int64_t* timed) {
    if (cls < 0 || cls >= PROF_CLASSES) VB_FAIL(VB_E_INVALID, "prof_read: class %d", cls);
    VB_HIP(hipDeviceSynchronize());
    double ms = 0; int64_t n = 0;
    for (const ProfRec& r : g_prof_recs) {
        if (r.cls != cls) continue;
        float t = 0.f;
        VB_HIP(hipEventElapsedTime(&t, g_prof_ev[r.ev], g_prof_ev[r.ev + 1]));
        ms += t; ++n;
    }
    *ms_sum = ms; *flops = g_prof_flops[cls]; *bytes = g_prof_bytes[cls]; *launches = g_prof_launches[cls]; *timed = n;
    return VB_OK;
}

int vb_ctx_create(int device, vb_ctx** out) {
    if (!out) VB_FAIL(VB_E_INVALID, "ctx_create: null out");
    int n = 0;
    VB_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) VB_FAIL(VB_E_INVALID, "ctx_create: device %d of %d", device, n);
    VB_HIP(hipSetDevice(device));
    vb_ctx* c = new vb_ctx();
    c->device = device;
    memset(&c->cfg, 0, sizeof(c->cfg));
    memset(&c->w, 0, sizeof(c->w));
    *out = c;
    return VB_OK;
}
int vb_ctx_destroy(vb_ctx* ctx) {
    if (ctx)
        for (SampleGraph& g : ctx->graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
        }
    delete ctx;
    return VB_OK;
}

int vb_dit_load(vb_ctx* ctx, const vb_dit_config* cfg, const vb_dit_weights* w) {
    if (!ctx || !cfg || !w) VB_FAIL(VB_E_INVALID, "dit_load: null argument");
    if (cfg->depth < 1 || cfg->depth > VB_MAX_DEPTH) VB_FAIL(VB_E_INVALID, "dit_load: depth %d", cfg->depth);
    if (cfg->np != 1 && cfg->np != 2) VB_FAIL(VB_E_INVALID, "dit_load: np %d", cfg->np);
    if (cfg->hidden % cfg->heads || cfg->hidden / cfg->heads != 96)
        VB_FAIL(VB_E_INVALID, "dit_load: head_dim %d unsupported (kernels are built for 96)", cfg->hidden / (cfg->heads ? cfg->heads : 1));
    if (cfg->hidden % cfg->num_experts || (cfg->hidden / cfg->num_experts) % 8) VB_FAIL(VB_E_INVALID, "dit_load: band width must be a multiple of 8");
    if (cfg->context_dim != cfg->hidden) VB_FAIL(VB_E_INVALID, "dit_load: context_dim must equal hidden_size (vocal2music_moe.py:367-373)");
    if (cfg->num_experts > 16) VB_FAIL(VB_E_INVALID, "dit_load: num_experts %d > 16", cfg->num_experts);
    if (cfg->hidden > 768) VB_FAIL(VB_E_INVALID, "dit_load: hidden %d > 768 (router register tile)", cfg->hidden);
    // captured sampler graphs hold the OLD weight pointers and launch geometry in their kernel nodes: a (re)load drops them all
    for (SampleGraph& g : ctx->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    ctx->graphs.clear();
    ctx->cfg = *cfg;
    ctx->w = *w;
    ctx->dit_loaded = true;
    return VB_OK;
}
size_t vb_dit_cond_bytes(const vb_dit_config* cfg, int B, int n_branch, int T, int L) {
    return carve_cond(nullptr, *cfg, B, n_branch, T, L).total;
}
size_t vb_dit_workspace_bytes(const vb_dit_config* cfg, int B, int n_branch, int T, int L) {
    return carve_ws(nullptr, *cfg, B, n_branch, T, L).total;
}
int vb_dit_precompute_cond(vb_ctx* ctx, const float* t5, const int64_t* midi, const int64_t* beats, int B, int n_branch, int T,
                           int T_mel, int L, void* cond, void* ws, void* stream) {
    if (!ctx || !ctx->dit_loaded) VB_FAIL(VB_E_STATE, "precompute_cond: DiT not loaded");
    if (n_branch < 1 || n_branch > 2 || B < 1) VB_FAIL(VB_E_INVALID, "precompute_cond: B=%d n_branch=%d", B, n_branch);
    VB_HIP(hipSetDevice(ctx->device));
    RoctxRange rr("vb_dit_precompute_cond");
    return dit_precompute(ctx, t5, midi, beats, B, n_branch, T, T_mel, L, cond, ws, (hipStream_t)stream);
}
int vb_dit_forward(vb_ctx* ctx, const float* x, const int64_t* t_idx, const void* cond, const vb_noise* noise, int B, int n_branch,
                   int T, int L, float* v_out, int32_t* route_out, void* ws, void* stream) {
    if (!ctx || !ctx->dit_loaded) VB_FAIL(VB_E_STATE, "dit_forward: DiT not loaded");
    VB_HIP(hipSetDevice(ctx->device));        // the calling host thread may be new (one thread per stream): bind it to the context's GPU
    return dit_forward(ctx, x, t_idx, cond, noise, 0, nullptr, B, n_branch, T, L, v_out, route_out, ws, true, nullptr, nullptr,
                       (hipStream_t)stream);
}
int vb_euler_cfg_step(float* x, const float* v, int B, int64_t per_item, float cfg_scale, float dt, int has_uncond, void* stream) {
    return launch_euler_cfg(x, v, B, per_item, cfg_scale, nullptr, nullptr, dt, has_uncond, (hipStream_t)stream);
}
// the launches of one sampler call after its tables are in place: tabulation of the per-step conditioning vectors, then n_steps x
// [step bookkeeping, one network evaluation of both CFG branches, Euler + guidance update]   (cfm1_audio_sampler.py:107-116)
static int sample_steps(vb_ctx* ctx, float* x, const void* cond, int B, int n_branch, int T, int L, int n_steps, float cfg_scale,
                        const vb_noise* noise, float* traj, void* ws, hipStream_t st) {
    const vb_dit_config& c = ctx->cfg;
    WsL s = carve_ws(ws, c, B, n_branch, T, L);
    const int Beff = B * n_branch;
    const int64_t per = (int64_t)c.in_channels * T;
    // (a kernel, not hipMemsetAsync: the captured step loop then consists of kernel nodes only)
    VB_TRY(launch_fill_f32(reinterpret_cast<float*>(s.vt), (int64_t)s.n_vt * c.np / 2, 0.f, st));
    // The timestep embedding, every block's adaLN modulation and the high-level gate logits depend on (t_k, caption)
    // only: tabulate them for ALL steps in four launches instead of four GEMVs inside every network evaluation.
    const bool tab = n_steps <= PRE_STEPS;
    const int D = c.hidden, MODW = s.MODW;
    if (tab) {
        const vb_dit_weights& w = ctx->w;
        CondL cd = carve_cond(const_cast<void*>(cond), c, B, n_branch, T, L);
        VB_TRY(launch_gemv_rows_idx(w.t_freq_table, 256, s.t_table, nullptr, 0, 1, w.t_mlp0_w, w.t_mlp0_b, n_steps, D, 256, 0, s.temb0_s, D, st, T_FREQ_ROWS));
        VB_TRY(launch_gemv_rows(s.temb0_s, D, nullptr, 0, 1, w.t_mlp2_w, w.t_mlp2_b, n_steps, D, D, 1, s.temb_s, D, st));
        VB_TRY(launch_iota_div(s.row_step, n_steps * Beff, Beff, st));
        if (w.adaln_wp) {
            // [steps x samples][768] x [19968][768]^T in split-bf16 (fp32-class) on the MFMA GEMM: 0.2 ms instead of 2.2 ms per call
            const int rows = n_steps * Beff;
            const int64_t apl = (int64_t)rows * D;
            VB_TRY(launch_silu_sum_planes(s.temb_s, cd.cemb, rows, D, Beff, s.modA, apl, st));
            GemmArgs g;
            g.A = s.modA; g.a_plane = apl; g.lda = D; g.B = (const bf16_t*)w.adaln_wp; g.b_plane = (int64_t)MODW * D; g.ldb = D;
            g.M = rows; g.N = MODW; g.K = D; g.nseg = 3; g.epi = EPI_F32; g.bias = w.adaln_b; g.out32 = s.mod_s; g.ldc32 = MODW;
            VB_TRY(launch_gemm(g, st));
        } else {
            VB_TRY(launch_gemv_rows_idx(s.temb_s, D, s.row_step, cd.cemb, D, Beff, w.adaln_w, w.adaln_b, n_steps * Beff, MODW, D, 1, s.mod_s, MODW, st));
        }
        VB_TRY(launch_gemv_rows(s.temb_s, D, nullptr, 0, 1, w.hl_w, w.hl_b, n_steps, c.depth * 2, D, 0, s.hl_s, c.depth * 2, st));
    }
    {
        const int N = (int)s.n_tok;
        if (bucket_router_counts_ok(N) && !vb_tune().bucket_count_launch)
            VB_TRY(launch_fill_f32(reinterpret_cast<float*>(bucket_counts(s.perm, N, 0)), bucket_counts_ints(N), 0.f, st));
    }
    // FinalLayer, CFG combination, Euler update and the step counter's advance as ONE launch per step (round 5; VB_EULER_LAUNCH=1 keeps the three
    // launches: same arithmetic, bit-identical)
    // (the ping-pong count tables stay right only while EVERY block evaluation runs router(table e & 1) -> place(clears table
    //  (e + 1) & 1) with one G / N / knob state - true here because fused_router / w2_pair are uniform over the blocks and the clear above is
    //  inside the captured graph; a per-block switch would need its own clear.  The device step counter s.step ends a call at n_steps with the
    //  fused Euler launch and at n_steps - 1 with the separate launches; nothing reads it after the call, launch_step_ctl resets it at k == 0.)
    const bool fuse = euler_fusable(ctx, n_branch);
    for (int k = 0; k < n_steps; ++k) {
        RoctxRange rs("euler_step");
        if (!fuse || k == 0) VB_TRY(launch_step_ctl(s.step, s.t_idx_cur, s.t_table, n_steps, Beff, k == 0, st));
        EulerFuse ef{x, cfg_scale, s.dt_table, k, s.step, s.t_idx_cur, s.t_table, n_steps};
        VB_TRY(dit_forward(ctx, x, s.t_idx_cur, cond, noise, k, s.step, B, n_branch, T, L, s.v, nullptr, ws, false,
                           tab ? s.mod_s + (size_t)k * Beff * MODW : nullptr, tab ? s.hl_s + (size_t)k * c.depth * 2 : nullptr, st, k * c.depth,
                           fuse ? &ef : nullptr));
        if (!fuse) VB_TRY(launch_euler_cfg(x, s.v, B, per, cfg_scale, s.dt_table, s.step, 0.f, n_branch == 2, st));
        if (traj) VB_HIP(hipMemcpyAsync(traj + (size_t)(k + 1) * B * per, x, (size_t)B * per * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return VB_OK;
}

int vb_sample_cfg(vb_ctx* ctx, float* x, const void* cond, int B, int n_branch, int T, int L, int n_steps,
                  const int64_t* t_idx_table, const float* dt_table, float cfg_scale, const vb_noise* noise, float* traj, void* ws,
                  void* stream) {
    if (!ctx || !ctx->dit_loaded) VB_FAIL(VB_E_STATE, "sample_cfg: DiT not loaded");
    if (n_steps < 1 || n_steps > 1024) VB_FAIL(VB_E_INVALID, "sample_cfg: n_steps=%d", n_steps);
    VB_HIP(hipSetDevice(ctx->device));
    RoctxRange rr("vb_sample_cfg");
    hipStream_t st = (hipStream_t)stream;
    const vb_dit_config& c = ctx->cfg;
    WsL s = carve_ws(ws, c, B, n_branch, T, L);
    const int64_t per = (int64_t)c.in_channels * T;
    // (tables may live on the host or on the device; a host caller must keep them alive until the stream has consumed them)
    VB_HIP(hipMemcpyAsync(s.t_table, t_idx_table, (size_t)n_steps * sizeof(int64_t), hipMemcpyDefault, st));
    VB_HIP(hipMemcpyAsync(s.dt_table, dt_table, (size_t)n_steps * sizeof(float), hipMemcpyDefault, st));
    if (traj) VB_HIP(hipMemcpyAsync(traj, x, (size_t)B * per * sizeof(float), hipMemcpyDeviceToDevice, st));
    // the noise key travels through device memory (the router reads it behind the step counter): nothing a replayed graph bakes in
    VB_TRY(launch_sampler_params(s.step, noise ? noise->seed : 0, noise ? noise->clip_base : 0, noise ? noise->nfe : 0, st));

    // ---- the step loop as ONE hipGraph (cfm1_audio_sampler.py:107-116 is ~3000 dependent launches at 50 steps): a call whose
    // buffers and shape were seen before replays the captured, instantiated graph - one host call instead of thousands, which is
    // what bounds small batches and several concurrent streams (the HIP runtime serialises launches of different host threads).
    // Eager when: a trajectory or injected noise arrays are requested (parity path), the per-launch HIP-event profiler is on, the
    // stream cannot capture (legacy default stream), VB_NO_GRAPH is set, or the key is new (its first call also warms every
    // kernel's one-time attributes outside a capture).
    const bool graphable = !vb_tune().no_graph && g_prof_mask == 0 && !traj && !(noise && noise->g1) && n_steps <= PRE_STEPS && st != nullptr;
    if (graphable) {
        SampleGraph* e = nullptr;
        for (SampleGraph& gph : ctx->graphs)
            if (gph.x == x && gph.cond == cond && gph.ws == ws && gph.B == B && gph.nb == n_branch && gph.T == T && gph.L == L &&
                gph.n_steps == n_steps && gph.cfg_scale == cfg_scale && gph.tune_gen == vb_tune_generation()) { e = &gph; break; }
        if (!e) {
            if (ctx->graphs.size() >= 8) {          // evict the least recently used entry
                size_t lru = 0;
                for (size_t i = 1; i < ctx->graphs.size(); ++i) if (ctx->graphs[i].last_use < ctx->graphs[lru].last_use) lru = i;
                if (ctx->graphs[lru].exec) (void)hipGraphExecDestroy(ctx->graphs[lru].exec);
                if (ctx->graphs[lru].graph) (void)hipGraphDestroy(ctx->graphs[lru].graph);
                ctx->graphs.erase(ctx->graphs.begin() + lru);
            }
            SampleGraph n; n.x = x; n.cond = cond; n.ws = ws; n.B = B; n.nb = n_branch; n.T = T; n.L = L; n.n_steps = n_steps; n.cfg_scale = cfg_scale;
            n.tune_gen = vb_tune_generation();      // a captured graph bakes the knob-dependent kernel selection in
            ctx->graphs.push_back(n);
            e = &ctx->graphs.back();
        }
        e->last_use = ++ctx->graph_clock;
        e->seen += 1;
        if (!e->exec && !e->failed && e->seen >= 2) {
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = sample_steps(ctx, x, cond, B, n_branch, T, L, n_steps, cfg_scale, noise, nullptr, ws, st);
                hipGraph_t gr = nullptr;
                const hipError_t ee = hipStreamEndCapture(st, &gr);
                if (rc == VB_OK && ee == hipSuccess && gr && hipGraphInstantiate(&e->exec, gr, nullptr, nullptr, 0) == hipSuccess) {
                    e->graph = gr;
                } else {
                    if (gr) (void)hipGraphDestroy(gr);
                    e->exec = nullptr; e->failed = true;
                    (void)hipGetLastError();
                }
            } else {
                e->failed = true;
                (void)hipGetLastError();
            }
        }
        if (e->exec) {
            VB_HIP(hipGraphLaunch(e->exec, st));
            return VB_OK;
        }
    }
    return sample_steps(ctx, x, cond, B, n_branch, T, L, n_steps, cfg_scale, noise, traj, ws, st);
}
int vb_sample_graphs(vb_ctx* ctx) {
    int n = 0;
    if (ctx) for (const SampleGraph& g : ctx->graphs) n += g.exec != nullptr;
    return n;
}

// ---- T5 text encoder -------------------------------------------------------------------------------
struct T5Ws { float* h; bf16_t* nrm; bf16_t* qkv; bf16_t* att; bf16_t* ff; size_t total; };
static T5Ws carve_t5(void* base, const vb_t5_config& c, int B, int L) {
    T5Ws o;
    Carver cv(base);
    const size_t R = (size_t)B * L, inner = (size_t)c.heads * c.d_kv;
    o.h = cv.take<float>(R * c.d_model);
    o.nrm = cv.take<bf16_t>(2 * R * c.d_model);
    o.qkv = cv.take<bf16_t>(2 * R * 3 * inner);
    o.att = cv.take<bf16_t>(2 * R * inner);
    o.ff = cv.take<bf16_t>(2 * R * c.d_ff);
    o.total = cv.off;
    return o;
}
int vb_t5_load(vb_ctx* ctx, const vb_t5_config* cfg, const vb_t5_weights* w) {
    if (!ctx || !cfg || !w) VB_FAIL(VB_E_INVALID, "t5_load: null argument");
    if (cfg->layers < 1 || cfg->layers > VB_T5_MAX_LAYERS) VB_FAIL(VB_E_INVALID, "t5_load: layers %d", cfg->layers);
    if (cfg->d_kv != 64) VB_FAIL(VB_E_INVALID, "t5_load: d_kv %d unsupported (attention kernel is built for 64)", cfg->d_kv);
    if (cfg->d_model % 64 || cfg->d_ff % 64 || cfg->d_model > 1024) VB_FAIL(VB_E_INVALID, "t5_load: d_model %d / d_ff %d", cfg->d_model, cfg->d_ff);
    ctx->t5cfg = *cfg;
    ctx->t5w = *w;
    ctx->t5_loaded = true;
    return VB_OK;
}
size_t vb_t5_workspace_bytes(const vb_t5_config* cfg, int B, int L) { return carve_t5(nullptr, *cfg, B, L).total; }
int vb_t5_encode(vb_ctx* ctx, const int64_t* ids, int B, int L, float* out, void* ws, void* stream) {
    if (!ctx || !ctx->t5_loaded) VB_FAIL(VB_E_STATE, "t5_encode: T5 not loaded");
    VB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const vb_t5_config& c = ctx->t5cfg;
    const vb_t5_weights& w = ctx->t5w;
    if (L > w.pos_len) VB_FAIL(VB_E_INVALID, "t5_encode: L=%d exceeds the position-bias table (%d)", L, w.pos_len);
    T5Ws s = carve_t5(ws, c, B, L);
    const int R = B * L, D = c.d_model, inner = c.heads * c.d_kv;
    VB_TRY(launch_gather_rows(ids, w.embed, R, D, c.vocab, s.h, st));
    Planes nrm = mkp(s.nrm, (int64_t)R * D, 2), qkv = mkp(s.qkv, (int64_t)R * 3 * inner, 2), att = mkp(s.att, (int64_t)R * inner, 2);
    Planes ff = mkp(s.ff, (int64_t)R * c.d_ff, 2);
    for (int i = 0; i < c.layers; ++i) {
        const vb_t5_layer& lw = w.layers[i];
        // x += O(softmax(Q K^T + bias) V),  Q/K/V from T5LayerNorm(x)   (T5LayerSelfAttention)
        VB_TRY(launch_rmsnorm_mod(s.h, lw.ln0, nullptr, nullptr, 0, R, D, L, c.eps, nrm, st));
        GemmArgs g;
        g.A = nrm.p; g.a_plane = nrm.plane; g.lda = D; g.B = (const bf16_t*)lw.wqkv; g.b_plane = (int64_t)3 * inner * D; g.ldb = D;
        g.M = R; g.N = 3 * inner; g.K = D; g.nseg = 3; g.epi = EPI_PLANES; g.out = qkv; g.ldc = 3 * inner;
        VB_TRY(launch_gemm(g, st));
        VB_TRY(launch_t5_attention(qkv, w.pos_bias, w.pos_len, B, L, c.heads, c.d_kv, att, st));
        g = GemmArgs();
        g.A = att.p; g.a_plane = att.plane; g.lda = inner; g.B = (const bf16_t*)lw.wo; g.b_plane = (int64_t)D * inner; g.ldb = inner;
        g.M = R; g.N = D; g.K = inner; g.nseg = 3; g.epi = EPI_RESID_GATE; g.out32 = s.h; g.ldc32 = D; g.gate = w.ones; g.gate_ld = 0; g.T = L;
        VB_TRY(launch_gemm(g, st));
        // x += Wo(gelu_new(Wi0 n) * Wi1 n),  n = T5LayerNorm(x)   (T5LayerFF, gated-gelu)
        VB_TRY(launch_rmsnorm_mod(s.h, lw.ln1, nullptr, nullptr, 0, R, D, L, c.eps, nrm, st));
        g = GemmArgs();
        g.A = nrm.p; g.a_plane = nrm.plane; g.lda = D; g.B = (const bf16_t*)lw.wi; g.b_plane = (int64_t)2 * c.d_ff * D; g.ldb = D;
        g.M = R; g.N = 2 * c.d_ff; g.K = D; g.nseg = 3; g.epi = EPI_GEGLU; g.out = ff; g.ldc = c.d_ff;
        VB_TRY(launch_gemm(g, st));
        g = GemmArgs();
        g.A = ff.p; g.a_plane = ff.plane; g.lda = c.d_ff; g.B = (const bf16_t*)lw.wo_ff; g.b_plane = (int64_t)D * c.d_ff; g.ldb = c.d_ff;
        g.M = R; g.N = D; g.K = c.d_ff; g.nseg = 3; g.epi = EPI_RESID_GATE; g.out32 = s.h; g.ldc32 = D; g.gate = w.ones; g.gate_ld = 0; g.T = L;
        VB_TRY(launch_gemm(g, st));
    }
    VB_TRY(launch_rmsnorm_mod(s.h, w.final_ln, nullptr, nullptr, 0, R, D, L, c.eps, nrm, st));
    VB_TRY(launch_planes_to_f32(nrm, (int64_t)R * D, out, st));
    return VB_OK;
}

// ---- log-mel front-end ------------------------------------------------------------------------------
static inline int mel_im_off(const vb_mel_config& c) { return (c.n_fft / 2 + 1 + 3) / 4 * 4; }
struct MelWs { float* X; float* spec; int T, J, pad, pad2; size_t total; };
static MelWs carve_mel(void* base, const vb_mel_config& c, int B, int L, int center) {
    MelWs o;
    Carver cv(base);
    o.pad = (c.n_fft - c.hop) / 2; o.pad2 = center ? c.n_fft / 2 : 0;
    const int pad = o.pad + o.pad2;
    o.T = L + 2 * pad >= c.n_fft ? 1 + (L + 2 * pad - c.n_fft) / c.hop : 0;
    o.J = o.T + c.n_fft / c.hop - 1;
    o.X = cv.take<float>((size_t)B * c.hop * o.J);
    o.spec = cv.take<float>((size_t)B * o.T * 2 * mel_im_off(c));
    o.total = cv.off;
    return o;
}
int vb_melnet_load(vb_ctx* ctx, const vb_mel_config* cfg, const float* dft_w, const float* basis_t) {
    if (!ctx || !cfg || !dft_w || !basis_t) VB_FAIL(VB_E_INVALID, "melnet_load: null argument");
    if (cfg->hop < 1 || cfg->n_fft < cfg->hop || cfg->n_fft % cfg->hop || cfg->n_fft % 2 || (cfg->n_fft - cfg->hop) % 2)
        VB_FAIL(VB_E_INVALID, "melnet_load: n_fft %d must be an even multiple of the hop %d (and n_fft - hop even)", cfg->n_fft, cfg->hop);
    if (cfg->n_fft / cfg->hop - 1 > 64) VB_FAIL(VB_E_INVALID, "melnet_load: n_fft / hop = %d too large", cfg->n_fft / cfg->hop);
    if (cfg->n_mels < 1 || cfg->n_mels > 256) VB_FAIL(VB_E_INVALID, "melnet_load: n_mels %d", cfg->n_mels);
    ctx->melcfg = *cfg; ctx->mel_dft = dft_w; ctx->mel_basis_t = basis_t; ctx->mel_loaded = true;
    return VB_OK;
}
int vb_melnet_frames(const vb_mel_config* cfg, int L, int center) { return cfg ? carve_mel(nullptr, *cfg, 1, L, center).T : 0; }
size_t vb_melnet_workspace_bytes(const vb_mel_config* cfg, int B, int L, int center) { return cfg ? carve_mel(nullptr, *cfg, B, L, center).total : 0; }
int vb_melnet_forward(vb_ctx* ctx, const float* wav, int B, int L, int center, float* mel, float* spec, void* ws, void* stream) {
    if (!ctx || !ctx->mel_loaded) VB_FAIL(VB_E_STATE, "melnet_forward: front-end not loaded");
    if (!wav || !ws || (!mel && !spec) || B < 1) VB_FAIL(VB_E_INVALID, "melnet_forward: bad argument");
    VB_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const vb_mel_config& c = ctx->melcfg;
    MelWs s = carve_mel(ws, c, B, L, center);
    if (s.T < 1) VB_FAIL(VB_E_INVALID, "melnet_forward: %d samples (+ 2 x %d) are shorter than one %d-sample frame", L, s.pad + s.pad2, c.n_fft);
    const int nb = c.n_fft / 2 + 1, im_off = mel_im_off(c), Co4 = 2 * im_off;
    float* sp = spec ? spec : s.spec;
    VB_TRY(launch_stft_frames(wav, B, L, c.hop, s.pad, s.pad2, s.J, s.X, st));
    ConvArgs a;
    a.x = s.X; a.x_bstride = (int64_t)c.hop * s.J; a.Ci = c.hop; a.T_in = s.J; a.w = ctx->mel_dft; a.Co = Co4; a.ksize = c.n_fft / c.hop;
    a.dil = 1; a.pad = 0; a.out = sp; a.out_bstride = (int64_t)s.T * Co4; a.T_out = s.T; a.out_transposed = 1; a.B = B;
    VB_TRY(launch_conv1d(a, st));
    if (mel) VB_TRY(launch_mel_tail(sp, B, s.T, Co4, nb, im_off, ctx->mel_basis_t, c.n_mels, mel, st));
    return VB_OK;
}

int vb_net_load(vb_ctx* ctx, int which, const vb_net_op* ops, int n_ops, const vb_buf_desc* bufs, int n_bufs, int in_channels,
                int out_channels, int in_tmul, int out_tmul) {
    if (!ctx || which < 0 || which > 2 || !ops || n_ops < 1 || in_tmul < 1 || out_tmul < 1) VB_FAIL(VB_E_INVALID, "net_load: bad argument");
    NetProgram& n = ctx->nets[which];
    n.ops.assign(ops, ops + n_ops);
    n.bufs.assign(bufs, bufs + n_bufs);
    for (int i = 0; i < n_ops; ++i) {
        const vb_net_op& o = ops[i];
        const int ids[5] = {o.x, o.out, o.res, o.stats, o.w_buf};
        for (int id : ids)
            if (id >= n_bufs || (id < -3)) VB_FAIL(VB_E_INVALID, "net_load: op %d references buffer %d of %d", i, id, n_bufs);
    }
    n.in_ch = in_channels; n.out_ch = out_channels; n.in_tmul = in_tmul; n.out_tmul = out_tmul; n.loaded = true;
    return VB_OK;
}
size_t vb_net_workspace_bytes(vb_ctx* ctx, int which, int B, int T) {
    if (!ctx || which < 0 || which > 2 || !ctx->nets[which].loaded) return 0;
    return net_ws_bytes(ctx->nets[which], B, T, nullptr);
}
int vb_vae_decode(vb_ctx* ctx, const float* z, int B, int T, float* mel, void* ws, void* stream) {
    if (!ctx) VB_FAIL(VB_E_INVALID, "vae_decode: null ctx");
    RoctxRange rr("vb_vae_decode");
    return net_run(ctx, VB_NET_VAE, z, B, T, mel, ws, (hipStream_t)stream);
}
int vb_vae_encode(vb_ctx* ctx, const float* mel, int B, int T, float* moments, void* ws, void* stream) {
    if (!ctx) VB_FAIL(VB_E_INVALID, "vae_encode: null ctx");
    RoctxRange rr("vb_vae_encode");
    return net_run(ctx, VB_NET_VAE_ENCODER, mel, B, T, moments, ws, (hipStream_t)stream);
}
int vb_hifigan_forward(vb_ctx* ctx, const float* mel, int B, int T, float* wav, void* ws, void* stream) {
    if (!ctx) VB_FAIL(VB_E_INVALID, "hifigan_forward: null ctx");
    RoctxRange rr("vb_hifigan_forward");
    return net_run(ctx, VB_NET_VOCODER, mel, B, T, wav, ws, (hipStream_t)stream);
}
int vb_crossfade_windows(const float* parts, const int32_t* starts, int nw, int B, int C, int n, int T, float* out, void* stream) {
    if (!parts || !starts || !out || nw < 1 || B < 1 || C < 1 || n < 1 || T < 1) VB_FAIL(VB_E_INVALID, "crossfade_windows: null pointer or nw/B/C/n/T < 1");
    return launch_crossfade_windows(parts, starts, nw, B, C, n, T, out, (hipStream_t)stream);
}
int vb_hifigan_forward_chunked(vb_ctx* ctx, const float* mel, int B, int T, int chunk, int halo, float* wav, void* ws, float* scratch_in,
                               float* scratch_out, void* stream) {
    if (!ctx) VB_FAIL(VB_E_INVALID, "hifigan_forward_chunked: null ctx");
    if (chunk < 1 || halo < 0) VB_FAIL(VB_E_INVALID, "hifigan_forward_chunked: chunk=%d halo=%d", chunk, halo);
    NetProgram& n = ctx->nets[VB_NET_VOCODER];
    if (!n.loaded) VB_FAIL(VB_E_STATE, "hifigan_forward_chunked: no vocoder loaded");
    hipStream_t st = (hipStream_t)stream;
    RoctxRange rr("vb_hifigan_forward_chunked");
    if (T <= chunk + 2 * halo) return net_run(ctx, VB_NET_VOCODER, mel, B, T, wav, ws, st);
    VB_HIP(hipSetDevice(ctx->device));
    const int hop = n.out_tmul, rows_in = B * n.in_ch, rows_out = B * n.out_ch;
    for (int s = 0; s < T; s += chunk) {
        // frames [lo, hi) = the chunk with its context; every row (clip, channel) of the slice is gathered into a contiguous tensor, the
        // generator runs on it, and the samples of [s, e) are scattered into the whole-clip waveform (strided 2-D copies, no kernel)
        const int e = s + chunk < T ? s + chunk : T;
        const int lo = s - halo > 0 ? s - halo : 0, hi = e + halo < T ? e + halo : T;
        const int tc = hi - lo;
        VB_HIP(hipMemcpy2DAsync(scratch_in, (size_t)tc * sizeof(float), mel + lo, (size_t)T * sizeof(float), (size_t)tc * sizeof(float), rows_in,
                                hipMemcpyDeviceToDevice, st));
        VB_TRY(net_run(ctx, VB_NET_VOCODER, scratch_in, B, tc, scratch_out, ws, st));
        VB_HIP(hipMemcpy2DAsync(wav + (size_t)s * hop, (size_t)T * hop * sizeof(float), scratch_out + (size_t)(s - lo) * hop,
                                (size_t)tc * hop * sizeof(float), (size_t)(e - s) * hop * sizeof(float), rows_out, hipMemcpyDeviceToDevice, st));
    }
    return VB_OK;
}

// ---- unit kernels --------------------------------------------------------------------------
int vb_rmsnorm_modulate(const float* h, const float* w, const float* shift, const float* scale, int mod_ld, int rows, int D, int T,
                        float eps, void* out_planes, int np, void* stream) {
    return launch_rmsnorm_mod(h, w, shift, scale, mod_ld, rows, D, T, eps, mkp((bf16_t*)out_planes, (int64_t)rows * D, np),
                              (hipStream_t)stream);
}
int vb_router_top1(const float* logits, const float* gumbel, int N, int E, int32_t* idx, void* stream) {
    return launch_router_top1(logits, gumbel, N, E, idx, (hipStream_t)stream);
}
int vb_route_bucket_scratch_ints(int N, int E) { return bucket_scratch_ints(N, E); }
int vb_route_bucket_pairs(const int32_t* ic, const int32_t* ia, int N, int E, int32_t* group_off, int32_t* perm, int32_t* pair_off,
                          int32_t* pair_pa, void* stream) {
    if (!pair_off || !pair_pa) VB_FAIL(VB_E_INVALID, "route_bucket_pairs: null pair outputs");
    return launch_bucket(ic, ia, N, E, group_off, perm, (hipStream_t)stream, pair_off, pair_pa);
}
int vb_route_bucket(const int32_t* ic, const int32_t* ia, int N, int E, int32_t* group_off, int32_t* perm, void* stream) {
    return launch_bucket(ic, ia, N, E, group_off, perm, (hipStream_t)stream);
}
int vb_gemm_bf16(const void* A, const void* Bw, const float* bias, int M, int N, int K, int np, float* C, void* stream) {
    GemmArgs g;
    g.A = (const bf16_t*)A; g.a_plane = (int64_t)M * K; g.lda = K; g.B = (const bf16_t*)Bw; g.b_plane = (int64_t)N * K; g.ldb = K;
    g.M = M; g.N = N; g.K = K; g.nseg = np == 2 ? 3 : 1; g.epi = EPI_F32; g.bias = bias; g.out32 = C; g.ldc32 = N;
    return launch_gemm(g, (hipStream_t)stream);
}
int vb_grouped_swiglu(const void* u, const int32_t* perm, const int32_t* group_off, int G, int n_slots, const void* w13,
                      const void* w2, const float* row_scale, int D, int H, int np, void* hidden, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int nseg = np == 2 ? 3 : 1;
    GemmArgs g;
    g.A = (const bf16_t*)u; g.a_plane = (int64_t)n_slots * D; g.lda = D; g.a_rows = perm; g.B = (const bf16_t*)w13;
    g.b_plane = (int64_t)G * 2 * H * D; g.ldb = D; g.b_group_stride = (int64_t)2 * H * D; g.M = n_slots; g.N = 2 * H; g.K = D;
    g.nseg = nseg; g.ngroups = G; g.group_off = group_off; g.epi = EPI_SWIGLU; g.out = mkp((bf16_t*)hidden, (int64_t)n_slots * H, np);
    g.ldc = H;
    VB_TRY(launch_gemm(g, st));
    g = GemmArgs();
    g.A = (const bf16_t*)hidden; g.a_plane = (int64_t)n_slots * H; g.lda = H; g.B = (const bf16_t*)w2; g.b_plane = (int64_t)G * D * H;
    g.ldb = H; g.b_group_stride = (int64_t)D * H; g.M = n_slots; g.N = D; g.K = H; g.nseg = nseg; g.ngroups = G;
    g.group_off = group_off; g.epi = EPI_SCATTER_F32; g.out32 = out; g.ldc32 = D; g.rows_out = perm; g.row_scale = row_scale;
    return launch_gemm(g, st);
}
int vb_attention(const void* q, const void* k, const void* vt, const void* ky, const void* vyt, const float* cross_w, int B, int T,
                 int Tpad, int L, int Lpad, int H, int hd, int np, void* out, void* stream) {
    AttnArgs a;
    const int64_t ND = (int64_t)B * T * H * hd;
    a.q = wpl(q, ND, np); a.k = wpl(k, ND, np); a.vt = wpl(vt, (int64_t)B * H * hd * Tpad, np);
    a.ky = wpl(ky, (int64_t)B * L * H * hd, np); a.vyt = wpl(vyt, (int64_t)B * H * hd * Lpad, np);
    a.cross_w = cross_w; a.out = wpl(out, ND, np); a.B = B; a.T = T; a.Tpad = Tpad; a.L = L; a.Lpad = Lpad; a.H = H; a.hd = hd;
    a.has_self = k != nullptr; a.has_cross = ky != nullptr; a.kv_batch_mod = 0; a.scale = 1.0f / sqrtf((float)hd);
    return launch_attention(a, (hipStream_t)stream);
}
int vb_conv1d_f32(const float* x, const float* w, const float* bias, int B, int Ci, int T_in, int Co, int ksize, int dil, int pad,
                  int tr_stride, int tr_pad, int tr_k, int T_out, int in_act, float in_slope, const float* res, float* out,
                  const void* w_x3, int ci_pad, void* stream) {
    ConvArgs a;
    if (w_x3) {
        const int phases = tr_stride > 1 ? tr_stride : 1;
        const int ntaps = tr_stride > 1 ? (tr_k + tr_stride - 1) / tr_stride : ksize;
        a.wp = (const bf16_t*)w_x3; a.Ci_pad = ci_pad; a.wp_plane = (int64_t)phases * ntaps * Co * ci_pad;
    }
    a.x = x; a.x_bstride = (int64_t)Ci * T_in; a.Ci = Ci; a.T_in = T_in; a.w = w; a.bias = bias; a.Co = Co; a.ksize = ksize;
    a.dil = dil; a.pad = pad; a.in_act = in_act; a.in_slope = in_slope; a.out = out; a.out_bstride = (int64_t)Co * T_out;
    a.T_out = T_out; a.res = res; a.res_bstride = (int64_t)Co * T_out; a.B = B; a.tr_stride = tr_stride; a.tr_pad = tr_pad; a.tr_k = tr_k;
    return launch_conv1d(a, (hipStream_t)stream);
}
int vb_conv1d_f32_mf(const float* x, const float* w, const float* w_mf, const float* bias, int B, int Ci, int T_in, int Co, int ksize, int dil,
                     int pad, int T_out, int in_act, float in_slope, const float* res, float alpha, float beta, float* out, void* stream) {
    if (!x || !w || !w_mf || !out || B < 1 || T_in < 1) VB_FAIL(VB_E_INVALID, "conv1d_f32_mf: null pointer or B/T < 1");
    ConvArgs a;
    a.x = x; a.x_bstride = (int64_t)Ci * T_in; a.Ci = Ci; a.T_in = T_in; a.w = w; a.w_mf = w_mf; a.bias = bias; a.Co = Co; a.ksize = ksize;
    a.dil = dil; a.pad = pad; a.in_act = in_act; a.in_slope = in_slope; a.out = out; a.out_bstride = (int64_t)Co * T_out;
    a.T_out = T_out; a.res = res; a.res_bstride = (int64_t)Co * T_out; a.B = B; a.alpha = alpha; a.beta = beta;
    return launch_conv1d(a, (hipStream_t)stream);
}
int vb_respair_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, int B, int C, int T, int k, int dil,
                   float slope, float alpha, float beta, float* out, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !out || B < 1 || T < 1) VB_FAIL(VB_E_INVALID, "respair_f32: null pointer or B/T < 1");
    RespairF32Args a;
    a.x = x; a.out = out; a.B = B; a.C = C; a.T = T; a.k = k; a.dil = dil; a.w1 = w1; a.w2 = w2; a.b1 = b1; a.b2 = b2;
    a.slope = slope; a.alpha = alpha; a.beta = beta;
    return launch_respair_f32(a, (hipStream_t)stream);
}
int vb_respair_f32_mf(const float* x, const float* w1_mf, const float* b1, const float* w2_mf, const float* b2, int B, int C, int T, int k, int dil,
                      float slope, float alpha, float beta, float* out, void* stream) {
    if (!x || !w1_mf || !b1 || !w2_mf || !b2 || !out || B < 1 || T < 1) VB_FAIL(VB_E_INVALID, "respair_f32_mf: null pointer or B/T < 1");
    RespairF32Args a;
    a.x = x; a.out = out; a.B = B; a.C = C; a.T = T; a.k = k; a.dil = dil; a.w1 = w1_mf; a.w2 = w2_mf; a.b1 = b1; a.b2 = b2;
    a.slope = slope; a.alpha = alpha; a.beta = beta;
    return launch_respair_f32w(a, (hipStream_t)stream);
}
int vb_fill_gumbel(float* out, int B, int n_branch, int T, int width, uint64_t seed, int64_t clip_base, int nfe, int block, int gate,
                   void* stream) {
    return launch_fill_gumbel(out, B, n_branch, T, width, seed, clip_base, nfe, nullptr, block, gate, (hipStream_t)stream);
}
int vb_cast_planes(const float* x, int64_t n, void* out, int np, void* stream) {
    return launch_cast_planes(x, n, mkp((bf16_t*)out, n, np), (hipStream_t)stream);
}

}  // extern "C"
