// Shared device/host helpers for libversband_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A "planes" tensor: bf16 hi plane (+ optional lo plane holding the rounding
// residual).  value = hi (+ lo).  np == 2 is the parity ("bf16x3") mode: every
// GEMM runs hi*hi + lo*hi + hi*lo on the bf16 MFMA pipe, which gives ~2^-17
// relative operand error (fp32-class) at 3/16 of the cost of the f32 MFMA.
struct Planes {
    bf16_t* p;
    int64_t plane;   // element offset between plane 0 and plane 1
    int np;
};

__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }          // RNE
__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }

__device__ __forceinline__ void split_bf16(float v, bf16_t& hi, bf16_t& lo) {
    hi = f2bf(v);
    lo = f2bf(v - bf2f(hi));
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (~10 instructions): the SwiGLU epilogues run it
// twice per output quad in the shadow of nothing (one wave per SIMD in the fused band-expert kernel)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- host side -------------------------------------------------------------
extern thread_local char g_vb_err[512];
#ifndef VB_OK
#define VB_OK 0
#define VB_E_INVALID (-1)
#define VB_E_HIP (-2)
#define VB_E_STATE (-3)
#endif
#define VB_FAIL(code, ...)                              \
    do {                                                \
        snprintf(g_vb_err, sizeof(g_vb_err), __VA_ARGS__); \
        return (code);                                  \
    } while (0)
#define VB_CHECK_LAUNCH()                                                                   \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess) VB_FAIL(VB_E_HIP, "%s:%d launch: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
    } while (0)
#define VB_HIP(x)                                                                           \
    do {                                                                                    \
        hipError_t e__ = (x);                                                               \
        if (e__ != hipSuccess) VB_FAIL(VB_E_HIP, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
    } while (0)
#define VB_TRY(x)                 \
    do {                          \
        int r__ = (x);            \
        if (r__ != VB_OK) return r__; \
    } while (0)

// Tuning / A-B knobs (environment variables VB_*), read ONCE per process instead of at every launch; tools and tests that flip a
// knob at run time call vb_tune_reload() (exported, not part of the product ABI) after changing the environment.
struct VbTune {
    float attn_defer_thr = 8.f;
    int router_tpw = 0, gemm_small = 11, gemm_small_tiles = 200, gemm_tile = -1, gemm_variant = 1, gemm_ablate = 0, gemm_nchunk = 0, gemm_p8 = -1, gemm_p8_mask = 0, gemm_p8_direct = 0, gemm_p8_p16 = 0;
    int conv_ablate = 0, attn_ablate = 0, attn_variant = -1, gemm_pk_f32 = 0, gemm_pk = 0, gemm_p8_ring = 0;
    bool conv_direct_epi = false, gate_unfolded = false, stem_f32 = false, band_unfused = false, score_fused = false, no_graph = false;
    int w2_pair = 1;
    bool qkv_p16_off = false, no_xcd_groups = false, qkv_vt16_off = false, rmsnorm_generic = false;
    int wide_resid = 1, big_tile_min_k = 384, conv_mf_occ = 2;
    bool proj_in_conv = false, conv_gemm_off = false, final_gemm = false, router_generic = false, band_epi_old = false;
    bool conv_f32_old = false, gemm_p8_off = false, conv_f32_rt_taps = false, bucket_count_launch = false, euler_launch = false, conv_mf_off = false;
};
const VbTune& vb_tune();
unsigned vb_tune_generation();      // bumped by vb_tune_reload(): anything that bakes knob-dependent kernel choices in (captured graphs) keys on it

// "first launch of this kernel on the CURRENT device": per-device once-flags for hipFuncSetAttribute (a process normally owns one
// GPU, but nothing here may silently depend on that)
// Several host threads (one per stream) may reach their first launch of a kernel together: exactly one of them sets the
// attribute (compare-exchange), the others wait the few microseconds until it is set - nobody launches before it is.
struct OnceFlags { std::atomic<int> state[64] = {}; };      // per device: 0 = untouched, 1 = being set, 2 = done
static inline void vb_set_max_lds_once(OnceFlags& f, const void* kernel, int bytes) {
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 63;
    if (f.state[d].load(std::memory_order_acquire) == 2) return;
    int expect = 0;
    if (f.state[d].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        f.state[d].store(2, std::memory_order_release);
        return;
    }
    while (f.state[d].load(std::memory_order_acquire) != 2) {}
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
