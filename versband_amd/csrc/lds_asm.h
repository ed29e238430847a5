// LDS fragment reads that hipcc does not count (gfx950).
//
// In a kernel that has LDS-DMA (global_load_lds) in flight hipcc (ROCm 7.2) never emits a partial `s_waitcnt lgkmcnt(N)` in front of
// the first use of a ds_read result: every such wait is lgkmcnt(0) (all 150 kernels of this library that DMA their tiles show
// nothing else, the register-staged ones use lgkmcnt(1..7) freely).  A software-pipelined fragment loop - request the fragments of
// step s+1, then multiply step s - therefore waits for BOTH steps' reads in front of step s: the LDS round trip it was written to
// hide is exposed every other step.  Reads issued from inline asm are outside the compiler's bookkeeping (the guide, "what hipcc
// does not do: count its memory operations"); the kernel waits for them itself with the exact count:
//
//     lds_rd32<OFF>(v, addr);            request (addr = 32-bit LDS byte address in a VGPR, OFF = immediate byte offset < 65536)
//     LDS_WAIT(N); lds_pin(v);           at most N younger LDS operations may still be in flight; v may be used from here on
//
// lds_pin() is an empty asm that redefines v: every use of v is data-dependent on it and so stays behind the (volatile, hence
// ordered) wait.  LDS operations of one wave return in order, so "N younger ones in flight" is exact.
#pragma once
#include <stdint.h>
#include <type_traits>

__device__ __forceinline__ unsigned lds_u32(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
template <int OFF> __device__ __forceinline__ void lds_rd32(float& v, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is 16 bits");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
#define LDS_WAIT(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N))
__device__ __forceinline__ void lds_pin(float& v) { asm volatile("" : "+v"(v)); }

// 16-byte form (bf16 MFMA fragments)
typedef unsigned lds_u32x4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ void lds_rd128(lds_u32x4& v, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_pin(lds_u32x4& v) { asm volatile("" : "+v"(v)); }
// stores from inline asm: an ordinary LDS access in a kernel with LDS-DMA in flight makes hipcc wait vmcnt(0) in front of it (it must
// assume the DMA writes the same bytes) - that would drain a counted-vmcnt ring; the kernel has waited for the pieces it touches itself
template <int OFF> __device__ __forceinline__ void lds_wr128(unsigned addr, const lds_u32x4& v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_write offset is 16 bits");
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void lds_wr32(unsigned addr, float v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_write offset is 16 bits");
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF));
}
__device__ __forceinline__ float lds_lrelu_bits(unsigned u, float slope) {
    const float v = __builtin_bit_cast(float, u);
    return fmaxf(v, v * slope);
}
// LeakyReLU in place on one 16-byte quad
template <int OFF> __device__ __forceinline__ void lds_lrelu128_request(lds_u32x4& v, unsigned addr) { lds_rd128<OFF>(v, addr); }
__device__ __forceinline__ lds_u32x4 lds_lrelu128_apply(const lds_u32x4& v, float slope) {
    lds_u32x4 r;
    r.x = __builtin_bit_cast(unsigned, lds_lrelu_bits(v.x, slope)); r.y = __builtin_bit_cast(unsigned, lds_lrelu_bits(v.y, slope));
    r.z = __builtin_bit_cast(unsigned, lds_lrelu_bits(v.z, slope)); r.w = __builtin_bit_cast(unsigned, lds_lrelu_bits(v.w, slope));
    return r;
}
template <int I, int N, class F> __device__ __forceinline__ void lds_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); lds_static_for<I + 1, N>(f); }
}
