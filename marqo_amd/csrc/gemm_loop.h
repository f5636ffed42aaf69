// Building blocks shared by the bf16 GEMM main loops (gemm_bf16.hip: both operands through the LDS ring; gemm_wd.hip: the W operand global -> VGPR).
#pragma once
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

constexpr int BK = 64;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_wave_base) {
    // 16 B per lane; LDS destination = wave-uniform base (M0) + lane * 16; source = descriptor base + voff (per lane) + soff (scalar)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(uintptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read16(unsigned addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return __builtin_bit_cast(bf16x8, v);
}

// The compiler takes an inline asm's outputs as valid the moment the statement has executed, and a register-only consumer (an MFMA, a v_dot2c)
// has no ordering against the hand-placed `s_waitcnt lgkmcnt(0)` asm: the optimiser may sink it to right behind the ds_read that defines its operand —
// it did exactly that with the LN_APPLY statistics (40 v_dot2c moved into the loop latch, in front of the wait; tests/test_gemm_isa.py caught it).
// Passing the registers through an EMPTY asm behind the wait ties their consumers to it by data flow (volatile asms keep their order); no instruction.
template <class T>
__device__ __forceinline__ void landed(T& v) { asm volatile("" : "+v"(v)); }

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): every index is a compile-time constant inside f (register arrays stay
// registers; a run-time counter that the unroller has to fold first sent the fragment arrays to scratch)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

}  // namespace
