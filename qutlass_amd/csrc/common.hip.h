// Shared device helpers for the gfx950 kernels.  CDNA4 only: wave64, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qamd {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned short v8u16 __attribute__((ext_vector_type(8)));

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Wave-uniform value the compiler can keep in an SGPR.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 128-bit raw buffer descriptor over [base, base+bytes): out-of-range lanes load 0 / drop stores.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// One LDS-DMA piece: 64 lanes x 16 B, LDS destination = lds (wave-uniform) + lane*16.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, void* lds, int voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds, 16, voffset, 0, 0, 0);
}

// fp32 -> bf16 bits, round-to-nearest-even (NaN kept quiet).
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  r = (f != f) ? (u | 0x00400000u) : r;
  return r >> 16;
}
// two fp32 -> packed bf16x2, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// XCD-aware, bijective block remap: blocks that the dispatcher places on one XCD (b % 8) get a
// contiguous range of logical ids, so the tiles they cover share A/B panels in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

}  // namespace qamd
