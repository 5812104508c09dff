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

// Grouped raster of the persistent kernels: tile t -> (tile row, tile column); groups of 4 tile rows, walked column by column (row fastest) inside a
// group, the last group has tiles_m % 4 rows.  magic = raster_magic(tiles_n) = ceil(2^32 / (4 tiles_n)), worked out by the host: t / group is then one
// s_mul_hi (exact or one too high for every t < 2^31: fixed up from the sign of the remainder), and the division by the group's row count (1 .. 4) is a
// shift or the constant reciprocal of 3 -- about 20 scalar instructions where the two 32-bit divisions took ~90, per decode; a persistent workgroup
// decodes twice before its first MFMA ([r4] profiles/ab_lib_gemm_r4bc_magic_decode.txt).
inline uint32_t raster_magic(int tiles_n) {
  const unsigned long long d = 4ull * (unsigned long long)tiles_n;
  return (uint32_t)(((1ull << 32) + d - 1) / d);
}
__host__ __device__ __forceinline__ uint32_t mulhi_u32(uint32_t a, uint32_t b) { return (uint32_t)(((unsigned long long)a * b) >> 32); }
// (host-callable so that tests/test_cabi_and_host.py can sweep it against the plain divisions through qutlass_amd_debug_raster_decode)
__host__ __device__ __forceinline__ void raster_decode(int t, int tiles_m, int tiles_n, uint32_t magic, int& tile_m, int& tile_n) {
  const int group = 4 * tiles_n;
  int gid = (int)mulhi_u32((uint32_t)t, magic);
  int rem = t - gid * group;
  if (rem < 0) { gid -= 1; rem += group; }
  const int first_m = gid * 4;
  const int left = tiles_m - first_m, gsz = left < 4 ? left : 4;
  const int q = gsz == 3 ? (int)(mulhi_u32((uint32_t)rem, 0xAAAAAAABu) >> 1) : rem >> (gsz >> 1);
  tile_m = first_m + (rem - q * gsz);
  tile_n = q;
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
