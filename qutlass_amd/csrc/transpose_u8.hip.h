// Byte-matrix transpose: in (K, M) row-major -> out (M, K) row-major, 1-byte elements (e4m3 codes).
// Pre-pass of matmul_mxf8_bf16_nn (the reference's ColumnMajor-A MXFP8 GEMM, gemm.cu:388-434): the
// scaled MFMA wants each lane's K run contiguous, so A^T is re-laid once (HBM-bound: 2 B per element)
// into a caller workspace and the TN kernel runs on it.
//
// One workgroup = 128 k-rows x 128 m-columns.  Both sides of the copy touch FEW rows per wave instruction with
// MANY bytes each (a transpose whose stores scatter 16 bytes over 64 rows thrashes the TLB: 13.8 us for 4096^2):
//   load   : lane -> 4 m-bytes of one k-row, a wave instruction = 2 rows x 128 B; 16 rows per thread
//   regs   : 4x4 byte transposes with v_perm_b32 -> for each of the lane's 4 m-columns, 4 dwords of 4 consecutive k
//   LDS    : tile [m][k] (row stride 132 B), dword writes, then row reads
//   store  : lane -> 16 bytes of one m-row, a wave instruction = 8 rows x 128 B
#pragma once
#include "common.hip.h"

namespace qamd {

struct TransposeParams {
  const uint8_t* in;   // (K, M)
  uint8_t* out;        // (M, K)
  int K, M;
};

__device__ __forceinline__ void transpose4x4_u8(const uint32_t r0, const uint32_t r1, const uint32_t r2,
                                                const uint32_t r3, uint32_t (&c)[4]) {
  // r_i = 4 consecutive columns of row i  ->  c_j = column j over rows 0..3 (byte 0 = row 0)
  const uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u);   // r0.b0 r1.b0 r0.b1 r1.b1
  const uint32_t t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);   // r0.b2 r1.b2 r0.b3 r1.b3
  const uint32_t u0 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
  const uint32_t u1 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
  c[0] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
  c[1] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
  c[2] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
  c[3] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
}

template <int UNIT = 0>   // (a template only so that the kernel is emitted by the one translation unit that launches it)
__global__ __launch_bounds__(256) void transpose_u8_kernel(const TransposeParams p) {
  constexpr int LROW = 128 + 4;   // bytes per LDS row (33 dwords)
  __shared__ __attribute__((aligned(16))) uint8_t tile[128 * LROW];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
  {
    const int mq = tid & 31, kq = tid >> 5;             // 4 m-bytes, 16 k-rows
    const int gm = m0 + 4 * mq;
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int gk = k0 + 16 * kq + i;
      r[i] = (gm < p.M && gk < p.K) ? *(const uint32_t*)(p.in + (size_t)gk * p.M + gm) : 0u;   // M % 4 == 0 (host-checked)
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t o[4];   // o[j]: m = 4 mq + j, k = 16 kq + 4q .. +3
      transpose4x4_u8(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3], o);
#pragma unroll
      for (int j = 0; j < 4; ++j) *(uint32_t*)(tile + (4 * mq + j) * LROW + 16 * kq + 4 * q) = o[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = ps * 32 + (tid >> 3), ch = (tid & 7) * 16;
    const int gm = m0 + row, gk = k0 + ch;
    if (gm < p.M && gk < p.K) {                          // K % 16 == 0 (host-checked)
      const uint32_t* s = (const uint32_t*)(tile + row * LROW + ch);
      const v4i v = {(int)s[0], (int)s[1], (int)s[2], (int)s[3]};
      *(v4i*)(p.out + (size_t)gm * p.K + gk) = v;
    }
  }
}

}  // namespace qamd
