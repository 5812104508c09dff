// Byte-matrix transpose: in (K, M) row-major -> out (M, K) row-major, 1-byte elements (e4m3 codes).
// Pre-pass of matmul_mxf8_bf16_nn (the reference's ColumnMajor-A MXFP8 GEMM, gemm.cu:388-434): the
// scaled MFMA wants each lane's K run contiguous, so A^T is re-laid once (HBM-bound: 2 B per element,
// 6 us for 4096 x 4096) into a caller workspace and the TN kernel runs on it.
//
// One workgroup = 64 k-rows x 256 m-columns.  Wave w owns k = 16w..16w+15; lane l owns m = 4l..4l+3:
// every load instruction of a wave is one 256-byte row segment (whole lines), the 16 x 4 byte block is
// transposed in registers with v_perm_b32 (8 per 4x4 block) and leaves as four 16-byte stores, one
// per m-row; the four waves of a workgroup complete 64 contiguous bytes of each output row.
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

__global__ __launch_bounds__(256) void transpose_u8_kernel(const TransposeParams p) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 256 + lane * 4;
  const int k0 = blockIdx.y * 64 + w * 16;
  if (m0 >= p.M || k0 >= p.K) return;   // M % 4 == 0 and K % 16 == 0 (host-checked): blocks are whole
  uint32_t r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = *(const uint32_t*)(p.in + (size_t)(k0 + i) * p.M + m0);
  uint32_t o[4][4];   // o[q][j]: m = m0 + j, k = k0 + 4q..4q+3
#pragma unroll
  for (int q = 0; q < 4; ++q) transpose4x4_u8(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3], o[q]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v4i v = {(int)o[0][j], (int)o[1][j], (int)o[2][j], (int)o[3][j]};
    *(v4i*)(p.out + (size_t)(m0 + j) * p.K + k0) = v;
  }
}

}  // namespace qamd
