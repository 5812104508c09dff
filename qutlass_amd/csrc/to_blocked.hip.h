// Block-scale swizzle (to_blocked): (rows, cols) row-major bytes -> 128x4 tiled layout,
//   out[(rb*CB + cb)*512 + (r%32)*16 + ((r%128)/32)*4 + c%4] = in[r][c]   (zero padded)
// Replaces qutlass/utils.py:160-193 (torch path) and :16-133 (Triton kernel).  Pure byte
// permutation, HBM-bound (2 B per scale byte).
//
// One workgroup = one 128-row tile x TC/4 column tiles (TC input columns; TC = 128 for large matrices, down to 16 so
// that small ones -- a 4096 x 128 activation scale matrix is 512 KiB -- still spread over the chip: 32 workgroups of
// 16 KiB each took 4.5 us, launch latency plus 20 serial iterations per thread).  The 128 x TC-byte slab is read with
// dword loads into LDS (out-of-range -> 0), then every thread assembles 16-byte output lines (4 x ds_read_b32 from the
// rows r, r+32, r+64, r+96) and stores them fully coalesced: consecutive lanes write consecutive 16 bytes of the
// output tile stream.
#pragma once
#include "common.hip.h"

namespace qamd {

struct BlockedParams {
  const uint8_t* in;
  uint8_t* out;
  int rows, cols;   // input shape
  int RB, CB;       // ceil(rows/128), ceil(cols/4)
};

template <int TC>   // input columns (bytes) per workgroup = TC / 4 column tiles
__global__ __launch_bounds__(256) void to_blocked_kernel(const BlockedParams p) {
  constexpr int CT = TC / 4;              // column tiles per workgroup
  constexpr int LROW = TC + 4;            // LDS row stride (bytes), +4 breaks the 32-row bank pattern
  __shared__ __attribute__((aligned(16))) uint8_t slab[128 * LROW];

  const int tid = threadIdx.x;
  const int cgroups = (p.CB + CT - 1) / CT;
  const int rb = blockIdx.x / cgroups, cg = blockIdx.x % cgroups;
  const int r0 = rb * 128, c0 = cg * TC;
  const bool vec_ok = (p.cols % 4) == 0;   // dword loads stay inside a row and are 4-byte aligned

  // ---- load 128 rows x 128 bytes, 4 bytes per thread-iteration (32 lanes cover one row) ------
  for (int it = tid; it < 128 * (TC / 4); it += 256) {
    const int r = it / (TC / 4), c4 = (it % (TC / 4)) * 4;
    const int gr = r0 + r, gc = c0 + c4;
    uint32_t v = 0;
    if (gr < p.rows && gc < p.cols) {
      const uint8_t* src = p.in + (size_t)gr * p.cols + gc;
      if (vec_ok) {
        v = *(const uint32_t*)src;
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (gc + b < p.cols) v |= (uint32_t)src[b] << (8 * b);
      }
    }
    *(uint32_t*)(slab + r * LROW + c4) = v;
  }
  __syncthreads();

  // ---- emit: CT column tiles x 32 lines of 16 bytes ------------------------------------------
  for (int it = tid; it < CT * 32; it += 256) {
    const int ct = it / 32, i = it % 32;
    const int cb = cg * CT + ct;
    if (cb >= p.CB) continue;
    v4i o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = *(const int*)(slab + (q * 32 + i) * LROW + ct * 4);
    *(v4i*)(p.out + ((size_t)rb * p.CB + cb) * 512 + i * 16) = o;
  }
}

}  // namespace qamd
