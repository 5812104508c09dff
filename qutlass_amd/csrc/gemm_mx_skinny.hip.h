// Small-batch MXFP4 GEMM for gfx950 (decode shapes, M <= 32 per tile):  D[M,N] (bf16) = alpha * (A . SFA) (B . SFB)^T
//
// Replaces matmul_host_ada_mxf4_bf16_tn (qutlass/csrc/gemm_ada.cu:30-135: 16x16x256 tiles, one warp per CTA, 5 stages,
// scales UN-swizzled row-major (rows, K/32), cutlass_extensions/gemm/threadblock/mx_mma_multistage.h:418-448) and is also
// what matmul_mxf4_bf16_tn dispatches to for M <= 32 (SWZ = true: to_blocked scales).
//
// The problem is weight-bandwidth bound (0.53 B per weight element, every byte of B read once), not MFMA bound, so
// the tiled kernel's LDS staging only adds latency.  Here one workgroup owns 32 rows of B (one 32x32 MFMA tile of the
// output per 32 rows of A) and splits K over its 8 waves; a wave loads its operands straight from global memory in MFMA
// layout -- lane (row, half) takes 16-byte chunk 4*half + j of a 128-byte row segment for k-slice j, so the four
// slices of a segment consume whole cache lines -- with ALL loads of its K range issued up front, runs
// v_mfma_scale_f32_32x32x64_f8f6f4 on them (scale dword = the 4 K-block scales of the lane's row, op_sel = j), and
// the 8 partial 32x32 tiles are summed through LDS.  No barrier inside the K loop, one at the reduction.
#pragma once
#include "common.hip.h"

namespace qamd {

struct SkinnyParams {
  const uint8_t* A;      // (M, K/2)
  const uint8_t* B;      // (N, K/2)
  const uint8_t* SFA;    // SWZ: to_blocked layout; else row-major (M, K/32)
  const uint8_t* SFB;
  const float* alpha;
  uint16_t* D;           // (M, N) bf16
  int M, N, K;
  uint32_t a_bytes, b_bytes, sfa_bytes, sfb_bytes;
  int ldd;               // row stride of D in elements (= N unless the launch covers a column range of a wider D)
};

// MAP2 = false: lane (row, half g) takes chunk 4g + j for k-slice j (two separate 16-byte pieces of a line per load);
// MAP2 = true : chunk 2j + g, so one load instruction reads 32 contiguous bytes of each of its 32 lines.
template <bool SWZ, int NWAVES, int SEG = 2, bool MAP2 = false>   // SEG: 128-byte row segments (256 K elements) per wave per trip
__global__ __launch_bounds__(NWAVES * 64) void gemm_mx_skinny_kernel(const SkinnyParams p) {
  __shared__ __attribute__((aligned(16))) float part[NWAVES][32][33];
  const float alpha_k = *p.alpha;     // [r4] fetched here, not behind the K loop: the decode kernels last 4-6 us and this is a memory round trip

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int i32 = lane & 31, g = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int rowbytes = p.K >> 1;
  const int nseg = (rowbytes + 127) >> 7;          // 128-byte segments per row (the last may be half: K % 256 == 128)
  const int KB = p.K >> 5;                         // scale columns
  const int CB = (KB + 3) >> 2;

  const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A + a_off, p.a_bytes - a_off);   // rows past M / N fall off the end -> 0
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);
  const __amdgpu_buffer_rsrc_t rSA = make_rsrc(p.SFA, p.sfa_bytes), rSB = make_rsrc(p.SFB, p.sfb_bytes);
  constexpr int CSTEP = MAP2 ? 32 : 16;            // byte step between the chunks of k-slices j, j + 1
  const int voff = i32 * rowbytes + g * (MAP2 ? 16 : 64);   // row i32, chunk 4g (+ j)  |  chunk g (+ 2j)
  // scale dword of (row, segment s, half g): K-blocks 8s + 4g .. +3   (MAP2: both dwords of the segment, h = 0 / 1)
  int soffA, soffB;
  {
    const int ra = m0 + i32, rb = n0 + i32;
    const int gg = MAP2 ? 0 : g;
    if (SWZ) {
      soffA = (ra >> 7) * CB * 512 + (ra & 31) * 16 + ((ra & 127) >> 5) * 4 + gg * 512;  // + s * 1024: column tile 2s + g
      soffB = (rb >> 7) * CB * 512 + (rb & 31) * 16 + ((rb & 127) >> 5) * 4 + gg * 512;
    } else {
      soffA = ra * KB + gg * 4;                                                          // + s * 8
      soffB = rb * KB + gg * 4;
    }
  }
  constexpr int OOB = 0x7f000000;
  const bool rowA_ok = SWZ ? true : (m0 + i32 < p.M);   // row-major scales have no padding rows
  const bool rowB_ok = SWZ ? true : (n0 + i32 < p.N);

  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int s0 = wave * SEG; s0 < nseg; s0 += NWAVES * SEG) {
    v4i fa[SEG][4], fb[SEG][4];
    int sa[SEG][2], sb[SEG][2];
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
      const int s = s0 + u;
      const int base = s * 128;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // chunks past the end of the row (K tail / segment past K) must read 0, not the next row
        const int cbyte = MAP2 ? g * 16 + j * 32 : g * 64 + j * 16;
        const int v = (s < nseg && base + cbyte < rowbytes) ? voff + j * CSTEP : OOB;
        fa[u][j] = __builtin_amdgcn_raw_buffer_load_b128(rA, v, base, 0);
        fb[u][j] = __builtin_amdgcn_raw_buffer_load_b128(rB, v, base, 0);
      }
      const int step = SWZ ? s * 1024 : s * 8;
      if (MAP2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                      // dword h = K-blocks 8s + 4h .. +3; the lane uses bytes g, 2 + g
          const bool col_ok = (8 * s + 4 * h) < KB;
          const int hoff = SWZ ? h * 512 : h * 4;
          const int da = __builtin_amdgcn_raw_buffer_load_b32(rSA, (col_ok && rowA_ok) ? soffA + step + hoff : OOB, 0, 0);
          const int db = __builtin_amdgcn_raw_buffer_load_b32(rSB, (col_ok && rowB_ok) ? soffB + step + hoff : OOB, 0, 0);
          sa[u][h] = (int)((uint32_t)da >> (8 * g));
          sb[u][h] = (int)((uint32_t)db >> (8 * g));
        }
      } else {
        const bool col_ok = (8 * s + 4 * g) < KB;          // K % 128 == 0: a scale dword is in or out as a whole
        sa[u][0] = __builtin_amdgcn_raw_buffer_load_b32(rSA, (col_ok && rowA_ok) ? soffA + step : OOB, 0, 0);
        sb[u][0] = __builtin_amdgcn_raw_buffer_load_b32(rSB, (col_ok && rowB_ok) ? soffB + step : OOB, 0, 0);
        sa[u][1] = sa[u][0];
        sb[u][1] = sb[u][0];
      }
    }
    asm volatile("" :: "s"(alpha_k));   // alpha is waited for HERE, behind the operand loads just issued (left alone, its load is sunk below the loop)
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const v4i a = fa[u][j], b = fb[u][j];
        const v8i A8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, B8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
        // srcA = B fragment, srcB = A fragment (as in gemm_mx.hip.h): acc[4q+e] = D[m = i32][n = 8q + 4g + e]
        if (MAP2) {     // slice j = K-block 2j + g: byte (2j & 3) of the lane's shifted dword j >> 1
          if (j == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 0, sb[u][0], 0, sa[u][0]);
          if (j == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 2, sb[u][0], 2, sa[u][0]);
          if (j == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 0, sb[u][1], 0, sa[u][1]);
          if (j == 3) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 2, sb[u][1], 2, sa[u][1]);
        } else {
          if (j == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 0, sb[u][0], 0, sa[u][0]);
          if (j == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 1, sb[u][0], 1, sa[u][0]);
          if (j == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 2, sb[u][0], 2, sa[u][0]);
          if (j == 3) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 3, sb[u][0], 3, sa[u][0]);
        }
      }
    }
  }

  // ---- cross-wave reduction: part[wave][m][n] -------------------------------------------------------------------
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave][i32][8 * q + 4 * g + e] = acc[4 * q + e];
  __syncthreads();
  const float alpha = alpha_k;
  for (int idx = tid; idx < 32 * 8; idx += NWAVES * 64) {   // 32 rows x 8 quads of 4 columns
    const int m = idx >> 3, nq = (idx & 7) * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NWAVES; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += part[w][m][nq + e];
    if (m0 + m < p.M && n0 + nq < p.N) {   // N % 8 == 0 (host-checked): a quad is in or out as a whole
      v2i o;
      o[0] = (int)pack_bf16x2(s[0] * alpha, s[1] * alpha);
      o[1] = (int)pack_bf16x2(s[2] * alpha, s[3] * alpha);
      *(v2i*)(p.D + (size_t)(m0 + m) * p.ldd + n0 + nq) = o;
    }
  }
}

template <bool SWZ, int NW = 8, int SEG = 2, bool MAP2 = false>
inline void launch_skinny(const SkinnyParams& p, hipStream_t s) {
  hipLaunchKernelGGL((gemm_mx_skinny_kernel<SWZ, NW, SEG, MAP2>), dim3((p.N + 31) / 32, (p.M + 31) / 32), dim3(NW * 64), 0, s, p);
}

}  // namespace qamd
