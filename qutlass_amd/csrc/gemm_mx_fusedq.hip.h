// Decode-time activation path in ONE launch (M <= 32):  D[M,N] (bf16) = alpha * Q(x . h) (B . SFB)^T
//
// The reference runs three launches per linear layer -- fusedQuantizeMx (qutlass/__init__.py:149-180), to_blocked
// (qutlass/utils.py:160-193), matmul_mxf4_bf16_tn (qutlass/__init__.py:34-76) -- and at batch sizes 1..32 each of them is
// nothing but launch latency (README.md:136-148: "Actual" vs "Ideal").  Here the small-batch GEMM of gemm_mx_skinny.hip.h
// quantises its own A operand: a workgroup owns 32 rows of the weight B and splits K over its 8 waves, exactly as there; but
// instead of loading packed e2m1 + e8m0 for A, each wave loads the bf16 activations of its K range, rotates them per 32-element
// group on the bf16 MFMA (transposed, as quantize.hip.h does), derives the e8m0 scale (abs-max or Quest), rounds to e2m1 and
// hands the codes to the scaled FP4 MFMA in registers.  Every workgroup repeats the quantisation of the (tiny) activation
// matrix -- 32 x K bf16 -- which costs nothing against a launch; the weight loads of a wave are issued BEFORE the
// quantisation starts, so the HBM latency of the weight stream hides behind it.
//
// Arithmetic = fused_quantize_kernel<32, false, METHOD, false, HWCVT> (same MFMA, same operand layout per (row, group), same
// reduction order) followed by gemm_mx_skinny_kernel (same K split, same reduction): the output is bit-identical to the
// three-launch path (tests/test_gpu_round3.py).
#pragma once
#include "gemm_mx_skinny.hip.h"
#include "quantize.hip.h"

namespace qamd {

struct FusedQParams {
  const uint16_t* x;     // (M, K) bf16 activations
  const uint16_t* h;     // (32, 32) bf16 rotation, row-major (y = x_g . h)
  const uint8_t* B;      // (N, K/2) packed e2m1
  const uint8_t* SFB;    // to_blocked e8m0 scales of B
  const float* alpha;
  uint16_t* D;           // (M, N) bf16
  int M, N, K;
  uint32_t x_bytes, b_bytes, sfb_bytes;
};

// VR = activation rows per rotation tile (4, 8, 16 or 32 >= M).  The bf16 MFMA that rotates the activations always works on 32
// "rows"; with fewer real rows a tile takes 32 / VR scale groups of each row instead (virtual row v = r * (32 / VR) + G'), so a
// 256-element K segment of the activations costs 8 / (32 / VR) rotate + quantize chains instead of 8 -- at M <= 4 ONE chain per
// segment.  (The chain, ~250 VALU instructions that every workgroup repeats, is what the first version of this kernel -- lane = row,
// 8 chains per segment whatever M -- lost against two separate launches: 10.9 vs 8.2 us at M = 16, profiles/ab_blocked_quant_r3.txt.)
// The codes and scale bytes of a segment go through a wave-private LDS tile [row][group][16 bytes] and come back in the operand
// layout of the scaled FP4 MFMA: lane (row, g) reads the 16 code bytes of group 4 g + j for k-slice j and the scale dword of
// groups 4 g .. 4 g + 3; rows >= VR read zeros.
template <int METHOD, bool HWCVT, int VR, int NWAVES = 8>
__global__ __launch_bounds__(NWAVES * 64) void gemm_mx_fusedq_kernel(const FusedQParams p) {
  static_assert(VR == 4 || VR == 8 || VR == 16 || VR == 32, "rows per rotation tile");
  constexpr int GPS = 32 / VR;          // scale groups of one row per rotation tile
  constexpr int NPASS = 8 / GPS;        // rotation tiles per 256-element segment
  constexpr int CROW = 8 * 16 + 16;     // staged code row: 8 groups x 16 bytes + pad
  __shared__ __attribute__((aligned(16))) float part[NWAVES][32][33];
  const float alpha_k = *p.alpha;     // [r4] fetched here, not behind the K loop: the decode kernels last 4-6 us and this is a memory round trip
  constexpr int HROW = 32 * 2 + 16;   // padded H^T row stride (bytes), as in quantize.hip.h
  __shared__ __attribute__((aligned(16))) char hT[32 * HROW];
  __shared__ __attribute__((aligned(16))) char cs_all[NWAVES][VR * CROW];      // codes of one segment: [row][group][16]
  __shared__ __attribute__((aligned(16))) uint8_t ss_all[NWAVES][VR * 8];      // scale bytes: [row][group]

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int i32 = lane & 31, g = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int rowbytes = p.K >> 1;                   // packed B row
  const int nseg = (rowbytes + 127) >> 7;          // 256-element K segments (K % 128 == 0: the last may be half)
  const int KB = p.K >> 5, CB = (KB + 3) >> 2;
  char* cs = cs_all[wave];
  uint8_t* ss = ss_all[wave];

  const uint32_t b_off = (uint32_t)n0 * rowbytes, x_off = (uint32_t)m0 * (uint32_t)p.K * 2u;
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);    // rows past N fall off the end -> 0
  const __amdgpu_buffer_rsrc_t rX = make_rsrc((const uint8_t*)p.x + x_off, p.x_bytes - x_off);   // rows past M -> 0
  const __amdgpu_buffer_rsrc_t rSB = make_rsrc(p.SFB, p.sfb_bytes);
  const int voffB = i32 * rowbytes + g * 64;       // row i32, chunk 4g (+ j)
  // rotation tile: virtual row i32 = (activation row vr, local group vg); pass ps covers groups ps * GPS + vg
  const int vr = i32 / GPS, vg = i32 % GPS;
  const int voffX = vr * p.K * 2 + vg * 64 + g * 16;   // bytes: row vr, group vg, half g (+ 32 per kc, + GPS * 64 per pass, + 512 per segment)
  const int rb = n0 + i32;
  const int soffB = (rb >> 7) * CB * 512 + (rb & 31) * 16 + ((rb & 127) >> 5) * 4 + g * 512;   // + s * 1024: column tile 2s + g
  constexpr int OOB = 0x7f000000;

  // ---- weight loads of this wave's FIRST segment go out before anything else ------------------------------------------------
  v4i fb[4];
  int sb = 0;
  auto load_b = [&](int s) __attribute__((always_inline)) {
    const int base = s * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = (s < nseg && base + g * 64 + j * 16 < rowbytes) ? voffB + j * 16 : OOB;
      fb[j] = __builtin_amdgcn_raw_buffer_load_b128(rB, v, base, 0);
    }
    sb = __builtin_amdgcn_raw_buffer_load_b32(rSB, (s < nseg && (8 * s + 4 * g) < KB) ? soffB + s * 1024 : OOB, 0, 0);
  };
  // activations of segment s: pass ps, chunk kc -> 16 bytes per lane (the X^T operand layout of quantize.hip.h)
  v4i xr[NPASS][2];
  auto load_x = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const int k0 = s * 256 + (ps * GPS + vg) * 32;   // first element of the lane's group
        const int v = (s < nseg && k0 < p.K) ? voffX + ps * GPS * 64 + kc * 32 : OOB;
        xr[ps][kc] = __builtin_amdgcn_raw_buffer_load_b128(rX, v, s * 512, 0);
      }
  };
  int s = wave;
  load_b(s);
  load_x(s);
  asm volatile("" :: "s"(alpha_k));   // alpha is waited for HERE, behind the first loads (left alone, its load is sunk below the K loop)

  // ---- H^T image in LDS (hT[j][k] = h[k][j]), then this lane's two MFMA fragments into registers -----------------------------
  if (tid < 256) {
    uint16_t hv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hv[i] = p.h[i * 256 + tid];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = i * 256 + tid, k = idx >> 5, j = idx & 31;
      *(uint16_t*)(hT + j * HROW + k * 2) = hv[i];
    }
  }
  __syncthreads();
  v8bf hf[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) hf[kc] = *(const v8bf*)(hT + i32 * HROW + (kc * 16 + g * 8) * 2);

  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (; s < nseg; s += NWAVES) {
    // ---- rotate + quantise the segment: NPASS tiles of 32 virtual rows; codes (8 bytes per lane) and scale bytes -> LDS ----------
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      v16f y;
#pragma unroll
      for (int r = 0; r < 16; ++r) y[r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[kc], __builtin_bit_cast(v8bf, xr[ps][kc]), y, 0, 0, 0);
      // y[4q+e] = (x_g . h)[virtual row i32][8q + 4 g + e]       (quantize.hip.h, same operand layout)
      float scale;
      if (METHOD == METHOD_ABSMAX) {
        // max is order-independent: a depth-4 tree instead of a 16-long dependent chain (two waves per SIMD hide little latency)
        float t8[8], t4[4];
#pragma unroll
        for (int r = 0; r < 8; ++r) t8[r] = fmaxf(fabsf(y[2 * r]), fabsf(y[2 * r + 1]));
#pragma unroll
        for (int r = 0; r < 4; ++r) t4[r] = fmaxf(t8[2 * r], t8[2 * r + 1]);
        float m = fmaxf(fmaxf(t4[0], t4[1]), fmaxf(t4[2], t4[3]));
        m = xhalf_max(m);
        scale = m + 1e-8f;
      } else {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s1 += y[r];
          s2 = fmaf(y[r], y[r], s2);
        }
        s1 = xhalf_add(s1);
        s2 = xhalf_add(s2);
        const float mean = s1 * 0.03125f;
        const float var = fmaf(-mean, mean, s2 * 0.03125f);
        scale = 1.0f;
        if (var >= 0.f) scale = (float)((double)sqrtf(var) * (2.92247856 / 6.) + 1e-8);
      }
      const uint32_t e8 = (__float_as_uint(scale) >> 23) & 0xffu;
      const int sh = 127 - (int)e8;
      float t[16];
      if (METHOD == METHOD_ABSMAX) {
        const float f3 = ldexpf(3.0f, sh);
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = y[r] * f3;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = ldexpf(y[r], sh);
      }
      const uint32_t P = e2m1_pack8<HWCVT, true>(t), Q = e2m1_pack8<HWCVT, true>(t + 8);   // (NaN -> 0x7 like the stand-alone quantizer: quantize.hip.h nan_to_pinf)
      auto sw = __builtin_amdgcn_permlane32_swap(P, Q, false, false);
      const uint32_t X = sw[0], Y = sw[1];
      v2i o;   // bytes 8 g .. 8 g + 7 of the group's 16 code bytes
      o[0] = (int)((X & 0xffffu) | (Y << 16));
      o[1] = (int)((X >> 16) | (Y & 0xffff0000u));
      const int G = ps * GPS + vg;
      *(v2i*)(cs + vr * CROW + G * 16 + g * 8) = o;
      if (g == 0) ss[vr * 8 + G] = (uint8_t)e8;
    }
    const int snext = s + NWAVES;
    load_x(snext);                          // xr is dead: the next segment's activations
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): the wave's own LDS writes landed (wave-private tile)
    __builtin_amdgcn_wave_barrier();
    // ---- MFMA operands: lane (row i32, g), k-slice j <- the 16 code bytes of group 4 g + j, the scale dword of groups 4 g .. 4 g + 3 ----
    v4i fa[4];
    uint32_t sa = 0;
    if (i32 < VR) {
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[j] = *(const v4i*)(cs + i32 * CROW + (4 * g + j) * 16);
      sa = *(const uint32_t*)(ss + i32 * 8 + 4 * g);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[j] = v4i{0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const v4i a = fa[j], b = fb[j];
      const v8i A8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, B8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
      if (j == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 0, sb, 0, (int)sa);
      if (j == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 1, sb, 1, (int)sa);
      if (j == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 2, sb, 2, (int)sa);
      if (j == 3) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc, 4, 4, 3, sb, 3, (int)sa);
    }
    load_b(snext);
    __builtin_amdgcn_wave_barrier();        // (LDS is in order per wave: the reads above precede the next segment's writes)
  }

  // ---- cross-wave reduction and epilogue: gemm_mx_skinny_kernel's --------------------------------------------------------------
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave][i32][8 * q + 4 * g + e] = acc[4 * q + e];
  __syncthreads();
  const float alpha = alpha_k;
  for (int idx = tid; idx < 32 * 8; idx += NWAVES * 64) {
    const int m = idx >> 3, nq = (idx & 7) * 4;
    float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NWAVES; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e) sm[e] += part[w][m][nq + e];
    if (m0 + m < p.M && n0 + nq < p.N) {
      v2i ov;
      ov[0] = (int)pack_bf16x2(sm[0] * alpha, sm[1] * alpha);
      ov[1] = (int)pack_bf16x2(sm[2] * alpha, sm[3] * alpha);
      *(v2i*)(p.D + (size_t)(m0 + m) * p.N + n0 + nq) = ov;
    }
  }
}

}  // namespace qamd
