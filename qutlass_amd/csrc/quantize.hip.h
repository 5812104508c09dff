// Fused rotate + quantize kernels for gfx950 (HBM-bound; 2 B/elem in, ~0.53-0.66 B/elem out).
//
//   fusedQuantizeMx : y = x_g . h (bf16 x bf16 -> fp32, x viewed as (numel/R, R), h a RUNTIME RxR
//                     matrix), then per 32 values an e8m0 scale (abs-max or Quest) + 32 e2m1 codes
//                     [+ 32 clip-mask bits].  Replaces qutlass/csrc/fused_quantize_mx.cu:64-207,
//                     fused_quantize_mx_mask.cu:62-123 and the arithmetic of
//                     cutlass_extensions/epilogue/threadblock/epilogue_quant.h:460-812, :1087-1230.
//   fusedQuantizeNv : same rotation, per 16 values an e4m3 scale relative to a global scale.
//                     Replaces fused_quantize_nv.cu:109-252 / epilogue_quant.h:1560-2128.
//
// CDNA4 mapping.  The rotation is computed TRANSPOSED on the bf16 MFMA:
//   D^T (32 j x 32 rows) = H^T (32 j x 16 k) . X^T (16 k x 32 rows)   [v_mfma_f32_32x32x16_bf16]
// so that after the MFMA lane (row = l&31, half = l>>5) holds 16 of the 32 values of ONE scale
// group of ONE row (j = 8q + 4*half + e) in registers: the group reduction is 15 in-register ops
// plus ONE wavefront exchange with lane l^32 (v_permlane32_swap), every lane of the wave is busy
// (the reference leaves 3 of 4 threads idle and round-trips accumulators through shared memory),
// and the X^T operand is exactly 16 contiguous bytes of the row per lane, loaded straight from
// global memory (buffer_load_dwordx4, out-of-range rows read as zero).
#pragma once
#include "common.hip.h"

namespace qamd {

enum { METHOD_QUEST = 0, METHOD_ABSMAX = 1 };

struct QuantParams {
  const uint16_t* x;   // bf16, numel
  const uint16_t* h;   // bf16, R x R row-major
  uint8_t* out;        // packed e2m1, numel/2
  uint8_t* out_sf;     // e8m0 (MX, numel/32) or e4m3 (NV, numel/16), flat group order
  uint32_t* out_mask;  // MX quest-with-mask: one u32 per 32-group (may be null)
  const float* global_scale;  // NV only
  int64_t numel;
  int ntiles;          // ceil(numel / (max(R,32) * 32)): tiles of 32 rows x max(R,32) elements
  // BLK kernels only (fusedQuantize{Mx,Nv}Blocked): the scales go straight into the to_blocked() layout of the logical
  // (sf_rows, sf_cols) scale matrix -- sf_rows = numel / K, sf_cols = K / 32 (MX) or K / 16 (NV) -- padding zero-filled
  int sf_rows, sf_cols;
};

// byte offset of scale (row, col) in the 128x4-tiled block-scale layout (qutlass/utils.py:60-64, :190-193); CB = ceil(cols / 4)
__device__ __forceinline__ uint32_t blocked_sf_offset(uint32_t row, uint32_t col, uint32_t CB) {
  return ((row >> 7) * CB + (col >> 2)) * 512u + (row & 31u) * 16u + ((row & 127u) >> 5) * 4u + (col & 3u);
}

// --- e2m1 encoders -------------------------------------------------------------------------------
// Software RTNE-satfinite encoder (semantics of PTX cvt.rn.satfinite.e2m1x2.f32; oracle:
// orc_e2m1_encode).  Uses the fp32 adder as the rounder: adding 2^22 / 2^23 / 2^24 rounds |t| to a
// multiple of 0.5 / 1 / 2 with ties-to-even, which is exactly the e2m1 grid in [0,2) / [2,4) / [4,6].
__device__ __forceinline__ uint32_t e2m1_encode_sw(float t) {
  const uint32_t sign = (__float_as_uint(t) >> 28) & 8u;
  float a = fminf(fabsf(t), 6.0f);          // NaN -> 6 (fminf returns the non-NaN operand)
  const bool ge2 = a >= 2.0f, ge4 = a >= 4.0f;
  const float magic = ge4 ? 16777216.0f : (ge2 ? 8388608.0f : 4194304.0f);
  const uint32_t k = __float_as_uint(a + magic) - __float_as_uint(magic);
  return sign | (k + (ge4 ? 4u : (ge2 ? 2u : 0u)));
}

// Hardware converter v_cvt_scalef32_pk_fp4_f32: two fp32 -> one byte, lo -> low nibble.  Enabled only after tools/probe verified
// it against the oracle on device (scale operand 1.0).  [r3] The scale operand DIVIDES by a power of two exactly, before the
// rounding: cvt(y, 2^-sh) == cvt(ldexp(y, sh), 1.0) on 220 000 pairs incl. every rounding tie of the e2m1 grid, saturating values
// and products in the fp32 denormal range (tests/native/cvt_scale_probe.hip, profiles/cvt_scale_probe_r3.txt) -- so a
// power-of-two block scale costs no instruction of its own.
template <int BYTE>
__device__ __forceinline__ uint32_t e2m1_pack2_hw(uint32_t old, float lo, float hi, float scale = 1.0f) {
  return __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(old, lo, hi, scale, BYTE);
}

// [r5] NaN -> +inf in one instruction (v_min_f32 returns its non-NaN operand; every other value, +-inf included, passes unchanged).  The reference's convert,
// cvt.rn.satfinite.e2m1x2.f32, maps a NaN of EITHER sign to +6 = 0x7 (epilogue_quant.h:77-97; oracle orc_e2m1_encode); v_cvt_scalef32_pk_fp4_f32 keeps the NaN's
// sign (0x7 or 0xf) -- and the NaNs an MFMA rotation makes of a NaN / inf activation come out negative (tests/test_gpu_round5.py, tools/dbg_special_quant.py).
#ifndef QAMD_Q_NANFIX
#define QAMD_Q_NANFIX 1
#endif
__device__ __forceinline__ float nan_to_pinf(float v) {
#if QAMD_Q_NANFIX
  float r;
  asm("v_min_f32 %0, 0x7f800000, %1" : "=v"(r) : "v"(v));
  return r;
#else
  return v;
#endif
}

// 8 values -> one dword.  `scale` (a power of two; HWCVT only) divides the values inside the convert.  NANFIX: the forward quantizers' exact NaN code (above).
template <bool HWCVT, bool NANFIX = false>
__device__ __forceinline__ uint32_t e2m1_pack8(const float* t, float scale = 1.0f) {
  if (HWCVT) {
    uint32_t r = 0;
    if constexpr (NANFIX) {
      r = e2m1_pack2_hw<0>(r, nan_to_pinf(t[0]), nan_to_pinf(t[1]), scale);
      r = e2m1_pack2_hw<1>(r, nan_to_pinf(t[2]), nan_to_pinf(t[3]), scale);
      r = e2m1_pack2_hw<2>(r, nan_to_pinf(t[4]), nan_to_pinf(t[5]), scale);
      r = e2m1_pack2_hw<3>(r, nan_to_pinf(t[6]), nan_to_pinf(t[7]), scale);
      return r;
    }
    r = e2m1_pack2_hw<0>(r, t[0], t[1], scale);
    r = e2m1_pack2_hw<1>(r, t[2], t[3], scale);
    r = e2m1_pack2_hw<2>(r, t[4], t[5], scale);
    r = e2m1_pack2_hw<3>(r, t[6], t[7], scale);
    return r;
  } else {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r |= e2m1_encode_sw(t[i]) << (4 * i);
    return r;
  }
}

// t[0..N) = a[0..N) * s as N/2 v_pk_mul_f32 (4 cycles per wave for two products, the same issue cost as one v_mul_f32:
// tests/native/valu_probe.hip; the streaming quantisers are VALU-issue-bound at small rotation sizes)
typedef float v2f __attribute__((ext_vector_type(2)));
template <int N, typename V>
__device__ __forceinline__ void scale_pk(const V& a, int a0, float s, float* t) {
#pragma unroll
  for (int i = 0; i < N; i += 2) {
    const v2f r = v2f{a[a0 + i], a[a0 + i + 1]} * v2f{s, s};
    t[i] = r[0];
    t[i + 1] = r[1];
  }
}

// combine a per-lane partial with the partner lane l^32 (one v_permlane32_swap + one op)
__device__ __forceinline__ float xhalf_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_add(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ uint32_t xhalf_or(uint32_t v) {
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return r[0] | r[1];
}

// [r6] Quest sums of one group in the REFERENCE's order: one thread adds elements 0, 1, 2, ... in turn (epilogue_quant.h:521-528 MX, :1621-1628 NV; nvcc contracts
// `c_sum2 += c_val * c_val` into an fma).  The kernel's own order (a lane's values, then the partner half) gives sums that differ from those in the last bits -- which
// decides the SIGN of the variance of a (nearly) constant group, i.e. whether the reference takes its `var >= 0` arm (MX: scale 1.0 otherwise) or stores a NaN scale
// byte (NV).  Such groups (variance below 2^-10 of the mean square, NaN included; never on real activations) are re-summed here; NQ quads per lane, a lane holds
// elements 8 q + 4 half + e of the group (v_permlane32_swap of a value with itself leaves {half 0's, half 1's} in both lanes).
template <int NQ, typename V>
__device__ __forceinline__ void quest_sums_in_reference_order(const V& a, int a0, float& s1, float& s2) {
  s1 = 0.f;
  s2 = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    float lo[4], hi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t b = __float_as_uint(a[a0 + 4 * q + e]);
      auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
      lo[e] = __uint_as_float(r[0]);
      hi[e] = __uint_as_float(r[1]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1 += lo[e]; s2 = fmaf(lo[e], lo[e], s2); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1 += hi[e]; s2 = fmaf(hi[e], hi[e], s2); }
  }
}

// OCP e4m3fn encode of a non-negative finite fp32, RNE, saturating at 448 (oracle: orc_e4m3_encode).
__device__ __forceinline__ uint32_t e4m3_encode_pos(float a) {
  if (!(a < 448.0f)) return (a != a) ? 0x7Fu : 0x7Eu;
  // quantum 2^(max(e,-6)-3): add-magic rounding at that binade
  uint32_t u = __float_as_uint(a);
  int e = (int)(u >> 23) - 127;
  e = e < -6 ? -6 : e;
  const float magic = __uint_as_float((uint32_t)(e + 20 + 127) << 23);   // 2^(e+20): ulp = 2^(e-3)
  const float r = (a + magic) - magic;                                   // RNE to the e4m3 grid
  if (r < 0.015625f) return (uint32_t)(r * 512.0f);                      // subnormal: m * 2^-9
  const uint32_t ru = __float_as_uint(r);
  return (((ru >> 23) - 120u) << 3) | ((ru >> 20) & 7u);
}
__device__ __forceinline__ float e4m3_decode_pos(uint32_t b) {
  const uint32_t e = b >> 3, m = b & 7;
  if (b == 0x7Fu) return __uint_as_float(0x7fc00000u);   // [r5] the format's NaN (a NaN group statistic): `SF > 0` is then false and the group's multiplier 0, as in the reference
  return e ? __uint_as_float(((e + 120u) << 23) | (m << 20)) : (float)m * 0.001953125f;
}

// -------------------------------------------------------------------------------------------------
// Kernel.  One wave owns one 32-row tile at a time (grid-stride over tiles).
//   R      : rotation size (MX: 32/64/128, NV: 16/32/64/128)
//   NV     : false = MX (e8m0 per 32), true = NV (e4m3 per 16)
//   METHOD : METHOD_QUEST / METHOD_ABSMAX
//   MASK   : MX quest only: also emit the clip mask
//   HWCVT  : use v_cvt_scalef32_pk_fp4_f32 for the final RTNE
//   BLK    : scale bytes are written in the to_blocked() layout (GEMM-ready: no separate swizzle launch) instead of flat
// -------------------------------------------------------------------------------------------------
// The kernel's body as a device function of (workgroup index, workgroup count): fused_quantize_kernel below is its plain launch; [r6] the one-launch decode layer
// (gemm_mx_os.hip.h gemm_mx_os16_fq_kernel) runs it on its first few workgroups.  PAD = false: the zero padding of the blocked scale layout is left out (a reader that
// only looks at the rows it wrote).
template <int R, bool NV, int METHOD, bool MASK, bool HWCVT, bool BLK = false, bool PAD = true>
__device__ __forceinline__ void fused_quantize_body(const QuantParams& p, const int bid, const int nblk) {
  constexpr int RP = (R < 32) ? 32 : R;         // rotation padded to one MFMA j-tile (R=16: block-diag)
  constexpr int KC = RP / 16;                   // 16-wide k chunks per row
  constexpr int JT = RP / 32;                   // 32-wide j tiles per row = MX groups per row
  // [r4] R >= 64: H sits in LDS as it is in memory (row k, column j; staged with 16-byte copies) and the MFMA fragments come out of it with
  // transposing reads (ds_read_b64_tr_b16, the addressing of quartet_bwd.hip.h) -- the transposed image cost every thread R*R/256 two-byte loads
  // and as many two-byte LDS writes before the first tile could be rotated (R = 128: ~2 us of a 10 us kernel at 4096^2).  Row stride = 16 or 48
  // dwords (mod 64) keeps the transposing reads of a half wave on 64 different banks.
  constexpr bool HTR = RP >= 64;
  constexpr int HROW = HTR ? RP * 2 + 64 : RP * 2 + 16;   // H (HTR) / H^T row stride in LDS (bytes)
  __shared__ __attribute__((aligned(16))) char hT[RP * HROW];
  // R >= 64: a tile's rows are 128 / 256 bytes apart, and a load in MFMA layout (lane = row, 16 bytes) touches 32 lines
  // for 32 bytes each -- every line four to eight times over the tile's loads, from an L1 that the other waves' tiles
  // have long flushed (R = 128: 11.6 us for 4096^2 against 6.9 us at R = 32).  Those sizes load their tile as ONE
  // contiguous run (64 lanes x 16 bytes = 1 KiB per instruction), stage it in a wave-private LDS area and read the MFMA
  // fragments from there.
  constexpr bool STAGED = RP >= 64;
  constexpr int XROW = RP * 2 + 16;              // staged row stride (bytes)
  __shared__ __attribute__((aligned(16))) char xs_all[STAGED ? 4 * 32 * XROW : 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int row = lane & 31, half = lane >> 5;

  // x as rows of RP elements (for R = 16 two rotation rows share one 32-element "row")
  const int64_t ngroups = p.numel / (NV ? 16 : 32);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (uint32_t)(p.numel * 2));

  const int wave_global = bid * 4 + wave, nwaves = nblk * 4;
  v4i xnext[RP / 16];
  // per-lane byte offset inside a tile and the step between a lane's loads: MFMA layout (row, half; 32 bytes apart) or,
  // staged, chunk lane + 64 i of the tile's contiguous 32 * RP * 2 bytes
  const int lane_off = STAGED ? lane * 16 : row * RP * 2 + half * 16;
  constexpr int LSTEP = STAGED ? 1024 : 32;
  char* xs = xs_all + (STAGED ? wave * 32 * XROW : 0);
  {   // [r2] the first tile's loads go out BEFORE H is staged: the two memory round trips overlap (4096^2 cold: 9.03 -> 8.46 us at
      // R = 32, 10.08 -> 9.31 at R = 64, 13.18 -> 12.54 at R = 128; profiles/ab_stream_ops_r2.txt)
    const int xoff0 = (wave_global < p.ntiles) ? (int)((int64_t)wave_global * 32 * RP * 2) + lane_off : 0x7f000000;
#pragma unroll
    for (int kc = 0; kc < RP / 16; ++kc) xnext[kc] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff0 + kc * LSTEP, 0, 0);
  }
  const float gscale = NV ? *p.global_scale : 1.0f;
  const uint32_t sfCB = BLK ? ((uint32_t)p.sf_cols + 3u) >> 2 : 0u;
  if (BLK && PAD) {
    // zero padding of the blocked layout (rows up to a multiple of 128, columns up to a multiple of 4), as to_blocked writes it
    const uint32_t prow = ((uint32_t)p.sf_rows + 127u) & ~127u, pcol = sfCB * 4u;
    const uint32_t n1 = (prow - (uint32_t)p.sf_rows) * pcol, cpad = pcol - (uint32_t)p.sf_cols, n2 = (uint32_t)p.sf_rows * cpad;
    // padding rows: one dword (the four columns of a column tile) per store; padding columns of real rows: bytes
    const uint32_t nd = n1 >> 2;
    for (uint32_t i = (uint32_t)bid * 256u + tid; i < nd + n2; i += (uint32_t)nblk * 256u) {
      if (i < nd) {
        const uint32_t r = (uint32_t)p.sf_rows + i / sfCB, cb = i % sfCB;
        *(uint32_t*)(p.out_sf + blocked_sf_offset(r, 4u * cb, sfCB)) = 0u;
      } else {
        const uint32_t j = i - nd, r = j / cpad, c = (uint32_t)p.sf_cols + j % cpad;
        p.out_sf[blocked_sf_offset(r, c, sfCB)] = 0;
      }
    }
  }

  // ---- H^T image in LDS: hT[j][k] = h[k][j]; R = 16 becomes blockdiag(h, h) so that one 32-wide
  //      MFMA tile rotates two adjacent 16-element rows at once
  if (R < 32) {
    // [r4] as below: all four loads of a thread before the first LDS write (the conditional load -> write loop made four trips to memory one after
    // the other: NV R = 16 paid ~3 us more than MX R = 32 per launch -- 4096^2 cold 11.0 against 8.5 us for the same bytes)
    constexpr int NE = RP * RP / 256;
    uint16_t hv[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int idx = i * 256 + tid, k = idx / RP, j = idx % RP;
      hv[i] = p.h[(k & 15) * R + (j & 15)];                      // (always in range: R x R entries)
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int idx = i * 256 + tid, k = idx / RP, j = idx % RP;
      *(uint16_t*)(hT + j * HROW + k * 2) = ((k >> 4) == (j >> 4)) ? hv[i] : (uint16_t)0;
    }
  } else {
    // ALL loads of a thread issued before the first LDS write (with load -> write per element a thread of the R = 128
    // kernel made its 64 trips to memory one after the other: 12.0 us for 4096^2 against 7.7 us at R = 32, and 7.8 us for a
    // 16-row input).  Consecutive lanes take consecutive j: 2-byte loads coalesce to whole lines, and the transposed
    // writes hT[j][k] land HROW = 2 RP + 16 bytes apart, 16 distinct banks per 16 lanes.  (16-byte loads with 8
    // transposed 2-byte writes each were tried: the writes of a wave then fall on 2 banks, 15.7 us.)
    if constexpr (HTR) {
      constexpr int NCH = RP * RP / 8 / 256, CPRH = RP / 8;          // 16-byte chunks per thread, chunks per row of h
      v4i hc[NCH];
#pragma unroll
      for (int i = 0; i < NCH; ++i) hc[i] = *(const v4i*)(p.h + (i * 256 + tid) * 8);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = i * 256 + tid;
        *(v4i*)(hT + (c / CPRH) * HROW + (c % CPRH) * 16) = hc[i];
      }
    } else {
      constexpr int NE = RP * RP / 256;
      uint16_t hv[NE];
#pragma unroll
      for (int i = 0; i < NE; ++i) hv[i] = p.h[i * 256 + tid];           // idx = i * 256 + tid = k * RP + j, R == RP here
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int idx = i * 256 + tid, k = idx / RP, j = idx % RP;
        *(uint16_t*)(hT + j * HROW + k * 2) = hv[i];
      }
    }
  }
  __syncthreads();

  // BLK: (sf_row, sf_rem) = this lane's RP-element row r_abs = tile * 32 + row as (logical row, RP-row within it); one division
  // here, then a carry-propagating add per tile
  constexpr uint32_t SF_GPR = NV ? 2 * JT : JT;                       // scale groups per RP-element row
  uint32_t sf_row = 0, sf_rem = 0, sf_qstep = 0, sf_rstep = 0, sf_rpr = 1;
  if (BLK) {
    sf_rpr = (uint32_t)p.sf_cols / SF_GPR;                            // RP-element rows per logical row (K / RP)
    const uint32_t r0 = (uint32_t)wave_global * 32u + (uint32_t)row, step = (uint32_t)nwaves * 32u;
    sf_row = r0 / sf_rpr;
    sf_rem = r0 - sf_row * sf_rpr;
    sf_qstep = step / sf_rpr;
    sf_rstep = step - sf_qstep * sf_rpr;
  }
  for (int tile = wave_global; tile < p.ntiles; tile += nwaves) {
    // R >= 64: keep H^T in LDS instead of letting the compiler hoist its R*R/256 fragments into registers across the
    // tile loop (R = 128: 128 VGPRs, 204 in total -> 2 waves per SIMD and no latency hiding; re-reading 32 KiB of LDS
    // per 8-KiB tile is nowhere near a limit).  The opaque offset makes the address loop-variant for the optimiser.
    int hoist_guard = 0;
    if (RP >= 64) asm volatile("" : "+v"(hoist_guard));
    const int64_t r_abs = (int64_t)tile * 32 + row;
    // BLK: position of this lane's RP-element row in the logical scale matrix; advanced incrementally at the end of the loop body
    // (a division per tile cost a third more VALU instructions in a kernel that is issue-bound at small R)
    const uint32_t sf_col0 = BLK ? sf_rem * SF_GPR : 0u;
    // X^T operand: lane (row, half), chunk kc -> x[r_abs][16 kc + 8 half .. +8)  (16 bytes).  Software pipeline: the
    // loads of the wave's NEXT tile are issued before this tile is computed (tiles past the end fall off the buffer
    // descriptor and read 0), so the HBM latency of tile i+1 hides behind the MFMAs / epilogue of tile i.
    v8bf xf[KC];
    if (STAGED) {
      constexpr int CPR = RP / 8;                // 16-byte chunks per row
#pragma unroll
      for (int i = 0; i < KC; ++i) {
        const int q = i * 64 + lane;             // chunk of the tile held in xnext[i]
        *(v4i*)(xs + (q / CPR) * XROW + (q % CPR) * 16) = xnext[i];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the wave's own writes landed (wave-private area)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) xf[kc] = *(const v8bf*)(xs + row * XROW + (2 * kc + half) * 16);
    } else {
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) xf[kc] = __builtin_bit_cast(v8bf, xnext[kc]);
    }
    {
      const int64_t on = (int64_t)(tile + nwaves) * 32 * RP * 2 + lane_off;
      const int xoffn = (tile + nwaves < p.ntiles) ? (int)on : 0x7f000000;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) xnext[kc] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoffn + kc * LSTEP, 0, 0);
    }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      v16f acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        v8bf hf;
        if constexpr (HTR) {   // column j = 32 jt + row of h, rows k = 16 kc + 8 half .. + 7: two transposing reads of 4 rows each
          typedef short v4s_ __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(3))) v4s_* lds_v4s_t;
          const char* tp = hT + hoist_guard + (8 * half + ((lane & 15) >> 2) + 16 * kc) * HROW + (((lane & 31) >> 4) * 16 + (lane & 3) * 4) * 2 + jt * 64;
          const v4s_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)tp);
          const v4s_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(tp + 4 * HROW));
          const v8u16 hv8 = {(uint16_t)lo[0], (uint16_t)lo[1], (uint16_t)lo[2], (uint16_t)lo[3], (uint16_t)hi[0], (uint16_t)hi[1], (uint16_t)hi[2], (uint16_t)hi[3]};
          hf = __builtin_bit_cast(v8bf, hv8);
        } else {
          hf = *(const v8bf*)(hT + hoist_guard + (jt * 32 + row) * HROW + (kc * 16 + half * 8) * 2);
        }
        if constexpr (R < 32) {
          // [r5] R = 16: the two 16-element rows that share this 32-wide tile get an accumulator EACH.  Summed into one (rounds 1-4), the zero blocks of
          // blockdiag(h, h) met the OTHER row's inputs -- 0 x NaN / 0 x inf = NaN: one non-finite activation poisoned its neighbour's whole group, which the
          // reference's per-row GEMM cannot do (tests/test_gpu_round5.py).  Step kc feeds columns j = 16 kc .. + 15 only = registers 8 kc .. + 7 of a lane.
          const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          const v16f part = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, xf[kc], z, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 8; ++r) acc[8 * kc + r] = part[8 * kc + r];
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, xf[kc], acc, 0, 0, 0);
        }
      }
      // acc[4q+e] = y[r_abs][32 jt + 8q + 4 half + e]
      const int64_t grp32 = r_abs * JT + jt;   // 32-element group index in the flat output

      if (!NV) {
        // ------------------------------ MX: e8m0 per 32 ----------------------------------------
        float scale;
        // nan_risk: the group may hold NaNs.  A NaN activation makes ALL outputs of its rotation NaN (0 x NaN included), an inf makes them +-inf or NaN: the
        // maximum (which ignores NaNs) is then exactly 0 or inf, the variance NaN or inf -- conditions that are free to test and (all-zero groups aside) never
        // true on real activations.  Only such groups pay for the NaN -> 0x7 fix-up below (1-5 % of the kernel when done unconditionally, profiles/ab_stream_r5n_*).
        bool nan_risk;
        if (METHOD == METHOD_ABSMAX) {
          float m = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(acc[r]));
          m = xhalf_max(m);
          scale = m + 1e-8f;
          nan_risk = m == 0.f;
        } else {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            s1 += acc[r];
            s2 = fmaf(acc[r], acc[r], s2);
          }
          s1 = xhalf_add(s1);
          s2 = xhalf_add(s2);
          float mean = s1 * 0.03125f;
          float var = fmaf(-mean, mean, s2 * 0.03125f);
          if (__builtin_expect(!(var > s2 * (0.03125f * 0.0009765625f)), 0)) {   // [r6] sign of the variance at stake: the reference's summation order
            quest_sums_in_reference_order<4>(acc, 0, s1, s2);
            mean = s1 * 0.03125f;
            var = fmaf(-mean, mean, s2 * 0.03125f);
          }
          scale = 1.0f;
          if (var >= 0.f) scale = (float)((double)sqrtf(var) * (2.92247856 / 6.) + 1e-8);
          nan_risk = !(var >= 0.f);
        }
        const uint32_t e8 = (__float_as_uint(scale) >> 23) & 0xffu;   // floor to 2^e, keep exponent
        int sh = 127 - (int)e8;                                       // y / 2^(e8-127) == ldexp(y, sh)
        // [r5] an infinite scale (a group that holds +-inf, or whose sum of squares overflows): the reference divides by it -- finite / inf = 0, inf / inf = NaN -> 0x7
        // (epilogue_quant.h:546-550, :565-569).  Neither the pre-scaled multiply (inf * 2^-128 = inf -> +-6) nor the convert's own scale operand (2^128 = inf:
        // every code comes out 7) does that, so this rare arm multiplies by 0 first (0 or NaN, the same two outcomes) and converts with unit scale.
        const bool inf_scale = e8 == 255u;
        if (__builtin_expect(inf_scale, 0)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = acc[r] * 0.0f;
          sh = 0;
        }
        if (__builtin_expect(inf_scale || nan_risk, 0)) {   // NaN -> +inf -> code 0x7 (nan_to_pinf above; a NaN never passes the clip test either way)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = nan_to_pinf(acc[r]);
        }
        float t[16];
        uint32_t mbits = 0;
        float cs = 1.0f;   // scale operand of the hardware convert
        if (METHOD == METHOD_ABSMAX && !MASK) {
          // (y * 2^sh) * 3 == y * (3 * 2^sh): the power-of-two scaling is exact, so one multiply by the pre-scaled
          // constant rounds exactly like the reference's two steps (scale >= 1e-8 keeps 3 * 2^sh finite; results in the
          // denormal range quantise to 0 either way)
          scale_pk<16>(acc, 0, ldexpf(3.0f, sh), t);
        } else if (HWCVT) {
          // [r3] y / 2^(e8-127) happens inside the convert (see e2m1_pack2_hw): no v_ldexp_f32 per element.  The clip test
          // |y * 2^sh| < 6 is |y| < 6 * 2^-sh (power-of-two scaling of either side is exact).
          cs = inf_scale ? 1.0f : __uint_as_float(e8 << 23);   // e8 >= 100: scale >= 1e-8
          if (MASK) {
            const float lim = 6.0f * cs;
#pragma unroll
            for (int r = 0; r < 16; ++r) mbits |= (fabsf(acc[r]) < lim ? 1u : 0u) << (8 * (r >> 2) + 4 * half + (r & 3));
          }
          if (METHOD == METHOD_ABSMAX) {
            scale_pk<16>(acc, 0, 3.0f, t);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = acc[r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = ldexpf(acc[r], sh);
            if (MASK) mbits |= (fabsf(v) < 6.0f ? 1u : 0u) << (8 * (r >> 2) + 4 * half + (r & 3));
            if (METHOD == METHOD_ABSMAX) v = v * 3.0f;
            t[r] = v;
          }
        }
        // bytes: q-th group of 4 values -> group bytes 4q + 2 half, 4q + 2 half + 1
        const uint32_t P = e2m1_pack8<HWCVT>(t, cs);       // halfwords H[0+half], H[2+half]
        const uint32_t Q = e2m1_pack8<HWCVT>(t + 8, cs);   // halfwords H[4+half], H[6+half]
        auto sw = __builtin_amdgcn_permlane32_swap(P, Q, false, false);
        const uint32_t X = sw[0], Y = sw[1];
        // half 0: X = own P, Y = partner P ; half 1: X = partner Q, Y = own Q
        v2i o;
        o[0] = (int)((X & 0xffffu) | (Y << 16));
        o[1] = (int)((X >> 16) | (Y & 0xffff0000u));
        const bool ok = grp32 < ngroups;
        if (ok) {
          *(v2i*)(p.out + grp32 * 16 + half * 8) = o;
          if (half == 0) p.out_sf[BLK ? (int64_t)blocked_sf_offset(sf_row, sf_col0 + jt, sfCB) : grp32] = (uint8_t)e8;
        }
        if (MASK) {
          const uint32_t mm = xhalf_or(mbits);
          if (ok && half == 0) p.out_mask[grp32] = mm;
        }
      } else {
        // ------------------------------ NV: e4m3 per 16 ----------------------------------------
        // lane holds j = 8q + 4 half + e: q in {0,1} belong to 16-group 2*grp32, q in {2,3} to +1
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          float v8[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) v8[r] = acc[sub * 8 + r];
          float out_scale;
          uint32_t sfb;
          bool nan_risk;   // (as in the MX arm)
          if (METHOD == METHOD_ABSMAX) {
            float m = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) m = fmaxf(m, fabsf(v8[r]));
            m = xhalf_max(m);
            nan_risk = m == 0.f || !(m < __builtin_inff());
            float sf = gscale * (m * (1.0f / 6.0f));
            sfb = e4m3_encode_pos(sf);
            sf = e4m3_decode_pos(sfb);
            out_scale = (sf != 0.f) ? __frcp_rn(sf * __frcp_rn(gscale)) : 0.0f;
          } else {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              s1 += v8[r];
              s2 = fmaf(v8[r], v8[r], s2);
            }
            s1 = xhalf_add(s1);
            s2 = xhalf_add(s2);
            float mean = s1 * 0.0625f;
            // var < 0 can only come from fp32 rounding on a (nearly) constant group.  [r6] The reference takes sqrt of it and stores the NaN it
            // gets as the scale byte 0x7f (epilogue_quant.h:1631-1640: no guard; `scale_q > 0` is then false, the multiplier 0 and every code +-0):
            // so does this kernel -- rounds 1-5 clamped the variance at 0.  A NaN variance (NaN / inf activations) takes the same path.
            float var = fmaf(-mean, mean, s2 * 0.0625f);
            if (__builtin_expect(!(var > s2 * (0.0625f * 0.0009765625f)), 0)) {   // whether it is negative depends on the order of the additions: the reference's
              quest_sums_in_reference_order<2>(v8, 0, s1, s2);
              mean = s1 * 0.0625f;
              var = fmaf(-mean, mean, s2 * 0.0625f);
            }
            nan_risk = !(var >= 0.f);
            const float sc = (float)((double)sqrtf(var) * (2.92247856 / 6.) + 1e-8);
            sfb = e4m3_encode_pos(sc);
            const float sq = e4m3_decode_pos(sfb);
            out_scale = (sq > 0.f) ? __frcp_rn(sq) : 0.0f;
          }
          float t[8];
          scale_pk<8>(v8, 0, out_scale, t);
          if (__builtin_expect(nan_risk, 0)) {
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = nan_to_pinf(t[r]);
          }
          // 8 values = bytes {0,1,4,5} + 2*half of the 8-byte group: halfwords H[half], H[2+half]
          const uint32_t P = e2m1_pack8<HWCVT>(t);
          auto sw = __builtin_amdgcn_permlane32_swap(P, P, false, false);
          // half 0: sw[0] = own P (H0,H2), sw[1] = partner P (H1,H3)
          // half 1: sw[0] = partner P (H0,H2), sw[1] = own P (H1,H3)
          const uint32_t X = sw[0], Y = sw[1];
          const uint32_t d0 = (X & 0xffffu) | (Y << 16);          // bytes 0..3
          const uint32_t d1 = (X >> 16) | (Y & 0xffff0000u);      // bytes 4..7
          const int64_t grp16 = grp32 * 2 + sub;
          if (grp16 < ngroups) {
            *(uint32_t*)(p.out + grp16 * 8 + half * 4) = half ? d1 : d0;
            if (half == 0) p.out_sf[BLK ? (int64_t)blocked_sf_offset(sf_row, sf_col0 + 2 * jt + sub, sfCB) : grp16] = (uint8_t)sfb;
          }
        }
      }
    }
    if (BLK) {   // next tile of this wave: r_abs += nwaves * 32
      sf_rem += sf_rstep;
      const uint32_t carry = sf_rem >= sf_rpr ? 1u : 0u;
      sf_rem -= carry ? sf_rpr : 0u;
      sf_row += sf_qstep + carry;
    }
  }
}

template <int R, bool NV, int METHOD, bool MASK, bool HWCVT, bool BLK = false>
__global__ __launch_bounds__(256) void fused_quantize_kernel(const QuantParams p) {
  fused_quantize_body<R, NV, METHOD, MASK, HWCVT, BLK>(p, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace qamd
