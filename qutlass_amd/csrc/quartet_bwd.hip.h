// QAT-backward data-prep kernels for gfx950.  All four are HBM-bound byte movers
// with a block reduction; they replace qutlass/csrc/quartet_bwd_sm120.cu:237-734 of the reference:
//
//   backward_t_bf16                    x (B,N,M) bf16          -> abs-max MXFP4 of x^T rotated per 32 along N
//   backward_qt_bf16                   MXFP4 (B,N,M) + alpha   -> the same on the dequantised operand
//   backward_bf16_square_double_mxfp8  x (m,n) bf16            -> e4m3 with ONE e8m0 per 32x32 block, emitted row- and column-wise
//   mxfp4_transpose_mxfp8              MXFP4 (m,n)             -> transposed e4m3 (n,m) with e8m0 per 32 along m
//
// CDNA4 mapping: the transposing quantisers stage a [32 n][64 m] bf16 tile per wave in LDS (coalesced 16-byte
// row loads in, 2-byte column reads out) so that the rotation runs on the same transposed bf16 MFMA as
// quantize.hip.h (lane = one output row, 16 of the 32 group values in registers, one v_permlane32_swap per
// reduction); format conversions are the hardware ones (v_cvt_scalef32_pk_bf16_fp4, v_cvt_scalef32_pk_fp8_bf16,
// v_cvt_scalef32_pk_fp4_f32), whose power-of-two scale operand performs the block scaling for free.
#pragma once
#include "common.hip.h"
#include "quantize.hip.h"

namespace qamd {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

// ----------------------------------------------------------------------------------------------------------------
// backward_t_bf16 / backward_qt_bf16
// ----------------------------------------------------------------------------------------------------------------
struct BwdTParams {
  const uint16_t* x;        // T: bf16 (B, N, M)
  const uint8_t* xq;        // QT: packed e2m1 (B, N, M/2)
  const uint8_t* xs;        // QT: e8m0 (B, N, M/32)
  const uint16_t* h;        // bf16 32 x 32, row-major
  const float* alpha;       // QT only (device scalar)
  uint8_t* out;             // e2m1 (B, M, N/2)
  uint8_t* out_sf;          // e8m0 (B, M, N/32)
  int B, N, M;
  int tiles_m;              // ceil(M / 64); B * (N/32) * tiles_m < 2^31 (host-checked): the kernel indexes tiles in 32 bits
};

// One wave = one [32 n][64 m] tile = one scale group for 64 output rows; the 8 waves of a workgroup take 8 consecutive
// scale groups (n0 .. n0 + 255) of the SAME 64 rows m, so the workgroup's output is one whole 128-byte line of e2m1 and
// 8 scale bytes per row: staged through LDS and stored as full lines (written per wave it was 16 bytes of each of 32
// lines per store instruction, and single scale bytes 128 bytes apart).
template <bool QT, bool HWCVT>
__global__ __launch_bounds__(512) void bwd_quant_t_kernel(const BwdTParams p) {
  constexpr int LROW = 64 * 2 + 16;  // LDS row stride (bytes): 16-byte aligned rows (ds_write_b128); the two lane halves read rows
                                     // 8 apart = 288 dwords = bank offset 32, so their 64-byte column runs never collide
  constexpr int HROW = 32 * 2 + 16;
  constexpr int OROW = 128 + 16;     // staged output row: 8 groups x 16 bytes + pad
  __shared__ __attribute__((aligned(16))) char tile_s[8][32 * LROW];
  __shared__ __attribute__((aligned(16))) char hT[32 * HROW];
  __shared__ __attribute__((aligned(16))) char out_s[64 * OROW];
  __shared__ __attribute__((aligned(16))) uint8_t sf_s[64 * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int row = lane & 31, half = lane >> 5;

  char* ts = tile_s[wave];
  const float alpha = QT ? *p.alpha : 1.0f;
  const int G = p.N >> 5;
  const int ngb = (G + 7) >> 3;                       // blocks of 8 scale groups
  const int ntw = p.B * p.tiles_m * ngb;              // workgroup tiles (host-checked < 2^31)
  // workgroup tile tw -> (b, tm, gb), gb fastest; 32-bit arithmetic: 64-bit div/mod is ~100 SALU ops each
  auto decode = [&](int tw, int& b, int& m0, int& g0) __attribute__((always_inline)) {
    const unsigned q1 = (unsigned)tw / (unsigned)ngb;
    g0 = (int)((unsigned)tw - q1 * (unsigned)ngb) * 8;
    b = (int)(q1 / (unsigned)p.tiles_m);
    m0 = (int)(q1 - (unsigned)b * (unsigned)p.tiles_m) * 64;
  };

  // global -> registers for one tile (software pipeline: the next tile's loads are in flight while this one is
  // rotated and quantised).  T: lane -> row lane/8 (+8 per pass), 16-byte chunk lane%8 (8 m): one pass = 8 rows x 128 B.
  // QT: lane -> row lane/2, 16-byte half (32 codes = one input scale group) + its e8m0 byte.
  v4i ld[QT ? 1 : 4];
  uint32_t ld_e = 0;
  auto load_tile = [&](int tw) __attribute__((always_inline)) {
    int b, m0, g0;
    decode(tw, b, m0, g0);
    const int g = g0 + wave;
    const bool live = tw < ntw && g < G;
    const int n0 = g * 32;
    if (!QT) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int r = ps * 8 + (lane >> 3), c = (lane & 7) * 8;
        ld[ps] = v4i{0, 0, 0, 0};
        if (live && m0 + c < p.M)   // M % 8 == 0 (host-checked): a chunk is in or out as a whole
          ld[ps] = *(const v4i*)(p.x + ((int64_t)b * p.N + n0 + r) * p.M + m0 + c);
      }
    } else {
      const int r = lane >> 1, c = (lane & 1) * 32;
      ld[0] = v4i{0, 0, 0, 0};
      ld_e = 127;
      if (live && m0 + c < p.M) {     // M % 32 == 0 (host-checked)
        const int64_t rowi = (int64_t)b * p.N + n0 + r;
        ld[0] = *(const v4i*)(p.xq + rowi * (p.M >> 1) + ((m0 + c) >> 1));
        ld_e = p.xs[rowi * (p.M >> 5) + ((m0 + c) >> 5)];
      }
    }
  };
  // [r2] T: the first tile's loads are issued BEFORE the rotation matrix is staged (the two memory round trips overlap: 13.7 -> 12.9 us
  // cold at 4096^2).  QT keeps them after it: its tile is 1/4 of the bytes and hoisting measured +4 % warm (profiles/ab_stream_ops_r2.txt)
  if (!QT) load_tile(blockIdx.x);
  {   // hT[j][k] = h[k][j]; [r2] both loads of a thread before the first LDS write (the loop form waited for each load in turn)
    uint16_t hv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) hv[i] = p.h[i * 512 + tid];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = i * 512 + tid, k = idx >> 5, j = idx & 31;
      *(uint16_t*)(hT + j * HROW + k * 2) = hv[i];
    }
  }
  __syncthreads();
  v8bf hf[2];   // H^T operand of the two K = 16 MFMAs (runtime matrix, loaded once)
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) hf[kc] = *(const v8bf*)(hT + row * HROW + (kc * 16 + half * 8) * 2);
  if (QT) load_tile(blockIdx.x);
  for (int tw = blockIdx.x; tw < ntw; tw += gridDim.x) {   // uniform over the workgroup: barriers inside are safe
    int b, m0, g0;
    decode(tw, b, m0, g0);

    // ---- stage the [32 n][64 m] bf16 tile in LDS ---------------------------------------------------------------
    if (!QT) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int r = ps * 8 + (lane >> 3), c = (lane & 7) * 8;
        *(v4i*)(ts + r * LROW + c * 2) = ld[ps];
      }
    } else {
      const int r = lane >> 1, c = (lane & 1) * 32;
      const float sc = __uint_as_float(ld_e ? (ld_e << 23) : 0x00400000u);   // 2^(e-127); e = 0 -> 2^-127 (denormal)
      v4i* d = (v4i*)(ts + r * LROW + c * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t w = (uint32_t)ld[0][q];
        v4i o;
        o[0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 0));
        o[1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 1));
        o[2] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 2));
        o[3] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 3));
        d[q] = o;
      }
    }
    load_tile(tw + gridDim.x);   // next tile's rows: in flight during the rotation / quantisation below
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes landed (wave-private tile)
    __builtin_amdgcn_wave_barrier();

#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      const int mloc = mh * 32 + row;
      // X^T operand: lane (m, half), chunk kc -> x[n0 + 16 kc + 8 half + i][m], i = 0..7
      v16f acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        v8u16 xv;
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = *(const uint16_t*)(ts + (kc * 16 + half * 8 + i) * LROW + mloc * 2);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[kc], __builtin_bit_cast(v8bf, xv), acc, 0, 0, 0);
      }
      // acc[4q+e] = y[m][8q + 4 half + e]   (quartet_bwd_sm120.cu:304-323 / :407-426)
      float amax = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(acc[r]));
      amax = xhalf_max(amax);
      float scale = QT ? amax / alpha : amax;
      const uint32_t sb = __float_as_uint(scale) & 0x7f800000u;
      scale = __uint_as_float(sb);
      const float mult = QT ? 3.0f / (scale * alpha) : 3.0f / scale;
      float tq[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) tq[r] = acc[r] * mult;
      const uint32_t P = e2m1_pack8<HWCVT>(tq);
      const uint32_t Q = e2m1_pack8<HWCVT>(tq + 8);
      auto sw = __builtin_amdgcn_permlane32_swap(P, Q, false, false);
      const uint32_t X = sw[0], Y = sw[1];
      v2i o;
      o[0] = (int)((X & 0xffffu) | (Y << 16));
      o[1] = (int)((X >> 16) | (Y & 0xffff0000u));
      *(v2i*)(out_s + mloc * OROW + wave * 16 + half * 8) = o;
      if (half == 0) sf_s[mloc * 8 + wave] = (uint8_t)(sb >> 23);
    }
    __syncthreads();   // the workgroup's [64 m][8 groups] output tile is staged (and every wave is done with its bf16 tile)
    {
      const int r = tid >> 3, pc = tid & 7;            // 512 pieces of 16 bytes: row r, group g0 + pc
      const int m = m0 + r, g = g0 + pc;
      if (m < p.M && g < G) {
        const int64_t grp = ((int64_t)b * p.M + m) * G + g;
        *(v4i*)(p.out + grp * 16) = *(const v4i*)(out_s + r * OROW + pc * 16);
      }
      if (tid < 64 && m0 + tid < p.M) {
        uint8_t* dst = p.out_sf + ((int64_t)b * p.M + m0 + tid) * G + g0;
        if ((G & 7) == 0) {
          *(v2i*)dst = *(const v2i*)(sf_s + tid * 8);   // 8-byte aligned: G % 8 == 0 and g0 % 8 == 0
        } else {
          for (int k = 0; k < 8 && g0 + k < G; ++k) dst[k] = sf_s[tid * 8 + k];
        }
      }
    }
    __syncthreads();   // staging area free for the next tile
  }
}

// ----------------------------------------------------------------------------------------------------------------
// backward_bf16_square_double_mxfp8: one workgroup = 128 x 128 elements = 4 x 4 blocks; wave w = rows 32w..32w+31.
// ----------------------------------------------------------------------------------------------------------------
struct SqParams {
  const uint16_t* x;   // bf16 (m, n)
  uint8_t* y;          // e4m3 (m, n)
  uint8_t* row_sf;     // e8m0 (m, n/32)
  uint8_t* col_sf;     // e8m0 (n, m_pad/32)
  int m, n;            // m: rows of x that exist; n a multiple of 128 (host-checked)
  int m_pad;           // rows of the outputs, a multiple of 128 >= m: rows m .. m_pad-1 are treated as zeros IN the kernel ([r3]: the
                       // reference pads x with torch.nn.functional.pad first, qutlass/__init__.py:288-290 -- a full extra copy of the operand)
};

// exponent byte of encode_e8m0_shiftm8 (quartet_bwd_sm120.cu:503-509): amax is a bf16 value held in fp32
__device__ __forceinline__ uint32_t e8m0_shift7(float amax) {
  return amax == 0.0f ? 127u : ((__float_as_uint(amax) >> 23) - 7u) & 0xffu;
}
__device__ __forceinline__ float e8m0_scale(uint32_t e) { return __uint_as_float(e ? (e << 23) : 0x00400000u); }

template <int UNIT = 0>   // (a template only so that the kernel is emitted by the one translation unit that launches it)
__global__ __launch_bounds__(256) void bwd_square_double_mxfp8_kernel(const SqParams p) {
  __shared__ uint8_t es[4][4];   // [wave = row block][column block]
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int tiles_n = p.n >> 7;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  const int r0 = ti * 128 + wave * 32, c0 = tj * 128;
  // lane -> row lane/16 (+4 per pass), 16-byte chunk lane%16 (8 columns); column block j = (lane%16)/4
  const int lr = lane >> 4, lc = (lane & 15) * 8;
  v4i v[8];
  float amax = 0.f;
#pragma unroll
  for (int ps = 0; ps < 8; ++ps) {
    const int row = r0 + ps * 4 + lr;
    v[ps] = row < p.m ? *(const v4i*)(p.x + (int64_t)row * p.n + c0 + lc) : v4i{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t w = (uint32_t)v[ps][q];
      amax = fmaxf(amax, fmaxf(fabsf(__uint_as_float(w << 16)), fabsf(__uint_as_float(w & 0xffff0000u))));
    }
  }
  // reduce over the lanes of the same column block: lane = 16 a + 4 j + c  ->  xor 1, 2 (c) and 16, 32 (a)
  amax = fmaxf(amax, __shfl_xor(amax, 1));
  amax = fmaxf(amax, __shfl_xor(amax, 2));
  amax = fmaxf(amax, __shfl_xor(amax, 16));
  amax = xhalf_max(amax);
  const uint32_t e = e8m0_shift7(amax);
  const float qs = e8m0_scale(e);
#pragma unroll
  for (int ps = 0; ps < 8; ++ps) {
    // (hipcc 7.2 folds the four conversions of one v4i into two when the sources are vector elements: it selects the
    //  same source register for both halves -- keep the sources as opaque scalars)
    int s0 = v[ps][0], s1 = v[ps][1], s2 = v[ps][2], s3 = v[ps][3];
    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    i16x2 lo = {0, 0}, hi = {0, 0};
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2, s0), qs, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2, s1), qs, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2, s2), qs, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2, s3), qs, true);
    const v2i o = {__builtin_bit_cast(int, lo), __builtin_bit_cast(int, hi)};
    *(v2i*)(p.y + (int64_t)(r0 + ps * 4 + lr) * p.n + c0 + lc) = o;
  }
  // scales: lanes 0, 4, 8, 12 hold the exponents of column blocks 0..3 of this wave's row block
  const uint32_t e0 = __shfl(e, 0), e1 = __shfl(e, 4), e2 = __shfl(e, 8), e3 = __shfl(e, 12);
  const uint32_t packed = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
  if (lane < 32) *(uint32_t*)(p.row_sf + (int64_t)(r0 + lane) * (p.n >> 5) + tj * 4) = packed;
  if (lane == 0) *(uint32_t*)&es[wave][0] = packed;
  __syncthreads();
  if (tid < 128) {   // column c0 + tid: the four row blocks of this workgroup are 4 consecutive bytes
    const int j = tid >> 5;
    const uint32_t col = es[0][j] | (es[1][j] << 8) | (es[2][j] << 16) | (es[3][j] << 24);
    *(uint32_t*)(p.col_sf + (int64_t)(c0 + tid) * (p.m_pad >> 5) + ti * 4) = col;
  }
}

// ----------------------------------------------------------------------------------------------------------------
// mxfp4_transpose_mxfp8: one workgroup = 128 m x 256 n; wave w dequantises rows 32w..32w+31 into LDS as bf16,
// then every lane owns 4 columns: 32 strided 2-byte reads, amax, requantise, 32 contiguous output bytes.
// ----------------------------------------------------------------------------------------------------------------
struct TrParams {
  const uint8_t* xq;   // packed e2m1 (m, n/2)
  const uint8_t* xs;   // e8m0 (m, n/32)
  uint8_t* y;          // e4m3 (n, m)
  uint8_t* out_sf;     // e8m0 (n, m_pad/32)
  int m, n;            // m: rows of x_fp4 / scales that exist; n % 256 == 0 (host-checked)
  int m_pad;           // row extent of the outputs (y is (n, m_pad)), a multiple of 128 >= m: rows m .. m_pad-1 count as zero codes with
                       // unit scales IN the kernel ([r3]: the reference pads x_fp4 with a copy and writes 1.0 into the caller's scale
                       // tensor first, qutlass/__init__.py:299-307 "TODO: padding in kernel")
};

// NC = columns (n) per workgroup: 256, or 128 (half the LDS, 4 workgroups per CU: the kernel is one round of workgroups,
// so its duration is ONE workgroup's load -> transpose -> store latency chain, and smaller tiles shorten it)
template <int NC>
__global__ __launch_bounds__(256) void mxfp4_transpose_mxfp8_kernel(const TrParams p) {
  constexpr int LROW = NC * 2 + 16;    // bf16 row of NC columns + pad
  constexpr int LPR = NC / 32;         // lanes per input row (16 bytes = 32 codes = one input scale group each)
  constexpr int RPP = 64 / LPR;        // rows per load pass
  constexpr int CPL = NC / 64;         // columns per lane
  __shared__ __attribute__((aligned(16))) char ts_all[4][32 * LROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int tiles_n = p.n / NC;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  const int r0 = ti * 128 + wave * 32, c0 = tj * NC;
  char* ts = ts_all[wave];
#pragma unroll
  for (int ps = 0; ps < 32 / RPP; ++ps) {
    const int r = ps * RPP + lane / LPR, c = (lane % LPR) * 32;
    const int64_t rowi = r0 + r;
    const bool live = rowi < p.m;
    const v4i v = live ? *(const v4i*)(p.xq + rowi * (p.n >> 1) + ((c0 + c) >> 1)) : v4i{0, 0, 0, 0};
    const float sc = live ? e8m0_scale(p.xs[rowi * (p.n >> 5) + ((c0 + c) >> 5)]) : 1.0f;
    v4i* d = (v4i*)(ts + r * LROW + c * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t w = (uint32_t)v[q];
      v4i o;
      o[0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 0));
      o[1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 1));
      o[2] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 2));
      o[3] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 3));
      d[q] = o;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  // columns lane, lane + 64, ... (consecutive lanes = consecutive 2-byte LDS addresses)
  v4i oq[CPL][2];
  uint8_t oe[CPL];
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    const int col = cc * 64 + lane;
    uint32_t pr[16];   // pr[i] = bf16 of rows 2i (low half) and 2i+1 (high half)
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t a = *(const uint16_t*)(ts + (2 * i) * LROW + col * 2);
      const uint32_t b = *(const uint16_t*)(ts + (2 * i + 1) * LROW + col * 2);
      pr[i] = a | (b << 16);
      amax = fmaxf(amax, fmaxf(fabsf(__uint_as_float(a << 16)), fabsf(__uint_as_float(b << 16))));
    }
    const uint32_t e = e8m0_shift7(amax);
    const float qs = e8m0_scale(e);
    v4i o[2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      i16x2 w = {0, 0};
      w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, __builtin_bit_cast(bf16x2, pr[2 * q]), qs, false);
      w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, __builtin_bit_cast(bf16x2, pr[2 * q + 1]), qs, true);
      o[q >> 2][q & 3] = __builtin_bit_cast(int, w);
    }
    oq[cc][0] = o[0];
    oq[cc][1] = o[1];
    oe[cc] = (uint8_t)e;
  }
  // The lane's 32 output bytes per column are a quarter of a 128-byte output line (the other three quarters belong to
  // the other waves) and its scale byte sits 128 bytes from its neighbour's: written straight from here that is 64
  // partial lines per store instruction (14.2 us for 4096^2, 23 % of the HBM roofline).  Stage the workgroup's
  // [NC n][128 m] fp8 tile and its [NC][4] scale bytes in LDS (the bf16 staging area is dead by now) and write whole
  // lines / one dword of scales per output row.
  __syncthreads();
  constexpr int OROW = 128 + 16;                       // staged output row: 128 m bytes + pad
  char* os = &ts_all[0][0];
  uint8_t* es = (uint8_t*)os + NC * OROW;              // [NC][4]
  static_assert(NC * OROW + NC * 4 <= 4 * 32 * LROW, "output staging fits the bf16 staging area");
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    const int col = cc * 64 + lane;
    *(v4i*)(os + col * OROW + wave * 32) = oq[cc][0];
    *(v4i*)(os + col * OROW + wave * 32 + 16) = oq[cc][1];
    es[col * 4 + wave] = oe[cc];
  }
  __syncthreads();
  const int m0 = ti * 128;
#pragma unroll
  for (int ps = 0; ps < NC / 32; ++ps) {               // NC * 8 16-byte pieces: row = piece / 8, chunk = piece % 8
    const int piece = ps * 256 + tid, row = piece >> 3, ch = piece & 7;
    const v4i v = *(const v4i*)(os + row * OROW + ch * 16);
    *(v4i*)(p.y + (int64_t)(c0 + row) * p.m_pad + m0 + ch * 16) = v;
  }
  if (tid < NC) *(uint32_t*)(p.out_sf + (int64_t)(c0 + tid) * (p.m_pad >> 5) + (m0 >> 5)) = *(const uint32_t*)(es + tid * 4);
}

}  // namespace qamd
