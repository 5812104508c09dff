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
  int abl;                  // lab build only (bwd_quant_tw_kernel): leave out 1 = the global loads, 2 = the unit stores, 4 = MFMA + quantisation, 8 = staging, 16 = the scale-byte stores
};
#if QAMD_BENCH
#define QAMD_BWD_ABL(b) (p.abl & (b))
#else
#define QAMD_BWD_ABL(b) false
#endif

// One wave = one [32 n][64 m] tile = one scale group for 64 output rows; the 8 waves of a workgroup take 8 consecutive
// scale groups (n0 .. n0 + 255) of the SAME 64 rows m, so the workgroup's output is one whole 128-byte line of e2m1 and
// 8 scale bytes per row: staged through LDS and stored as full lines (written per wave it was 16 bytes of each of 32
// lines per store instruction, and single scale bytes 128 bytes apart).
//
// [r3] Instruction diet.  The first version was bound by instruction ISSUE, not by memory: at 8192^2 its 32 wave tiles per SIMD cost ~230
// vector instructions each (PMC: SQ_ACTIVE_INST_VALU = 16 of its 23-32 us; tests/native/valu_probe.hip has the per-instruction costs), and when
// those were halved the kernel did not move -- the address arithmetic had gone to the SCALAR unit, which a CU has ONE of for its four SIMDs
// (~300 scalar instructions per tile and wave, every one of the 8 waves repeating the same tile decode with its three integer divisions,
// twice).  Both are cut now:
//   * tile coordinates advance incrementally (carry-propagating adds, no division in the loop) and are computed once per tile; every address
//     is a wave-uniform 64-bit base + a per-lane 32-bit offset computed ONCE (buffer loads / stores; rows and columns past the tensor fall
//     off the descriptor) -- the per-lane 64-bit multiplies of the first version were ~25 vector instructions per tile;
//   * the X^T operand comes out of the LDS tile with ds_read_b64_tr_b16 (4 reads per 32 rows instead of 16 two-byte
//     reads + 8 v_perm_b32);
//   * no division per group: the e8m0 byte is the exponent field of amax (T) or of amax / alpha (QT), and for two
//     positive normal floats the exponent of RN(a / b) is the exponent field of the INTEGER bits(a) - bits(b) + bits(1.0)
//     (the mantissa borrow is exactly the "a's mantissa < b's" case; RN cannot carry the quotient of two 24-bit
//     mantissas up to the next power of two: the largest quotient below 2 is 2 - 2^-23, itself representable);
//     the multiplier 3 / scale (T) resp. 3 / (scale * alpha) (QT) is RN(3 / alpha) * 2^-E exactly, so the elements are
//     multiplied by the kernel-wide constant (v_pk_mul_f32, two per instruction) and 2^-E is the scale operand of the
//     convert (quantize.hip.h: e2m1_pack2_hw).  Zero / denormal / huge amax or alpha (the reference's 3/0 = inf,
//     0 * inf = NaN -> code 7 behaviour for all-zero groups included) take the original division path, per wave.
// A ring of 2-4 prefetched tiles per wave (more bytes in flight) was measured and is SLOWER (QT 8192^2: 23 -> 28 us): not latency-bound.

// [r5] QT operands under an input scale byte of 255: the reference multiplies the decoded code by the bf16 +inf (bits 255 << 7): +-inf, or NaN for a zero code.  The
// hardware convert with an infinite scale operand returns NaN for EVERY code, so such a group (never produced by the quantizers) is decoded with unit scale and
// patched: nonzero -> +-inf, zero -> NaN.  w2 = two bf16 of cvt_scalef32_pk_bf16_fp4(w, 1.0, k).
__device__ __forceinline__ uint32_t qt_times_inf(uint32_t w2) {
  const uint32_t lo = w2 & 0xffffu, hi = w2 >> 16;
  const uint32_t l = (lo & 0x7fffu) ? ((lo & 0x8000u) | 0x7f80u) : 0x7fc0u, h = (hi & 0x7fffu) ? ((hi & 0x8000u) | 0x7f80u) : 0x7fc0u;
  return l | (h << 16);
}
template <bool QT, bool HWCVT>
__global__ __launch_bounds__(512) void bwd_quant_t_kernel(const BwdTParams p) {
#ifndef QAMD_BWD_LROW_QT
#define QAMD_BWD_LROW_QT 144
#endif
  // LDS row stride (bytes) of the [32 n][64 m] bf16 tile.  T: 192 -- its staging writes are whole 128-byte rows (conflict-free at any stride) and
  // the tr reads want the stride = 16 or 48 dwords (mod 64).  QT: a lane pair writes the 128-byte row as 2 x 4 ds_write_b128 and 4 rows per 8-lane
  // group: conflict-free needs stride = 4 (mod 16) dwords, which the tr reads cannot have at the same time -- 144 keeps the writes clean.
  constexpr int LROW = QT ? QAMD_BWD_LROW_QT : 192;
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
  // Tile order: m fastest.  T -- the workgroups that run together then read whole input rows (every tile takes full 128-byte lines of 256 rows
  // 2 M bytes apart; with the n-blocks fastest the resident set touched 2 KB of each of 8192 rows: 8192^2 cold 42.1 -> 37.6 us).  QT -- a tile
  // reads 32 of the 128 bytes of each of its input lines, so the SIB = 4 m-tiles that share the lines are dispatched as one unit: workgroup ids go
  // round the 8 XCDs, and ids 32 j + 8 s + x (s = 0..3) -- same XCD x, one dispatch window -- take the four siblings of unit 8 j + x; the XCD's L2
  // then fetches each line once (with plain m-fastest order the siblings sat on four XCDs: 24.5 -> 31.4 us at 8192^2), and the units that run
  // together spread over all line columns, i.e. all L2 / HBM channels (n-blocks fastest kept them on four columns).  The host makes
  // gridDim.x a multiple of 32 for QT.
  // Unit coordinates (b, o, i), all wave-uniform: i = m-tile (T) / quad of m-tiles (QT), o = block of 8 scale groups.  A workgroup walks units
  // u0, u0 + step, ...: the step is decomposed once, the walk is two carry-propagating adds.  The two 64-bit tile origins ride along: `in` =
  // element index of the tile's first input element for wave 0, (b N + 256 gb) M + 64 tm, and `grp` = index of its first output scale group,
  // (b M + 64 tm) G + 8 gb -- both linear in (b, o, i), so a step is one add and a carry a second.
  constexpr int SIB = QT ? 4 : 1;
  const int sib = QT ? (int)(blockIdx.x & 31u) >> 3 : 0;
  const int n_i = (p.tiles_m + SIB - 1) / SIB, n_o = ngb;
  const int64_t in_b = (int64_t)p.N * p.M, in_i = 64 * SIB, in_o = 256ll * p.M, gr_b = (int64_t)p.M * G, gr_i = 64ll * SIB * G, gr_o = 8;
  struct Tile { int b, o, i; int64_t in, grp; };
  auto split = [&](unsigned u) __attribute__((always_inline)) {
    const unsigned q = u / (unsigned)n_i;
    Tile r{(int)(q / (unsigned)n_o), (int)(q % (unsigned)n_o), (int)(u % (unsigned)n_i), 0, 0};
    r.in = r.b * in_b + r.o * in_o + r.i * in_i;
    r.grp = r.b * gr_b + r.o * gr_o + r.i * gr_i;
    return r;
  };
  const Tile step = split(QT ? gridDim.x >> 2 : gridDim.x);
  const int64_t in_c1 = in_o - n_i * in_i, in_c2 = in_b - n_o * in_o, gr_c1 = gr_o - n_i * gr_i, gr_c2 = gr_b - n_o * gr_o;
  auto advance = [&](Tile t) __attribute__((always_inline)) {
    t.i += step.i;
    const bool c1 = t.i >= n_i;
    t.i -= c1 ? n_i : 0;
    t.o += step.o + (c1 ? 1 : 0);
    const bool c2 = t.o >= n_o;
    t.o -= c2 ? n_o : 0;
    t.b += step.b + (c2 ? 1 : 0);
    t.in += step.in + (c1 ? in_c1 : 0) + (c2 ? in_c2 : 0);
    t.grp += step.grp + (c1 ? gr_c1 : 0) + (c2 ? gr_c2 : 0);
    return t;
  };

  // ---- per-lane offsets, computed once -------------------------------------------------------------------------
  // T : lane -> row lane/8 (+8 per pass, a wave-uniform soffset), 16-byte chunk lane%8 (8 m): one pass = 8 rows x 128 B.
  // QT: lane -> row lane/2, 16-byte half (32 codes = one input scale group) + its e8m0 byte.
  const uint32_t rowb = QT ? (uint32_t)p.M >> 1 : (uint32_t)p.M * 2u;   // input row stride in bytes (host-checked: 32 rows < 2^31 bytes)
  const int lcol = QT ? (lane & 1) * 32 : (lane & 7) * 8;               // first m of the lane's chunk inside the tile
  const uint32_t ld_off = QT ? (uint32_t)(lane >> 1) * rowb + (uint32_t)(lane & 1) * 16u : (uint32_t)(lane >> 3) * rowb + (uint32_t)(lane & 7) * 16u;
  const uint32_t lds_off = QT ? (uint32_t)(lane >> 1) * ((uint32_t)p.M >> 5) + (uint32_t)(lane & 1) : 0u;
  const uint32_t st_off = (uint32_t)(tid >> 3) * (uint32_t)G * 16u + (uint32_t)(tid & 7) * 16u;   // output piece: row tid/8, group tid%8
  const uint32_t OOB = 0x80000000u;                                     // past every descriptor below (their extents are < 2^31)

  // global -> registers for one tile (software pipeline: the next tile's loads are in flight while this one is rotated and quantised)
  // the wave's own 32 input rows start 32 wave rows below the tile origin
  const char* xw = QT ? (const char*)p.xq + (int64_t)wave * 32 * (p.M >> 1) : (const char*)p.x + (int64_t)wave * 32 * p.M * 2;
  const char* sw = QT ? (const char*)p.xs + (int64_t)wave * 32 * (p.M >> 5) : nullptr;
  v4i ld[QT ? 1 : 4];
  uint8_t ld_e = 0;   // (kept as the loaded byte and widened at its use: widened here, the zero-extension lands behind the load at the loop
                      //  latch, and with it the wait for the load just issued)
  auto load_tile = [&](const Tile t) __attribute__((always_inline)) {
    const int m0 = (t.i * SIB + sib) * 64, g = t.o * 8 + wave;
    const bool live = t.b < p.B && g < G;
    const uint32_t voff = (m0 + lcol < p.M) ? ld_off : OOB;           // M % 8 (T) / % 32 (QT) == 0: a chunk is in or out as a whole
    if (!QT) {
      const __amdgpu_buffer_rsrc_t r = make_rsrc(xw + t.in * 2, live ? 32u * rowb : 0u);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) ld[ps] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)(ps * 8 * rowb), 0);
    } else {
      const __amdgpu_buffer_rsrc_t r = make_rsrc(xw + (t.in >> 1), live ? 32u * rowb : 0u);
      const __amdgpu_buffer_rsrc_t re = make_rsrc(sw + (t.in >> 5), live ? 32u * ((uint32_t)p.M >> 5) : 0u);
      ld[0] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
      ld_e = __builtin_amdgcn_raw_buffer_load_b8(re, (int)((m0 + lcol < p.M) ? lds_off : OOB), 0, 0);   // dropped lanes read 0: their codes are 0 too
    }
  };
  Tile cur = split(QT ? (blockIdx.x >> 5) * 8u + (blockIdx.x & 7u) : blockIdx.x);
  cur.in += sib * 64;                    // this workgroup's sibling: the same m-tile of every unit it walks
  cur.grp += (int64_t)sib * 64 * G;
  // [r2] T: the first tile's loads are issued BEFORE the rotation matrix is staged (the two memory round trips overlap: 13.7 -> 12.9 us
  // cold at 4096^2).  QT keeps them after it: its tile is 1/4 of the bytes and hoisting measured +4 % warm (profiles/ab_stream_ops_r2.txt)
  if (!QT) load_tile(cur);
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
  if (QT) load_tile(cur);

  // kernel-wide constants of the division-free scale path (see the header comment)
  const uint32_t alpha_bits = __float_as_uint(alpha);
  const bool alpha_fast = !QT || (alpha_bits >= 0x30800000u && alpha_bits <= 0x4e800000u);   // 2^-30 <= alpha <= 2^30, positive, finite
  const uint32_t Kexp = QT ? alpha_bits - 0x3f800000u : 0u;             // bits(amax) - Kexp: exponent field = that of RN(amax / alpha)
  const float c3 = QT ? 3.0f / alpha : 3.0f;                            // RN(3 / alpha): the multiplier's mantissa
  // transposing LDS read: the 16 lanes of a group supply the four 8-byte pieces of each of 4 consecutive n rows of a 16-column block and
  // receive one column each (lane%16), 4 n values -- half a K = 16 MFMA operand
  const char* tr_ptr = ts + (8 * half + ((lane & 15) >> 2)) * LROW + (((lane & 31) >> 4) * 16 + (lane & 3) * 4) * 2;

  while (cur.b < p.B) {   // uniform over the workgroup: barriers inside are safe
    const int m0 = (cur.i * SIB + sib) * 64, g0 = cur.o * 8;
    const int64_t grp0 = cur.grp;
    if (QT && m0 >= p.M) {   // a sibling past the last m-tile (tiles_m not a multiple of 4): nothing staged, nothing stored
      cur = advance(cur);
      load_tile(cur);
      continue;
    }

    // ---- stage the [32 n][64 m] bf16 tile in LDS ---------------------------------------------------------------
    if (!QT) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int r = ps * 8 + (lane >> 3), c = (lane & 7) * 8;
        *(v4i*)(ts + r * LROW + c * 2) = ld[ps];
      }
    } else {
      const int r = lane >> 1, c = (lane & 1) * 32;
      const uint32_t e = ld_e;
      // [r5] the reference builds the scale as the bf16 with bits e << 7 (quartet_bwd_sm120.cu:369-371): byte 0 is 0.0 (NOT 2^-127: the operand is zero whatever
      // its code), byte 255 is +inf (operand +-inf, 0 x inf = NaN).  [r6] v_cvt_scalef32_pk_bf16_fp4 reads only the EXPONENT of its scale operand
      // (tests/native/cvt_mant_probe.hip), so 0.0f acts as 2^-127 there: for byte 0 the magnitudes of the code word are cleared instead (sign nibbles
      // kept: code x 0.0 = +-0.0), which makes a tile whose rows all carry scale byte 0 come out with amax 0 as in the reference
      const float sc = e == 255u ? 1.0f : __uint_as_float(e << 23);
      v4i* d = (v4i*)(ts + r * LROW + c * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t w = e ? (uint32_t)ld[0][q] : ((uint32_t)ld[0][q] & 0x88888888u);   // [r6] scale byte 0: +-0 (see sc above)
        v4i o;
        o[0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 0));
        o[1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 1));
        o[2] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 2));
        o[3] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 3));
        if (__builtin_expect(e == 255u, 0)) {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = (int)qt_times_inf((uint32_t)o[k]);
        }
        d[q] = o;
      }
    }
    cur = advance(cur);
    load_tile(cur);   // next tile's rows: in flight during the rotation / quantisation below
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
        typedef short v4s_ __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s_* lds_v4s_t;
        const v4s_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(tr_ptr + (16 * kc) * LROW + mh * 64));
        const v4s_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(tr_ptr + (16 * kc + 4) * LROW + mh * 64));
        const v8u16 xv = {(uint16_t)lo[0], (uint16_t)lo[1], (uint16_t)lo[2], (uint16_t)lo[3], (uint16_t)hi[0], (uint16_t)hi[1], (uint16_t)hi[2], (uint16_t)hi[3]};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[kc], __builtin_bit_cast(v8bf, xv), acc, 0, 0, 0);
      }
      // acc[4q+e] = y[m][8q + 4 half + e]   (quartet_bwd_sm120.cu:304-323 / :407-426)
      float amax = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(acc[r]));
      amax = xhalf_max(amax);
      // e8m0 byte + multiplier.  Fast path (see the header comment): amax in [2^-60, 2^60] and alpha in [2^-30, 2^30].
      const uint32_t ab = __float_as_uint(amax);
      uint32_t sb = (ab - Kexp) & 0x7f800000u;
      float mfac = c3, cs = __uint_as_float(sb);
      const bool fast = HWCVT && alpha_fast && ab >= 0x21800000u && ab <= 0x5d800000u;
      const bool slow_wave = __builtin_amdgcn_ballot_w64(!fast) != 0;
      if (slow_wave) {   // wave-uniform: the reference's arithmetic as written (quartet_bwd_sm120.cu:304-323)
        float scale = QT ? amax / alpha : amax;
        const uint32_t sbs = __float_as_uint(scale) & 0x7f800000u;
        scale = __uint_as_float(sbs);
        const float mult = QT ? 3.0f / (scale * alpha) : 3.0f / scale;
        sb = fast ? sb : sbs;
        mfac = fast ? mfac : mult;
        cs = fast ? cs : 1.0f;
      }
      float tq[16];
      scale_pk<16>(acc, 0, mfac, tq);   // (software encoder, !HWCVT: never the fast path, cs == 1 and unused)
      if (slow_wave) {
        // 0 * inf (all-zero group: 3 / 0 = inf) and inf * 0 are NaN, which cvt.rn.satfinite.e2m1x2 encodes as +6 whatever its sign; the
        // gfx950 convert keeps the NaN's sign bit, and v_pk_mul_f32 hands back a NEGATIVE quiet NaN here (codes 0xF): make it the positive one
#pragma unroll
        for (int r = 0; r < 16; ++r) tq[r] = (tq[r] != tq[r]) ? __uint_as_float(0x7fc00000u) : tq[r];
      }
      const uint32_t P = e2m1_pack8<HWCVT>(tq, cs);
      const uint32_t Q = e2m1_pack8<HWCVT>(tq + 8, cs);
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
      // 512 pieces of 16 bytes: row tid/8, group g0 + tid%8.  Rows past M fall off the descriptor (its extent is the tile's valid rows).
      const int rows = min(64, p.M - m0);
      const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + grp0 * 16, (uint32_t)rows * (uint32_t)G * 16u - (uint32_t)g0 * 16u);
      const uint32_t so = (g0 + (tid & 7) < G) ? st_off : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(*(const v4i*)(out_s + (tid >> 3) * OROW + (tid & 7) * 16), ro, (int)so, 0, 0);
      if (tid < 64 && m0 + tid < p.M) {
        uint8_t* dst = p.out_sf + grp0 + (int64_t)tid * G;
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
// [r4] bwd_quant_tw_kernel: the same arithmetic with WAVE-OWNED output lines -- no workgroup barrier in the tile loop.
//
// The kernel above spreads the 8 scale groups of an output line over the 8 waves of a workgroup and meets twice per unit at a barrier (stage the
// workgroup's output tile, free it again).  Here ONE WAVE walks NG scale groups (32 NG n) of its 64 output rows itself, one [32 n][64 m] tile after
// the other with the next tile's rows in flight, collects its 64 x 16 NG bytes of e2m1 + 64 x NG scale bytes in a private LDS area and stores whole
// 128-byte (NG = 8) or 64-byte (NG = 4) line segments.  Waves never wait for each other; the four sibling m-tiles that share QT's 128-byte input
// lines are the four waves of one workgroup.  What bounds it is instruction issue, not bytes (a wave issues one instruction per 4 cycles:
// ~145 VALU + the LDS round trips per 2048-element tile), so the walk keeps its addresses incrementally (one 64-bit add per tile, the divisions
// once per unit) and the LDS footprint is cut to 10 - 11.5 KB per wave: NG = 4 fits 16 (QT) / 12 (T) waves on a CU.
template <bool QT, bool HWCVT, int NG>
__global__ __launch_bounds__(256) void bwd_quant_tw_kernel(const BwdTParams p) {
  constexpr int LROW = QT ? QAMD_BWD_LROW_QT : 192;
  constexpr int HROW = 32 * 2 + 16;
  constexpr int OROW = NG * 16 + 16;
  static_assert(32 * HROW <= 32 * LROW, "the staged H^T borrows the first wave's tile area");
  __shared__ __attribute__((aligned(16))) char tile_s[4][32 * LROW];
  __shared__ __attribute__((aligned(16))) char out_s[4][64 * OROW];
  __shared__ __attribute__((aligned(16))) uint8_t sf_s[4][64 * NG];

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int row = lane & 31, half = lane >> 5;
  char* ts = tile_s[wave];
  char* os = out_s[wave];
  uint8_t* ss = sf_s[wave];
  const float alpha = QT ? *p.alpha : 1.0f;
  const int G = p.N >> 5;
  const int n_o = (G + NG - 1) / NG, n_i = p.tiles_m;
  const uint32_t OOB = 0x80000000u;

  v8bf hf[2];     // H^T fragments of the lane: filled below, after the first tile's loads are out
  const uint32_t rowb = QT ? (uint32_t)p.M >> 1 : (uint32_t)p.M * 2u;   // input row stride in bytes
  const int lcol = QT ? (lane & 1) * 32 : (lane & 7) * 8;
  const uint32_t ld_off = QT ? (uint32_t)(lane >> 1) * rowb + (uint32_t)(lane & 1) * 16u : (uint32_t)(lane >> 3) * rowb + (uint32_t)(lane & 7) * 16u;
  const uint32_t lds_off = QT ? (uint32_t)(lane >> 1) * ((uint32_t)p.M >> 5) + (uint32_t)(lane & 1) : 0u;
  const uint32_t alpha_bits = __float_as_uint(alpha);
  const bool alpha_fast = !QT || (alpha_bits >= 0x30800000u && alpha_bits <= 0x4e800000u);
  const uint32_t Kexp = QT ? alpha_bits - 0x3f800000u : 0u;
  const float c3 = QT ? 3.0f / alpha : 3.0f;
  const char* tr_ptr = ts + (8 * half + ((lane & 15) >> 2)) * LROW + (((lane & 31) >> 4) * 16 + (lane & 3) * 4) * 2;

  // the wave's walk: units (b, block of NG groups o, m-tile i), i fastest (< 2^31: host-checked), tiles k = 0 .. ng-1 inside a unit.  The cursor
  // L is one tile ahead of the arithmetic (its rows are in flight); e0 = element index of the tile's first input element, advanced by 32 rows
  // per tile and recomputed once per unit.
  const uint32_t U = (uint32_t)((int64_t)p.B * n_o * n_i);
  const uint32_t stride = gridDim.x * 4u;
  const int64_t estep = (int64_t)32 * p.M;
  struct Cur { uint32_t u; int k, ng, b, g0, m0; int64_t e0; };
  auto decode = [&](Cur& c) __attribute__((always_inline)) {
    c.k = 0;
    if (c.u >= U) { c.ng = 0; c.b = 0; c.g0 = 0; c.m0 = 0; c.e0 = 0; return; }
    uint32_t i = c.u % (uint32_t)n_i, q = c.u / (uint32_t)n_i;
    uint32_t o = q % (uint32_t)n_o;
    q /= (uint32_t)n_o;
    c.b = uniform((int)q);
    c.m0 = uniform((int)i * 64);
    c.g0 = uniform((int)o * NG);
    c.ng = uniform(min(NG, G - c.g0));
    c.e0 = ((int64_t)c.b * p.N + (int64_t)c.g0 * 32) * p.M + c.m0;
  };
  // (A register ring with the unit's NG tiles in flight instead of one, non-temporal loads / stores and a group-block-fastest unit order were
  // measured and dropped: profiles/ab_bwd_r4k_variants.txt, ab_bwd_abl_r4l_nt_order.txt.)
  constexpr int SLOTS = 1;
  v4i ld[SLOTS][QT ? 1 : 4];
  uint8_t ld_e[SLOTS] = {};
  auto load_tile = [&](const int s, const bool live_in, const int m0, const int64_t e0) __attribute__((always_inline)) {
    const bool live = live_in && !QAMD_BWD_ABL(1);
    const uint32_t voff = (live && m0 + lcol < p.M) ? ld_off : OOB;
    if (!QT) {
      const __amdgpu_buffer_rsrc_t r = make_rsrc((const char*)p.x + e0 * 2, live ? 32u * rowb : 0u);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) ld[s][ps] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)(ps * 8 * rowb), 0);
    } else {
      const __amdgpu_buffer_rsrc_t r = make_rsrc((const char*)p.xq + (e0 >> 1), live ? 32u * rowb : 0u);
      const __amdgpu_buffer_rsrc_t re = make_rsrc((const char*)p.xs + (e0 >> 5), live ? 32u * ((uint32_t)p.M >> 5) : 0u);
      ld[s][0] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
      ld_e[s] = __builtin_amdgcn_raw_buffer_load_b8(re, (int)((live && m0 + lcol < p.M) ? lds_off : OOB), 0, 0);
    }
  };
  // ---- stage the [32 n][64 m] bf16 tile of slot s in the wave's LDS area ------------------------------------------
  auto stage = [&](const int s) __attribute__((always_inline)) {
    if (QAMD_BWD_ABL(8)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (!QT) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int r = ps * 8 + (lane >> 3), c = (lane & 7) * 8;
        *(v4i*)(ts + r * LROW + c * 2) = ld[s][ps];
      }
    } else {
      const int r = lane >> 1, c = (lane & 1) * 32;
      const uint32_t e = ld_e[s];
      const float sc = e == 255u ? 1.0f : __uint_as_float(e << 23);   // (byte 0 -> 0.0: the bf16 with bits e << 7, as the reference builds it; byte 255 = +inf: qt_times_inf)
      v4i* d = (v4i*)(ts + r * LROW + c * 2);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const uint32_t w = e ? (uint32_t)ld[s][0][qq] : ((uint32_t)ld[s][0][qq] & 0x88888888u);
        v4i ov;
        ov[0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 0));
        ov[1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 1));
        ov[2] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 2));
        ov[3] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 3));
        if (__builtin_expect(e == 255u, 0)) {
#pragma unroll
          for (int k = 0; k < 4; ++k) ov[k] = (int)qt_times_inf((uint32_t)ov[k]);
        }
        d[qq] = ov;
      }
    }
  };
  // ---- rotate + quantise the staged tile = scale group k of the unit --------------------------------------------------
  auto compute = [&](const int k) __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): the wave's own LDS writes landed
    __builtin_amdgcn_wave_barrier();
    if (!QAMD_BWD_ABL(4)) {
    // both 32-row halves of the tile: the MFMAs and the amax reductions first, ONE wave-uniform decision for the pair, then a straight-line arm
    // for both (two independent chains the scheduler can interleave -- a branch per half ends the scheduling region four times per tile)
    v16f acc[2];
    float amax[2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mh][r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        typedef short v4s_ __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s_* lds_v4s_t;
        const v4s_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(tr_ptr + (16 * kc) * LROW + mh * 64));
        const v4s_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(tr_ptr + (16 * kc + 4) * LROW + mh * 64));
        const v8u16 xv = {(uint16_t)lo[0], (uint16_t)lo[1], (uint16_t)lo[2], (uint16_t)lo[3], (uint16_t)hi[0], (uint16_t)hi[1], (uint16_t)hi[2], (uint16_t)hi[3]};
        acc[mh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[kc], __builtin_bit_cast(v8bf, xv), acc[mh], 0, 0, 0);
      }
    }
    bool fast[2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      // [r5] the reference's chain from 0 (sixteen NaNs -- inf - inf under the rotation -- reduce to 0), written as the chain: it compiles to 8 v_max3_f32 with |x| operands.
      // The balanced tree that stood here cost 23 instructions per half: every leaf |x| was first quieted by a v_max_f32 |x|, |x| of its own (llvm.maxnum on a maybe-sNaN).
      float am = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(acc[mh][r]));
      amax[mh] = xhalf_max(am);
      const uint32_t ab = __float_as_uint(amax[mh]);
      fast[mh] = HWCVT && alpha_fast && ab >= 0x21800000u && ab <= 0x5d800000u;
    }
    const bool slow_wave = __builtin_amdgcn_ballot_w64(!(fast[0] && fast[1])) != 0;
    auto emit = [&](const int mh, auto slow_c) __attribute__((always_inline)) {
      constexpr bool SLOW = decltype(slow_c)::value;
      const int mloc = mh * 32 + row;
      const uint32_t ab = __float_as_uint(amax[mh]);
      uint32_t sb = (ab - Kexp) & 0x7f800000u;
      float mfac = c3, cs = __uint_as_float(sb);
      if (SLOW) {   // the reference's arithmetic as written (quartet_bwd_sm120.cu:304-323) for the lanes outside the fast range
        float scale = QT ? amax[mh] / alpha : amax[mh];
        const uint32_t sbs = __float_as_uint(scale) & 0x7f800000u;
        scale = __uint_as_float(sbs);
        const float mult = QT ? 3.0f / (scale * alpha) : 3.0f / scale;
        sb = fast[mh] ? sb : sbs;
        mfac = fast[mh] ? mfac : mult;
        cs = fast[mh] ? cs : 1.0f;
      }
      float tq[16];
      scale_pk<16>(acc[mh], 0, mfac, tq);
      if (SLOW) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tq[r] = (tq[r] != tq[r]) ? __uint_as_float(0x7fc00000u) : tq[r];
      }
      const uint32_t P = e2m1_pack8<HWCVT>(tq, cs);
      const uint32_t Q = e2m1_pack8<HWCVT>(tq + 8, cs);
      auto sw = __builtin_amdgcn_permlane32_swap(P, Q, false, false);
      const uint32_t X = sw[0], Y = sw[1];
      v2i ov;
      ov[0] = (int)((X & 0xffffu) | (Y << 16));
      ov[1] = (int)((X >> 16) | (Y & 0xffff0000u));
      *(v2i*)(os + mloc * OROW + k * 16 + half * 8) = ov;
      if (half == 0) ss[mloc * NG + k] = (uint8_t)(sb >> 23);
    };
    if (!slow_wave) {
      emit(0, std::false_type{});
      emit(1, std::false_type{});
    } else {
      emit(0, std::true_type{});
      emit(1, std::true_type{});
    }
    }
    __builtin_amdgcn_wave_barrier();   // (the transposing reads of this tile were consumed: the next group may be staged)
  };
  // ---- the wave's 64 output rows: 16 NG-byte line segments + NG scale bytes per row --------------------------------------
  auto store_unit = [&](const int b, const int g0, const int m0, const int ng) __attribute__((always_inline)) {
    if (QAMD_BWD_ABL(2)) return;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int rows = min(64, p.M - m0);
    const int64_t grp0 = ((int64_t)b * p.M + m0) * G + g0;    // first output scale group of the unit
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + grp0 * 16, (uint32_t)rows * (uint32_t)G * 16u - (uint32_t)g0 * 16u);
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const int pc = it * 64 + lane, r = pc / NG, kk = pc % NG;
      const uint32_t so = (kk < ng) ? (uint32_t)r * (uint32_t)G * 16u + (uint32_t)kk * 16u : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(*(const v4i*)(os + r * OROW + kk * 16), ro, (int)so, 0, 0);
    }
    if (lane < rows && !QAMD_BWD_ABL(16)) {
      uint8_t* dst = p.out_sf + grp0 + (int64_t)lane * G;
      if ((G & (NG - 1)) == 0) {
        if (NG == 8) *(v2i*)dst = *(const v2i*)(ss + lane * NG);
        else if (NG == 2) *(uint16_t*)dst = *(const uint16_t*)(ss + lane * NG);
        else *(uint32_t*)dst = *(const uint32_t*)(ss + lane * NG);
      } else {
        for (int kk = 0; kk < ng; ++kk) dst[kk] = ss[lane * NG + kk];
      }
    }
    __builtin_amdgcn_wave_barrier();   // the LDS reads above are issued before the next unit's writes (in-order LDS): only the compiler needs the fence
  };

  // (QT reads every e8m0 line once per XCD -- FETCH_SIZE 50.3 MB for 35.7 MB of input at 8192^2 -- because a row block's m-tiles go round the 8 XCDs with
  // the workgroup ids.  Giving each XCD whole row blocks, as a contiguous eighth of the walk or block by block, brings FETCH_SIZE to 36.0 MB and is
  // SLOWER, 26.6 -> 31.8 / 31.2 us cold: the two 64-byte halves of an output line then come from different L2s and WRITE_SIZE grows from 37.8 to
  // 52.8 / 58.2 MB.  A third order that keeps the line-sharing workgroups AND the two group blocks of an output line together in one XCD's dispatch
  // order leaves cold loads where they were (14.1 us: not byte-bound) and slows the stores (12.2 -> 13.4 us).  profiles/pmc_stream_ops_r4.txt,
  // ab_bwd_r4s_xcd_eighths.txt, ab_bwd_r4t_xcd_rowblocks.txt, ab_bwd_abl_r4aa_superblock_order.txt.)
  const uint32_t wg = blockIdx.x;
  Cur L;
  L.u = uniform((int)(wg * 4u + (uint32_t)wave));
  decode(L);
  load_tile(0, L.u < U, L.m0, L.e0);
  // (the first tile's rows are in flight while H is staged: the two memory round trips overlap, as in fused_quantize_kernel)
  {   // hT[j][k] = h[k][j], staged in the first wave's tile area and read once
    char* hT = tile_s[0];
    uint16_t hv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hv[i] = p.h[i * 256 + tid];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = i * 256 + tid, k = idx >> 5, j = idx & 31;
      *(uint16_t*)(hT + j * HROW + k * 2) = hv[i];
    }
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) hf[kc] = *(const v8bf*)(hT + row * HROW + (kc * 16 + half * 8) * 2);
    __syncthreads();
  }

  while (L.u < U) {
    const int k = L.k, ng = L.ng, b = L.b, g0 = L.g0, m0 = L.m0;
    stage(0);
    // the registers are free again: the next tile of the walk (this unit's next group, or the first group of the wave's next unit)
    if (L.k + 1 < L.ng) { L.k += 1; L.e0 += estep; }
    else { L.u += stride; decode(L); }
    load_tile(0, L.u < U, L.m0, L.e0);
    compute(k);
    if (k + 1 == ng) store_unit(b, g0, m0, ng);
  }
}

// ----------------------------------------------------------------------------------------------------------------
// [r5] bwd_qt_ring_kernel: bwd_quant_tw_kernel<QT> with the INPUT read as whole 128-byte lines.
//
// The four waves of a workgroup are the four m-tiles that share QT's e2m1 lines; in bwd_quant_tw_kernel each of them loads its own 32 bytes of every line (a load
// instruction that touches 32 lines for 1 KiB; tests/native/xpose_traffic_ubench.hip: that access shape alone costs 32 us at 8192^2 cold against 17 with whole lines).
// Here the workgroup fetches a step -- scale group k of the unit, [32 n][256 m] = 32 lines + 32 x 8 scale bytes -- cooperatively: wave w issues ONE LDS-DMA piece of 8
// whole lines (wave 0 a dword piece for the scale bytes as well) into a ring of R slots, R - 1 steps ahead of the arithmetic; a step starts with s_waitcnt vmcnt (own
// piece) + one s_barrier of the four waves (all pieces; and the slot of the step before is free for the piece issued next), then every wave takes its 32-byte quarter of
// each row + the scale byte out of the slot (16-byte chunks XOR-swizzled by 2 (row % 4): 8 lanes = 4 rows x 2 chunks hit 8 different chunks) and goes on exactly as
// bwd_quant_tw_kernel: bf16 tile, transposing reads, MFMAs, division-free scales, wave-owned output segments.  Needs M % 128 == 0.
template <bool HWCVT, int NG, int R, bool XR>
__global__ __launch_bounds__(256) void bwd_qt_ring_kernel(const BwdTParams p) {
  constexpr int LROW = QAMD_BWD_LROW_QT;
  constexpr int HROW = 32 * 2 + 16;
  constexpr int OROW = NG * 16 + 16;
  constexpr int SLOT = 4096 + 256;
  static_assert(32 * HROW <= 32 * LROW, "the staged H^T borrows the first wave's tile area");
  __shared__ __attribute__((aligned(16))) char tile_s[4][32 * LROW];
  __shared__ __attribute__((aligned(16))) char out_s[4][64 * OROW];
  __shared__ __attribute__((aligned(16))) uint8_t sf_s[4][64 * NG];
  __shared__ __attribute__((aligned(16))) char ring_s[R][SLOT];

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int row = lane & 31, half = lane >> 5;
  char* ts = tile_s[wave];
  char* os = out_s[wave];
  uint8_t* ss = sf_s[wave];
  const float alpha = *p.alpha;
  const int G = p.N >> 5;
  const int n_o = (G + NG - 1) / NG, n_iq = (p.tiles_m + 3) >> 2;
  const uint32_t OOB = 0x80000000u;
  const uint32_t rowb = (uint32_t)p.M >> 1, srowb = (uint32_t)p.M >> 5;
  const uint32_t alpha_bits = __float_as_uint(alpha);
  const bool alpha_fast = alpha_bits >= 0x30800000u && alpha_bits <= 0x4e800000u;
  const uint32_t Kexp = alpha_bits - 0x3f800000u;
  const float c3 = 3.0f / alpha;
  const char* tr_ptr = ts + (8 * half + ((lane & 15) >> 2)) * LROW + (((lane & 31) >> 4) * 16 + (lane & 3) * 4) * 2;

  // ---- the workgroup's walk: units (b, block of NG groups o, quad of m-tiles i), i fastest; steps k = 0 .. ng - 1 inside a unit.  XR: workgroup id 128 q + 8 j + x
  // takes unit 128 q + 16 x + j -- the 16 quads along m that share the 128-byte lines of input scale bytes on ONE XCD.
  const uint32_t U = (uint32_t)((int64_t)p.B * n_o * n_iq), U128 = U & ~127u;
  struct Cur { uint32_t u; int k, ng, b, g0, mq0; int64_t e0; };
  auto decode = [&](Cur& c) __attribute__((always_inline)) {
    c.k = 0;
    if (c.u >= U) { c.ng = 0; c.b = 0; c.g0 = 0; c.mq0 = 0; c.e0 = 0; return; }
    const uint32_t u = (XR && c.u < U128) ? (c.u & ~127u) + ((c.u & 7u) << 4) + ((c.u & 127u) >> 3) : c.u;
    uint32_t i = u % (uint32_t)n_iq, q = u / (uint32_t)n_iq;
    uint32_t o = q % (uint32_t)n_o;
    q /= (uint32_t)n_o;
    c.b = uniform((int)q);
    c.mq0 = uniform((int)i * 256);
    c.g0 = uniform((int)o * NG);
    c.ng = uniform(min(NG, G - c.g0));
    c.e0 = ((int64_t)c.b * p.N + (int64_t)c.g0 * 32) * p.M + c.mq0;
  };
  auto advance = [&](Cur& c) __attribute__((always_inline)) {
    if (c.k + 1 < c.ng) { c.k += 1; c.e0 += (int64_t)32 * p.M; }
    else { c.u += gridDim.x; decode(c); }
  };
  // ---- one step's fetch: wave w -> rows 8 w .. 8 w + 7 (lane / 8), LDS chunk lane % 8 of the row holds input chunk (lane % 8) ^ 2 (row % 4)
  const int cg = (lane & 7) ^ (((lane >> 3) & 3) << 1);
  const uint32_t q_off = (uint32_t)(lane >> 3) * rowb + (uint32_t)cg * 16u;
  const uint32_t e_off = (uint32_t)(lane >> 1) * srowb + (uint32_t)(lane & 1) * 4u;
  auto fetch = [&](const Cur& f, const int slot) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t r = make_rsrc((const char*)p.xq + (f.e0 >> 1) + (int64_t)wave * 8 * rowb, 8u * rowb);
    dma16(r, ring_s[slot] + wave * 1024, (int)((f.mq0 + 32 * cg < p.M) ? q_off : OOB));   // (dropped chunks land as zeros: code 0 under scale byte 0)
    if (wave == 0) {
      const __amdgpu_buffer_rsrc_t re = make_rsrc((const char*)p.xs + (f.e0 >> 5), 32u * srowb);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(re, (lds_ptr_t)(ring_s[slot] + 4096), 4, (int)((f.mq0 + 128 * (lane & 1) < p.M) ? e_off : OOB), 0, 0, 0);
    }
  };
  // the lane's quarter-row out of a slot: row lane / 2, input chunk 2 wave + lane % 2 (one input scale group) + its e8m0 byte
  const int tr_ = lane >> 1, tc_ = 2 * wave + (lane & 1);
  const int t_off = tr_ * 128 + ((tc_ ^ ((tr_ & 3) << 1)) << 4), t_eoff = 4096 + tr_ * 8 + tc_;
  v4i ld;
  uint32_t ld_e = 0;

  auto stage = [&]() __attribute__((always_inline)) {
    const int r = lane >> 1, c = (lane & 1) * 32;
    const uint32_t e = ld_e;
    const float sc = e == 255u ? 1.0f : __uint_as_float(e << 23);   // (byte 0 -> 0.0, byte 255 = +inf: see bwd_quant_t_kernel)
    v4i* d = (v4i*)(ts + r * LROW + c * 2);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const uint32_t w = e ? (uint32_t)ld[qq] : ((uint32_t)ld[qq] & 0x88888888u);
      v4i ov;
      ov[0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 0));
      ov[1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 1));
      ov[2] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 2));
      ov[3] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 3));
      if (__builtin_expect(e == 255u, 0)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ov[k] = (int)qt_times_inf((uint32_t)ov[k]);
      }
      d[qq] = ov;
    }
  };
  v8bf hf[2];
  auto compute = [&](const int k) __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): the wave's own LDS writes landed
    __builtin_amdgcn_wave_barrier();
    v16f acc[2];
    float amax[2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mh][r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        typedef short v4s_ __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s_* lds_v4s_t;
        const v4s_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(tr_ptr + (16 * kc) * LROW + mh * 64));
        const v4s_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(tr_ptr + (16 * kc + 4) * LROW + mh * 64));
        const v8u16 xv = {(uint16_t)lo[0], (uint16_t)lo[1], (uint16_t)lo[2], (uint16_t)lo[3], (uint16_t)hi[0], (uint16_t)hi[1], (uint16_t)hi[2], (uint16_t)hi[3]};
        acc[mh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[kc], __builtin_bit_cast(v8bf, xv), acc[mh], 0, 0, 0);
      }
    }
    bool fast[2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      float am = 0.f;   // (the reference's chain from 0: 8 v_max3_f32 with |x| operands)
#pragma unroll
      for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(acc[mh][r]));
      amax[mh] = xhalf_max(am);
      const uint32_t ab = __float_as_uint(amax[mh]);
      fast[mh] = HWCVT && alpha_fast && ab >= 0x21800000u && ab <= 0x5d800000u;
    }
    const bool slow_wave = __builtin_amdgcn_ballot_w64(!(fast[0] && fast[1])) != 0;
    auto emit = [&](const int mh, auto slow_c) __attribute__((always_inline)) {
      constexpr bool SLOW = decltype(slow_c)::value;
      const int mloc = mh * 32 + row;
      const uint32_t ab = __float_as_uint(amax[mh]);
      uint32_t sb = (ab - Kexp) & 0x7f800000u;
      float mfac = c3, cs = __uint_as_float(sb);
      if (SLOW) {   // the reference's arithmetic as written (quartet_bwd_sm120.cu:407-426) for the lanes outside the fast range
        float scale = amax[mh] / alpha;
        const uint32_t sbs = __float_as_uint(scale) & 0x7f800000u;
        scale = __uint_as_float(sbs);
        const float mult = 3.0f / (scale * alpha);
        sb = fast[mh] ? sb : sbs;
        mfac = fast[mh] ? mfac : mult;
        cs = fast[mh] ? cs : 1.0f;
      }
      float tq[16];
      scale_pk<16>(acc[mh], 0, mfac, tq);
      if (SLOW) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tq[r] = (tq[r] != tq[r]) ? __uint_as_float(0x7fc00000u) : tq[r];
      }
      const uint32_t P = e2m1_pack8<HWCVT>(tq, cs);
      const uint32_t Q = e2m1_pack8<HWCVT>(tq + 8, cs);
      auto sw = __builtin_amdgcn_permlane32_swap(P, Q, false, false);
      const uint32_t X = sw[0], Y = sw[1];
      v2i ov;
      ov[0] = (int)((X & 0xffffu) | (Y << 16));
      ov[1] = (int)((X >> 16) | (Y & 0xffff0000u));
      *(v2i*)(os + mloc * OROW + k * 16 + half * 8) = ov;
      if (half == 0) ss[mloc * NG + k] = (uint8_t)(sb >> 23);
    };
    if (!slow_wave) {
      emit(0, std::false_type{});
      emit(1, std::false_type{});
    } else {
      emit(0, std::true_type{});
      emit(1, std::true_type{});
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto store_unit = [&](const int b, const int g0, const int m0, const int ng) __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int rows = min(64, p.M - m0);
    const int64_t grp0 = ((int64_t)b * p.M + m0) * G + g0;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + grp0 * 16, (uint32_t)rows * (uint32_t)G * 16u - (uint32_t)g0 * 16u);
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const int pc = it * 64 + lane, r = pc / NG, kk = pc % NG;
      const uint32_t so = (kk < ng) ? (uint32_t)r * (uint32_t)G * 16u + (uint32_t)kk * 16u : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(*(const v4i*)(os + r * OROW + kk * 16), ro, (int)so, 0, 0);
    }
    if (lane < rows) {
      uint8_t* dst = p.out_sf + grp0 + (int64_t)lane * G;
      if ((G & (NG - 1)) == 0) {
        if (NG == 8) *(v2i*)dst = *(const v2i*)(ss + lane * NG);
        else *(uint32_t*)dst = *(const uint32_t*)(ss + lane * NG);
      } else {
        for (int kk = 0; kk < ng; ++kk) dst[kk] = ss[lane * NG + kk];
      }
    }
    __builtin_amdgcn_wave_barrier();
  };

  Cur C, F;
  C.u = F.u = blockIdx.x;
  decode(C);
  decode(F);
  int fs = 0, cs = 0, inflight = 0;   // slot of the next fetch / of the current step; steps fetched and not yet taken
#pragma unroll
  for (int i = 0; i < R - 1; ++i) {
    if (F.u < U) { fetch(F, fs); advance(F); fs = fs + 1 == R ? 0 : fs + 1; ++inflight; }
  }
  {   // hT[j][k] = h[k][j], staged in the first wave's tile area and read once
    char* hT = tile_s[0];
    uint16_t hv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hv[i] = p.h[i * 256 + tid];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = i * 256 + tid, k = idx >> 5, j = idx & 31;
      *(uint16_t*)(hT + j * HROW + k * 2) = hv[i];
    }
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) hf[kc] = *(const v8bf*)(hT + row * HROW + (kc * 16 + half * 8) * 2);
    __syncthreads();
  }
  while (C.u < U) {   // uniform over the workgroup
    // own piece of this step landed: the pieces of the R - 2 later steps may stay in flight (VMEM returns in order; stores issued in between only make the wait longer);
    // at the tail of the walk fewer pieces follow, so everything is waited for
    if (inflight == R - 1 && R > 2) {
      if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (R - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 2) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");   // all four pieces are in the slot; every wave is done with the slot of the step before
    --inflight;
    if (F.u < U) { fetch(F, fs); advance(F); fs = fs + 1 == R ? 0 : fs + 1; ++inflight; }
    const int m0 = C.mq0 + 64 * wave;
    if (m0 < p.M) {
      // (inline asm: behind a plain LDS read of the ring the compiler waits for EVERY LDS-DMA in flight -- it cannot tell the slots apart -- i.e. for the piece issued two
      //  lines up, a whole memory round trip per step)
      const uint32_t slot_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring_s[cs]);
      asm volatile("ds_read_b128 %0, %2\n\tds_read_u8 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ld), "=&v"(ld_e) : "v"(slot_a + (uint32_t)t_off), "v"(slot_a + (uint32_t)t_eoff) : "memory");
      stage();
      compute(C.k);
      if (C.k + 1 == C.ng) store_unit(C.b, C.g0, m0, C.ng);
    }
    cs = cs + 1 == R ? 0 : cs + 1;
    advance(C);
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
  int abl;             // lab build only: leave out 1 = the row-scale stores, 2 = the column-scale stores, 4 = the data stores
};

// exponent byte of encode_e8m0_shiftm8 (quartet_bwd_sm120.cu:503-509): amax is a bf16 value held in fp32
__device__ __forceinline__ uint32_t e8m0_shift7(float amax) {
  return amax == 0.0f ? 127u : ((__float_as_uint(amax) >> 23) - 7u) & 0xffu;
}
__device__ __forceinline__ float e8m0_scale(uint32_t e) { return __uint_as_float(e ? (e << 23) : 0x00400000u); }

// [r4] WC = column tiles per workgroup (4 WC waves): the row scales of a workgroup leave as 4 WC contiguous bytes per row.  The 4 MB of scale bytes cost the
// WC = 1 form 15 % of its time as 4-byte stores into as many different lines (8192^2 cold 46.8 us, 40.1 without them, the row scales 4.9 of the 6.8:
// profiles/ab_sq_abl_r4ab.txt); walking a 4-wave workgroup along n instead (row dwords collected in registers) lost more in occupancy than it saved
// (profiles/ab_sq_abl_r4ac_walk.txt).
template <int WC = 1, int TPW = 1>   // WC column tiles side by side (4 waves each), TPW tiles per wave one after the other: 128 WC TPW columns per workgroup
__global__ __launch_bounds__(256 * WC) void bwd_square_double_mxfp8_kernel(const SqParams p) {
  constexpr int CT = WC * TPW;                                     // column tiles per workgroup
  static_assert(CT == 1 || CT == 4 || CT == 8, "row pieces of 4, 16 or 32 bytes");
  __shared__ __attribute__((aligned(16))) uint8_t es[4][4 * CT];   // [row block][column block]
  const int tid = threadIdx.x, lane = tid & 63, wv = uniform(tid >> 6), wave = wv & 3, wcol = wv >> 2;
  const int tiles_n = p.n / (128 * CT);
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  const int r0 = ti * 128 + wave * 32;
  // lane -> row lane/16 (+4 per pass), 16-byte chunk lane%16 (8 columns); column block j = (lane%16)/4
  const int lr = lane >> 4, lc = (lane & 15) * 8;
  // [r3] block maximum on the packed bf16 bit patterns (sign stripped, v_pk_max_u16: 2 instructions per dword instead of 3; NaN inputs excepted, the order
  // of the patterns is the order of the magnitudes)
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  v4i v[TPW][8];
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp) {      // all loads of the wave first
    const int c0 = (tj * CT + wcol * TPW + tp) * 128;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int row = r0 + ps * 4 + lr;
      v[tp][ps] = row < p.m ? *(const v4i*)(p.x + (int64_t)row * p.n + c0 + lc) : v4i{0, 0, 0, 0};
    }
  }
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp) {
    const int ct = wcol * TPW + tp, c0 = (tj * CT + ct) * 128;
    u16x2 mx = {0, 0};
#pragma unroll
    for (int ps = 0; ps < 8; ++ps)
#pragma unroll
      for (int q = 0; q < 4; ++q) mx = __builtin_elementwise_max(mx, __builtin_bit_cast(u16x2, (uint32_t)v[tp][ps][q] & 0x7fff7fffu));
    float amax = __uint_as_float((uint32_t)(mx[0] > mx[1] ? mx[0] : mx[1]) << 16);
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
      int s0 = v[tp][ps][0], s1 = v[tp][ps][1], s2 = v[tp][ps][2], s3 = v[tp][ps][3];
      asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
      i16x2 lo = {0, 0}, hi = {0, 0};
      lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2, s0), qs, false);
      lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2, s1), qs, true);
      hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2, s2), qs, false);
      hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2, s3), qs, true);
      const v2i o = {__builtin_bit_cast(int, lo), __builtin_bit_cast(int, hi)};
      if (!QAMD_BWD_ABL(4)) *(v2i*)(p.y + (int64_t)(r0 + ps * 4 + lr) * p.n + c0 + lc) = o;
    }
    // scales: lanes 0, 4, 8, 12 hold the exponents of column blocks 0..3 of this wave's row block
    const uint32_t e0 = __shfl(e, 0), e1 = __shfl(e, 4), e2 = __shfl(e, 8), e3 = __shfl(e, 12);
    const uint32_t packed = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
    if (CT == 1 && lane < 32 && !QAMD_BWD_ABL(1)) *(uint32_t*)(p.row_sf + (int64_t)(r0 + lane) * (p.n >> 5) + tj * 4) = packed;
    if (lane == 0) *(uint32_t*)&es[wave][4 * ct] = packed;
  }
  __syncthreads();
  if (CT > 1 && tid < 128 && !QAMD_BWD_ABL(1)) {   // row ti * 128 + tid: the 4 CT column blocks of this workgroup are 4 CT consecutive bytes
    uint8_t* dst = p.row_sf + (int64_t)(ti * 128 + tid) * (p.n >> 5) + tj * 4 * CT;
#pragma unroll
    for (int q = 0; q < CT / 4; ++q) *(v4i*)(dst + 16 * q) = *(const v4i*)&es[tid >> 5][16 * q];
  }
  if (!QAMD_BWD_ABL(2)) {
    for (int idx = tid; idx < 128 * CT; idx += 256 * WC) {   // column tj * 128 CT + idx: the four row blocks of this workgroup are 4 consecutive bytes
      const int j = idx >> 5;
      const uint32_t col = es[0][j] | (es[1][j] << 8) | (es[2][j] << 16) | (es[3][j] << 24);
      *(uint32_t*)(p.col_sf + (int64_t)(tj * 128 * CT + idx) * (p.m_pad >> 5) + ti * 4) = col;
    }
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
  int abl;             // lab build only: 1 = leave out the scale stores
};

// NC = columns (n) per workgroup: 256, or 128 (half the LDS, 4 workgroups per CU: the kernel is one round of workgroups,
// so its duration is ONE workgroup's load -> transpose -> store latency chain, and smaller tiles shorten it)
// [r4] MR = rows (m) per workgroup, MR / 32 waves: 256 doubles the scale piece a workgroup writes per output row (8 bytes instead of 4: the 4-byte pieces cost
// 8 % at 8192^2, profiles/ab_sf_stores_r4ah.txt) at the same LDS per wave.
template <int NC, int MR = 128>
__global__ __launch_bounds__(MR * 2) void mxfp4_transpose_mxfp8_kernel(const TrParams p) {
  constexpr int NWV = MR / 32, NTH = MR * 2;      // waves, threads
  constexpr int LROW = NC * 2 + 64;    // bf16 row of NC columns + pad: 16 dwords (mod 64), so the 4 rows x 32 bytes that each of the two 16-lane groups of a
                                       // half wave gathers with ds_read_b64_tr_b16 fall on 64 different banks
  constexpr int LPR = NC / 32;         // lanes per input row (16 bytes = 32 codes = one input scale group each)
  constexpr int RPP = 64 / LPR;        // rows per load pass
  constexpr int CPL = NC / 64;         // columns per lane
  __shared__ __attribute__((aligned(16))) char ts_all[NWV][32 * LROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int tiles_n = p.n / NC;
  // [r3] NC = 128: a tile reads 64 of the 128 bytes of each of its input lines, the tile next to it (tj ^ 1) the other 64.  Workgroup ids go round
  // the 8 XCDs, so the two landed on different XCDs and each L2 fetched the whole line: FETCH_SIZE 82 MB for 35.6 MB of input at 8192^2
  // (profiles/pmc_stream_ops_r3.txt).  Pairs now sit on one XCD, one dispatch slot apart (tiles_n is even: n % 256 == 0).
  // ([r4] The e8m0 lines of the input rows are still fetched once per XCD (FETCH_SIZE 50.3 MB for 35.7 MB at 8192^2); keeping a row block on one XCD
  // removes that and costs more on the write side -- the 4 scale bytes a tile writes per output row share their line with 31 other row blocks, which then
  // sit on 8 different L2s: WRITE_SIZE 69.4 -> 85.8 MB, 28.4 -> 28.9 / 29.2 us cold.  profiles/ab_transpose_r4{s,t}_*.txt, pmc_stream_ops_r4.txt.)
  unsigned t = blockIdx.x;
  if (NC == 128 && t < (gridDim.x & ~15u)) {
    const unsigned x = t & 7u, k = t >> 3;
    t = 2u * ((k >> 1) * 8u + x) + (k & 1u);
  }
  const int ti = (int)(t / (unsigned)tiles_n), tj = (int)(t % (unsigned)tiles_n);
  const int r0 = ti * MR + wave * 32, c0 = tj * NC;
  char* ts = ts_all[wave];
  bool nan_in = false;
#pragma unroll
  for (int ps = 0; ps < 32 / RPP; ++ps) {
    const int r = ps * RPP + lane / LPR, c = (lane % LPR) * 32;
    const int64_t rowi = r0 + r;
    const bool live = rowi < p.m;
    const v4i v = live ? *(const v4i*)(p.xq + rowi * (p.n >> 1) + ((c0 + c) >> 1)) : v4i{0, 0, 0, 0};
    const uint32_t se = live ? p.xs[rowi * (p.n >> 5) + ((c0 + c) >> 5)] : 127u;
    // [r5] input scale byte 255 is NaN (`__nv_cvt_e8m0_to_bf16raw`, quartet_bwd_sm120.cu:658-660): all 32 operands of the group become NaN, which the reference's
    // fmaxf block maximum ignores and its e4m3 convert turns into 0x7f.  The bit-pattern maximum below cannot ignore a NaN, so a wave that has seen such a byte
    // takes the exact arm there (wave-uniform, never taken on the quantizers' own outputs).
    nan_in |= se == 255u;
    const float sc = se == 255u ? __uint_as_float(0x7fc00000u) : e8m0_scale(se);
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
  const bool wave_nan = __builtin_amdgcn_ballot_w64(nan_in) != 0;
  // Columns lane, lane + 64, ...  [r3] The lane's 32 m values of a column come out of the tile with 8 transposing reads (ds_read_b64_tr_b16: the 16
  // lanes of a group supply the 8-byte pieces of 4 rows x 16 columns and receive one column each, rows 2i, 2i + 1 already paired in a register)
  // instead of 32 two-byte reads + 16 packs -- PMC had the LDS instruction issue busy 17 of the kernel's 23 us at 8192^2 -- and the block
  // maximum is taken on the packed bf16 bit patterns (sign stripped, v_pk_max_u16: for non-NaN values the order of the patterns is the order of
  // the magnitudes; [r5] a wave that met an input scale byte of 255 = NaN operands takes the exact arm, `wave_nan`).
  typedef short v4s_ __attribute__((ext_vector_type(4)));
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) v4s_* lds_v4s_t;
  const char* tr_ptr = ts + ((lane & 15) >> 2) * LROW + ((lane >> 4) * 16 + (lane & 3) * 4) * 2;
  v4i oq[CPL][2];
  uint8_t oe[CPL];
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    uint32_t pr[16];   // pr[i] = bf16 of rows 2i (low half) and 2i+1 (high half)
    u16x2 mx = {0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const v4s_ t4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(tr_ptr + 4 * q * LROW + cc * 128));
      const v2i w2 = __builtin_bit_cast(v2i, t4);
      pr[2 * q] = (uint32_t)w2[0];
      pr[2 * q + 1] = (uint32_t)w2[1];
      mx = __builtin_elementwise_max(mx, __builtin_bit_cast(u16x2, pr[2 * q] & 0x7fff7fffu));
      mx = __builtin_elementwise_max(mx, __builtin_bit_cast(u16x2, pr[2 * q + 1] & 0x7fff7fffu));
    }
    float amax = __uint_as_float((uint32_t)(mx[0] > mx[1] ? mx[0] : mx[1]) << 16);
    uint32_t nanrows = 0;   // bit r: row r of this column's block is NaN
    if (wave_nan) {   // the maximum over the NON-NaN magnitudes (fmaxf semantics); the NaN rows are written as 0x7f below, whatever the convert makes of their sign
      uint32_t m16 = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        uint32_t lo16 = pr[i] & 0x7fffu, hi16 = (pr[i] >> 16) & 0x7fffu;
        const bool nl = lo16 > 0x7f80u, nh = hi16 > 0x7f80u;
        nanrows |= (nl ? 1u : 0u) << (2 * i) | (nh ? 1u : 0u) << (2 * i + 1);
        lo16 = nl ? 0u : lo16; hi16 = nh ? 0u : hi16;
        m16 = max(m16, max(lo16, hi16));
      }
      amax = __uint_as_float(m16 << 16);
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
    if (wave_nan && nanrows) {   // byte r of the lane's 32 output bytes = row r
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        uint32_t v = (uint32_t)o[d >> 2][d & 3];
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if ((nanrows >> (4 * d + b)) & 1u) v = (v & ~(0xffu << (8 * b))) | (0x7fu << (8 * b));
        o[d >> 2][d & 3] = (int)v;
      }
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
  constexpr int OROW = MR + 16;                        // staged output row: MR m bytes + pad
  char* os = &ts_all[0][0];
  uint8_t* es = (uint8_t*)os + NC * OROW;              // [NC][NWV]
  static_assert(NC * OROW + NC * NWV <= NWV * 32 * LROW, "output staging fits the bf16 staging area");
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    const int col = cc * 64 + lane;
    *(v4i*)(os + col * OROW + wave * 32) = oq[cc][0];
    *(v4i*)(os + col * OROW + wave * 32 + 16) = oq[cc][1];
    es[col * NWV + wave] = oe[cc];
  }
  __syncthreads();
  const int m0 = ti * MR;
#pragma unroll
  for (int ps = 0; ps < NC / 32; ++ps) {               // NC * MR / 16 16-byte pieces: row = piece / (MR / 16), chunk = piece % (MR / 16)
    const int piece = ps * NTH + tid, row = piece / (MR / 16), ch = piece % (MR / 16);
    const v4i v = *(const v4i*)(os + row * OROW + ch * 16);
    *(v4i*)(p.y + (int64_t)(c0 + row) * p.m_pad + m0 + ch * 16) = v;
  }
  if (tid < NC && !QAMD_BWD_ABL(1)) {
    uint8_t* dst = p.out_sf + (int64_t)(c0 + tid) * (p.m_pad >> 5) + (m0 >> 5);
    if (MR == 128) *(uint32_t*)dst = *(const uint32_t*)(es + tid * 4);
    else *(v2i*)dst = *(const v2i*)(es + tid * 8);
  }
}

// ----------------------------------------------------------------------------------------------------------------
// [r4] mxfp4_transpose_mxfp8_tw_kernel: the transposer with WAVE-OWNED output lines (the layout of bwd_quant_tw_kernel above).
//
// The kernel above is one shot per workgroup: load 128 m x NC n, two workgroup barriers, store -- 4096 workgroups in four rounds at 8192^2, each
// a serial load -> LDS -> transpose -> barrier -> stage -> barrier -> store chain.  Here a wave walks units of [32 MCH m] x [64 n]: one chunk of
// 32 m rows x 64 n (32 input bytes per row; the four waves of a workgroup are the four 64-column blocks of the same 128-byte input lines)
// after the other with the next chunk's rows in flight, dequantises it into a private bf16 tile, takes each of its 64 columns out with
// transposing reads (lane = column), requantises the 32 m values of the column to e4m3 + one e8m0 byte, and collects 64 n x 32 MCH bytes in a
// private output area that leaves as whole 128-byte lines (MCH = 4) or 64-byte segments (MCH = 2).  No workgroup barrier anywhere.
template <int MCH>
__global__ __launch_bounds__(256) void mxfp4_transpose_mxfp8_tw_kernel(const TrParams p) {
  constexpr int LROW = 64 * 2 + 64;    // as above (NC = 64): the transposing reads of a half wave fall on 64 different banks
  constexpr int OROW = MCH * 32 + 16;
  __shared__ __attribute__((aligned(16))) char tile_s[4][32 * LROW];
  __shared__ __attribute__((aligned(16))) char out_s[4][64 * OROW];
  __shared__ __attribute__((aligned(16))) uint8_t es_s[4][64 * MCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  char* ts = tile_s[wave];
  char* os = out_s[wave];
  uint8_t* es = es_s[wave];
  const uint32_t OOB = 0x80000000u;
  const uint32_t rowb = (uint32_t)p.n >> 1, srow = (uint32_t)p.n >> 5;     // input row strides in bytes: codes, scales
  const uint32_t n_j = (uint32_t)p.n >> 6, n_i = (uint32_t)p.m_pad / (32u * MCH);
  const uint32_t U = n_i * n_j, stride = gridDim.x * 4u;
  const uint32_t ld_off = (uint32_t)(lane >> 1) * rowb + (uint32_t)(lane & 1) * 16u;
  const uint32_t lds_off = (uint32_t)(lane >> 1) * srow + (uint32_t)(lane & 1);

  typedef short v4s_ __attribute__((ext_vector_type(4)));
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) v4s_* lds_v4s_t;
  const char* tr_ptr = ts + ((lane & 15) >> 2) * LROW + ((lane >> 4) * 16 + (lane & 3) * 4) * 2;

  // the walk: units u = (ti, tj), tj fastest (the workgroup's four waves = four neighbouring column blocks), chunks ch = 0 .. MCH-1 inside a unit
  struct Cur { uint32_t u; int ch, row0, c0; };
  auto decode = [&](Cur& c) __attribute__((always_inline)) {
    c.ch = 0;
    if (c.u >= U) { c.row0 = 0; c.c0 = 0; return; }
    const uint32_t tj = c.u % n_j, ti = c.u / n_j;
    c.row0 = uniform((int)(ti * 32u * MCH));
    c.c0 = uniform((int)(tj * 64u));
  };
  v4i ld;
  uint8_t ld_e = 0;
  auto load_chunk = [&](const Cur& c) __attribute__((always_inline)) {
    const int rows_live = c.u < U ? max(0, min(32, p.m - c.row0)) : 0;     // rows m .. m_pad-1 of the padded problem: zero codes
    const int64_t roff = (int64_t)c.row0;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(p.xq + roff * rowb + (c.c0 >> 1), (uint32_t)rows_live * rowb);
    const __amdgpu_buffer_rsrc_t re = make_rsrc(p.xs + roff * srow + (c.c0 >> 5), (uint32_t)rows_live * srow);
    ld = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(rows_live ? ld_off : OOB), 0, 0);
    ld_e = __builtin_amdgcn_raw_buffer_load_b8(re, (int)(rows_live ? lds_off : OOB), 0, 0);
  };
  Cur L;
  L.u = uniform((int)(blockIdx.x * 4u + (uint32_t)wave));
  decode(L);
  load_chunk(L);
  while (L.u < U) {
    const int ch = L.ch, row0 = L.row0, c0 = L.c0;
    {   // ---- dequantise the chunk into the wave's bf16 tile: lane = (row lane / 2, 32 codes = one input scale group)
      const int r = lane >> 1, c = (lane & 1) * 32;
      const float sc = e8m0_scale(ld_e);
      v4i* d = (v4i*)(ts + r * LROW + c * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t w = (uint32_t)ld[q];
        v4i o;
        o[0] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 0));
        o[1] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 1));
        o[2] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 2));
        o[3] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, sc, 3));
        d[q] = o;
      }
    }
    // the registers are free again: the next chunk of the walk
    if (L.ch + 1 < MCH) { L.ch += 1; L.row0 += 32; }
    else { L.u += stride; decode(L); }
    load_chunk(L);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    {   // ---- column `lane` of the tile: 32 m values as 16 packed pairs, block maximum on the bf16 bit patterns, e4m3 + e8m0 (as above)
      uint32_t pr[16];
      u16x2 mx = {0, 0};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const v4s_ t4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t)(tr_ptr + 4 * q * LROW));
        const v2i w2 = __builtin_bit_cast(v2i, t4);
        pr[2 * q] = (uint32_t)w2[0];
        pr[2 * q + 1] = (uint32_t)w2[1];
        mx = __builtin_elementwise_max(mx, __builtin_bit_cast(u16x2, pr[2 * q] & 0x7fff7fffu));
        mx = __builtin_elementwise_max(mx, __builtin_bit_cast(u16x2, pr[2 * q + 1] & 0x7fff7fffu));
      }
      const float amax = __uint_as_float((uint32_t)(mx[0] > mx[1] ? mx[0] : mx[1]) << 16);
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
      *(v4i*)(os + lane * OROW + ch * 32) = o[0];
      *(v4i*)(os + lane * OROW + ch * 32 + 16) = o[1];
      es[lane * MCH + ch] = (uint8_t)e;
    }
    __builtin_amdgcn_wave_barrier();   // (the transposing reads of this chunk were consumed: the next one may be staged)
    if (ch + 1 < MCH) continue;
    // ---- the wave's 64 output rows n: 32 MCH bytes each + MCH scale bytes ---------------------------------------------
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    {
      const int m0 = row0 - 32 * (MCH - 1);
      const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.y + (int64_t)c0 * p.m_pad + m0, 64u * (uint32_t)p.m_pad - (uint32_t)m0);
#pragma unroll
      for (int it = 0; it < MCH * 2; ++it) {
        const int piece = it * 64 + lane, rown = piece / (MCH * 2), pc = piece % (MCH * 2);
        __builtin_amdgcn_raw_buffer_store_b128(*(const v4i*)(os + rown * OROW + pc * 16), ro, (int)((uint32_t)rown * (uint32_t)p.m_pad + (uint32_t)pc * 16u), 0, 0);
      }
      uint8_t* dst = p.out_sf + (int64_t)(c0 + lane) * (p.m_pad >> 5) + (m0 >> 5);
      if (MCH == 4) *(uint32_t*)dst = *(const uint32_t*)(es + lane * 4);
      else *(uint16_t*)dst = *(const uint16_t*)(es + lane * 2);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace qamd
