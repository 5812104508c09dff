// NVFP4 GEMM for gfx950:  D[M,N] (bf16) = alpha * (A . SFA) (B . SFB)^T, e2m1 data, e4m3fn scale per
// 16 K-elements (to_blocked layout).  Replaces matmul_host_nvf4_bf16_tn (qutlass/csrc/gemm.cu:250-326).
//
// CDNA4's block-scaled MFMA only applies E8M0 scales per 32 elements, so NVFP4 (E4M3 per 16) is NOT
// native.  e2m1 x e4m3 has up to 6 significant bits: exact in f16/bf16, not in fp8/fp6.  This kernel
// therefore keeps the reference's exact semantics (its tests assert bit-equality with the fp64
// dequant-matmul oracle, tests/nvfp4_test.py:224) by dequantising on the fly to f16 -- value =
// cvt(e2m1) [v_cvt_scalef32_pk_f16_fp4, scale 1.0] * e4m3 scale [v_pk_mul_f16], both exact -- and
// running v_mfma_f32_32x32x16_f16.  Its roofline is therefore the 16-bit MFMA peak (~2.5 PF dense),
// not the FP4 peak.  Data movement (LDS-DMA stages, swizzle, swapped operand roles, LDS-staged
// whole-line epilogue, XCD-aware raster) is identical to gemm_mx.hip.h.
#pragma once
#include <algorithm>
#include <cmath>

#include "common.hip.h"

namespace qamd {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

#ifndef QAMD_CTX_MAGIC_DECODE
#define QAMD_CTX_MAGIC_DECODE 0
#endif
#ifndef QAMD_NV_KERNARG_EARLY
#define QAMD_NV_KERNARG_EARLY 0
#endif
struct NvGemmParams {
  const uint8_t* A;
  const uint8_t* B;
  const uint8_t* SFA;
  const uint8_t* SFB;
  const float* alpha;
  uint16_t* D;
  int M, N, K;
  int ldd;         // row stride of D in elements (= N unless the launch covers a column range of a wider D: operands >= 2 GiB, capi.hip)
  int tiles_m, tiles_n;
  uint32_t raster_magic;   // persistent kernel (gemm_nvf4_pk.hip.h): raster_magic(tiles_n) of common.hip.h, set by launch_nvf4_pk
  uint32_t a_bytes, b_bytes, sfa_bytes, sfb_bytes;
  uint32_t* dbg;   // bench only: block 0 writes {shader cycles, 100 MHz ticks} of its K loop
  float* ws;       // [r3] split-K: fp32 partials ws[z][M][N] (caller scratch, capi.hip nvf4_impl); null = single pass
  int splits;      //      K ranges per tile (1 = single pass); the launch has tiles x splits workgroups, range z = blockIdx.x / tiles
  int kt_per;      //      K stages (of 256 elements) per range: even, so that a range starts on LDS buffer 0
  // [r4] persistent kernel (gemm_nvf4_pk.hip.h): stream-K over the last sk_tiles tiles of the raster (0 = none: whole tiles only)
  int sk_tiles;
  float* sk_ws;                  //      parked fp32 accumulators, 256 KiB per range boundary (slot = workgroup index, 1 .. grid - 1)
  unsigned long long* sk_flags;  //      one arrival flag per slot; == sk_tag: parked (the consumer resets it to 0)
  unsigned long long sk_tag;     //      per-launch number (capi.hip next_launch_tag): the flags need no initialisation
};

template <int BM_, int BN_, int WAVES_M_, int WAVES_N_>
struct NvCfg {
  static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
  static constexpr int NWAVES = WAVES_M * WAVES_N, THREADS = NWAVES * 64;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, MT = WTM / 32, NT = WTN / 32;
  static constexpr int ROWB = 128;      // bytes of K per row per stage = 256 elements = 16 scale groups
  static constexpr int SCT = 4;         // scale column tiles (4 groups of 16) per stage
  static constexpr int SA_TILES = (BM + 127) / 128, SB_TILES = (BN + 127) / 128;
  static constexpr int PA = SA_TILES * SCT, PB = SB_TILES * SCT;
  static constexpr int NSIA = PA / 2, NSIB = PB / 2;
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  static constexpr int SA_BYTES = NSIA * 1024, SB_BYTES = NSIB * 1024;
  static constexpr int OFF_B = A_BYTES, OFF_SA = A_BYTES + B_BYTES, OFF_SB = OFF_SA + SA_BYTES;
  static constexpr int STAGE_BYTES = OFF_SB + SB_BYTES;
  static constexpr int NA = BM / 8 / NWAVES, NB = BN / 8 / NWAVES;
  static constexpr int SROW = BN * 2;
  static constexpr int LDS_BYTES = (2 * STAGE_BYTES > BM * SROW) ? 2 * STAGE_BYTES : BM * SROW;
  static constexpr int SPW = (NSIA + NSIB + NWAVES - 1) / NWAVES;   // 1-KiB scale pieces per wave (2 for the 4-wave config)
};

// 4 e4m3 scale bytes (one dword of the blocked layout) -> two packed-f16 pairs, exact: the hardware convert (OCP e4m3fn: sign honoured, 0x7f / 0xff = NaN), two
// instructions.  [r5] Rounds 1-4 built the f16 bits with integer operations here ((byte & 0x7f) << 7, times 2^8): the same value for every byte the quantizer emits,
// but the sign bit was dropped and 0x7f decoded to 480 where the persistent kernel (v_cvt_scalef32_pk_f16_fp8 since round 4) and the oracle say NaN -- one call
// could change meaning with the shape-dependent dispatch (tests/test_gpu_round5.py::test_matmul_nvf4_nan_scale_bytes_decode_the_same_in_every_kernel).
__device__ __forceinline__ void e4m3x4_to_f16(uint32_t d, h2_t& s01, h2_t& s23) {
  s01 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d, 1.0f, false);
  s23 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d, 1.0f, true);
}

// one dword = 8 e2m1 (element 2b = low nibble of byte b) -> 8 f16 scaled by s (broadcast pair)
__device__ __forceinline__ h8_t dq8(uint32_t w, h2_t s) {
  const h2_t a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 0) * s;
  const h2_t b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 1) * s;
  const h2_t c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 2) * s;
  const h2_t d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 3) * s;
  return h8_t{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// SPLIT: the workgroup walks K stages [z kt_per, (z + 1) kt_per) of its tile only and stores the raw fp32 accumulators to ws[z] (splitk_reduce_kernel sums
// the ranges in z order, applies alpha and rounds: gemm_mx.hip.h).  For outputs of a few dozen 128x128 tiles and a long K, where the only other way to
// give every CU work is 64x64 tiles whose 32x32 wave tiles dequantise two fragments per MFMA.
template <class C, bool SPLIT = false, bool FENCED = false>
__global__ __launch_bounds__(C::THREADS) void gemm_nvf4_kernel(const NvGemmParams p) {
  constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT;
  __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
#if QAMD_NV_KERNARG_EARLY   // one scalar-load round for the kernel arguments (as gemm_mx_deepp_kernel); prepared at the end of round 4, off in the product, not measured yet
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"(p.ws), "s"(p.splits), "s"(p.kt_per));
#endif
  const float alpha_k = SPLIT ? 1.0f : *p.alpha;       // [r4] fetched here, not where the epilogue starts (a memory round trip on every workgroup's critical path)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wave_m = wave / C::WAVES_N, wave_n = wave % C::WAVES_N;
  const int i32 = lane & 31, g = lane >> 5;

  int tile_m, tile_n;
  const int nb = p.tiles_m * p.tiles_n;
  const int z = SPLIT ? (int)blockIdx.x / nb : 0;
  {
    const int b2 = xcd_remap((int)blockIdx.x - z * nb, nb);
#if QAMD_CTX_MAGIC_DECODE   // (prepared at the end of round 4, not the product's choice yet: gemm_mx.hip.h GemmCtx)
    raster_decode(b2, p.tiles_m, p.tiles_n, p.raster_magic, tile_m, tile_n);
#else
    constexpr int GM = 4;
    const int group = GM * p.tiles_n;
    const int gid = b2 / group;
    const int first_m = gid * GM;
    const int gsz = min(p.tiles_m - first_m, GM);
    tile_m = first_m + (b2 % group) % gsz;
    tile_n = (b2 % group) / gsz;
#endif
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int rowbytes = p.K >> 1;
  const int KT = (rowbytes + C::ROWB - 1) / C::ROWB;
  const int CB = (p.K / 16 + 3) >> 2;
  const bool ktail = (rowbytes % C::ROWB) != 0;

  const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A + a_off, p.a_bytes - a_off);
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);
  const uint32_t sa_off = (uint32_t)(m0 >> 7) * CB * 512, sb_off = (uint32_t)(n0 >> 7) * CB * 512;
  const __amdgpu_buffer_rsrc_t rSA = make_rsrc(p.SFA + sa_off, p.sfa_bytes - sa_off);
  const __amdgpu_buffer_rsrc_t rSB = make_rsrc(p.SFB + sb_off, p.sfb_bytes - sb_off);

  int voffA[C::NA], voffB[C::NB], chA[C::NA], chB[C::NB];
#pragma unroll
  for (int t = 0; t < C::NA; ++t) {
    const int q = wave * C::NA + t, row = 8 * q + (lane >> 3);
    chA[t] = (lane & 7) ^ ((row >> 1) & 7);
    voffA[t] = row * rowbytes + chA[t] * 16;
  }
#pragma unroll
  for (int t = 0; t < C::NB; ++t) {
    const int q = wave * C::NB + t, row = 8 * q + (lane >> 3);
    chB[t] = (lane & 7) ^ ((row >> 1) & 7);
    voffB[t] = row * rowbytes + chB[t] * 16;
  }
  int voffS[C::SPW], colS[C::SPW];
#pragma unroll
  for (int e = 0; e < C::SPW; ++e) {
    const int sp = wave + e * C::NWAVES;             // scale piece (1 KiB = two 512-byte tiles, one per lane half)
    voffS[e] = 0x7fffffff;
    colS[e] = 0;
    if (sp < C::NSIA + C::NSIB) {
      const bool isB = sp >= C::NSIA;
      const int s = isB ? sp - C::NSIA : sp;
      const int pp = 2 * s + g;
      colS[e] = pp % C::SCT;
      voffS[e] = ((pp / C::SCT) * CB + colS[e]) * 512 + i32 * 16;
    }
  }

  auto issue_stage = [&](int kt, int buf) {
    char* st = smem + buf * C::STAGE_BYTES;
    const int soff = kt * C::ROWB;
    const bool tail = ktail && (kt == KT - 1);
#pragma unroll
    for (int t = 0; t < C::NA; ++t) {
      int v = voffA[t];
      if (tail && (soff + chA[t] * 16 >= rowbytes)) v = 0x7fffffff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(st + (wave * C::NA + t) * 1024), 16, v, soff, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < C::NB; ++t) {
      int v = voffB[t];
      if (tail && (soff + chB[t] * 16 >= rowbytes)) v = 0x7fffffff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(st + C::OFF_B + (wave * C::NB + t) * 1024), 16, v, soff, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < C::SPW; ++e) {
      const int sp = wave + e * C::NWAVES;
      if (sp < C::NSIA + C::NSIB) {
        int v = voffS[e];
        if (kt * C::SCT + colS[e] >= CB) v = 0x7fffffff;
        const int ssoff = kt * C::SCT * 512;
        if (sp < C::NSIA)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rSA, (lds_ptr_t)(st + C::OFF_SA + sp * 1024), 16, v, ssoff, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rSB, (lds_ptr_t)(st + C::OFF_SB + (sp - C::NSIA) * 1024), 16, v, ssoff, 0, 0);
      }
    }
  };

  // fragment chunk c = 4g + j (j = 0..3); its two 16-groups 2c, 2c+1 live in scale column tile
  // 2g + (j>>1), bytes 2(j&1), 2(j&1)+1 of the row's dword
  const int sw = (i32 >> 1) & 7;
  int rdA[4], rdB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * g + j;
    rdA[j] = (wave_m * C::WTM + i32) * C::ROWB + ((c ^ sw) << 4);
    rdB[j] = C::OFF_B + (wave_n * C::WTN + i32) * C::ROWB + ((c ^ sw) << 4);
  }
  const int rbaseA = (BM >= 128) ? 0 : (m0 & 127), rbaseB = (BN >= 128) ? 0 : (n0 & 127);
  int rdSA[MT], rdSB[NT];   // address of column tile 2g; tile 2g+1 is +512
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int r = rbaseA + wave_m * C::WTM + 32 * t;
    rdSA[t] = C::OFF_SA + ((r >> 7) * C::SCT + 2 * g) * 512 + i32 * 16 + ((r & 127) >> 5) * 4;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r = rbaseB + wave_n * C::WTN + 32 * t;
    rdSB[t] = C::OFF_SB + ((r >> 7) * C::SCT + 2 * g) * 512 + i32 * 16 + ((r & 127) >> 5) * 4;
  }

  v16f acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // A stage is 16 k-steps of 16 elements: step s = (j, u) with chunk j = s / 4 (scale column tile jj = j / 2, byte pair
  // jl = j % 2) and dword u = s % 4.  Software pipeline with TWO fragment register sets: the converts of step s + 1 write
  // set (s + 1) & 1 while the MFMAs of step s read set s & 1.
  //
  // [r3] Who issues when.  A 32x32x16 f16 MFMA holds the matrix pipe for 32 cycles and the 4-5 converts / multiplies that belong to it issue in
  // 16-20 of them -- but only if they sit BETWEEN the MFMAs in program order: with one wave per SIMD (the 256-row tiles) there is no second wave
  // to fill the shadow, and the scheduler's own order was runs of 4-7 MFMAs followed by runs of 16-24 converts, i.e. the two units took turns
  // (13 100 cycles per stage = 8192 of MFMA + 5000 of VALU; zero-filled operands at 2.4 GHz and random ones at the power-limited 1.95 GHz took
  // the same CYCLES).  For those tiles (FENCE) every MFMA is followed by its share of the next step's dequantisation and a scheduling fence;
  // the second scale column tile of a stage is converted two row sets per step in the shadows of steps 1..4.  8192^3: 784 -> 736 us, -5 ... -8 % on
  // every shape that runs 256-row tiles, bit-identical (tools/ab_nvf4.py).  Also measured: the stage hand-off moved INTO the stage (after step 8
  // nobody reads the stage's LDS buffer any more, so wait + barrier + re-issue at step 11 and the next stage's first scales / first step in the
  // shadows of steps 11..15, i.e. nothing exposed at the top of a stage): correct, and no faster (0.949 against 0.938 of the old time) -- the
  // kernel is at the socket power limit once the two units overlap; not kept.
  // The smaller tiles run several workgroups per CU, whose waves fill each other's shadows, and lose 5 % to a fixed order: they keep the
  // plain loop.
  constexpr bool FENCE = BM >= 256 || FENCED;   // FENCED: lab, the fixed order for a smaller tile (one workgroup per CU after a K split)
  constexpr int NSLOT = MT * NT, NH = 2 * (MT + NT);
  h2_t sa[2][MT][2], sb[2][NT][2];     // [jj & 1][.][jl]
  v4i ca[2][MT], cb[2][NT];            // [j & 1]
  h8_t fa[2][MT], fb[2][NT];           // [s & 1]
  auto load_scale1 = [&](const char* st, const int jj, const int f) __attribute__((always_inline)) {   // one row set's dword of 4 e4m3 scales -> f16 pairs
    if (f < MT) e4m3x4_to_f16(*(const uint32_t*)(st + rdSA[f] + jj * 512), sa[jj & 1][f][0], sa[jj & 1][f][1]);
    else e4m3x4_to_f16(*(const uint32_t*)(st + rdSB[f - MT] + jj * 512), sb[jj & 1][f - MT][0], sb[jj & 1][f - MT][1]);
  };
  auto load_scales = [&](const char* st, const int jj) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < MT + NT; ++f) load_scale1(st, jj, f);
  };
  auto load_chunks = [&](const char* st, const int j) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < MT; ++t) ca[j & 1][t] = *(const v4i*)(st + rdA[j] + t * 32 * C::ROWB);
#pragma unroll
    for (int t = 0; t < NT; ++t) cb[j & 1][t] = *(const v4i*)(st + rdB[j] + t * 32 * C::ROWB);
  };
  // half of one fragment of step s: dword bytes 2 hh, 2 hh + 1 -> elements 4 hh .. 4 hh + 3 (2 converts + 2 packed multiplies)
  auto dq_half = [&](const int s, const int f, const int hh) __attribute__((always_inline)) {
    const int j = s >> 2, u = s & 3, jj = j >> 1, jl = j & 1;
    const bool isA = f < MT;
    const int t = isA ? f : f - MT;
    const _Float16 sc = isA ? sa[jj & 1][t][jl][u >> 1] : sb[jj & 1][t][jl][u >> 1];
    const uint32_t w = (uint32_t)(isA ? ca[j & 1][t][u] : cb[j & 1][t][u]);
    const h2_t s2 = {sc, sc};
    h2_t lo, hi;
    if (hh == 0) {
      lo = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 0) * s2;
      hi = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 1) * s2;
    } else {
      lo = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 2) * s2;
      hi = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, 1.0f, 3) * s2;
    }
    h8_t& d = isA ? fa[s & 1][t] : fb[s & 1][t];
    d[4 * hh + 0] = lo[0]; d[4 * hh + 1] = lo[1]; d[4 * hh + 2] = hi[0]; d[4 * hh + 3] = hi[1];
  };
  auto dq_step = [&](const int s) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < MT + NT; ++f) { dq_half(s, f, 0); dq_half(s, f, 1); }
  };
  auto mfma_slot = [&](const int s, const int i) __attribute__((always_inline)) {
    const int m = i / NT, n = i % NT;
    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[s & 1][n], fa[s & 1][m], acc[m][n], 0, 0, 0);
  };

  const int kt_begin = SPLIT ? z * p.kt_per : 0, kt_end = SPLIT ? min(KT, kt_begin + p.kt_per) : KT;   // (kt_per is even: a range starts on buffer 0)
  issue_stage(kt_begin, 0);
  asm volatile("" :: "s"(alpha_k));   // alpha is waited for HERE, behind the first stage's DMA (left alone, its load is sunk to the epilogue)
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* st = smem + (kt & 1) * C::STAGE_BYTES;
    // [r4] FENCE tiles (one workgroup per CU): this stage's first LDS reads go out BEFORE the next stage's DMA burst (8-14 instructions of ~16 cycles each
    // that the first MFMA of the stage was waiting behind)
    if (!FENCE && kt + 1 < kt_end) issue_stage(kt + 1, (kt + 1) & 1);
    load_scales(st, 0);
    load_chunks(st, 0);
    if (FENCE && kt + 1 < kt_end) issue_stage(kt + 1, (kt + 1) & 1);
    dq_step(0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) {
        if (!FENCE && s == 1) load_scales(st, 1);                              // second scale column tile: used from step 8
        if ((s & 3) == 0 && (s >> 2) + 1 < 4) load_chunks(st, (s >> 2) + 1);   // raw chunk of j + 1: one chunk (4 steps) ahead
        if (!FENCE) dq_step(s + 1);
      }
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        mfma_slot(s, i);
        if (FENCE) {
          if (s + 1 < 16) {
#pragma unroll
            for (int h = i * NH / NSLOT; h < (i + 1) * NH / NSLOT; ++h) dq_half(s + 1, h >> 1, h & 1);
          }
          // this stage's second scale column tile (needed by the dequantisation of step 8, which runs in the shadows of step 7): two row sets
          // per step in steps 1..4 instead of one burst of ~80 instructions
          if (s >= 1 && s <= 4 && (i == NSLOT / 2 - 1 || i == NSLOT - 1)) {
            const int f = 2 * (s - 1) + (i == NSLOT - 1 ? 1 : 0);
            if (f < MT + NT) load_scale1(st, 1, f);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  if constexpr (SPLIT) {   // raw fp32 partial of this K range: a lane owns 4 consecutive columns per q -> 16-byte stores (layout of gemm_mx.hip.h epilogue_partial)
    float* base = p.ws + (size_t)z * p.M * p.N;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int grow = m0 + wave_m * C::WTM + 32 * m + i32;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gcol = n0 + wave_n * C::WTN + 32 * n + 8 * q + 4 * g;
          if (grow < p.M && gcol < p.N) {
            const v4f v = {acc[m][n][4 * q + 0], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
            *(v4f*)(base + (size_t)grow * p.N + gcol) = v;
          }
        }
    }
    return;
  }
  const float alpha = alpha_k;
  __syncthreads();
  // the epilogue re-derives lane / thread id (v_mbcnt) instead of keeping them live across the K loop: with 256 accumulators
  // + two fragment sets the 256x256 tile is at the 256-VGPR limit and the three id registers were spilled to scratch
  const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int etid = wave * 64 + elane, ei32 = elane & 31, eg = elane >> 5;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wave_m * C::WTM + 32 * m + ei32;
        const int cg = (wave_n * C::WTN + 32 * n + 8 * q + 4 * eg) >> 2;
        v2i w;
        w[0] = pack_bf16x2(acc[m][n][4 * q + 0] * alpha, acc[m][n][4 * q + 1] * alpha);
        w[1] = pack_bf16x2(acc[m][n][4 * q + 2] * alpha, acc[m][n][4 * q + 3] * alpha);
        *(v2i*)(smem + row * C::SROW + ((cg ^ (row & 15)) << 3)) = w;
      }
  __syncthreads();
  constexpr int CPR = BN / 8;
  constexpr int RPP = C::THREADS / CPR;
  const int chunk = etid % CPR, r0 = etid / CPR;
  const int gcol = n0 + chunk * 8;
#pragma unroll 4
  for (int pss = 0; pss < BM / RPP; ++pss) {
    const int row = pss * RPP + r0;
    const int grow = m0 + row;
    if (grow < p.M && gcol < p.N) {
      v4i v = *(const v4i*)(smem + row * C::SROW + ((((2 * chunk) ^ (row & 15)) & ~1) << 3));
      if (row & 1) v = v4i{v[2], v[3], v[0], v[1]};
      *(v4i*)(p.D + (size_t)grow * p.ldd + gcol) = v;
    }
  }
}

// ================================================================================================
// v2: dequantise ONCE per workgroup into f16 LDS tiles.
//
// The kernel above converts every fragment in the wave that consumes it, so an A chunk is converted by
// all WAVES_N waves of its row block and a B chunk by all WAVES_M: 3x redundant VALU work, ~900 VALU
// instructions per 128 MFMAs per wave -- the VALU and the matrix pipe are both saturated
// (tests/native/ubench.hip "valu": cvt and pk_mul issue at full rate, 2 waves x (8 MFMA + 64 VALU) take
// 1.3x the MFMA-only time).  Here each thread owns ONE operand row of the tile: per 64-element K stage it
// loads the row's 32 packed bytes + its 4 e4m3 scales from global memory (two stages ahead, into
// registers), converts them once (32 cvt + 32 pk_mul), and writes 128 bytes of f16 into the LDS stage
// with the same chunk ^ ((row>>1)&7) swizzle the MX kernels use.  The MFMA side is then the plain f16
// kernel: per K=16 step 6 ds_read_b128 feed 8 v_mfma_f32_32x32x16_f16 -- the same LDS bytes per MFMA cycle
// as the FP4 kernel.  One barrier per stage; no LDS-DMA (the packed source is 16 KiB per stage).
// ================================================================================================
template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, int ABL_ = 0>
struct NvLdsCfg {
  static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
  static constexpr int ABL = ABL_;       // bench-only ablations: 1 no convert/ds_write, 2 no fragment reads, 4 no MFMA, 8 no barrier
  static constexpr int NWAVES = WAVES_M * WAVES_N, THREADS = NWAVES * 64;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, MT = WTM / 32, NT = WTN / 32;
  static constexpr int KS = 64;          // K elements per stage = one scale column tile (4 groups of 16)
  static constexpr int ROWB = 128;       // f16 bytes per row per stage
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, OFF_B = A_BYTES;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int SROW = BN * 2;
  static constexpr int LDS_BYTES = (2 * STAGE_BYTES > BM * SROW) ? 2 * STAGE_BYTES : BM * SROW;
  static_assert(BM + BN == THREADS, "one operand row per thread");
  static_assert(BM % 64 == 0, "A/B producer split must be wave-uniform");
};

struct NvPacked {   // one row-stage in flight: 64 e2m1 + 4 e4m3
  v4i d0, d1;
  uint32_t s;
};

template <class C>
__global__ __launch_bounds__(C::THREADS) void gemm_nvf4_lds_kernel(const NvGemmParams p) {
  constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT;
  __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
  const float alpha_k = *p.alpha;     // [r4] fetched here, not where the epilogue starts

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wave_m = wave / C::WAVES_N, wave_n = wave % C::WAVES_N;
  const int i32 = lane & 31, g = lane >> 5;

  int tile_m, tile_n;
  {
    const int nb = p.tiles_m * p.tiles_n;
    const int b2 = xcd_remap(blockIdx.x, nb);
#if QAMD_CTX_MAGIC_DECODE
    raster_decode(b2, p.tiles_m, p.tiles_n, p.raster_magic, tile_m, tile_n);
#else
    constexpr int GM = 4;
    const int group = GM * p.tiles_n;
    const int gid = b2 / group;
    const int first_m = gid * GM;
    const int gsz = min(p.tiles_m - first_m, GM);
    tile_m = first_m + (b2 % group) % gsz;
    tile_n = (b2 % group) / gsz;
#endif
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int rowbytes = p.K >> 1;
  const int KT = (p.K + C::KS - 1) / C::KS;   // == number of scale column tiles
  const int CB = KT;

  // ---- producer role: thread -> one operand row of the tile (waves 0..BM/64-1: A, the rest: B) -------
  const bool prodB = wave >= BM / 64;
  const int prow = prodB ? tid - BM : tid;
  const uint32_t base_off = prodB ? (uint32_t)n0 * rowbytes : (uint32_t)m0 * rowbytes;
  const uint32_t tot_bytes = prodB ? p.b_bytes : p.a_bytes;
  const __amdgpu_buffer_rsrc_t rD = make_rsrc((prodB ? p.B : p.A) + base_off, tot_bytes - base_off);
  const __amdgpu_buffer_rsrc_t rS = make_rsrc(prodB ? p.SFB : p.SFA, prodB ? p.sfb_bytes : p.sfa_bytes);
  const int voffD = prow * rowbytes;                       // rows past M / N fall off the descriptor -> 0
  const int grow = (prodB ? n0 : m0) + prow;               // scale row in the padded blocked matrix
  const int voffS = (grow >> 7) * CB * 512 + (grow & 31) * 16 + ((grow & 127) >> 5) * 4;
  const int wrow = (prodB ? C::OFF_B : 0) + prow * C::ROWB;   // LDS row base of this thread's row
  const int wsw = (prow >> 1) & 7;
  constexpr int OOB = 0x7fffffff;

  auto load_stage = [&](int kt, NvPacked& P) __attribute__((always_inline)) {
    const int soff = kt * 32;
    int v0 = (soff < rowbytes) ? voffD : OOB;
    int v1 = (soff + 16 < rowbytes) ? voffD + 16 : OOB;
    int vs = (kt < KT) ? voffS : OOB;
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(vs));       // keep the loop branch-free (see gemm_mx.hip.h)
    P.d0 = __builtin_amdgcn_raw_buffer_load_b128(rD, v0, soff, 0);
    P.d1 = __builtin_amdgcn_raw_buffer_load_b128(rD, v1, soff, 0);
    P.s = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rS, vs, kt * 512, 0);
  };
  // convert dwords [u0, u1) of the row-stage and store them as f16 chunks of LDS stage `buf`
  auto convert_part = [&](const NvPacked& P, int kt, int buf, const int u0, const int u1) __attribute__((always_inline)) {
    // scale bytes of groups past K (K % 64 == 32 tail) are layout padding: force them to 0
    const int valid = p.K / 16 - 4 * kt;                   // wave-uniform
    const uint32_t smask = valid >= 4 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
    h2_t s01, s23;
    e4m3x4_to_f16(P.s & smask, s01, s23);
    char* st = smem + buf * C::STAGE_BYTES + wrow;
#pragma unroll
    for (int u = u0; u < u1; ++u) {
      const uint32_t w = (uint32_t)(u < 4 ? P.d0[u & 3] : P.d1[u & 3]);
      const int grp = u >> 1;
      const _Float16 s = grp == 0 ? s01[0] : grp == 1 ? s01[1] : grp == 2 ? s23[0] : s23[1];
      const h8_t f = dq8(w, h2_t{s, s});
      *(h8_t*)(st + ((u ^ wsw) << 4)) = f;
    }
  };

  // ---- consumer role: fragment addresses (chunk 2s+g of K-step s) ---------------------------------
  const int sw = (i32 >> 1) & 7;
  int rdA[4], rdB[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = 2 * s + g;
    rdA[s] = (wave_m * C::WTM + i32) * C::ROWB + ((c ^ sw) << 4);
    rdB[s] = C::OFF_B + (wave_n * C::WTN + i32) * C::ROWB + ((c ^ sw) << 4);
  }

  v16f acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  auto read_frags = [&](int buf, int s, h8_t (&fa)[MT], h8_t (&fb)[NT]) __attribute__((always_inline)) {
    const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[t] = *(const h8_t*)(st + rdA[s] + t * 32 * C::ROWB);
#pragma unroll
    for (int t = 0; t < NT; ++t) fb[t] = *(const h8_t*)(st + rdB[s] + t * 32 * C::ROWB);
  };
  auto mfma_step = [&](const h8_t (&fa)[MT], const h8_t (&fb)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[n], fa[m], acc[m][n], 0, 0, 0);
  };

  // stage kt: MFMAs on LDS[kt&1]; the row-stage kt+1 (in Pn) is converted into LDS[(kt+1)&1] in four
  // parts between the K-steps; then Pn is refilled with stage kt+3.
  auto stage = [&](int kt, NvPacked& Pn) __attribute__((always_inline)) {
    const int buf = kt & 1, nxt = buf ^ 1;
    h8_t fa[2][MT], fb[2][NT];
    if (C::ABL & 2) {
#pragma unroll
      for (int t = 0; t < MT; ++t) fa[0][t] = fa[1][t] = h8_t{(_Float16)1, (_Float16)2, (_Float16)0, (_Float16)1, (_Float16)3, (_Float16)1, (_Float16)2, (_Float16)1};
#pragma unroll
      for (int t = 0; t < NT; ++t) fb[0][t] = fb[1][t] = h8_t{(_Float16)1, (_Float16)1, (_Float16)2, (_Float16)1, (_Float16)0, (_Float16)1, (_Float16)2, (_Float16)3};
    } else
    read_frags(buf, 0, fa[0], fb[0]);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");       // Pn landed (the 3 loads of stage kt+2 may still fly)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3 && !(C::ABL & 2)) read_frags(buf, s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
      if (!(C::ABL & 1)) convert_part(Pn, kt + 1, nxt, 2 * s, 2 * s + 2);
      else if (s == 0) asm volatile("" ::"v"(Pn.d0), "v"(Pn.d1), "v"(Pn.s));
      if (!(C::ABL & 4)) mfma_step(fa[s & 1], fb[s & 1]);
      else {
#pragma unroll
        for (int t = 0; t < MT; ++t) asm volatile("" ::"v"(fa[s & 1][t]));
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" ::"v"(fb[s & 1][t]));
      }
    }
    load_stage(kt + 3, Pn);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(C::ABL & 8)) __syncthreads();
  };

  NvPacked P0, P1;
  load_stage(0, P0);
  load_stage(1, P1);
  asm volatile("" :: "s"(alpha_k));   // alpha is waited for HERE, behind the first loads
  asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  convert_part(P0, 0, 0, 0, 8);
  load_stage(2, P0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  // invariant at stage kt: LDS[kt&1] holds stage kt; stage kt+1 is in P[(kt+1)&1], stage kt+2 in P[kt&1]
  const uint64_t dbg_c0 = p.dbg ? __builtin_readcyclecounter() : 0, dbg_r0 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    stage(kt, P1);
    stage(kt + 1, P0);
  }
  if (kt < KT) stage(kt, P1);
  if (p.dbg && blockIdx.x == 0 && tid == 0) {
    p.dbg[0] = (uint32_t)(__builtin_readcyclecounter() - dbg_c0);
    p.dbg[1] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - dbg_r0);
    p.dbg[2] = (uint32_t)KT;
  }

  const float alpha = alpha_k;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wave_m * C::WTM + 32 * m + i32;
        const int cg = (wave_n * C::WTN + 32 * n + 8 * q + 4 * g) >> 2;
        v2i w;
        w[0] = pack_bf16x2(acc[m][n][4 * q + 0] * alpha, acc[m][n][4 * q + 1] * alpha);
        w[1] = pack_bf16x2(acc[m][n][4 * q + 2] * alpha, acc[m][n][4 * q + 3] * alpha);
        *(v2i*)(smem + row * C::SROW + ((cg ^ (row & 15)) << 3)) = w;
      }
  __syncthreads();
  constexpr int CPR = BN / 8;
  constexpr int RPP = C::THREADS / CPR;
  const int chunk = tid % CPR, r0 = tid / CPR;
  const int gcol = n0 + chunk * 8;
#pragma unroll 4
  for (int pss = 0; pss < BM / RPP; ++pss) {
    const int row = pss * RPP + r0;
    const int grow2 = m0 + row;
    if (grow2 < p.M && gcol < p.N) {
      v4i v = *(const v4i*)(smem + row * C::SROW + ((((2 * chunk) ^ (row & 15)) & ~1) << 3));
      if (row & 1) v = v4i{v[2], v[3], v[0], v[1]};
      *(v4i*)(p.D + (size_t)grow2 * p.ldd + gcol) = v;
    }
  }
}

// ================================================================================================
// Small batch (M <= 32 per tile): split-K kernel without LDS staging, the NVFP4 twin of gemm_mx_skinny.hip.h.
// One workgroup = 32 rows of B x 32 rows of A; the 8 waves split K in 128-byte row segments (256 elements); a wave
// loads its packed operands and e4m3 scale dwords straight from global memory with all loads of its K range in
// flight, dequantises to f16 exactly as the tiled kernel does (cvt + one pk_mul per pair) and feeds
// v_mfma_f32_32x32x16_f16; partial 32x32 tiles are summed through LDS.  Weight-bandwidth bound.
// ================================================================================================
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void gemm_nvf4_skinny_kernel(const NvGemmParams p) {
  __shared__ __attribute__((aligned(16))) float part[NWAVES][32][33];
  const float alpha_k = *p.alpha;     // [r4] fetched here, not behind the K loop (decode shapes: a 5-10 us kernel)
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int i32 = lane & 31, g = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int rowbytes = p.K >> 1;
  const int nseg = (rowbytes + 127) >> 7;
  const int CB = (p.K / 16 + 3) >> 2;              // scale column tiles (4 groups of 16) per row

  const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A + a_off, p.a_bytes - a_off);
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);
  const __amdgpu_buffer_rsrc_t rSA = make_rsrc(p.SFA, p.sfa_bytes), rSB = make_rsrc(p.SFB, p.sfb_bytes);
  const int voff = i32 * rowbytes + g * 64;
  const int ra = m0 + i32, rb = n0 + i32;
  // scale dword of (row, column tile ct): to_blocked layout; lane half g uses tiles 4s + 2g and 4s + 2g + 1
  const int soffA = (ra >> 7) * CB * 512 + (ra & 31) * 16 + ((ra & 127) >> 5) * 4;
  const int soffB = (rb >> 7) * CB * 512 + (rb & 31) * 16 + ((rb & 127) >> 5) * 4;
  constexpr int OOB = 0x7f000000;

  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int s = wave; s < nseg; s += NWAVES) {
    v4i ca[4], cb[4];
    uint32_t da[2], db[2];
    const int base = s * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = (base + g * 64 + j * 16 < rowbytes) ? voff + j * 16 : OOB;   // K tail: chunks past the row read 0
      ca[j] = __builtin_amdgcn_raw_buffer_load_b128(rA, v, base, 0);
      cb[j] = __builtin_amdgcn_raw_buffer_load_b128(rB, v, base, 0);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int ct = 4 * s + 2 * g + jj;
      const int so = (ct < CB) ? ct * 512 : OOB;   // scale tiles past K are layout padding: read 0 (0 x 0, never NaN)
      da[jj] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rSA, soffA + so, 0, 0);
      db[jj] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rSB, soffB + so, 0, 0);
    }
    asm volatile("" :: "s"(alpha_k));   // alpha is waited for HERE, behind the operand loads just issued (left alone, its load is sunk below the loop)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      // groups past K inside the last tile (K % 64 == 32): mask their scale bytes
      const int valid = p.K / 16 - 4 * (4 * s + 2 * g + jj);
      const uint32_t smask = valid >= 4 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
      h2_t sa[2], sb[2];
      e4m3x4_to_f16(da[jj] & smask, sa[0], sa[1]);
      e4m3x4_to_f16(db[jj] & smask, sb[0], sb[1]);
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) {
        const int j = 2 * jj + jl;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const _Float16 xa = sa[jl][u >> 1], xb = sb[jl][u >> 1];
          const h8_t fa = dq8((uint32_t)ca[j][u], h2_t{xa, xa});
          const h8_t fb = dq8((uint32_t)cb[j][u], h2_t{xb, xb});
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc, 0, 0, 0);
        }
      }
    }
  }

#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave][i32][8 * q + 4 * g + e] = acc[4 * q + e];
  __syncthreads();
  const float alpha = alpha_k;
  for (int idx = tid; idx < 32 * 8; idx += NWAVES * 64) {
    const int m = idx >> 3, nq = (idx & 7) * 4;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NWAVES; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[e] += part[w][m][nq + e];
    if (m0 + m < p.M && n0 + nq < p.N) {
      v2i o;
      o[0] = (int)pack_bf16x2(sum[0] * alpha, sum[1] * alpha);
      o[1] = (int)pack_bf16x2(sum[2] * alpha, sum[3] * alpha);
      *(v2i*)(p.D + (size_t)(m0 + m) * p.ldd + n0 + nq) = o;
    }
  }
}

// variant: 3 = small-batch split-K kernel (auto for M <= 32); otherwise 0 = auto = 1 (per-wave dequant; measured equal or slightly faster than v2 on every shape,
// profiles/native_r1_nvfp4_ablation.log -- both are power-bound at the same wall time), 2 = v2 (LDS dequant)
// (Host launchers are compiled only into the translation unit that owns them -- capi.hip, QAMD_TU: a kernel template that
// an inline launcher merely MENTIONS is instantiated by the device pass of every unit that sees the launcher.)
#ifndef QAMD_TU
#define QAMD_TU 0
#endif
#ifndef QAMD_BENCH
#define QAMD_BENCH 0
#endif
#if QAMD_BENCH
#if QAMD_TU == 0 || QAMD_TU == 7
// bench-only ablations of the v2 kernel (variant 10 + b); returns false for any other variant.  (Not inline: the unit that
// owns it must emit it.)
bool launch_nvf4_ablation(NvGemmParams p, hipStream_t s, int variant) {
#define QAMD_NV_ABL(b)                                                                                          \
  if (variant == 10 + b) {                                                                                     \
    using C = NvLdsCfg<256, 256, 2, 4, b>;                                                                     \
    p.tiles_m = (p.M + C::BM - 1) / C::BM;                                                                     \
    p.tiles_n = (p.N + C::BN - 1) / C::BN;                                                                     \
    p.raster_magic = raster_magic(p.tiles_n);          \
    hipLaunchKernelGGL((gemm_nvf4_lds_kernel<C>), dim3(p.tiles_m * p.tiles_n), dim3(C::THREADS), 0, s, p);     \
    return true;                                                                                               \
  }
  QAMD_NV_ABL(1) QAMD_NV_ABL(2) QAMD_NV_ABL(3) QAMD_NV_ABL(4) QAMD_NV_ABL(5) QAMD_NV_ABL(6) QAMD_NV_ABL(8) QAMD_NV_ABL(9) QAMD_NV_ABL(11)
#undef QAMD_NV_ABL
  return false;
}
#else
bool launch_nvf4_ablation(NvGemmParams p, hipStream_t s, int variant);
#endif
#endif   // QAMD_BENCH

// [r4] the persistent 256x256 kernel with stream-K over a part-filled last round (gemm_nvf4_pk.hip.h; its own translation unit)
struct NvPkPlan { int grid, sk_tiles; };
inline bool nvpk_shape_ok(int64_t M, int64_t N, int64_t K) { return M > 0 && N > 0 && K % 256 == 0 && K >= 512; }
// Workgroups and stream-K region for T tiles of 256x256 on `cus` CUs.
//   T a multiple of cus, or no scratch, or less than one round: whole tiles only, BALANCED rounds (R = ceil(T / cus) tiles per workgroup on
//     ceil(T / R) workgroups rounded up to a multiple of 8 -- the rule of the MX persistent kernels, capi.hip deepp_grid)
//   otherwise: one workgroup per CU; the full rounds but the last as whole tiles, the last cus + T % cus tiles as an evenly split stream of K stages
//     (every workgroup walks (cus + T % cus) KT / cus stages: at most one cut tile at each end of its range)
inline NvPkPlan nvpk_plan(int64_t M, int64_t N, int64_t K, int cus, bool may_sk) {
  const int64_t T = ((M + 255) / 256) * ((N + 255) / 256);
  if (may_sk && T > cus && T % cus != 0 && (T % cus) * 8 <= (int64_t)cus * 7) return {cus, (int)(cus + T % cus)};
  const int64_t rounds = (T + cus - 1) / cus;
  const int64_t g = ((T + rounds - 1) / rounds + 7) / 8 * 8;
  return {(int)std::min<int64_t>(std::min<int64_t>(g, cus), T), 0};
}
constexpr int64_t NVPK_PART_BYTES = 256 * 256 * 4;   // one parked tile of fp32 accumulators
// scratch of a stream-K launch on `grid` workgroups: a parked tile per range boundary, then the arrival flags
inline int64_t nvpk_ws_bytes(int grid) { return (int64_t)grid * NVPK_PART_BYTES + (int64_t)grid * 8; }
hipError_t launch_nvf4_pk(NvGemmParams p, hipStream_t s, int grid, bool trace);

// Tile configuration of the auto rule (no GPU touched; also behind qutlass_amd_debug_nvf4_plan for the CPU tests):
//   -1 split-K skinny kernel, 0: 256x256, 1: 128x128, 2: 128x64, 3: 64x64, 4: 256x128 on four waves of 128x64
// [r3] Where 128x128 tiles fill the chip, the three large configurations are priced round by round -- a per-tile kernel runs
// ceil(tiles / slots) rounds, and what a part-filled last round costs decides most shapes that are not powers of two.  Round times in us at
// K = 4096 (tools/calib_tiles.py, profiles/calib_tiles_r3.txt: every candidate forced on 70 shapes), scaled with K:
//   256x256 (one per CU)   81 while <= 3/4 of the CUs have a tile, 101 for a full round (the part's electrical limit)
//   256x128 (one per CU)   51 ... 59.8 likewise
//   128x128 (two per CU)   a full round of 2 x CUs tiles 64.7; a remainder of <= CUs tiles runs one per CU: 34 ... 37; one tile more: 55 ... 64.7
// The rule before: the 256x128 tile whenever its round occupancy beat that of the 128x128 grid -- 2560 x 4096 x 4096 ran 106 us (two rounds of
// 256x128) where 160 tiles of 256x256 take 84.5; 1024 x 4096 ran on 128 tiles, 54.7 us against 35.5.
// (outputs of >= 3/4 x CUs tiles of 128x128 with M, N > 128; below that: nvf4_plan)
inline int nvf4_big_cfg(int64_t M, int64_t N, int64_t K, int cus, double* t_us = nullptr) {
  auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  const double sk = (double)std::max<int64_t>(K, 256) / 4096.0, fix = 5.0;   // per round: `fix` us of prologue / epilogue + a part proportional to K
  auto rt = [&](double t16) { return fix + (t16 - fix) * sk; };
  auto ramp = [&](double lo, double hi, double o) { return lo + (hi - lo) * std::min(1.0, std::max(0.0, (o - 0.75) / 0.25)); };
  const double c = (double)cus;
  // 256x256
  const int64_t t0 = tiles(256, 256), r0 = (t0 - 1) / cus, l0 = t0 - r0 * cus;
  double T0 = r0 * rt(101.0) + rt(ramp(81.0, 101.0, l0 / c));   // the per-tile kernel (K % 256 != 0 or K < 512)
  // [r4] K % 256 == 0: the persistent kernel (gemm_nvf4_pk.hip.h) in balanced rounds -- R = ceil(tiles / CUs) tiles per workgroup on G = ceil(tiles / R)
  // workgroups.  A tile costs 5 us + 5.5 us per K stage with every CU busy and less while CUs idle (socket power limit: the busy ones clock higher),
  // f(G / CUs) through (0, .70) (.5, .73) (.625, .76) (.75, .84) (.875, .91) (1, 1); 3.5 us per launch.  Fitted to the forced-variant calibration
  // profiles/calib_tiles_nvf4_r4b.txt (64 shapes of this regime: the modelled time is within 3.4 % of the measured one, and the modelled choice is the
  // best measured candidate on all 64 -- 14 226 us against 14 525 for the round-3 rule on the same shapes, e.g. 6144 x 4096 x 14336 588 -> 529 us).
  if (nvpk_shape_ok(M, N, K)) {
    const int64_t R = (t0 + cus - 1) / cus;
    const int64_t G = std::min<int64_t>(std::min<int64_t>(((t0 + R - 1) / R + 7) / 8 * 8, cus), t0);
    static constexpr double X[6] = {0.0, 0.5, 0.625, 0.75, 0.875, 1.0}, Y[6] = {0.70, 0.73, 0.76, 0.84, 0.91, 1.0};
    const double o = (double)G / c;
    double f = 1.0;
    for (int i = 0; i < 5; ++i)
      if (o <= X[i + 1]) { f = Y[i] + (Y[i + 1] - Y[i]) * (o - X[i]) / (X[i + 1] - X[i]); break; }
    T0 = 3.5 + (double)R * (5.0 + 5.5 * (double)K / 256.0) * f;
  }
  // 256x128 on four waves
  const int64_t t4 = tiles(256, 128), r4 = (t4 - 1) / cus, l4 = t4 - r4 * cus;
  const double T4 = r4 * rt(59.8) + rt(ramp(51.0, 59.8, l4 / c));
  // 128x128, two per CU
  const int64_t t1 = tiles(128, 128), r1 = t1 / (2 * cus), l1 = t1 - r1 * 2 * cus;
  double T1 = r1 * rt(64.7);
  if (l1 > 0) T1 += l1 <= cus ? rt(34.0 + 3.0 * l1 / c) : rt(55.0 + 9.7 * (double)(l1 - cus) / c);   // (the CUs that hold two tiles set the time: a small spill costs almost the full pair round)
  int cfg = 0;
  double best = T0;
  if (M >= 256 && T4 < best * 0.985) { cfg = 4; best = T4; }   // (ties go to the larger tile)
  if (T1 < best * 0.985) { cfg = 1; best = T1; }
  if (t_us) *t_us = best;
  return cfg;
}

// [r3] The whole rule: tile configuration + number of K ranges (split-K needs caller scratch: may_split).  Shared by the launcher, by
// qutlass_amd_nvf4_splitk_workspace_bytes and by the CPU tests' debug entry (capi.hip); no GPU touched.
//
// Small and mid-size outputs are priced by a second model, fitted (least squares on log time, rms 7 %) to tools/calib_nv_small.py: 132 shapes
// M = 16 ... 1024 x the (N, K) of the reference's benchmark models x {skinny, 64x64, 128x64, 128x128 tiles} x {1, 2, 4, 8} K ranges
// (profiles/calib_nv_small_r3.txt):
//   tile kernels  a + g b kt / 16 [+ reduce pass]   kt = K stages (256 elements) per workgroup, g = 1 while every CU holds at most one workgroup,
//                                                   else ceil(workgroups / CUs) x e (several workgroups on a CU overlap each other: e < 1)
//                 128x128: a 4.1 b 29.2 e 0.91    128x64: a 3.2 b 20.3 e 0.85    64x64: a 3.6 b 13.8 e 0.78   (us; b per 16 stages = K 4096)
//   reduce pass   2.9 us + (S + 1) M N 4 bytes at 4.4 TB/s, and a split must win by 2 % (5 % against the large-output model); a K range is never shorter than 4 stages
//   skinny        2.5 + 4.7 ceil(workgroups of 32x32 / CUs) K / 4096 (refitted on the GPU-only re-take of the calibration, rms 6 %: profiles/calib_nv_small_r3_graph.txt), M <= 256
//                 only; always for M <= 32
// Against the calibration the chosen candidates sum to 10 353 us (best measured candidate per shape: 10 288; the occupancy thresholds this replaces:
// 11 057), e.g. 256 x 4096 x 14336 54.7 -> 39.2 us (128x128 tiles, 4 K ranges), 128 x 8192 x 28672 107 -> 67, 64 x 28672 x 4096 36.7 -> 26.6 (64x64 tiles
// instead of the skinny kernel), 512 x 5120 x 5120 47.0 -> 41.1 (160 tiles of 128x128 instead of 320 of 128x64, which put two on 64 CUs).
struct NvPlan { int cfg, splits, kt_per; };   // cfg as above (-1 skinny, [r6] -2 / -3 wave-owned small-batch kernel with 32 / 16 columns per workgroup, -4 / -5 its 16x16 / 32x16 decode form, -6 ... -9 the decode form with 32 / 48 / 56 / 56 (8 A rows) columns per workgroup, -10 / -11 the 32x32-MFMA kernel on 64x32 / 96x32 tiles); splits = K ranges actually launched (none empty), kt_per = stages per range (even)
// [r6] Does the wave-owned small-batch kernel (gemm_nvf4_os.hip.h) take the shape?  0 (no) or 32 (columns per workgroup; 16 is lab-only: the kernel is bound by its
// dequantisation instructions -- ~16 per MFMA -- not by bytes, so spreading the weight over twice the workgroups buys nothing: profiles/calib_nvos_r6u.txt).
// Measured against the plan before it (skinny / tile kernels / split-K with scratch), M = 1 ... 128:
//   K <= 4096 (one shot): up to THREE rounds of 32x32 tiles (N = 4096, M = 128: 11.5 -> 9.7 us; 6144 x 4096, M = 128 = 768 tiles: 16.8 -> 13.8; four rounds lose)
//   longer K (wave-owned rings, <= 64 stages): one tile per CU at most (4096 x 14336, M <= 64: 14.4-20.0 -> 13.4-15.5 us), two rounds up to 32 stages
//   (4096 x 8192, M = 128: 21.6 -> 18.0), and never fewer than a quarter of the CUs busy (a long K on few tiles stays with the split plans)
// [r6, third session] 1616 = its decode form, 16x16 tiles on v_mfma_f32_16x16x32_f16 (gemm_nvf4_os16_kernel): the same dequantisation instructions per weight element spread
// over twice the workgroups.  Whenever the 16x16 tiles fit one per CU it wins at every K and every workgroup count measured (K = 2048 ... 28672, 32 ... 256 workgroups:
// N = K = 4096, M <= 16: 5.3 -> 3.7-3.9 us; 4096 x 14336: 13.6 -> 9.1-9.7; 1024 x 14336, M = 32: 18.6 -> 8.8; x 28672, M = 16: 30.0 -> 16.4 -- also against the split-K
// plans); two per CU (N = 6144 / 8192, M <= 16) still 4-8 % ahead at K = 4096 and behind from K = 8192 on.  profiles/calib_nv16_r7.txt
inline int nv_os_plan(int64_t M, int64_t N, int64_t K, int cus) {
  const int64_t T32 = ((M + 31) / 32) * ((N + 31) / 32), KT = (K / 2 + 127) / 128;
  if (M > 128) return 0;
  const int64_t G16 = ((M + 15) / 16) * ((N + 15) / 16);
  if (G16 <= cus && KT <= 128) return 1616;
  // M <= 16 against a weight too wide for 16-column workgroups: the decode form with 32 / 48 / 56 columns per workgroup, the A dword dequantised once for 2 / 3 / 4 n-tiles
  // (1632 / 1648 / 1656; 856 = 56 columns with only A rows 0 ... 7 fetched, M <= 8: its 16 stages fit the LDS).  N = 11008 / 12288 x K = 4096: 9.2-10.8 -> 5.8-5.9 us (the
  // 32x32 kernel ran two rounds of tiles); 12288 x 5120 13.6-14.1 -> 8.7-9.2; 14336 x 4096 9.5-9.8 -> 7.6-7.7 (M <= 8) / 8.0-8.3; x 8192 19.2 -> 15.0; N = 6144 / 8192:
  // -5 % at K = 4096, -12 % at K = 8192; 5120^2 -12 ... -15 %.  profiles/calib_nv16w_r7.txt
  if (M <= 16 && KT <= 32) {
    for (int tn : {32, 48, 56})
      if ((N + tn - 1) / tn <= cus) return tn == 56 ? ((M <= 8 && KT <= 16) ? 856 : 1656) : 1600 + tn;
    // wider still (the fused up / gate projections, N = 24576 / 28672): 48 columns per workgroup in two or three rounds, K <= 4096 (24576 x 4096: 14.9-15.4 -> 13.2 us,
    // 28672 x 4096, M = 16: 20.6 -> 17.6; 16384: a tie, left alone)
    if (KT <= 16 && N >= 20480 && (N + 47) / 48 <= 3 * (int64_t)cus) return 1648;
  }
  // 3216 = the same with two m-tiles per workgroup (a B dword dequantised once for both): where 32x16 tiles fit one per CU (N = 4096, M = 17 ... 32: K = 4096 5.35 -> 4.75 us,
  // K = 14336 14.1 -> 11.7, K = 28672 30-36 -> 20)
  if (((M + 31) / 32) * ((N + 15) / 16) <= cus && KT <= 128) return 3216;
  // one round of 32x32 tiles: here; [r6, third session] several rounds (and the 64x32 form) are priced in nvf4_plan against the fitted models of the other kernels
  if (T32 <= cus && (KT <= 16 || (KT <= 64 && 4 * T32 >= cus))) return 32;
  return 0;
}
hipError_t launch_nvf4_os(NvGemmParams p, hipStream_t s, int tn);   // capi.hip (the NVFP4 unit)
inline NvPlan nvf4_plan(int64_t M, int64_t N, int64_t K, int cus, bool may_split) {
  if (const int tn = nv_os_plan(M, N, K, cus)) return {tn == 856 ? -9 : tn == 1656 ? -8 : tn == 1648 ? -7 : tn == 1632 ? -6 : tn == 3216 ? -5 : tn == 1616 ? -4 : tn == 16 ? -3 : -2, 1, 0};
  if (M <= 32) {   // one m-tile: the wave-owned 32x32 kernel in up to four rounds of tiles (1.75 + 4.05 us per round and 16 stages, K <= 8192) or the split-K kernel (2.53 + 4.66)
    const int64_t KT0 = (K / 2 + 127) / 128, R0 = (((N + 31) / 32) + cus - 1) / cus;
    return {(KT0 <= 32 && R0 <= 4) ? -2 : -1, 1, 0};
  }
  auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  const int KT = (int)((K / 2 + 127) / 128);
  static constexpr int BM[3] = {128, 128, 64}, BN[3] = {128, 64, 64};
  static constexpr double A[3] = {4.107, 3.241, 3.616}, B[3] = {29.172, 20.30, 13.831}, E[3] = {0.906, 0.853, 0.776};
  NvPlan best = {1, 1, 0};
  double best_t = 1e30;
  for (int c = 0; c < 3; ++c)
    for (int S = 1; S <= (may_split && N % 4 == 0 ? 8 : 1); S *= 2) {
      const int kt = 2 * ((KT + 2 * S - 1) / (2 * S)), S2 = (KT + kt - 1) / kt;
      if (S > 1 && (kt < 4 || S2 < 2)) continue;
      const double n = (double)(tiles(BM[c], BN[c]) * S2) / cus;
      const double g = n <= 1.0 ? 1.0 : std::ceil(n) * E[c];
      double t = A[c] + g * B[c] * kt / 16.0;
      if (S2 > 1) t = (t + 2.897 + (double)(S2 + 1) * M * N * 4.0 / 4.409e6) * 1.02;
      if (t < best_t) { best_t = t; best = {c + 1, S2, S2 > 1 ? kt : 0}; }
    }
  if (M <= 384) {
    const double wg = (double)(((M + 31) / 32) * ((N + 31) / 32));
    const double t_skinny = M <= 256 ? 2.53 + 4.66 * std::ceil(wg / cus) * K / 4096.0 : 1e30;
    // [r6] the wave-owned 32x32 kernel where nv_os_plan's occupancy rule passed the shape on (a part-filled last round, or M = 129 ... 256): up to four rounds of tiles with
    // K <= 4096 cost 1.75 + 4.05 us per round (N = 4096, M = 64 / 128: 5.7 / 9.8 us; 6144 x 4096, M = 128: 13.85) against 2.53 + 4.66 per round for the split-K kernel --
    // the dip scan found M = 192 slower than M = 256 at N = K = 4096 (16.8 vs 15.7 us) and M = 96 slower than M = 128 at N = 6144 (16.9 vs 13.9): profiles/dip_scan_r7z.txt
    const double rounds = std::ceil(wg / cus);
    // ... and the same on its wave-owned rings up to K = 8192 (+5 % per round: 2048 x 8192, M = 160 ... 256: 19.0-23.1 -> 17.0-17.8 us; 5120^2, M = 64 / 128: 14.1 / 19.9 -> 12.6 / 18.5)
    // (four rounds: 6144 x 4096, M = 160: 21.4 -> 17.7 us; where 64x64 tiles fit one round -- N = 4096, M = 256: 15.5 -- their model is lower and they stay)
    double t_os = 1e30, t_os64 = 1e30;
    if (KT <= 32 && rounds <= 4.0) t_os = 1.75 + 4.05 * rounds * KT / 16.0 * (KT > 16 ? 1.05 : 1.0);
    // ... and its 64x32 form (two m-tiles per workgroup, a B dword dequantised once for both; one slot per wave, refilled behind the stage's dequantisation): 6.6 us per round and
    // 16 stages up to K = 4096, 5.8 beyond -- N = 4096, M = 96 / 128: 9.6 -> 8.4-8.5 us (ONE round instead of two), x 8192: 17.3-17.8 -> 12.9-13.3; 5120^2, M = 64: 12.6 -> 9.1;
    // 8192^2, M = 64: 19.5 -> 15.6; two rounds only beyond K = 4096 (N = 4096, M = 256 at K = 4096: 16.1 against 15.5 on 64x64 tiles).  profiles/calib_nv6432_r7.txt
    {
      const double rounds64 = std::ceil((double)(((M + 63) / 64) * ((N + 31) / 32)) / cus);
      if (M > 32 && KT <= 64 && (rounds64 <= 1.0 || (rounds64 <= 2.0 && KT > 16))) t_os64 = 1.75 + (KT <= 16 ? 6.6 : 5.8) * rounds64 * KT / 16.0;
    }
    // ... and three m-tiles per workgroup (96x32 tiles: 10.7 vector instructions per MFMA; 18 KiB per stage, still one slot per wave): 9.2 us per round and 16 stages up to K = 4096,
    // 8.0 beyond -- N = 4096, M = 160 / 192: 13.6-13.9 -> 10.9-11.1 us (ONE round of 256 workgroups), x 8192: 24 -> 17.8-18.3; 6144 / 8192 x 4096, M = 96: 13.3 / 13.8 -> 10.9 / 11.4;
    // 8192^2, M = 96: 26 -> 20.2; two rounds: 8192 x 4096, M = 192: 22.2 -> 20.9.  profiles/calib_nv9632_r7.txt
    double t_os96 = 1e30;
    {
      const double rounds96 = std::ceil((double)(((M + 95) / 96) * ((N + 31) / 32)) / cus);
      if (M > 64 && KT <= 64 && rounds96 <= 2.0) t_os96 = 1.75 + (KT <= 16 ? 9.2 : 8.0) * rounds96 * KT / 16.0;
    }
    if (t_os96 < t_os64 && t_os96 < t_os && t_os96 < t_skinny && t_os96 < best_t) return {-11, 1, 0};
    if (t_os64 < t_os && t_os64 < t_skinny && t_os64 < best_t) return {-10, 1, 0};
    if (t_os < t_skinny && t_os < best_t) return {-2, 1, 0};
    if (t_skinny < best_t) return {-1, 1, 0};
  }
  if (M > 128 && N > 128 && tiles(128, 128) >= cus * 3 / 4) {   // the large-output model above; a split only where it beats that model's time by 5 %
    double tb;
    const int cb = nvf4_big_cfg(M, N, K, cus, &tb);
    if (!(best.splits > 1 && best_t < tb * 0.95)) return {cb, 1, 0};
  }
  return best;
}

#if QAMD_TU == 0 || QAMD_TU == 4
// Returns hipErrorInvalidValue for a variant this build does not know (the product library knows only 0 = auto).
// cus: compute units of the device (capi.hip chip_cus): every occupancy threshold below derives from it
// *splits_out (when given): the number of K ranges the launch wrote to p.ws -- the caller then runs splitk_reduce_kernel; 1 = D is final
inline hipError_t launch_nvf4_gemm(NvGemmParams p, hipStream_t s, int variant = 0, int cus = 256, int* splits_out = nullptr) {
  const int64_t want = cus * 3 / 4;
  if (splits_out) *splits_out = 1;
  int force_split = 0;
#if QAMD_BENCH
  bool fenced = false;
  if (variant >= 160 && variant < 170) {   // lab: 160 + S = 128x128 tiles on TWO waves of 128x64 (6 fragment dequantisations per 8 MFMAs), S K ranges
    force_split = variant - 160;
    variant = 8;
    if (force_split < 1) return hipErrorInvalidValue;
  }
  if (variant >= 150 && variant < 160) { fenced = true; variant -= 40; }   // lab: 150 + S = 128x128 tiles, S K ranges (1 = none), fixed MFMA / dequantisation order
  if (variant >= 100 && variant < 140) {   // lab: 100 + 10 cfg + S = tile cfg (1 128x128, 2 128x64, 3 64x64) with S K ranges (tools/calib_nv_small.py)
    force_split = variant % 10;
    variant = 4 + (variant - 100) / 10;    // 5 / 6 / 7 force the tile below
    if ((force_split < 2 && !fenced) || force_split < 1 || variant < 5 || variant > 7) return hipErrorInvalidValue;
  }
#endif
#if !QAMD_BENCH
  if (variant != 0) return hipErrorInvalidValue;
#endif
  // 128x128 tiles when a dimension is small OR when 256x256 tiles would leave most CUs without work
  const bool small = p.M <= 128 || p.N <= 128 || (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256) < want;
  // small batch: split-K kernel for M <= 64, and up to M = 128 while even 64x64 tiles would leave CUs idle (measured,
  // M = 128, K = 4096: N = 4096 11.8 us vs 17.3 us for 64x64 tiles; N = 14336 35.6 us vs 25.2 us for 128x64 tiles)
  const NvPlan plan = nvf4_plan(p.M, p.N, p.K, cus, p.ws != nullptr);
#if QAMD_BENCH
  if (variant == 46 || variant == 47) return launch_nvf4_os(p, s, variant == 47 ? 16 : 32);   // lab: force the wave-owned small-batch kernel (any K: rings beyond 4096)
  if (variant == 48) return launch_nvf4_os(p, s, 1616);                                       // lab: ... its 16x16 decode form (any M: rows in tiles of 16)
  if (variant == 49) return launch_nvf4_os(p, s, 3216);                                       // lab: ... with two m-tiles per workgroup (32x16)
  if (variant == 55) return launch_nvf4_os(p, s, 9632);                                       // lab: ... with three m-tiles per workgroup (96x32 tiles)
  if (variant == 54) return launch_nvf4_os(p, s, 6432);                                       // lab: the 32x32-MFMA kernel with two m-tiles per workgroup (64x32 tiles)
  if (variant >= 50 && variant <= 53) return launch_nvf4_os(p, s, variant == 50 ? 1632 : variant == 51 ? 1648 : variant == 52 ? 1656 : 856);   // lab: ... with 32 / 48 / 56 columns per workgroup (53: 56 columns, A rows 0 ... 7 only)
#endif
  if (variant == 0 && plan.cfg <= -2) return launch_nvf4_os(p, s, plan.cfg == -11 ? 9632 : plan.cfg == -10 ? 6432 : plan.cfg == -9 ? 856 : plan.cfg == -8 ? 1656 : plan.cfg == -7 ? 1648 : plan.cfg == -6 ? 1632 : plan.cfg == -5 ? 3216 : plan.cfg == -4 ? 1616 : plan.cfg == -3 ? 16 : 32);
  if (variant == 3 || (variant == 0 && plan.cfg < 0)) {
    hipLaunchKernelGGL((gemm_nvf4_skinny_kernel<8>), dim3((p.N + 31) / 32, (p.M + 31) / 32), dim3(512), 0, s, p);
    return hipSuccess;
  }
#if QAMD_BENCH
  if (variant == 1 && !small) {   // lab: the round-1 choice for 256x256 tiles, 8 waves of 128x64 (each A chunk dequantised by 4 waves, each B chunk by 2)
    using C = NvCfg<256, 256, 2, 4>;
    p.tiles_m = (p.M + C::BM - 1) / C::BM;
    p.tiles_n = (p.N + C::BN - 1) / C::BN;
    p.raster_magic = raster_magic(p.tiles_n);
    hipLaunchKernelGGL((gemm_nvf4_kernel<C>), dim3(p.tiles_m * p.tiles_n), dim3(C::THREADS), 0, s, p);
    return hipSuccess;
  }
#endif
#if QAMD_BENCH
  // lab: 42 = the persistent 256x256 kernel, whole tiles in balanced rounds; 43 = with stream-K over the last round (needs scratch); 44 / 45 = 42 / 43 + stage trace
  const bool pk_forced = variant >= 42 && variant <= 45;
  const bool pk_sk = variant == 43 || variant == 45, pk_trace = variant == 44 || variant == 45;
  if (pk_forced) variant = 41;
#else
  constexpr bool pk_forced = false, pk_trace = false;
#endif
  if (variant <= 1 || variant == 4 || (variant >= 5 && variant <= 9) || variant == 40 || variant == 41) {
    // tile choice by occupancy (as for the MX kernels): the largest tile that gives >= 192 workgroups
    auto tiles = [&](int bm, int bn) { return (int64_t)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
    int cfg = small ? 1 : 0;                                  // 0: 256x256, 1: 128x128, 2: 128x64, 3: 64x64
    if (variant == 0 || variant == 1) {
      // the pricing of nvf4_plan above ([r2] wave quantisation of the 256x256 grid, [r3] the 256x128 four-wave tile and the cost model)
      if (plan.cfg >= 0) cfg = plan.cfg;
    }
    if (variant == 5) cfg = 1;
    if (variant == 6) cfg = 2;
    if (variant == 7) cfg = 3;
#if QAMD_BENCH
    if (variant == 40) cfg = 4;   // lab: force the 256x128 tile on four waves of 128x64
    if (variant == 41) cfg = 0;   //      force the 256x256 tile (tools/calib_tiles.py)
    if (variant == 8) cfg = 8;    // lab: 128x128 tile on 2 waves of 128x64 (A dequantised by 2 waves, B by 1)
    if (variant == 9) cfg = 9;    //      128x128 tile on 2 waves of 64x128
#endif
    // [r3] split-K (caller scratch in p.ws, capi.hip checks its size against nvf4_plan): K ranges of an even number of 256-element stages, none empty
    int splits = 1;
    if (p.ws && p.N % 4 == 0) {
      splits = force_split ? force_split : (variant == 0 ? plan.splits : 1);
      const int KT = (p.K / 2 + 127) / 128;
      p.kt_per = 2 * ((KT + 2 * splits - 1) / (2 * splits));
      splits = (KT + p.kt_per - 1) / p.kt_per;
    }
    p.splits = splits;
    if (splits <= 1) p.ws = nullptr;
    if (splits_out) *splits_out = splits;
    // [r4] 256x256 tiles: the persistent kernel (lab: variant 41 keeps the per-tile kernel for A/B runs)
    if (cfg == 0 && nvpk_shape_ok(p.M, p.N, p.K) && (variant <= 1 || pk_forced)) {
      void* scratch = p.ws;   // (splits == 1 for this configuration: the caller's scratch is free for parked tiles; capi.hip checked its size)
#if QAMD_BENCH
      const bool may_sk = scratch != nullptr && pk_sk;   // stream-K: lab only (variants 43 / 45); measured equal to balanced rounds for this power-bound kernel
#else
      const bool may_sk = false;
#endif
      const NvPkPlan pl = nvpk_plan(p.M, p.N, p.K, cus, may_sk);
      p.sk_tiles = pl.sk_tiles;
      p.sk_ws = (float*)scratch;
      p.sk_flags = (unsigned long long*)((char*)scratch + (int64_t)pl.grid * NVPK_PART_BYTES);
      p.ws = nullptr;
      return launch_nvf4_pk(p, s, pl.grid, pk_trace);
    }
    if (cfg == 0) p.ws = nullptr;
#define QAMD_NV_LAUNCH_SPLIT(BM_, BN_, WM_, WN_)                                                               \
    if (splits > 1) {                                                                                          \
      using C = NvCfg<BM_, BN_, WM_, WN_>;                                                                     \
      p.tiles_m = (p.M + C::BM - 1) / C::BM;                                                                   \
      p.tiles_n = (p.N + C::BN - 1) / C::BN;                                                                   \
      p.raster_magic = raster_magic(p.tiles_n);          \
      hipLaunchKernelGGL((gemm_nvf4_kernel<C, true>), dim3(p.tiles_m * p.tiles_n * splits), dim3(C::THREADS), 0, s, p); \
      return hipSuccess;                                                                                       \
    }
#define QAMD_NV_LAUNCH(BM_, BN_, WM_, WN_)                                                                     \
    {                                                                                                          \
      using C = NvCfg<BM_, BN_, WM_, WN_>;                                                                     \
      p.tiles_m = (p.M + C::BM - 1) / C::BM;                                                                   \
      p.tiles_n = (p.N + C::BN - 1) / C::BN;                                                                   \
      p.raster_magic = raster_magic(p.tiles_n);          \
      hipLaunchKernelGGL((gemm_nvf4_kernel<C>), dim3(p.tiles_m * p.tiles_n), dim3(C::THREADS), 0, s, p);       \
      return hipSuccess;                                                                                       \
    }
    // [r2] 256x256 tiles: 4 waves of 128x128 -- every operand chunk is dequantised by 2 waves (8 waves of 128x64: A by 4, B by 2),
    // 64 converts / multiplies per 16 MFMAs instead of 48 per 8: +1.7 % (4096^3) .. +3.3 % (8192^3) in the steady state
    // (profiles/native_r2_nvsteady.log)
    if (cfg == 0) QAMD_NV_LAUNCH(256, 256, 2, 2)
    // [r3] 256x128 on four waves of 128x64: 6 fragment dequantisations per 8 MFMAs (0.75 per MFMA) against 4 per 4 for the 64x64 wave tiles of the 128x128 tile
    if (cfg == 4) QAMD_NV_LAUNCH(256, 128, 2, 2)
#if QAMD_BENCH
    if (cfg == 8) { QAMD_NV_LAUNCH_SPLIT(128, 128, 1, 2) QAMD_NV_LAUNCH(128, 128, 1, 2) }
    if (cfg == 9) QAMD_NV_LAUNCH(128, 128, 2, 1)
#endif
#if QAMD_BENCH
    if (cfg == 1 && fenced) {
      using C = NvCfg<128, 128, 2, 2>;
      p.tiles_m = (p.M + C::BM - 1) / C::BM;
      p.tiles_n = (p.N + C::BN - 1) / C::BN;
      p.raster_magic = raster_magic(p.tiles_n);
      if (splits > 1) hipLaunchKernelGGL((gemm_nvf4_kernel<C, true, true>), dim3(p.tiles_m * p.tiles_n * splits), dim3(C::THREADS), 0, s, p);
      else hipLaunchKernelGGL((gemm_nvf4_kernel<C, false, true>), dim3(p.tiles_m * p.tiles_n), dim3(C::THREADS), 0, s, p);
      return hipSuccess;
    }
#endif
    if (cfg == 1) { QAMD_NV_LAUNCH_SPLIT(128, 128, 2, 2) QAMD_NV_LAUNCH(128, 128, 2, 2) }
    if (cfg == 2) { QAMD_NV_LAUNCH_SPLIT(128, 64, 2, 2) QAMD_NV_LAUNCH(128, 64, 2, 2) }
    QAMD_NV_LAUNCH_SPLIT(64, 64, 2, 2)
    QAMD_NV_LAUNCH(64, 64, 2, 2)
#undef QAMD_NV_LAUNCH
#undef QAMD_NV_LAUNCH_SPLIT
  }
#if QAMD_BENCH
  if (variant >= 10 && launch_nvf4_ablation(p, s, variant)) return hipSuccess;
  if (variant != 2) return hipErrorInvalidValue;
  if (small) {
    using C = NvLdsCfg<128, 128, 2, 2>;
    p.tiles_m = (p.M + C::BM - 1) / C::BM;
    p.tiles_n = (p.N + C::BN - 1) / C::BN;
    p.raster_magic = raster_magic(p.tiles_n);
    hipLaunchKernelGGL((gemm_nvf4_lds_kernel<C>), dim3(p.tiles_m * p.tiles_n), dim3(C::THREADS), 0, s, p);
  } else {
    using C = NvLdsCfg<256, 256, 2, 4>;
    p.tiles_m = (p.M + C::BM - 1) / C::BM;
    p.tiles_n = (p.N + C::BN - 1) / C::BN;
    p.raster_magic = raster_magic(p.tiles_n);
    hipLaunchKernelGGL((gemm_nvf4_lds_kernel<C>), dim3(p.tiles_m * p.tiles_n), dim3(C::THREADS), 0, s, p);
  }
  return hipSuccess;
#else
  return hipErrorInvalidValue;
#endif
}
#endif   // QAMD_TU == 0 || QAMD_TU == 4

}  // namespace qamd
