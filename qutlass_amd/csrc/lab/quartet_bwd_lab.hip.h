// LAB ONLY (-DQAMD_BENCH=1): QAT-backward kernels that were built, are bit-identical to the product kernels and are NOT faster.  The product translation units do not read
// this file (tests/test_cabi_and_host.py: test_product_build_does_not_see_the_lab_sources).
#pragma once
#include "../quartet_bwd.hip.h"

namespace qamd {

// ----------------------------------------------------------------------------------------------------------------
// [r5] bwd_qt_panel_kernel: backward_qt_bf16 with WHOLE 128-byte lines on both sides.
//
// tests/native/xpose_traffic_ubench.hip moves the bytes of this op through LDS with no arithmetic at all and finds the tile shape, not the instruction count,
// setting the time: 32-byte input pieces (what a [32 n][64 m] wave tile reads of each e2m1 line) 32.5 us at 8192^2 cold, whole lines in and out 17.0 us, longer
// runs than a line no better (a load instruction pays per line it touches, whatever it uses of it).  So the unit here is [256 n][256 m]: wave w of 8 owns scale
// group w = 32 input rows x 128 bytes, fetched as four 1-KiB LDS-DMA pieces of 8 whole lines (+ one dword piece for the 32 x 8 scale bytes) into a wave-private
// raw panel, 16-byte chunks XOR-swizzled by row.  The wave takes the panel into registers in one go (lane = row, 8 bytes = 16 codes of each of the 8 32-column
// blocks: 16 registers), re-arms the DMA for its NEXT unit at once -- a whole unit of arithmetic hides that fetch -- and then walks the four [32 n][64 m]
// sub-tiles exactly as bwd_quant_t_kernel does: dequantise into the bf16 tile, transposing reads, two K = 16 MFMAs per 32 rows, division-free scales, the
// workgroup's [64 m][8 groups] output block staged and stored as whole lines + 8 scale bytes per row.  81 408 bytes of LDS: two workgroups per CU.
// Needs M % 128 == 0 (dword-aligned scale pieces).
// MEASURED (profiles/ab_bwd_r5w_panel_and_ring.txt): 8192^2 cold 25.2 us against 26.1 (wave-owned segments) and 22.4 (bwd_qt_ring_kernel), warm 22.0 against 19.0;
// 4096^2 cold 10.2 against 8.3.  Its memory side is the best of the three (cold - warm = 3 us) but eight workgroup barriers per unit and 16 waves per CU cost more
// than that saves.  Kept here as the record of the attempt.
template <bool HWCVT>
__global__ __launch_bounds__(512) void bwd_qt_panel_kernel(const BwdTParams p) {
  constexpr int LROW = 144;          // bf16 tile row stride: conflict-free 16-byte writes from 8 consecutive rows (36 dwords = 4 banks apart)
  constexpr int HROW = 32 * 2 + 16;
  constexpr int OROW = 128 + 16;
  constexpr int RAW = 4096 + 256;    // [32 n][128 B] e2m1 + [32 n][8] e8m0
  __shared__ __attribute__((aligned(16))) char raw_s[8][RAW];
  __shared__ __attribute__((aligned(16))) char tile_s[8][32 * LROW];
  __shared__ __attribute__((aligned(16))) char out_s[64 * OROW];   // (the staged H^T borrows it before the first unit)
  __shared__ __attribute__((aligned(16))) uint8_t sf_s[64 * 8];
  static_assert(32 * HROW <= 64 * OROW, "hT");

  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  const int row = lane & 31, half = lane >> 5;
  char* ts = tile_s[wave];
  char* raw = raw_s[wave];
  const float alpha = *p.alpha;
  const int G = p.N >> 5;
  const int ngb = (G + 7) >> 3, n_q = (p.M + 255) >> 8;
  const uint32_t OOB = 0x80000000u;
  const uint32_t rowb = (uint32_t)p.M >> 1, srowb = (uint32_t)p.M >> 5;

  // ---- units: (b, block of 8 groups o, quad of m-tiles i), i fastest.  Workgroup ids go round the 8 XCDs; id = 128 k + 8 j + x takes unit 128 k + 16 x + j,
  // so that the 16 units along m that share the 128-byte lines of input scale bytes sit on ONE XCD (they were fetched once per XCD: 1.41 x the input).
  const uint32_t U = (uint32_t)((int64_t)p.B * ngb * n_q), U128 = U & ~127u;
  struct Unit { int b, g0, m0; int64_t in, grp; };
  auto decode = [&](uint32_t id) __attribute__((always_inline)) {
    const uint32_t u = id < U128 ? (id & ~127u) + ((id & 7u) << 4) + ((id & 127u) >> 3) : id;
    const uint32_t i = u % (uint32_t)n_q, q = u / (uint32_t)n_q, o = q % (uint32_t)ngb, b = q / (uint32_t)ngb;
    Unit r;
    r.b = uniform((int)b); r.g0 = uniform((int)o * 8); r.m0 = uniform((int)i * 256);
    r.in = ((int64_t)r.b * p.N + (int64_t)r.g0 * 32) * p.M + r.m0;     // element index of the unit's first input element
    r.grp = ((int64_t)r.b * p.M + r.m0) * G + r.g0;                    // index of its first output scale group
    return r;
  };

  // ---- the wave's fetch: 4 x (8 rows x 128 B) + 1 x (32 rows x 8 B), straight into its raw panel ----------------------------
  // e2m1: lane -> row lane / 8 (+ 8 per piece), LDS slot lane % 8 holds chunk slot ^ (row % 8);  e8m0: lane -> row lane / 2, dword lane % 2
  const int cg = (lane & 7) ^ (lane >> 3);
  const uint32_t q_off = (uint32_t)(lane >> 3) * rowb + (uint32_t)cg * 16u;
  const uint32_t e_off = (uint32_t)(lane >> 1) * srowb + (uint32_t)(lane & 1) * 4u;
  auto fetch = [&](const uint32_t id) __attribute__((always_inline)) {
    if (id >= U) return;
    const Unit t = decode(id);
    const bool live = t.g0 + wave < G;
    const int64_t e0 = t.in + (int64_t)wave * 32 * p.M;
    const __amdgpu_buffer_rsrc_t r = make_rsrc((const char*)p.xq + (e0 >> 1), live ? 32u * rowb : 0u);
    const __amdgpu_buffer_rsrc_t re = make_rsrc((const char*)p.xs + (e0 >> 5), live ? 32u * srowb : 0u);
    const uint32_t vq = (t.m0 + 32 * cg < p.M) ? q_off : OOB;                   // (dropped chunks land as zeros: code 0 under scale byte 0)
    const uint32_t ve = (t.m0 + 128 * (lane & 1) < p.M) ? e_off : OOB;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(raw + ps * 1024), 16, (int)vq, (int)(ps * 8 * rowb), 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(re, (lds_ptr_t)(raw + 4096), 4, (int)ve, 0, 0, 0);
  };
  // ---- panel -> registers: lane = row lane % 32; of every 32-column block c it takes the 8-byte half hh (the halves swap every 8 rows: 16 lanes = rows
  // r .. r + 15 then read 16 different 8-byte pieces of the 8 slots) ---------------------------------------------------------------
  const int hh = half ^ ((row >> 3) & 1);
  const char* rd = raw + row * 128 + hh * 8;
  const int r7s = (row & 7) << 4;
  v2i qv[8], ev;
  auto take = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 8; ++c) qv[c] = *(const v2i*)(rd + ((c << 4) ^ r7s));
    ev = *(const v2i*)(raw + 4096 + row * 8);
  };

  uint32_t id = blockIdx.x;
  fetch(id);
  {   // hT[j][k] = h[k][j], staged in the output area and read once
    char* hT = out_s;
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
  v8bf hf[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) hf[kc] = *(const v8bf*)(out_s + row * HROW + (kc * 16 + half * 8) * 2);
  __syncthreads();   // every wave has its fragments: the output area is free

  const uint32_t alpha_bits = __float_as_uint(alpha);
  const bool alpha_fast = alpha_bits >= 0x30800000u && alpha_bits <= 0x4e800000u;
  const uint32_t Kexp = alpha_bits - 0x3f800000u;
  const float c3 = 3.0f / alpha;
  const char* tr_ptr = ts + (8 * half + ((lane & 15) >> 2)) * LROW + (((lane & 31) >> 4) * 16 + (lane & 3) * 4) * 2;
  const uint32_t st_off = (uint32_t)(tid >> 3) * (uint32_t)G * 16u + (uint32_t)(tid & 7) * 16u;   // output piece: row tid / 8, group tid % 8

  if (id < U) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    take();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fetch(id + gridDim.x);
  }
  while (id < U) {   // uniform over the workgroup
    const Unit t = decode(id);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m0 = t.m0 + mt * 64;
      const bool any = m0 < p.M;               // (a quad past the last m-tile: nothing staged, nothing stored -- uniform)
      if (any) {
        // ---- dequantise the [32 n][64 m] sub-tile: blocks 2 mt, 2 mt + 1; the lane's 16 codes of a block -> 32 bytes of bf16 at column 32 cb + 16 hh
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int c = 2 * mt + cb;
          const uint32_t e = ((uint32_t)ev[c >> 2] >> (8 * (c & 3))) & 0xffu;
          const float sc = e == 255u ? 1.0f : __uint_as_float(e << 23);   // (byte 0 -> 0.0, byte 255 = +inf: see bwd_quant_t_kernel)
          v4i* d = (v4i*)(ts + row * LROW + cb * 64 + hh * 32);
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const uint32_t w = (uint32_t)qv[c][qq];
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
        __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): the wave's own LDS writes landed (wave-private tile)
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
          *(v2i*)(out_s + mloc * OROW + wave * 16 + half * 8) = ov;
          if (half == 0) sf_s[mloc * 8 + wave] = (uint8_t)(sb >> 23);
        };
        if (!slow_wave) {
          emit(0, std::false_type{});
          emit(1, std::false_type{});
        } else {
          emit(0, std::true_type{});
          emit(1, std::true_type{});
        }
      }
      if (mt == 3) {
        // every block of this unit is in the bf16 tile or beyond: take the next unit's panel (its DMA went out a whole unit ago; the stores still in flight
        // are at least a sub-tile old) and re-arm the fetch for the one after it -- BEFORE this sub-tile's barrier and store
        if (id + gridDim.x < U) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          take();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          fetch(id + 2 * gridDim.x);
        }
      }
      // (bare barriers: __syncthreads() is s_waitcnt vmcnt(0) first, i.e. it would wait for the fetch issued two lines up and for every store in flight)
      asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");   // the workgroup's [64 m][8 groups] output block is staged
      if (any) {
        const int rows = min(64, p.M - m0);
        const int64_t grp0 = t.grp + (int64_t)mt * 64 * G;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + grp0 * 16, (uint32_t)rows * (uint32_t)G * 16u - (uint32_t)t.g0 * 16u);
        const uint32_t so = (t.g0 + (tid & 7) < G) ? st_off : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(*(const v4i*)(out_s + (tid >> 3) * OROW + (tid & 7) * 16), ro, (int)so, 0, 0);
        if (tid < 64 && m0 + tid < p.M) {
          uint8_t* dst = p.out_sf + grp0 + (int64_t)tid * G;
          if ((G & 7) == 0) {
            *(v2i*)dst = *(const v2i*)(sf_s + tid * 8);
          } else {
            for (int k = 0; k < 8 && t.g0 + k < G; ++k) dst[k] = sf_s[tid * 8 + k];
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");   // staging area free for the next sub-tile (its reads have returned: the stores took their data)
    }
    id += gridDim.x;
  }
}


// ----------------------------------------------------------------------------------------------------------------
// [r5] The same tile, PERSISTENT: a workgroup walks tiles t, t + grid, ... with the NEXT tile's rows (2 x 16 bytes + 2 scale bytes per lane) in flight while this one is
// transposed, requantised and stored -- the one-shot form exposes a cold memory round trip per tile to every wave (60 % of its wave cycles wait; cold - warm = 7 us at
// 8192^2).  Bare barriers (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() is s_waitcnt vmcnt(0) first, i.e. it would wait for the prefetch.
// MEASURED (profiles/ab_transpose_r5ac_persistent_prefetch.txt), byte-identical: 8192^2 cold 29.4 us against 28.3, warm 21.1 against 21.1; 2048 x 14336 / 14336 x 2048 cold
// 12.5-12.8 against 13.8-13.9; 16384 x 8192 warm 43.5 against 37.8.  Hiding the load latency buys nothing at 8192^2: cold - warm stays 7-8 us with the prefetch, i.e. the cold
// penalty is bandwidth (scattered 128-byte lines written while the input streams in), not exposed latency.  Kept here as the record.
template <int NC, int MR = 128>
__global__ __launch_bounds__(MR * 2) void mxfp4_transpose_mxfp8_pp_kernel(const TrParams p) {
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
  const unsigned ntiles = (unsigned)(p.m_pad / MR) * (unsigned)tiles_n;
  constexpr int NPS = 32 / RPP;
  struct Tile { int r0, c0, m0; };
  auto decode = [&](unsigned t) __attribute__((always_inline)) {
    if (NC == 128 && t < (ntiles & ~15u)) {   // (the pairing of the one-shot kernel: the two tiles that share input lines on one XCD; gridDim.x % 16 == 0 keeps a workgroup's XCD)
      const unsigned x = t & 7u, k = t >> 3;
      t = 2u * ((k >> 1) * 8u + x) + (k & 1u);
    }
    const int ti = (int)(t / (unsigned)tiles_n), tj = (int)(t % (unsigned)tiles_n);
    return Tile{ti * MR + wave * 32, tj * NC, ti * MR};
  };
  v4i vld[NPS];
  uint32_t sld[NPS];
  auto load = [&](unsigned t) __attribute__((always_inline)) {
    if (t >= ntiles) return;
    const Tile tl = decode(t);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int r = ps * RPP + lane / LPR, c = (lane % LPR) * 32;
      const int64_t rowi = tl.r0 + r;
      const bool live = rowi < p.m;
      vld[ps] = live ? *(const v4i*)(p.xq + rowi * (p.n >> 1) + ((tl.c0 + c) >> 1)) : v4i{0, 0, 0, 0};
      sld[ps] = live ? p.xs[rowi * (p.n >> 5) + ((tl.c0 + c) >> 5)] : 127u;
    }
  };
  char* ts = ts_all[wave];
  load(blockIdx.x);
  for (unsigned tcur = blockIdx.x; tcur < ntiles; tcur += gridDim.x) {
  const Tile tl = decode(tcur);
  const int c0 = tl.c0;
  bool nan_in = false;
  v4i vcur[NPS];
  uint32_t scur[NPS];
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) { vcur[ps] = vld[ps]; scur[ps] = sld[ps]; }
  load(tcur + gridDim.x);   // the next tile's rows: in flight until the next trip
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    const int r = ps * RPP + lane / LPR, c = (lane % LPR) * 32;
    const v4i v = vcur[ps];
    const uint32_t se = scur[ps];
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
  asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
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
  asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
  const int m0 = tl.m0;
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
  asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");   // the staged tile has been read (the stores took their data): the area is the next tile's bf16 staging
  }
}


}  // namespace qamd
