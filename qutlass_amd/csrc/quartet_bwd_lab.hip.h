// LAB ONLY (-DQAMD_BENCH=1): QAT-backward kernels that were built, are bit-identical to the product kernels and are NOT faster.  The product translation units do not read
// this file (tests/test_cabi_and_host.py: test_product_build_does_not_see_the_lab_sources).
#pragma once
#include "quartet_bwd.hip.h"

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


}  // namespace qamd
