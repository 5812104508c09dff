// Small-batch MXFP4 GEMM for gfx950 with the K split INSIDE the workgroup: a 32x32 (or 32x64 / 64x32) output tile per workgroup of four waves, wave w taking
// k-slice w of every 256-element K stage, one cross-wave sum through LDS at the end.  Replaces the M-bucketed small tiles of qutlass/csrc/gemm.cu:195-222
// (MmaTile 128x128x256 for M <= 16 ...) and the 16x16x256 tile of the reference's small-batch kernel (qutlass/csrc/gemm_ada.cu:127-129) for 8 < M <= 64 where the
// weight matrix is what moves: a batch of 64 rows against N = K = 4096 is 8.4 MB of weights and 1 MB of everything else.
//
// Why ([r6]).  The smallest tile of the ring kernels (gemm_mx_ringp) is 64x64 -- four waves of 32x32 -- so M <= 64 against N = 4096 gives 64
// workgroups on a 256-CU part, each pulling 256 KiB through ONE CU's LDS-DMA path (~40 B/clk/CU): 5.1 us per call, of which 1.6 are the launch floor, flat from
// M = 9 to M = 128 (profiles/dip_scan_r5ab.txt).  Splitting K across workgroups instead costs a second launch or a cross-CU hand-off (3-5 us either way: DESIGN.md
// section 7, MI355X_MICROARCH.md price list).  A 32x32 tile is the smallest the 32x32x64 scaled MFMA computes; with the stage's four k-slices dealt to the four
// waves every wave still issues LDS-DMA (four loaders reach the CU's rate, one does not), every wave runs one MFMA per stage, and 256 workgroups stream 128 KiB each.
//
// Data path = the ring kernels': K stages of 128 B per row, LDS-DMA pieces of 8 rows x 128 B with the 16-byte chunk XOR-swizzled by row at the SOURCE, to_blocked
// scale pieces of 512 B (128 rows x 4 K-blocks) fetched whole, rows past M / N and chunks past K off the end of the buffer descriptor (zeros).  An 8-deep LDS ring,
// the fragments of stage kt + 1 read into a second register set while the MFMA of stage kt issues.
// Results: every product is exact and each wave sums its k-slices in K order; the four partial sums are added as ((w0 + w1) + w2) + w3 in fp32 -- bit-identical
// to the other schedules wherever partial sums are exact (the reference's test regime), one fp32 rounding apart otherwise, like every split-K plan here
// ("Determinism").
#pragma once
#include "gemm_mx.hip.h"

namespace qamd {

// D: depth of the LDS ring.  These launches are LATENCY-bound, not bandwidth-bound: a workgroup streams 128 KiB in 16 stages of 8 KiB, and an LDS-DMA piece that misses
// the XCD's L2 takes ~1 500 - 2 000 cycles to land with the whole chip asking -- with 3 stages in flight (D = 4) the launch moved 16 B/clk/CU (4096^2 weights, M <= 64:
// 4.17 us); 7 stages in flight cover that latency at the CU's own rate.
template <int TM_, int TN_, int D_ = 8>
struct KsCfg {
  static constexpr int TM = TM_, TN = TN_, D = D_, ROWB = 128;
  static constexpr int MT = TM / 32, NT = TN / 32;
  static constexpr int NA = TM / 8, NB = TN / 8;              // 1-KiB data pieces per stage
  static constexpr int PA = NA / 4, PB = NB / 4;              // per wave
  static constexpr int LPS = PA + PB + 1;                     // DMA instructions per wave per stage (+ its scale piece)
  static constexpr int OFF_B = TM * ROWB, OFF_S = (TM + TN) * ROWB, STAGE = OFF_S + 4 * 1024;   // scale slots: A column tile 0, 1; B column tile 0, 1
  static constexpr int RED = 4 * MT * NT * 4096;              // cross-wave sum: [wave][tile] 32 x 32 fp32
  static constexpr int LDS_BYTES = D * STAGE > RED ? D * STAGE : RED;
  static_assert(TM % 32 == 0 && TN % 32 == 0 && TM <= 128 && TN <= 128, "tile");
  static_assert((D - 2) * LPS <= 63 && D % 2 == 0 && D >= 4, "vmcnt immediate; the ring is unrolled D times and the register sets alternate");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class C>
__global__ __launch_bounds__(256) void gemm_mx_ks_kernel(const GemmParams p) {
  constexpr int MT = C::MT, NT = C::NT, D = C::D, LPS = C::LPS;
  __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"(p.alpha));   // all scalar argument loads in one round
  const float alpha = *p.alpha;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), i32 = lane & 31, g = lane >> 5;
  // tile: workgroups that share a B column tile are neighbours (and stay on one XCD: xcd_remap)
  const int nb = p.tiles_m * p.tiles_n;
  const int b2 = xcd_remap((int)blockIdx.x, nb);
  const int m0 = uniform((b2 % p.tiles_m) * C::TM), n0 = uniform((b2 / p.tiles_m) * C::TN);
  const int rowbytes = p.K >> 1, KT = (rowbytes + C::ROWB - 1) / C::ROWB, CB = (p.K / 32 + 3) >> 2;
  const int tailbytes = rowbytes - (KT - 1) * C::ROWB;   // bytes per row of the last stage

  // ---- LDS-DMA sources --------------------------------------------------------------------------------------------------------------
  const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A + a_off, p.a_bytes - a_off), rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);
  // piece qq of an operand tile = rows 8 qq .. + 7; lane -> row 8 qq + (l >> 3), physical chunk l & 7 = logical chunk ^ ((row >> 1) & 7): only the parity of qq matters
  int vP[2], chP[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    chP[par] = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    vP[par] = (lane >> 3) * rowbytes + (chP[par] << 4);
  }
  const int rstep = 8 * rowbytes;
  // scale piece of this wave: waves 0, 1 -> A's column tiles 2 kt, 2 kt + 1; waves 2, 3 -> B's (lanes 0-31 carry the 512 bytes, lanes 32-63 load zeros into the slot's pad)
  const bool sIsB = wave >= 2;
  const uint32_t s_off = (uint32_t)((sIsB ? n0 : m0) >> 7) * CB * 512;
  const __amdgpu_buffer_rsrc_t rS = sIsB ? make_rsrc(p.SFB + s_off, p.sfb_bytes - s_off) : make_rsrc(p.SFA + s_off, p.sfa_bytes - s_off);
  const int vS = g == 0 ? i32 * 16 : (int)0x80000000;
  auto issue = [&](int kt, const int slot) __attribute__((always_inline)) {
    char* st = smem + slot * C::STAGE;
    int tail = (kt == KT - 1) ? tailbytes : C::ROWB;
    int oob = (kt < KT) ? 0 : -1;
    asm volatile("" : "+v"(tail), "+v"(oob));
    const int soff = kt * C::ROWB;
#pragma unroll
    for (int t = 0; t < C::PA + C::PB; ++t) {
      const bool isB = t >= C::PA;
      const int qq = wave + 4 * (isB ? t - C::PA : t);
      const int par = qq & 1;   // (= wave & 1)
      const int o = oob | ((chP[par] << 4) < tail ? 0 : -1);
      const int v = ((vP[par] + qq * rstep) & ~o) | ((int)0x80000000 & o);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rB : rA, (lds_ptr_t)(st + (isB ? C::OFF_B : 0) + qq * 1024), 16, v, soff, 0, 0);
    }
    const int ct = 2 * kt + (wave & 1);
    const int os = (kt < KT && ct < CB) ? 0 : -1;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, (lds_ptr_t)(st + C::OFF_S + wave * 1024), 16, (vS & ~os) | ((int)0x80000000 & os), ct * 512, 0, 0);
  };

  // ---- fragment / scale reads of this wave's k-slice ------------------------------------------------------------------------------------
  // row r = 32 t + i32 of the tile, logical chunk 4 g + wave (lane half g owns K-blocks 4 g .. 4 g + 3 of the stage), physical chunk ^ ((r >> 1) & 7)
  const int rdF = i32 * C::ROWB + (((4 * g + wave) ^ ((i32 >> 1) & 7)) << 4);
  // scale dword of row r: slot (column tile g) + (r & 31) * 16 + ((r & 127) >> 5) * 4; this wave's K-block is byte `wave` of it
  const int rdSA = C::OFF_S + g * 1024 + i32 * 16 + ((m0 & 127) >> 5) * 4;
  const int rdSB = C::OFF_S + (2 + g) * 1024 + i32 * 16 + ((n0 & 127) >> 5) * 4;
  const int sshift = 8 * wave;
  v4i fa[2][MT], fb[2][NT];
  int sa[2][MT], sb[2][NT];
  auto read_stage = [&](const int slot, const int set) __attribute__((always_inline)) {
    const char* st = smem + slot * C::STAGE;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      fa[set][t] = *(const v4i*)(st + rdF + t * 32 * C::ROWB);
      sa[set][t] = *(const int*)(st + rdSA + 4 * t);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      fb[set][t] = *(const v4i*)(st + C::OFF_B + rdF + t * 32 * C::ROWB);
      sb[set][t] = *(const int*)(st + rdSB + 4 * t);
    }
  };
  v16f acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

  // one stage: stage kt + 1 landed (own pieces) + barrier -> DMA of stage kt + D - 1 into the slot of stage kt - 1 (free: everyone's reads of it returned before this
  // barrier), reads of stage kt + 1 into the other register set, then the MFMAs of stage kt
  auto stage = [&](int kt, auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value, set = u & 1;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 3) * LPS) : "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    issue(kt + D - 1, (u + D - 1) % D);
    read_stage((u + 1) % D, set ^ 1);
    fence();
    __builtin_amdgcn_s_waitcnt(0xc07f | ((2 * (MT + NT)) << 8));   // lgkmcnt(2 (MT + NT)): the reads just issued may stay in flight, those of stage kt have returned
    fence();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const v4i a = fa[set][m], b = fb[set][n];
        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0}, acc[m][n], 4, 4,
                                                                    0, (int)((unsigned)sb[set][n] >> sshift), 0, (int)((unsigned)sa[set][m] >> sshift));
      }
    fence();
  };

#pragma unroll
  for (int s = 0; s < D - 1; ++s) issue(s, s);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * LPS) : "memory");   // stage 0 landed
  __builtin_amdgcn_s_barrier();
  fence();
  read_stage(0, 0);
  fence();
  for (int kt = 0; kt < KT; kt += D) {
    static_for<0, D>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if (u == 0 || kt + u < KT) stage(kt + u, uc);
    });
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // trailing (out-of-range) loads and the look-ahead reads: the ring becomes the sum's scratch
  __builtin_amdgcn_s_barrier();
  fence();

  // ---- cross-wave sum: [wave][tile][row][8 chunks of 4 fp32], chunk ^ (row & 7) (the 8 lanes of a ds_write_b128 group hit 8 chunks) ----------------
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(v4f*)(smem + ((wave * MT * NT + m * NT + n) * 32 + i32) * 128 + (((2 * q + g) ^ (i32 & 7)) << 4)) =
            v4f{acc[m][n][4 * q + 0], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
  __syncthreads();
  const int rr = tid >> 3, cq = tid & 7;   // row of the 32 x 32 tile, chunk of 4 columns
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      v4f s[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) s[w] = *(const v4f*)(smem + ((w * MT * NT + m * NT + n) * 32 + rr) * 128 + ((cq ^ (rr & 7)) << 4));
      v4f t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = ((s[0][e] + s[1][e]) + s[2][e]) + s[3][e];
      const int row = m0 + 32 * m + rr, col = n0 + 32 * n + 4 * cq;
      if (row < p.M && col < p.N) {
        v2i o;
        o[0] = (int)pack_bf16x2(t[0] * alpha, t[1] * alpha);
        o[1] = (int)pack_bf16x2(t[2] * alpha, t[3] * alpha);
        *(v2i*)(p.D + (size_t)row * p.ldd + col) = o;
      }
    }
}

}  // namespace qamd
