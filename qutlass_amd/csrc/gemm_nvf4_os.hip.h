// Small-batch NVFP4 GEMM for gfx950 on wave-owned K stages: the NVFP4 twin of gemm_mx_os.hip.h.  Replaces matmul_host_nvf4_bf16_tn's M-bucketed small tiles
// (qutlass/csrc/gemm.cu:250-326) for batches of at most a few dozen rows, where gemm_nvf4_skinny_kernel (gemm_nvf4.hip.h) was the plan: N = K = 4096, M = 1 ... 64
// took 5.6-7.3 us there against 3.2-3.7 us for the same bytes in MXFP4.
//
// One 32x32 (or 32x16) output tile per workgroup of EIGHT waves, two per SIMD -- the e2m1 -> f16 dequantisation is ~20 vector instructions per MFMA, and with two
// waves on a SIMD one wave's converts run beside the other's MFMAs.  Wave w owns K stages w, w + 8, w + 16, ... (256 elements = 128 B per row each): it issues the
// LDS-DMA pieces of its stages itself and is their only reader, so the K walk has no workgroup barrier; s_waitcnt vmcnt counts the wave's own pieces down.
// K <= 4096: all 16 stages have an LDS area of their own and are requested before the first MFMA; longer K (RING): each wave refills the two slots it owns as it
// consumes them.  Data path per stage: the pieces of gemm_mx_os (8 rows x 128 B, 16-byte chunk XOR-swizzled by row at the source; rows past M / N and chunks past K
// fall off the buffer descriptor), and the e4m3 scales as TWO dword pieces per operand: lane (row i32, half g) fetches the dwords of column tiles 4 kt + 2 g and
// 4 kt + 2 g + 1 of the to_blocked image -- the eight scale groups of the 64 bytes its fragments cover.
// Arithmetic = gemm_nvf4_skinny_kernel's: cvt(e2m1) x e4m3 scale in f16 (both exact), v_mfma_f32_32x32x16_f16, fp32 sums; each wave sums its stages in K order, the
// eight partial sums are added in wave order in fp32 -- bit-identical to the other kernels wherever partial sums are exact (the reference's test regime).
#pragma once
#include "gemm_nvf4.hip.h"

namespace qamd {

// MT = 2 ([r6, third session]; M = 65 ... 128 and beyond): two m-tiles of 32 rows per workgroup -- a B dword is dequantised once for both (12 instead of 16 vector
// instructions per MFMA) and a batch of 128 rows against N = 4096 is ONE round of 256 workgroups instead of two.  A 64x32 tile's stage is 13.5 KiB: one slot per wave; beyond
// K = 2048 the wave refills it as soon as its fragment reads have returned -- the refill's round trip runs behind the stage's ~300 dequantisation instructions.
template <int SPW_, int TN_ = 32, int MT_ = 1>   // stages (one shot) / slots (RING) per wave
struct NvOsCfg {
  static constexpr int NW = 8, MT = MT_, TM = 32 * MT_, TN = TN_, ROWB = 128, SPW = SPW_, NSLOT = NW * SPW_;
  static constexpr int OFF_B = TM * ROWB, OFF_S = (TM + 32) * ROWB, OFF_SB = OFF_S + MT * 512, STAGE = OFF_SB + 512;   // two 256-byte scale pieces (jj = 0, 1) per m-tile and for B
  static constexpr int NPA = TM / 8, NPB = TN / 8;
  static constexpr int LPS = NPA + NPB + 2 * MT + 2;
  static constexpr int RED = NW * TM * 128;
  static constexpr int LDS_BYTES = NSLOT * STAGE > RED ? NSLOT * STAGE : RED;
  static_assert(TN == 32 || TN == 16, "tile width");
  static_assert(MT >= 1 && MT <= 3, "m-tiles");   // (MT = 3: 96 rows -- 18 KiB per stage, eight slots = 144 KiB)
  static_assert(SPW >= 1 && SPW * LPS <= 63, "vmcnt immediate");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class C, bool RING = false>
__global__ __launch_bounds__(512) void gemm_nvf4_os_kernel(const NvGemmParams p) {
  constexpr int SPW = C::SPW, LPS = C::LPS, NW = C::NW, MT = C::MT;
  __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"(p.alpha));   // all scalar argument loads in one round
  const float alpha = *p.alpha;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), i32 = lane & 31, g = lane >> 5;
  const int nb = p.tiles_m * p.tiles_n;
  const int b2 = xcd_remap((int)blockIdx.x, nb);
  const int m0 = uniform((b2 % p.tiles_m) * C::TM), n0 = uniform((b2 / p.tiles_m) * C::TN);
  const int rowbytes = p.K >> 1, KT = (rowbytes + C::ROWB - 1) / C::ROWB;
  const int G16 = p.K >> 4, CB = (G16 + 3) >> 2;   // scale groups per row, column tiles of 4 groups
  const int tailbytes = rowbytes - (KT - 1) * C::ROWB;

  // ---- LDS-DMA sources (gemm_mx_os.hip.h) ------------------------------------------------------------------------------------------------
  const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A + a_off, p.a_bytes - a_off), rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);
  int vP[2], chP[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    chP[par] = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    vP[par] = (lane >> 3) * rowbytes + (chP[par] << 4);
  }
  const int rstep = 8 * rowbytes;
  const uint32_t sb_off = (uint32_t)(n0 >> 7) * CB * 512;
  const __amdgpu_buffer_rsrc_t rSA = make_rsrc(p.SFA, p.sfa_bytes), rSB = make_rsrc(p.SFB + sb_off, p.sfb_bytes - sb_off);
  const int rowB = (n0 & 127) + i32;   // (TN = 16: n0 is a multiple of 16 only; lanes past the 16 rows fetch some row's dword -- unused)
  int vSA[MT];   // A scale rows are addressed from the operand's start: a 96-row tile's m-tiles may lie in two 128-row scale tiles
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int rabs = m0 + 32 * t;
    vSA[t] = (rabs >> 7) * CB * 512 + 2 * g * 512 + i32 * 16 + ((rabs & 127) >> 5) * 4;
  }
  const int vSB = 2 * g * 512 + (rowB & 31) * 16 + ((rowB & 127) >> 5) * 4;

  auto issue = [&](const int kt, const int slot) __attribute__((always_inline)) {   // stage kt into slot `slot` of this wave (kt >= KT: every piece out of range -> zeros)
    char* st = smem + (wave * SPW + slot) * C::STAGE;
    int tail = (kt == KT - 1) ? tailbytes : C::ROWB;
    int oob = (kt < KT) ? 0 : -1;
    asm volatile("" : "+v"(tail), "+v"(oob));
    const int soff = kt * C::ROWB;
#pragma unroll
    for (int t = 0; t < C::NPA + C::NPB; ++t) {
      const bool isB = t >= C::NPA;
      const int qq = isB ? t - C::NPA : t, par = qq & 1;
      const int o = oob | ((chP[par] << 4) < tail ? 0 : -1);
      const int v = ((vP[par] + qq * rstep) & ~o) | ((int)0x80000000 & o);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rB : rA, (lds_ptr_t)(st + (isB ? C::OFF_B : 0) + qq * 1024), 16, v, soff, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int os = (kt < KT && 4 * kt + 2 * g + jj < CB) ? 0 : -1;   // a column tile past the operand's last one would read the next row tile's bytes
#pragma unroll
      for (int t = 0; t < MT; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rSA, (lds_ptr_t)(st + C::OFF_S + t * 512 + jj * 256), 4, ((vSA[t] + jj * 512) & ~os) | ((int)0x80000000 & os), kt * 2048, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rSB, (lds_ptr_t)(st + C::OFF_SB + jj * 256), 4, ((vSB + jj * 512) & ~os) | ((int)0x80000000 & os), kt * 2048, 0, 0);
    }
  };

#pragma unroll
  for (int j = 0; j < SPW; ++j) issue(wave + NW * j, j);

  const int sw = (i32 >> 1) & 7;
  v16f acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

  // one stage (kt) out of slot u; RING: kt_next (>= KT: zeros) is requested into the slot as soon as the reads have returned
  auto consume = [&](const int kt, const int u, const int kt_next) __attribute__((always_inline)) {
    const char* st = smem + (wave * SPW + u) * C::STAGE;
    v4i ca[MT][4], cb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int off = i32 * C::ROWB + (((4 * g + j) ^ sw) << 4);
#pragma unroll
      for (int t = 0; t < MT; ++t) ca[t][j] = *(const v4i*)(st + t * 32 * C::ROWB + off);
      cb[j] = *(const v4i*)(st + C::OFF_B + off);
    }
    uint32_t da[MT][2], db[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
      for (int t = 0; t < MT; ++t) da[t][jj] = *(const uint32_t*)(st + C::OFF_S + t * 512 + jj * 256 + lane * 4);
      db[jj] = *(const uint32_t*)(st + C::OFF_SB + jj * 256 + lane * 4);
    }
    fence();
    if constexpr (RING) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot is free ...
#pragma unroll
      for (int t = 0; t < MT; ++t) asm volatile("" : "+v"(ca[t][0]), "+v"(ca[t][1]), "+v"(ca[t][2]), "+v"(ca[t][3]), "+v"(da[t][0]), "+v"(da[t][1]));   // ... (every read pinned behind the wait)
      asm volatile("" : "+v"(cb[0]), "+v"(cb[1]), "+v"(cb[2]), "+v"(cb[3]), "+v"(db[0]), "+v"(db[1]));
      issue(kt_next, u);
      fence();
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      // groups past K inside the last column tile (K % 64 == 32): their scale bytes are layout padding -- masked to 0 (0 x 0, never NaN)
      const int valid = G16 - 4 * (4 * kt + 2 * g + jj);
      const uint32_t smask = valid >= 4 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
      h2_t sa[MT][2], sb[2];
#pragma unroll
      for (int t = 0; t < MT; ++t) e4m3x4_to_f16(da[t][jj] & smask, sa[t][0], sa[t][1]);
      e4m3x4_to_f16(db[jj] & smask, sb[0], sb[1]);
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) {
        const int j = 2 * jj + jl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const _Float16 xb = sb[jl][q >> 1];
          const h8_t fb = dq8((uint32_t)cb[j][q], h2_t{xb, xb});
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const _Float16 xa = sa[t][jl][q >> 1];
            const h8_t fa = dq8((uint32_t)ca[t][j][q], h2_t{xa, xa});
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc[t], 0, 0, 0);
          }
        }
      }
    }
    fence();
  };
  if constexpr (!RING) {
    static_for<0, SPW>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SPW - 1 - j) * LPS) : "memory");
      fence();
      consume(wave + NW * j, j, 0);
    });
  } else {
    for (int kt = wave; kt < KT; kt += NW * SPW) {
      static_for<0, SPW>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        if (u == 0 || kt + NW * u < KT) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SPW - 1) * LPS) : "memory");
          fence();
          consume(kt + NW * u, u, kt + NW * u + NW * SPW);
        }
      });
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every wave's reads of the stage areas are done: they become the sum's scratch
  fence();

  // ---- cross-wave sum: [wave][row][8 chunks of 4 fp32], chunk ^ (row & 7) -------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(v4f*)(smem + (wave * C::TM + 32 * t + i32) * 128 + (((2 * q + g) ^ (i32 & 7)) << 4)) = v4f{acc[t][4 * q + 0], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
  __syncthreads();
  for (int it = tid; it < C::TM * 8; it += 512) {
    const int rr = it >> 3, cq = it & 7;   // row of the TM x 32 tile, chunk of 4 columns
    v4f t = *(const v4f*)(smem + rr * 128 + ((cq ^ (rr & 7)) << 4));
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const v4f s = *(const v4f*)(smem + (w * C::TM + rr) * 128 + ((cq ^ (rr & 7)) << 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += s[e];
    }
    const int row = m0 + rr, col = n0 + 4 * cq;
    if (row < p.M && col < p.N && 4 * cq < C::TN) {
      v2i o;
      o[0] = (int)pack_bf16x2(t[0] * alpha, t[1] * alpha);
      o[1] = (int)pack_bf16x2(t[2] * alpha, t[3] * alpha);
      *(v2i*)(p.D + (size_t)row * p.ldd + col) = o;
    }
  }
}

// ---- [r6, third session] decode form: a 16x16 output tile per workgroup on v_mfma_f32_16x16x32_f16 -------------------------------------------------------------
// The 32x32 kernel above is bound by its dequantisation instructions (16 per MFMA: 4 converts + 4 packed multiplies per operand dword), not by bytes, and at M <= 16 half
// of its A fragment is rows that do not exist.  Here a workgroup owns 16 output columns and 16 rows: a 16x16x32 MFMA takes ONE dword (8 e2m1) per lane and operand, so the
// instructions per weight element are those of the 32-column kernel while N = 4096 spreads over 256 workgroups instead of 128 -- half the vector work per CU.
// Lane l = (row r = l & 15, quarter kq = l >> 4) reads the two 16-byte chunks kq and kq + 4 of its row (64 elements of the stage's 256); MFMA (h, d) takes dword d of
// chunk kq + 4 h from every lane -- a permutation of K inside the stage that A and B share.  Scales: ONE dword piece per operand and stage (lane l fetches row r's dword of
// column tile 4 kt + kq); the two scale groups of chunk kq + 4 h are bytes 2 (kq & 1), + 1 of column tile 4 kt + 2 h + (kq >> 1).
// MT = 2 (M = 17 ... 32): two m-tiles of 16 rows per workgroup -- a B dword is dequantised once and feeds both m-tiles' MFMAs (12 instead of 16 vector instructions
// per MFMA), still 16 columns per workgroup.
// TN > 16 (MT = 1; M <= 16 against a weight too wide for 16-column workgroups): ceil(TN / 16) n-tiles per workgroup -- the A dword is dequantised once for all of them
// (TN = 32: 12 vector instructions per MFMA, 48: 10.7, 56: 10) and the weight still runs one workgroup per CU (N = 8192: TN = 32; 12288: 48; 14336: 56 = 256 workgroups
// where the 32x32 kernel ran two rounds).  TN = 56: the fourth n-tile's columns 8 ... 15 read whatever the LDS holds behind the B area -- they only feed columns that are not
// stored.  NPA = 1 (M <= 8): only A rows 0 ... 7 are fetched (rows 8 ... 15 of the fragment read the first B rows: finite, and they only feed rows that are not stored) --
// that KiB per stage is what lets a 56-column tile's 16 stages (K = 4096) fit the LDS.
template <int SPW_, int MT_ = 1, int TN_ = 16, int NPA_ = 2 * MT_>
struct NvOs16Cfg {
  static constexpr int NW = 8, MT = MT_, TM = 16 * MT_, TN = TN_, ROWB = 128, SPW = SPW_, NSLOT = NW * SPW_;
  static constexpr int NT = (TN + 15) / 16, NPA = NPA_, NPB = TN / 8;
  static constexpr int OFF_B = NPA * 1024, OFF_S = OFF_B + TN * ROWB, OFF_SB = OFF_S + MT * 256, STAGE = OFF_SB + NT * 256;   // one 256-byte scale piece per m-tile and per n-tile
  static constexpr int LPS = NPA + NPB + MT + NT;
  static constexpr int RED = NW * TM * NT * 64;
  static constexpr int LDS_BYTES = NSLOT * STAGE > RED ? NSLOT * STAGE : RED;
  static_assert(MT == 1 || (MT == 2 && TN == 16), "two m-tiles: 16 columns only");
  static_assert(TN % 8 == 0 && TN >= 16 && TN <= 64, "tile width");
  static_assert(NPA == 2 * MT || (NPA == 1 && MT == 1), "A pieces");
  static_assert(SPW >= 1 && SPW * LPS <= 63, "vmcnt immediate");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class C, bool RING = false>
__global__ __launch_bounds__(512) void gemm_nvf4_os16_kernel(const NvGemmParams p) {
  constexpr int SPW = C::SPW, LPS = C::LPS, NW = C::NW, MT = C::MT, NT = C::NT;
  __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"(p.alpha));   // all scalar argument loads in one round
  const float alpha = *p.alpha;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), r16 = lane & 15, kq = lane >> 4;
  const int nb = p.tiles_m * p.tiles_n;
  const int b2 = xcd_remap((int)blockIdx.x, nb);
  const int m0 = uniform((b2 % p.tiles_m) * C::TM), n0 = uniform((b2 / p.tiles_m) * C::TN);
  const int rowbytes = p.K >> 1, KT = (rowbytes + C::ROWB - 1) / C::ROWB;
  const int G16 = p.K >> 4, CB = (G16 + 3) >> 2;   // scale groups per row, column tiles of 4 groups
  const int tailbytes = rowbytes - (KT - 1) * C::ROWB;

  // ---- LDS-DMA sources: pieces of 8 rows x 128 B, chunk ^ ((row >> 1) & 7) at the source (gemm_mx_os.hip.h) ----------------------------------------------
  const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A + a_off, p.a_bytes - a_off), rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);
  int vP[2], chP[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    chP[par] = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    vP[par] = (lane >> 3) * rowbytes + (chP[par] << 4);
  }
  const int rstep = 8 * rowbytes;
  // scale dwords: lane l fetches row l & 15 of column tile 4 kt + (l >> 4); B rows are addressed from the operand's start (a column tile of 56 may straddle two 128-row scale tiles)
  const uint32_t sa_off = (uint32_t)(m0 >> 7) * CB * 512;
  const __amdgpu_buffer_rsrc_t rSA = make_rsrc(p.SFA + sa_off, p.sfa_bytes - sa_off), rSB = make_rsrc(p.SFB, p.sfb_bytes);
  int vSA[MT], vSB[NT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int rowA = (m0 & 127) + 16 * t + r16;
    vSA[t] = kq * 512 + (rowA & 31) * 16 + (rowA >> 5) * 4;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int nr = n0 + 16 * t + r16;   // (an n-tile past TN: rows of the next workgroup's tile or past N -- unused / zeros)
    vSB[t] = (nr >> 7) * CB * 512 + kq * 512 + (nr & 31) * 16 + ((nr & 127) >> 5) * 4;
  }

  auto issue = [&](const int kt, const int slot) __attribute__((always_inline)) {   // stage kt into slot `slot` of this wave (kt >= KT: every piece out of range -> zeros)
    char* st = smem + (wave * SPW + slot) * C::STAGE;
    int tail = (kt == KT - 1) ? tailbytes : C::ROWB;
    int oob = (kt < KT) ? 0 : -1;
    asm volatile("" : "+v"(tail), "+v"(oob));
    const int soff = kt * C::ROWB;
#pragma unroll
    for (int t = 0; t < C::NPA + C::NPB; ++t) {
      const bool isB = t >= C::NPA;
      const int qq = isB ? t - C::NPA : t, par = qq & 1;
      const int o = oob | ((chP[par] << 4) < tail ? 0 : -1);
      const int v = ((vP[par] + qq * rstep) & ~o) | ((int)0x80000000 & o);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rB : rA, (lds_ptr_t)(st + (isB ? C::OFF_B : 0) + qq * 1024), 16, v, soff, 0, 0);
    }
    const int os = (kt < KT && 4 * kt + kq < CB) ? 0 : -1;   // a column tile past the operand's last one would read the next row tile's bytes
#pragma unroll
    for (int t = 0; t < MT; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rSA, (lds_ptr_t)(st + C::OFF_S + t * 256), 4, (vSA[t] & ~os) | ((int)0x80000000 & os), kt * 2048, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rSB, (lds_ptr_t)(st + C::OFF_SB + t * 256), 4, (vSB[t] & ~os) | ((int)0x80000000 & os), kt * 2048, 0, 0);
  };

#pragma unroll
  for (int j = 0; j < SPW; ++j) issue(wave + NW * j, j);

  const int sw = (r16 >> 1) & 7;
  v4f acc[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = v4f{0.f, 0.f, 0.f, 0.f};
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

  auto consume = [&](const int kt, const int u, const int kt_next) __attribute__((always_inline)) {
    const char* st = smem + (wave * SPW + u) * C::STAGE;
    v4i ca[MT][2], cb[NT][2];
    uint32_t da[MT][2], db[NT][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int off = r16 * C::ROWB + (((kq + 4 * h) ^ sw) << 4);
      const int soff = ((2 * h + (kq >> 1)) * 16 + r16) * 4;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        ca[t][h] = *(const v4i*)(st + t * 16 * C::ROWB + off);
        da[t][h] = *(const uint32_t*)(st + C::OFF_S + t * 256 + soff);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        cb[n][h] = *(const v4i*)(st + C::OFF_B + n * 16 * C::ROWB + off);
        db[n][h] = *(const uint32_t*)(st + C::OFF_SB + n * 256 + soff);
      }
    }
    fence();
    if constexpr (RING) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot is free ...
#pragma unroll
      for (int t = 0; t < MT; ++t) asm volatile("" : "+v"(ca[t][0]), "+v"(ca[t][1]), "+v"(da[t][0]), "+v"(da[t][1]));   // ... (every read pinned behind the wait)
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(cb[n][0]), "+v"(cb[n][1]), "+v"(db[n][0]), "+v"(db[n][1]));
      issue(kt_next, u);
      fence();
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // groups past K inside the last column tile (K % 64 == 32): their scale bytes are layout padding -- masked to 0 (0 x 0, never NaN)
      const int valid = G16 - 4 * (4 * kt + 2 * h + (kq >> 1));
      const uint32_t smask = valid >= 4 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
      h2_t sa[MT], sb[NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) sa[t] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((da[t][h] & smask) >> (16 * (kq & 1)), 1.0f, false);   // the two groups of chunk kq + 4 h
#pragma unroll
      for (int n = 0; n < NT; ++n) sb[n] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((db[n][h] & smask) >> (16 * (kq & 1)), 1.0f, false);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        h8_t fa[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const _Float16 xa = sa[t][d >> 1];
          fa[t] = dq8((uint32_t)ca[t][h][d], h2_t{xa, xa});
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const _Float16 xb = sb[n][d >> 1];
          const h8_t fb = dq8((uint32_t)cb[n][h][d], h2_t{xb, xb});
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb, fa[t], acc[t][n], 0, 0, 0);
        }
      }
    }
    fence();
  };
  if constexpr (!RING) {
    static_for<0, SPW>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SPW - 1 - j) * LPS) : "memory");
      fence();
      consume(wave + NW * j, j, 0);
    });
  } else {
    for (int kt = wave; kt < KT; kt += NW * SPW) {
      static_for<0, SPW>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        if (u == 0 || kt + NW * u < KT) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SPW - 1) * LPS) : "memory");
          fence();
          consume(kt + NW * u, u, kt + NW * u + NW * SPW);
        }
      });
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every wave's reads of the stage areas are done: they become the sum's scratch
  fence();

  // ---- cross-wave sum: [wave][row m][16 NT columns] fp32; a lane holds row m = 16 t + r16, columns 16 n + 4 kq .. + 3 (srcA = the B fragment) --------------
  constexpr int RROW = 64 * NT;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) *(v4f*)(smem + (wave * C::TM + 16 * t + r16) * RROW + n * 64 + kq * 16) = acc[t][n];
  __syncthreads();
  if (tid < 64 * MT * NT) {
    const int rr = (tid >> 2) % C::TM, n = (tid >> 2) / C::TM, cq = tid & 3;   // row of the tile, n-tile, chunk of 4 columns
    v4f t = *(const v4f*)(smem + rr * RROW + n * 64 + cq * 16);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const v4f s = *(const v4f*)(smem + (w * C::TM + rr) * RROW + n * 64 + cq * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += s[e];
    }
    const int row = m0 + rr, cl = 16 * n + 4 * cq, col = n0 + cl;
    if (row < p.M && col < p.N && cl < C::TN) {
      v2i o;
      o[0] = (int)pack_bf16x2(t[0] * alpha, t[1] * alpha);
      o[1] = (int)pack_bf16x2(t[2] * alpha, t[3] * alpha);
      *(v2i*)(p.D + (size_t)row * p.ldd + col) = o;
    }
  }
}

#if QAMD_TU == 0 || QAMD_TU == 4
// (not inline: the NVFP4 unit of capi.hip emits it; declared in gemm_nvf4.hip.h for launch_nvf4_gemm)
// [r6] the wave-owned small-batch kernel (gemm_nvf4_os.hip.h): K <= 2048 one stage per wave, K <= 4096 two, longer K two refilled slots per wave
hipError_t launch_nvf4_os(NvGemmParams p, hipStream_t s, int tn) {
  if (tn == 1616 || tn == 3216) {   // [r6] the decode form: 16x16 (3216: 32x16) tiles on the 16x16x32 MFMA -- one shot while the tile's K extent fits the LDS (K <= 8192 / 4096), refilled slots beyond
    const bool two = tn == 3216;
    p.tiles_m = two ? (p.M + 31) / 32 : (p.M + 15) / 16;
    p.tiles_n = (p.N + 15) / 16;
    const int KT = (p.K / 2 + 127) / 128;
    const dim3 grid(p.tiles_m * p.tiles_n), block(512);
    if (two) {
      if (KT <= 8) hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<1, 2>>), grid, block, 0, s, p);
      else if (KT <= 16) hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<2, 2>>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<2, 2>, true>), grid, block, 0, s, p);
      return hipSuccess;
    }
    if (KT <= 8) hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<1>>), grid, block, 0, s, p);
    else if (KT <= 16) hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<2>>), grid, block, 0, s, p);
    else if (KT <= 24) hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<3>>), grid, block, 0, s, p);
    else if (KT <= 32) hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<4>>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<4>, true>), grid, block, 0, s, p);
    return hipSuccess;
  }
  if (tn == 1632 || tn == 1648 || tn == 1656 || tn == 856) {   // [r6] 16 rows x 32 / 48 / 56 columns per workgroup (856: only A rows 0 ... 7 fetched, M <= 8); K <= 4096 one shot, two refilled slots per wave beyond
    const int cols = tn == 1632 ? 32 : tn == 1648 ? 48 : 56;
    p.tiles_m = (p.M + 15) / 16;
    p.tiles_n = (p.N + cols - 1) / cols;
    const int KT = (p.K / 2 + 127) / 128;
    const dim3 grid(p.tiles_m * p.tiles_n), block(512);
#define QAMD_NVW(TN_, NPA_)                                                                                                         \
    do {                                                                                                                            \
      if (KT <= 8) hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<1, 1, TN_, NPA_>>), grid, block, 0, s, p);                    \
      else if (KT <= 16) hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<2, 1, TN_, NPA_>>), grid, block, 0, s, p);              \
      else hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<2, 1, TN_, NPA_>, true>), grid, block, 0, s, p);                      \
    } while (0)
    if (tn == 1632) QAMD_NVW(32, 2);
    else if (tn == 1648) QAMD_NVW(48, 2);
    else if (tn == 856) QAMD_NVW(56, 1);
    else {   // 56 columns with all 16 A rows: 16 stages do not fit the LDS (164 KiB) -- one shot up to 8 stages, refilled slots beyond
      if (KT <= 8) hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<1, 1, 56, 2>>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((gemm_nvf4_os16_kernel<NvOs16Cfg<1, 1, 56, 2>, true>), grid, block, 0, s, p);
    }
#undef QAMD_NVW
    return hipSuccess;
  }
  if (tn == 6432 || tn == 9632) {   // [r6] two / three m-tiles per workgroup (64x32 / 96x32 tiles): one slot per wave -- one shot up to K = 2048, refilled beyond
    const int tm = tn == 9632 ? 96 : 64;
    p.tiles_m = (p.M + tm - 1) / tm;
    p.tiles_n = (p.N + 31) / 32;
    const int KT = (p.K / 2 + 127) / 128;
    const dim3 grid(p.tiles_m * p.tiles_n), block(512);
    if (tm == 96) {
      if (KT <= 8) hipLaunchKernelGGL((gemm_nvf4_os_kernel<NvOsCfg<1, 32, 3>, false>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((gemm_nvf4_os_kernel<NvOsCfg<1, 32, 3>, true>), grid, block, 0, s, p);
      return hipSuccess;
    }
    if (KT <= 8) hipLaunchKernelGGL((gemm_nvf4_os_kernel<NvOsCfg<1, 32, 2>, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_nvf4_os_kernel<NvOsCfg<1, 32, 2>, true>), grid, block, 0, s, p);
    return hipSuccess;
  }
  p.tiles_m = (p.M + 31) / 32;
  p.tiles_n = (p.N + tn - 1) / tn;
  const int KT = (p.K / 2 + 127) / 128;
  const dim3 grid(p.tiles_m * p.tiles_n), block(512);
#define QAMD_NVOS(SPW_, RING_)                                                                                                      \
  do {                                                                                                                              \
    if (tn == 16) hipLaunchKernelGGL((gemm_nvf4_os_kernel<NvOsCfg<SPW_, 16>, RING_>), grid, block, 0, s, p);                           \
    else hipLaunchKernelGGL((gemm_nvf4_os_kernel<NvOsCfg<SPW_, 32>, RING_>), grid, block, 0, s, p);                                     \
  } while (0)
  if (KT <= 8) QAMD_NVOS(1, false);
  else if (KT <= 16) QAMD_NVOS(2, false);
  else QAMD_NVOS(2, true);
#undef QAMD_NVOS
  return hipSuccess;
}
#endif

}  // namespace qamd
