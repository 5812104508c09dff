// Small-batch MXFP4 GEMM for gfx950, "one shot": a 32x32 output tile per workgroup of four waves whose WHOLE K extent is requested from memory before the first MFMA.
// Replaces the M-bucketed small tiles of qutlass/csrc/gemm.cu:195-222 and the small-batch kernel of qutlass/csrc/gemm_ada.cu:127-129 for M <= 64 against weights of
// K <= 4096 (the decode shapes of the reference's own sweep, README.md:121-123), where gemm_mx_ks.hip.h was the plan until now.
//
// Why ([r6]).  gemm_mx_ks walks the K stages through a 4-deep ring with one barrier of the four waves per stage: three stages of 8 KiB in flight per
// workgroup and ~250 cycles per stage for ONE MFMA per wave -- at N = K = 4096 the 16 stages stream the 8.4 MB of weights at 4.8 TB/s (4.07 us per call, of which
// ~2.3 are launch + first round trip + cross-wave sum + store).  A 32x32 tile with K <= 4096 is (32 + 32) rows x 2 KiB + 8 KiB of scale dwords = 136 KiB: it fits
// the LDS whole.  So there is no ring: wave w owns K stages w, w + 4, w + 8, ... -- issues every LDS-DMA piece of them up front (<= 40 per wave, all in flight), and
// takes each stage as it lands (s_waitcnt vmcnt counts its own pieces down; nobody else touches them: no barrier until the cross-wave sum).
//
// Data path per stage = gemm_mx_ks's (128 B per row, pieces of 8 rows x 128 B with the 16-byte chunk XOR-swizzled by row at the SOURCE; rows past M / N and chunks
// past K fall off the buffer descriptor and read zeros), except the scales: ONE dword piece per operand and stage -- lane l fetches the dword of row l % 32 in column
// tile 2 kt + l / 32 of the to_blocked image (bytes = K-blocks 4 (l / 32) .. + 3 of the stage), which is exactly the dword lane l's MFMAs take their scale byte from.
// RING ([r6], K > 4096): the same wave-owned stages through wave-owned SLOTS -- SPW per wave; as soon as a wave's reads of stage j have returned it requests its stage
// j + SPW into the slot just freed (the only consumer of a slot is the wave that filled it: still no barrier), so SPW - 1 of its stages are always in flight.
// Results: each wave sums its stages' four k-slices in K order into one accumulator; the four partial sums are added as ((w0 + w1) + w2) + w3 in fp32 -- bit-identical
// to the other schedules wherever partial sums are exact (the reference's test regime), one fp32 rounding apart otherwise, like every split-K plan here.
#pragma once
#include "gemm_mx.hip.h"
#include "quantize.hip.h"

namespace qamd {

// TN = 16 ([r6]): the workgroup owns 16 output columns -- only B rows 0 .. 15 of the tile are fetched (rows 16 .. 31 of the MFMA's B fragment read whatever the LDS
// holds: they only feed output columns 16 .. 31, which are not stored), so a weight of N columns spreads over N / 16 workgroups: N = 4096 fills 256 CUs instead of 128
// and every CU pulls half the bytes through its LDS-DMA path.
// [r6, third session] Two more axes of the same kernel:
//   EBITS = 8 -- MXFP8 (qutlass/csrc/gemm.cu:328-386 on small batches): a stage is 128 fp8 elements = ONE column tile of the to_blocked image, two MFMAs of 64;
//     the hardware's 8-bit fragment is "split" (registers 0-3 of lane half g = bytes 16 g .. + 15 of K-block 2 j, registers 4-7 the same of K-block 2 j + 1, the
//     scale byte of K-block 2 j + g taken from lane half g), so a lane reads chunks 4 j + 2 u + g and shifts its scale dword right by 8 g (op_sel 2 j).  AFMT = 1: A is
//     e5m2 (the extension entry qutlass_amd_matmul_mxf8_bf16_tn_fmt).  One scale piece per operand and stage (both lane halves fetch the row's dword; TM = 64: half g
//     fetches m-tile g's).
//   TM = 64 -- two m-tiles per workgroup, for batches whose 32x32 tiles no longer fit one per CU (M = 96 ... 128 against N = 4096: 256 tiles of 64x32).  Still wave-owned
//     K stages: the owner of a stage fetches the tile's 64 A rows and 32 B rows ONCE and runs both m-tiles' MFMAs on them -- no byte is fetched twice.
template <int SPW_, int TN_ = 32, int EBITS_ = 4, int TM_ = 32, int AFMT_ = 0>   // SPW: K stages per wave -- the kernel covers 4 SPW stages of 128 bytes per row (RING: any K)
struct OsCfg {
  static constexpr int TM = TM_, TN = TN_, EBITS = EBITS_, AFMT = AFMT_, ROWB = 128, SPW = SPW_, KTMAX = 4 * SPW_;
  static constexpr int MT = TM / 32;                                                        // m-tiles of 32 rows
  static constexpr int KSL = EBITS == 4 ? 4 : 2;                                            // MFMAs (k-slices of 64) per stage and m-tile
  static constexpr int NSA = EBITS == 4 ? MT : 1;                                           // A scale pieces per stage
  static constexpr int OFF_B = TM * ROWB, OFF_S = (TM + 32) * ROWB, OFF_SB = OFF_S + NSA * 256, STAGE = OFF_SB + 256;   // 256 B of scale dwords per piece (the B area keeps 32 rows: the fragment reads span them)
  static constexpr int NPA = TM / 8, NPB = TN / 8;                                          // A / B pieces per stage
  static constexpr int LPS = NPA + NPB + NSA + 1;                                           // LDS-DMA instructions per stage
  static_assert(TN == 32 || TN == 16, "tile width");
  static_assert(TM == 32 || TM == 64, "tile height");
  static_assert(EBITS == 4 || EBITS == 8, "element width");
  static_assert(AFMT == 0 || (AFMT == 1 && EBITS == 8), "A format: 0 = e2m1 / e4m3, 1 = e5m2 (MXFP8 only)");
  static constexpr int RED = 4 * TM * 128;                                                  // cross-wave sum: [wave] TM x 32 fp32
  static constexpr int LDS_BYTES = KTMAX * STAGE > RED ? KTMAX * STAGE : RED;
  static_assert(SPW >= 1 && SPW * LPS <= 63, "vmcnt immediate");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// RM: the scale operands are row-major (rows, K / 32) as matmul_ada_mxf4_bf16_tn hands them over (qutlass/csrc/gemm_ada.cu) instead of the to_blocked image
// RING: any K -- the wave's SPW slots are refilled as they are consumed (false: at most 4 SPW stages, every stage has a slot of its own)
template <class C, bool RM = false, bool RING = false>
__global__ __launch_bounds__(256) void gemm_mx_os_kernel(const GemmParams p) {
  constexpr int SPW = C::SPW, LPS = C::LPS, MT = C::MT, KSL = C::KSL, E8 = C::EBITS == 8;
  static_assert(!(RM && E8), "row-major scales: matmul_ada_mxf4_bf16_tn only");
  __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"(p.alpha));   // all scalar argument loads in one round
  const float alpha = *p.alpha;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), i32 = lane & 31, g = lane >> 5;
  // tile: workgroups that share a B row tile are neighbours (and stay on one XCD: xcd_remap)
  const int nb = p.tiles_m * p.tiles_n;
  const int b2 = xcd_remap((int)blockIdx.x, nb);
  const int m0 = uniform((b2 % p.tiles_m) * C::TM), n0 = uniform((b2 / p.tiles_m) * C::TN);
  const int rowbytes = E8 ? p.K : p.K >> 1, KT = (rowbytes + C::ROWB - 1) / C::ROWB, CB = (p.K / 32 + 3) >> 2;
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
  // scale dwords.  fp4: row (m0 | n0) + i32 of column tile 2 kt + g -- byte ((r % 32) * 16 + ((r % 128) / 32) * 4) of the 512-byte tile (qutlass/utils.py:60-64)
  // (RM: row r's eight scale bytes of stage kt are bytes 8 kt .. + 7 of its K / 32 -- the lane's dword is bytes 8 kt + 4 g .. + 3; rows past M / N lie past the descriptor).
  // fp8: the stage is column tile kt; both lane halves fetch the row's dword (TM = 64: lane half g fetches m-tile g's)
  const int KB = p.K >> 5;
  const uint32_t sa_off = RM ? (uint32_t)m0 * KB : (uint32_t)(m0 >> 7) * CB * 512, sb_off = RM ? (uint32_t)n0 * KB : (uint32_t)(n0 >> 7) * CB * 512;
  const __amdgpu_buffer_rsrc_t rSA = make_rsrc(p.SFA + sa_off, p.sfa_bytes - sa_off), rSB = make_rsrc(p.SFB + sb_off, p.sfb_bytes - sb_off);
  const int rowB = (n0 & 127) + i32;   // row of the 128-row scale tile (TN = 16: n0 is a multiple of 16 only; lanes past the 16 rows fetch some row's dword -- unused)
  const int mq = (m0 & 127) >> 5;      // 32-row slab of the 128-row scale tile the A tile starts in
  const int gcol = E8 ? 0 : g * 512;   // fp4: lane half g takes column tile 2 kt + g
  int vSA[C::NSA];
#pragma unroll
  for (int t = 0; t < C::NSA; ++t) {
    const int slab = E8 ? (MT == 2 ? g : 0) : t;   // m-tile whose dword this lane fetches with piece t
    vSA[t] = RM ? (32 * slab + i32) * KB + 4 * g : gcol + i32 * 16 + (mq + slab) * 4;
  }
  const int vSB = RM ? i32 * KB + 4 * g : gcol + (rowB & 31) * 16 + ((rowB & 127) >> 5) * 4;
  constexpr int SCW = E8 ? 512 : 1024;   // scale bytes per stage and 128-row tile

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
    // a column tile past the operand's last one would read the next row tile's bytes (RM: K-blocks past K / 32 the next row's)
    const int os = (kt < KT && (RM ? 8 * kt + 4 * g < KB : (E8 ? kt : 2 * kt + g) < CB)) ? 0 : -1;
#pragma unroll
    for (int t = 0; t < C::NSA; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rSA, (lds_ptr_t)(st + C::OFF_S + t * 256), 4, (vSA[t] & ~os) | ((int)0x80000000 & os), RM ? kt * 8 : kt * SCW, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rSB, (lds_ptr_t)(st + C::OFF_SB), 4, (vSB & ~os) | ((int)0x80000000 & os), RM ? kt * 8 : kt * SCW, 0, 0);
  };

  // ---- everything this wave will ever read (RING: its first SPW stages), requested now ---------------------------------------------------------
#pragma unroll
  for (int j = 0; j < SPW; ++j) issue(wave + 4 * j, j);

  // row i32 of an m-tile / of the B tile; fp4: logical chunk 4 g + js (lane half g owns K-blocks 4 g .. 4 g + 3 of the stage); fp8: chunks 4 js + 2 u + g (see OsCfg);
  // physical chunk ^ ((row >> 1) & 7)
  const int sw = (i32 >> 1) & 7;
  v16f acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

  // one stage out of slot u; RING: kt_next (>= KT: zeros) is requested into the slot as soon as the reads have returned
  auto consume = [&](const int u, const int kt_next) __attribute__((always_inline)) {
    const char* st = smem + (wave * SPW + u) * C::STAGE;
    v4i fa[MT][4], fb[4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {   // fp4: c4 = k-slice js; fp8: c4 = 2 js + u
      const int chunk = E8 ? 4 * (c4 >> 1) + 2 * (c4 & 1) + g : 4 * g + c4;
      const int off = i32 * C::ROWB + ((chunk ^ sw) << 4);
#pragma unroll
      for (int t = 0; t < MT; ++t) fa[t][c4] = *(const v4i*)(st + t * 32 * C::ROWB + off);
      fb[c4] = *(const v4i*)(st + C::OFF_B + off);
    }
    int sa[MT], sb;
#pragma unroll
    for (int t = 0; t < MT; ++t) sa[t] = *(const int*)(st + C::OFF_S + (E8 ? (MT == 2 ? (32 * t + i32) * 4 : lane * 4) : t * 256 + lane * 4));
    sb = *(const int*)(st + C::OFF_SB + lane * 4);
    fence();   // every read issued before the first MFMA (left alone, the compiler reads one k-slice at a time into the same registers: four exposed LDS round trips)
    if constexpr (RING) {
      if constexpr (MT == 1)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) :: "memory");   // the slot is free
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[MT - 1][0]), "+v"(fa[MT - 1][1]), "+v"(fa[MT - 1][2]), "+v"(fa[MT - 1][3]),
                     "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) :: "memory");
      issue(kt_next, u);
      fence();
    }
    if constexpr (E8) {   // "my" scale byte of K-block 2 js + g down to byte 2 js - ... op_sel 2 js picks it
#pragma unroll
      for (int t = 0; t < MT; ++t) sa[t] = (int)((unsigned)sa[t] >> (8 * g));
      sb = (int)((unsigned)sb >> (8 * g));
    }
    // cbsz = format of srcA (the B fragments), blgp = format of srcB (the A fragments): 4 = e2m1, 0 = e4m3, 1 = e5m2
    constexpr int FMT = E8 ? 0 : 4, FMTA = E8 ? C::AFMT : 4;
#pragma unroll
    for (int js = 0; js < KSL; ++js) {
      const v4i b = E8 ? fb[2 * js] : fb[js];
      const v4i bh = E8 ? fb[2 * js + 1] : v4i{0, 0, 0, 0};
      const v8i B8 = {b[0], b[1], b[2], b[3], bh[0], bh[1], bh[2], bh[3]};
      const int ops = E8 ? 2 * js : js;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const v4i a = E8 ? fa[t][2 * js] : fa[t][js];
        const v4i ah = E8 ? fa[t][2 * js + 1] : v4i{0, 0, 0, 0};
        const v8i A8 = {a[0], a[1], a[2], a[3], ah[0], ah[1], ah[2], ah[3]};
        if (ops == 0) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[t], FMT, FMTA, 0, sb, 0, sa[t]);
        if (ops == 1) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[t], FMT, FMTA, 1, sb, 1, sa[t]);
        if (ops == 2) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[t], FMT, FMTA, 2, sb, 2, sa[t]);
        if (ops == 3) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[t], FMT, FMTA, 3, sb, 3, sa[t]);
      }
    }
    fence();
  };
  if constexpr (!RING) {
    static_for<0, SPW>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SPW - 1 - j) * LPS) : "memory");   // this wave's stage j landed (its later stages may still be in flight)
      fence();
      consume(j, 0);
    });
  } else {
    // the wave's stages kt = wave + 4 j, j = 0, 1, ...: slot j % SPW; behind the wait for stage j exactly the SPW - 1 stages after it are outstanding (the refills
    // past K are zero-fill pieces: the count stays the same to the end)
    for (int kt = wave; kt < KT; kt += 4 * SPW) {
      static_for<0, SPW>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        if (u == 0 || kt + 4 * u < KT) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SPW - 1) * LPS) : "memory");
          fence();
          consume(u, kt + 4 * u + 4 * SPW);
        }
      });
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every wave's reads of the stage areas are done: they become the sum's scratch
  fence();

  // ---- cross-wave sum: [wave][row][8 chunks of 4 fp32], chunk ^ (row & 7) (the 8 lanes of a ds_write_b128 group hit 8 chunks) ------------------------
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(v4f*)(smem + (wave * C::TM + 32 * t + i32) * 128 + (((2 * q + g) ^ (i32 & 7)) << 4)) = v4f{acc[t][4 * q + 0], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
  __syncthreads();
  const int cq = tid & 7;   // chunk of 4 columns
#pragma unroll
  for (int h = 0; h < MT; ++h) {
    const int rr = (tid >> 3) + 32 * h;   // row of the tile
    v4f s[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) s[w] = *(const v4f*)(smem + (w * C::TM + rr) * 128 + ((cq ^ (rr & 7)) << 4));
    v4f t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = ((s[0][e] + s[1][e]) + s[2][e]) + s[3][e];
    const int row = m0 + rr, col = n0 + 4 * cq;
    if (row < p.M && col < p.N && 4 * cq < C::TN) {
      v2i o;
      o[0] = (int)pack_bf16x2(t[0] * alpha, t[1] * alpha);
      o[1] = (int)pack_bf16x2(t[2] * alpha, t[3] * alpha);
      *(v2i*)(p.D + (size_t)row * p.ldd + col) = o;
    }
  }
}

// ---- [r6, third session] decode form: a 16x16 output tile per workgroup on v_mfma_scale_f32_16x16x128_f8f6f4 ------------------------------------------------
// At M <= 16 half of the 32-row A tile above is rows that do not exist, and they are fetched all the same (they fall off the descriptor as zeros but still take their
// LDS-DMA pieces and LDS space): a workgroup of the 32x16 form pulls 48 rows x K / 2 bytes through its CU's LDS-DMA path.  The 16x16x128 MFMA takes 32 K elements per lane
// (lane l = row l & 15, K-block kq = l >> 4 of the four blocks of an MFMA), so a 16x16 tile needs 32 rows per stage: a third fewer bytes per CU, and a 4.5-KiB stage
// (one shot up to K = 8192).  Same wave-owned K stages: wave w owns stages w, w + 4, ...; no barrier in the K walk.
//   fp4: a stage is two MFMAs; MFMA h takes chunk 4 h + kq from lane (r, kq); its scale byte is byte kq of the row's dword of column tile 2 kt + h -- the dword shifted
//        right by 8 kq, op_sel 0.
//   fp8: a stage is ONE MFMA; the 8-bit fragment is "split" as in the 32x32x64 form: registers 0-3 = chunk kq, registers 4-7 = chunk 4 + kq, and the scale byte of K-block b
//        is taken from lane group b -- so lane (r, kq) supplies byte kq of the row's dword of column tile kt (again the dword shifted by 8 kq).
// One dword scale piece per operand and stage: lane l fetches the dword of row l & 15 in column tile 2 kt + ((l >> 4) & 1) (fp8: kt).
// TN (third template parameter of the configuration): output columns per workgroup, 16 ... 64 in pieces of 8 rows of B -- the A rows are fetched once for ceil(TN / 16) n-tiles,
// so a WIDE weight at M <= 16 still runs one workgroup per CU with 16 + TN rows per stage (N = 8192: TN = 32, 48 rows instead of the 32x32 form's 64; N = 14336: TN = 56 = 256
// workgroups, 72 rows -- the 32x64 K-split ring kernel fetched 96 and met at a barrier per stage).  TN = 56: the fourth n-tile's columns 8 ... 15 read whatever the LDS holds
// behind the B area -- they only feed output columns that are not stored.
template <int SPW_, int EBITS_ = 4, int AFMT_ = 0, int TN_ = 16>
struct Os16Cfg {
  static constexpr int TM = 16, TN = TN_, EBITS = EBITS_, AFMT = AFMT_, ROWB = 128, SPW = SPW_, KTMAX = 4 * SPW_;
  static constexpr int NT = (TN + 15) / 16, NPB = TN / 8, NSB = (NT + 1) / 2;   // n-tiles (MFMAs per k-slice), B pieces, B scale pieces (one per pair of n-tiles)
  static constexpr int OFF_B = TM * ROWB, OFF_S = (TM + TN) * ROWB, OFF_SB = OFF_S + 256, STAGE = OFF_SB + NSB * 256;
  static constexpr int LPS = 2 + NPB + 1 + NSB;
  static constexpr int RED = 4 * 16 * NT * 64;
  static constexpr int LDS_BYTES = KTMAX * STAGE > RED ? KTMAX * STAGE : RED;
  static_assert(TN % 8 == 0 && TN >= 16 && TN <= 64, "tile width");
  static_assert(EBITS == 4 || EBITS == 8, "element width");
  static_assert(AFMT == 0 || (AFMT == 1 && EBITS == 8), "A format: 0 = e2m1 / e4m3, 1 = e5m2 (MXFP8 only)");
  static_assert(SPW >= 1 && SPW * LPS <= 63, "vmcnt immediate");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// RM: row-major scale operands (rows, K / 32) -- matmul_ada_mxf4_bf16_tn; the lane's dword (K-blocks 4 c .. 4 c + 3 of its row) holds the same four bytes either way
template <class C, bool RING = false, bool RM = false>
__global__ __launch_bounds__(256) void gemm_mx_os16_kernel(const GemmParams p) {
  constexpr int SPW = C::SPW, LPS = C::LPS, E8 = C::EBITS == 8, NT = C::NT;
  static_assert(!(RM && E8), "row-major scales: matmul_ada_mxf4_bf16_tn only");
  __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"(p.alpha));   // all scalar argument loads in one round
  const float alpha = *p.alpha;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), r16 = lane & 15, kq = lane >> 4;
  const int nb = p.tiles_m * p.tiles_n;
  const int b2 = xcd_remap((int)blockIdx.x, nb);
  const int m0 = uniform((b2 % p.tiles_m) * C::TM), n0 = uniform((b2 / p.tiles_m) * C::TN);
  const int rowbytes = E8 ? p.K : p.K >> 1, KT = (rowbytes + C::ROWB - 1) / C::ROWB, CB = (p.K / 32 + 3) >> 2;
  const int tailbytes = rowbytes - (KT - 1) * C::ROWB;

  const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A + a_off, p.a_bytes - a_off), rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);
  int vP[2], chP[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    chP[par] = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    vP[par] = (lane >> 3) * rowbytes + (chP[par] << 4);
  }
  const int rstep = 8 * rowbytes;
  // scale dwords.  A: lane l fetches row l & 15 of column tile 2 kt + ((l >> 4) & 1) (fp8: kt) -- lanes 32-63 a second copy.  B: one piece per PAIR of n-tiles, lane half
  // g = l >> 5 fetching n-tile 2 pp + g; the rows of a column tile of 56 may straddle two 128-row scale tiles, so B rows are addressed from the operand's start.
  const int KB = p.K >> 5;
  const uint32_t sa_off = RM ? (uint32_t)m0 * KB : (uint32_t)(m0 >> 7) * CB * 512;
  const __amdgpu_buffer_rsrc_t rSA = make_rsrc(p.SFA + sa_off, p.sfa_bytes - sa_off), rSB = make_rsrc(p.SFB, p.sfb_bytes);
  const int rowA = (m0 & 127) + r16;
  const int ctl = E8 ? 0 : (kq & 1);   // column tile of the stage this lane fetches
  const int vSA = RM ? r16 * KB + 4 * ctl : ctl * 512 + (rowA & 31) * 16 + (rowA >> 5) * 4;
  int vSB[C::NSB];
#pragma unroll
  for (int pp = 0; pp < C::NSB; ++pp) {
    const int nr = n0 + 16 * (2 * pp + (lane >> 5)) + r16;   // B row (an n-tile past TN: rows of the next workgroup's tile or past N -- unused / zeros)
    vSB[pp] = RM ? nr * KB + 4 * ctl : (nr >> 7) * CB * 512 + ctl * 512 + (nr & 31) * 16 + ((nr & 127) >> 5) * 4;
  }
  constexpr int SCW = RM ? 8 : E8 ? 512 : 1024;   // scale bytes per stage and 128-row tile (RM: per row)

  auto issue = [&](const int kt, const int slot) __attribute__((always_inline)) {   // stage kt into slot `slot` of this wave (kt >= KT: every piece out of range -> zeros)
    char* st = smem + (wave * SPW + slot) * C::STAGE;
    int tail = (kt == KT - 1) ? tailbytes : C::ROWB;
    int oob = (kt < KT) ? 0 : -1;
    asm volatile("" : "+v"(tail), "+v"(oob));
    const int soff = kt * C::ROWB;
#pragma unroll
    for (int t = 0; t < 2 + C::NPB; ++t) {
      const bool isB = t >= 2;
      const int qq = isB ? t - 2 : t, par = qq & 1;
      const int o = oob | ((chP[par] << 4) < tail ? 0 : -1);
      const int v = ((vP[par] + qq * rstep) & ~o) | ((int)0x80000000 & o);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isB ? rB : rA, (lds_ptr_t)(st + (isB ? C::OFF_B : 0) + qq * 1024), 16, v, soff, 0, 0);
    }
    const int os = (kt < KT && (RM ? 8 * kt + 4 * ctl < KB : (E8 ? kt : 2 * kt + ctl) < CB)) ? 0 : -1;   // a column tile past the operand's last one would read the next row tile's bytes (RM: the next row's)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rSA, (lds_ptr_t)(st + C::OFF_S), 4, (vSA & ~os) | ((int)0x80000000 & os), kt * SCW, 0, 0);
#pragma unroll
    for (int pp = 0; pp < C::NSB; ++pp)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rSB, (lds_ptr_t)(st + C::OFF_SB + pp * 256), 4, (vSB[pp] & ~os) | ((int)0x80000000 & os), kt * SCW, 0, 0);
  };

#pragma unroll
  for (int j = 0; j < SPW; ++j) issue(wave + 4 * j, j);

  const int sw = (r16 >> 1) & 7;
  v4f acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

  auto consume = [&](const int u, const int kt_next) __attribute__((always_inline)) {
    const char* st = smem + (wave * SPW + u) * C::STAGE;
    v4i fa[2], fb[NT][2];
    int sa[2], sb[NT][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int off = r16 * C::ROWB + (((4 * h + kq) ^ sw) << 4);
      const int hs = E8 ? 0 : h;
      fa[h] = *(const v4i*)(st + off);
      sa[h] = *(const int*)(st + C::OFF_S + (hs * 16 + r16) * 4);   // the fetching lane (kq & 1 = column tile) put row r16 of column tile h at (h * 16 + r16) * 4
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        fb[t][h] = *(const v4i*)(st + C::OFF_B + t * 16 * C::ROWB + off);
        sb[t][h] = *(const int*)(st + C::OFF_SB + (t >> 1) * 256 + ((t & 1) * 32 + hs * 16 + r16) * 4);
      }
    }
    fence();
    if constexpr (RING) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(sa[0]), "+v"(sa[1]) :: "memory");   // the slot is free (LDS reads return in order: the A reads were issued ...
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(fb[t][0]), "+v"(fb[t][1]), "+v"(sb[t][0]), "+v"(sb[t][1]));   // ... and every B read is pinned behind the wait)
      issue(kt_next, u);
      fence();
    }
    constexpr int FMT = E8 ? 0 : 4, FMTA = E8 ? C::AFMT : 4;   // cbsz = format of srcA (the B fragment), blgp = format of srcB (the A fragment)
    if constexpr (E8) {
      const v8i A8 = {fa[0][0], fa[0][1], fa[0][2], fa[0][3], fa[1][0], fa[1][1], fa[1][2], fa[1][3]};
      const int xa = (int)((unsigned)sa[0] >> (8 * kq));
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const v8i B8 = {fb[t][0][0], fb[t][0][1], fb[t][0][2], fb[t][0][3], fb[t][1][0], fb[t][1][1], fb[t][1][2], fb[t][1][3]};
        acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B8, A8, acc[t], FMT, FMTA, 0, (int)((unsigned)sb[t][0] >> (8 * kq)), 0, xa);
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const v8i A8 = {fa[h][0], fa[h][1], fa[h][2], fa[h][3], 0, 0, 0, 0};
        const int xa = (int)((unsigned)sa[h] >> (8 * kq));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const v8i B8 = {fb[t][h][0], fb[t][h][1], fb[t][h][2], fb[t][h][3], 0, 0, 0, 0};
          acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B8, A8, acc[t], FMT, FMTA, 0, (int)((unsigned)sb[t][h] >> (8 * kq)), 0, xa);
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
      consume(j, 0);
    });
  } else {
    for (int kt = wave; kt < KT; kt += 4 * SPW) {
      static_for<0, SPW>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        if (u == 0 || kt + 4 * u < KT) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SPW - 1) * LPS) : "memory");
          fence();
          consume(u, kt + 4 * u + 4 * SPW);
        }
      });
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every wave's reads of the stage areas are done: they become the sum's scratch
  fence();

  // ---- cross-wave sum: [wave][row m][16 NT columns] fp32; a lane holds row m = r16, columns 16 t + 4 kq .. + 3 (srcA = the B fragment) ----------------------
  constexpr int RROW = 64 * NT;   // bytes per row of a wave's partial tile
#pragma unroll
  for (int t = 0; t < NT; ++t) *(v4f*)(smem + (wave * 16 + r16) * RROW + t * 64 + kq * 16) = acc[t];
  __syncthreads();
  if (tid < 64) {
    const int rr = tid >> 2, cq = tid & 3;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      v4f s[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) s[w] = *(const v4f*)(smem + (w * 16 + rr) * RROW + t * 64 + cq * 16);
      v4f x;
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = ((s[0][e] + s[1][e]) + s[2][e]) + s[3][e];
      const int row = m0 + rr, cl = 16 * t + 4 * cq, col = n0 + cl;
      if (row < p.M && col < p.N && cl < C::TN) {
        v2i o;
        o[0] = (int)pack_bf16x2(x[0] * alpha, x[1] * alpha);
        o[1] = (int)pack_bf16x2(x[2] * alpha, x[3] * alpha);
        *(v2i*)(p.D + (size_t)row * p.ldd + col) = o;
      }
    }
  }
}

#if QAMD_BENCH
// ---- [r6, third session] the decode LAYER y = Q(x h) W^T in one launch, without repeating the quantisation in every workgroup ----------------------------------
// gemm_mx_fusedq.hip.h (round 3) lets every workgroup rotate and quantise the whole activation matrix itself: fine at M <= 4, 8.1 us at M = 16 where two launches take 5.7.
// Here the first `nq` workgroups run the QUANTIZER's own body (quantize.hip.h fused_quantize_body: R = 32, blocked scales) on a quarter-tile share of x each and write the
// codes / scale bytes to caller scratch; every workgroup first requests its WEIGHT stages (they do not depend on x), then waits for the quantised activations at one
// arrival counter ({launch tag : 56, count : 8}: the scratch needs no initialisation), then requests its A stages and runs the decode form's K walk.  The producers are the
// lowest workgroup indices (dispatched first) and wait for nobody, so the consumers' spin cannot deadlock; it is bounded all the same (a timed-out workgroup stores nothing).
// Same arithmetic as fusedQuantizeMxBlocked + gemm_mx_os16_kernel: bit-identical to the two-launch path (24 shapes, profiles/calib_actpath_handoff_r7.txt).
// MEASURED AND NOT ADOPTED (lab build only, "gemm_variant" 580): with one producer (M = 1) the layer takes 5.5 us at N = K = 4096 against 5.2 for two launches, and every further
// producer adds ~1.3 us (M = 16: 26 us) -- a producer's release is a write-back of its XCD's L2 plus a compare-and-swap that executes at the memory side, and the consumers on
// the other seven XCDs see neither sooner than a round trip to memory.  (The first version -- every workgroup fencing, the consumers invalidating their L2 -- took 19 ... 38 us.)
// Eight private L2s make a grid-wide hand-off cost more than the launch it was meant to save.
struct FqOsParams {
  QuantParams q;                 // q.out / q.out_sf = g.A / g.SFA: the scratch
  GemmParams g;
  unsigned long long* flag;      // arrival counter (scratch)
  unsigned long long tag;        // (launch number & (2^56 - 1)) << 8
  int nq;                        // producer workgroups
};

template <class C, int METHOD>
__global__ __launch_bounds__(256) void gemm_mx_os16_fq_kernel(const FqOsParams P) {
  constexpr int SPW = C::SPW, NT = C::NT;
  static_assert(C::EBITS == 4, "MXFP4");
  constexpr int LB = C::NPB + C::NSB, LA = 2 + 1;   // LDS-DMA instructions per stage: weight side, activation side
  static_assert(SPW * LA <= 63 && SPW * LB <= 63, "vmcnt immediate");
  const GemmParams& p = P.g;
  __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
  __shared__ int s_ok;
  const float alpha = *p.alpha;
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), r16 = lane & 15, kq = lane >> 4;
  const int nb = p.tiles_n;      // one m-tile (M <= 16)
  const int b2 = xcd_remap((int)blockIdx.x, nb);
  const int n0 = uniform(b2 * C::TN);
  const int rowbytes = p.K >> 1, KT = (rowbytes + C::ROWB - 1) / C::ROWB, CB = (p.K / 32 + 3) >> 2;
  const int tailbytes = rowbytes - (KT - 1) * C::ROWB;

  const uint32_t b_off = (uint32_t)n0 * rowbytes;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A, p.a_bytes), rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);
  int vP[2], chP[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    chP[par] = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    vP[par] = (lane >> 3) * rowbytes + (chP[par] << 4);
  }
  const int rstep = 8 * rowbytes;
  const __amdgpu_buffer_rsrc_t rSA = make_rsrc(p.SFA, p.sfa_bytes), rSB = make_rsrc(p.SFB, p.sfb_bytes);
  const int ctl = kq & 1;
  const int vSA = ctl * 512 + (r16 & 31) * 16;
  int vSB[C::NSB];
#pragma unroll
  for (int pp = 0; pp < C::NSB; ++pp) {
    const int nr = n0 + 16 * (2 * pp + (lane >> 5)) + r16;
    vSB[pp] = (nr >> 7) * CB * 512 + ctl * 512 + (nr & 31) * 16 + ((nr & 127) >> 5) * 4;
  }
  auto issueB = [&](const int kt, const int slot) __attribute__((always_inline)) {
    char* st = smem + (wave * SPW + slot) * C::STAGE;
    int tail = (kt == KT - 1) ? tailbytes : C::ROWB;
    int oob = (kt < KT) ? 0 : -1;
    asm volatile("" : "+v"(tail), "+v"(oob));
    const int soff = kt * C::ROWB;
#pragma unroll
    for (int qq = 0; qq < C::NPB; ++qq) {
      const int par = qq & 1;
      const int o = oob | ((chP[par] << 4) < tail ? 0 : -1);
      const int v = ((vP[par] + qq * rstep) & ~o) | ((int)0x80000000 & o);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(st + C::OFF_B + qq * 1024), 16, v, soff, 0, 0);
    }
    const int os = (kt < KT && 2 * kt + ctl < CB) ? 0 : -1;
#pragma unroll
    for (int pp = 0; pp < C::NSB; ++pp)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rSB, (lds_ptr_t)(st + C::OFF_SB + pp * 256), 4, (vSB[pp] & ~os) | ((int)0x80000000 & os), kt * 1024, 0, 0);
  };
  auto issueA = [&](const int kt, const int slot) __attribute__((always_inline)) {
    char* st = smem + (wave * SPW + slot) * C::STAGE;
    int tail = (kt == KT - 1) ? tailbytes : C::ROWB;
    int oob = (kt < KT) ? 0 : -1;
    asm volatile("" : "+v"(tail), "+v"(oob));
    const int soff = kt * C::ROWB;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int o = oob | ((chP[qq] << 4) < tail ? 0 : -1);
      const int v = ((vP[qq] + qq * rstep) & ~o) | ((int)0x80000000 & o);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(st + qq * 1024), 16, v, soff, 0, 17);   // aux 17 = sc0 sc1: past the L2
    }
    const int os = (kt < KT && 2 * kt + ctl < CB) ? 0 : -1;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rSA, (lds_ptr_t)(st + C::OFF_S), 4, (vSA & ~os) | ((int)0x80000000 & os), kt * 1024, 0, 17);
  };

  // ---- the weights first: they do not depend on the activations -------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < SPW; ++j) issueB(wave + 4 * j, j);

  // ---- producers: the quantizer's body on this workgroup's share of x --------------------------------------------------------------------------
  if ((int)blockIdx.x < P.nq) fused_quantize_body<32, false, METHOD, false, true, true, false>(P.q, (int)blockIdx.x, P.nq);
  if ((int)blockIdx.x < P.nq) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's stores (and its weight pieces) are done
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                // ... and written back past this XCD's L2 (producers only: the consumers read the scratch with
  }                                                                   //     L2-bypassing loads instead of invalidating their L2 -- that would throw the weights out as well)
  __syncthreads();
  bool ok = true;
  if (tid == 0) {
    unsigned long long cur = __hip_atomic_load(P.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)blockIdx.x < P.nq) {
      for (;;) {
        const int cnt = ((cur & ~0xffull) == P.tag) ? (int)(cur & 0xffull) : 0;
        const unsigned long long want = P.tag | (unsigned long long)(cnt + 1);
        if (__hip_atomic_compare_exchange_strong(P.flag, &cur, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { cur = want; break; }
      }
    }
    const unsigned long long full = P.tag | (unsigned long long)P.nq;
    int spins = 0;
    while (cur != full && spins < (1 << 16)) {
      __builtin_amdgcn_s_sleep(1);
      cur = __hip_atomic_load(P.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++spins;
    }
    ok = cur == full;
    *(volatile int*)&s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  ok = uniform(*(volatile int*)&s_ok) != 0;
  if (!ok) return;   // (bounded wait ran out: nothing stored -- the tests compare every output)

  // ---- the quantised activations ------------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < SPW; ++j) issueA(wave + 4 * j, j);

  const int sw = (r16 >> 1) & 7;
  v4f acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  static_for<0, SPW>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SPW - 1 - j) * LA) : "memory");   // activation pieces of stage j landed (the weights long before)
    fence();
    const char* st = smem + (wave * SPW + j) * C::STAGE;
    v4i fa[2], fb[NT][2];
    int sa[2], sb[NT][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int off = r16 * C::ROWB + (((4 * h + kq) ^ sw) << 4);
      fa[h] = *(const v4i*)(st + off);
      sa[h] = *(const int*)(st + C::OFF_S + (h * 16 + r16) * 4);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        fb[t][h] = *(const v4i*)(st + C::OFF_B + t * 16 * C::ROWB + off);
        sb[t][h] = *(const int*)(st + C::OFF_SB + (t >> 1) * 256 + ((t & 1) * 32 + h * 16 + r16) * 4);
      }
    }
    fence();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v8i A8 = {fa[h][0], fa[h][1], fa[h][2], fa[h][3], 0, 0, 0, 0};
      const int xa = (int)((unsigned)sa[h] >> (8 * kq));
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const v8i B8 = {fb[t][h][0], fb[t][h][1], fb[t][h][2], fb[t][h][3], 0, 0, 0, 0};
        acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B8, A8, acc[t], 4, 4, 0, (int)((unsigned)sb[t][h] >> (8 * kq)), 0, xa);
      }
    }
    fence();
  });
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  fence();
  constexpr int RROW = 64 * NT;
#pragma unroll
  for (int t = 0; t < NT; ++t) *(v4f*)(smem + (wave * 16 + r16) * RROW + t * 64 + kq * 16) = acc[t];
  __syncthreads();
  if (tid < 64) {
    const int rr = tid >> 2, cq = tid & 3;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      v4f s[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) s[w] = *(const v4f*)(smem + (w * 16 + rr) * RROW + t * 64 + cq * 16);
      v4f x;
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = ((s[0][e] + s[1][e]) + s[2][e]) + s[3][e];
      const int cl = 16 * t + 4 * cq, col = n0 + cl;
      if (rr < p.M && col < p.N && cl < C::TN) {
        v2i o;
        o[0] = (int)pack_bf16x2(x[0] * alpha, x[1] * alpha);
        o[1] = (int)pack_bf16x2(x[2] * alpha, x[3] * alpha);
        *(v2i*)(p.D + (size_t)rr * p.ldd + col) = o;
      }
    }
  }
}
#endif   // QAMD_BENCH

}  // namespace qamd
