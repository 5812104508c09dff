// LAB-ONLY schedules of the block-scaled MX GEMM (libqutlass_amd_bench.so, -DQAMD_BENCH=1): included by gemm_mx.hip.h, inside
// namespace qamd, after GemmCtx.  None of them is reachable from the product library's dispatch rules; they stay compiled in the lab
// build because their measurements are part of the design record (DESIGN.md sections 3.3-3.4, 3.8) and the forced-tile parity tests
// (tests/_benchlib.py) run them.  The product's schedules are gemm_mx_ringp (gemm_mx.hip.h), gemm_mx_deepp / gemm_mx_deepp8 and the
// heterogeneous launch (gemm_mx_deepp.hip.h), gemm_mx_skinny_kernel (gemm_mx_skinny.hip.h).
//   gemm_mx_lockstep / _pingpong / _queue / _simple   8-wave (or 4-wave) 2-stage schedules of round 1
//   gemm_mx_deep / gemm_mx_deep8 (+ NN via v_perm)    per-tile predecessors of the persistent deep kernels
//   gemm_mx_regstage                                  deep tiling with the L2 -> LDS copy through registers
//   gemm_mx_ring                                      round-1 ring schedule (whole-stage reads after the barrier)
#pragma once

// -------------------------------------------------------------------------------------------------
// Schedule 1: 2-stage ring, one barrier per stage (all waves in lockstep).
// -------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void gemm_mx_lockstep(char* smem, const GemmParams& p) {
  GemmCtx<C> cx(smem, p);
  cx.issue_stage(0, 0);
  for (int kt = 0; kt < cx.KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // stage kt landed for every wave; everyone is done reading the other buffer
    if (kt + 1 < cx.KT && !(C::ABL & ABL_NO_DMA)) cx.issue_stage(kt + 1, (kt + 1) & 1);
    const int buf = kt & 1;
    cx.read_scales(buf);
    cx.read_frags(buf, 0);
    cx.mfma_slice(0);
    cx.read_frags(buf, 1);
    cx.mfma_slice(1);
    if (C::KSL == 4) {
      cx.read_frags(buf, 2 % C::KSL);
      cx.mfma_slice(2 % C::KSL);
      cx.read_frags(buf, 3 % C::KSL);
      cx.mfma_slice(3 % C::KSL);
    }
  }
  cx.epilogue();
}

// -------------------------------------------------------------------------------------------------
// Schedule 2: ping-pong.  The waves form two groups (wave < NWAVES/2 and the rest; waves w and
// w + NWAVES/2 share a SIMD).  Every stage is four blocks separated by workgroup barriers:
//     L0  ds_read scales + fragments of the first half of the k-slices, issue ALL LDS-DMA of stage kt+1
//     M0  MFMAs of the first half
//     L1  ds_read fragments of the second half; wait lgkmcnt(0) and vmcnt(0)
//     M1  MFMAs of the second half
// Group B runs one block behind group A (one extra barrier up front, one at the end for A), so on
// every SIMD one wave is in an MFMA block while its partner is in a load block.
// Hazards (barrier n of A pairs with barrier n of B, B's code being one block earlier):
//   RAW  stage kt+1 is first read in A's L0(kt+1), entered through the barrier that closes A's M1(kt)
//        and B's L1(kt); every wave executed vmcnt(0) for its own DMA at the end of its L1(kt).
//   WAR  DMA of stage kt+1 is first issued in A's L0(kt), entered through the barrier that closes
//        B's L1(kt-1), at whose end B waited lgkmcnt(0) for its last reads of that buffer.
// -------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void gemm_mx_pingpong(char* smem, const GemmParams& p) {
  GemmCtx<C> cx(smem, p);
  const bool groupB = ((cx.wave >> p.pp_shift) & 1) != 0;
  constexpr int H = C::KSL / 2;   // k-slices per half (2 fp4 / 1 fp8)

  cx.issue_stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (groupB) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  const bool prio = (p.pp_flags & 1) != 0;
  cx.trace();
  for (int kt = 0; kt < cx.KT; ++kt) {
    const int buf = kt & 1;
    // ---- L0 ----
    cx.read_scales(buf);
#pragma unroll
    for (int j = 0; j < H; ++j) cx.read_frags(buf, j);
    if (kt + 1 < cx.KT && !(C::ABL & ABL_NO_DMA)) cx.issue_stage(kt + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    cx.trace();
    __builtin_amdgcn_s_barrier();
    cx.trace();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M0 ----
    if (prio) __builtin_amdgcn_s_setprio(1);
    cx.mfma_slice(0);
    if (H == 2) cx.mfma_slice(1 % C::KSL);
    if (prio) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    cx.trace();
    __builtin_amdgcn_s_barrier();
    cx.trace();
    __builtin_amdgcn_sched_barrier(0);
    // ---- L1 ----
#pragma unroll
    for (int j = H; j < C::KSL; ++j) cx.read_frags(buf, j);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    cx.trace();
    __builtin_amdgcn_s_barrier();
    cx.trace();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M1 ----
    if (prio) __builtin_amdgcn_s_setprio(1);
    cx.mfma_slice(H);
    if (H == 2) cx.mfma_slice(3 % C::KSL);
    if (prio) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    cx.trace();
    __builtin_amdgcn_s_barrier();
    cx.trace();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!groupB) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  cx.trace();
  cx.epilogue();
  cx.trace();
  cx.trace_dump();
}

// -------------------------------------------------------------------------------------------------
// Schedule 3 ("queue"): per-wave software pipeline that keeps the MFMA queue fed.
//
// Measured on MI355X (tests/native trace): a wave ISSUES an MFMA in ~12 cycles
// and the matrix pipe works the queue off at 32 cycles per v_mfma_scale_f32_32x32x64 (fp4); but an
// LDS read whose destination registers are still sources of a queued MFMA does not complete until
// that MFMA has executed.  So the fragment registers are double-buffered per k-slice: slice s+1 is
// read into the OTHER register set right after the MFMAs of slice s were queued, and an empty asm
// keeps set s allocated across those reads so the compiler cannot reuse its registers.
//
// Per stage (4 k-slices, one workgroup barrier):
//     M0 ; R1 ; DMA(kt+1) second half
//     M1 ; R2
//     M2 ; R3
//     M3 ; wait own DMA(kt+1) + own reads ; BARRIER ; R0' (+ scales') ; DMA(kt+2) first half
// RAW  R0' of stage kt+1 follows the barrier that every wave reaches after vmcnt(0) for its DMA(kt+1).
// WAR  DMA(kt+2) overwrites the buffer of stage kt; it follows the barrier that every wave reaches
//      after lgkmcnt(0) for its last reads (R3) of stage kt.
// -------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void gemm_mx_queue(char* smem, const GemmParams& p) {
  static_assert(C::EBITS == 4, "queue schedule is written for fp4 (4 k-slices of one 16-byte chunk)");
  constexpr int MT = C::MT, NT = C::NT;
  GemmCtx<C> cx(smem, p);
  v4i fa[2][MT] = {}, fb[2][NT] = {};
  int sa[2][MT], sb[2][NT];

  auto read_slice = [&](int buf, int j, int set) __attribute__((always_inline)) {
    if ((C::ABL & ABL_NO_READS) && buf >= 0) {   // keep whatever the registers hold (opaque to the optimiser)
#pragma unroll
      for (int t = 0; t < MT; ++t) asm volatile("" : "+v"(fa[set][t]));
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(fb[set][t]));
      return;
    }
    const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[set][t] = *(const v4i*)(st + cx.rdA[j] + t * 32 * C::ROWB);
#pragma unroll
    for (int t = 0; t < NT; ++t) fb[set][t] = *(const v4i*)(st + cx.rdBd + cx.rdA[j] + t * 32 * C::ROWB);
  };
  auto read_scales = [&](int buf, int set) __attribute__((always_inline)) {
    const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < MT; ++t) sa[set][t] = *(const int*)(st + cx.rdSA[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) sb[set][t] = *(const int*)(st + cx.rdSB[t]);
  };
  auto keep = [&](int set) __attribute__((always_inline)) {   // pin the registers of a fragment set across the next slice's reads
#pragma unroll
    for (int t = 0; t < MT; ++t) asm volatile("" ::"v"(fa[set][t]));
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" ::"v"(fb[set][t]));
  };
  auto mfma = [&](int j, int set, int sset) __attribute__((always_inline)) {
    if (C::ABL & ABL_NO_MFMA) return;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const v4i a = fa[set][m], b = fb[set][n];
        const v8i A8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, B8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
        if (j == 0) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 0, sb[sset][n], 0, sa[sset][m]);
        if (j == 1) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 1, sb[sset][n], 1, sa[sset][m]);
        if (j == 2) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 2, sb[sset][n], 2, sa[sset][m]);
        if (j == 3) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 3, sb[sset][n], 3, sa[sset][m]);
      }
  };
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  const bool dma_on = !(C::ABL & ABL_NO_DMA);
  const bool pf_on = (p.pp_flags & 2) != 0;   // L2 warm-up loads (uniform)
  int pf = 0;

  // one stage; BUF = kt & 1 is a compile-time constant so every register-array index is static
  auto stage = [&](int kt, auto bufc) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    const int KT = cx.KT;
    // j = 0
    mfma(0, 0, BUF); fence();
    read_slice(BUF, 1, 1); keep(0); fence();
    cx.trace();                                                     // t1: M0 issued, R1 issued
    if (dma_on) cx.issue_stage_part(kt + 1, BUF ^ 1, 1, kt + 1 < KT);
    fence();
    cx.trace();                                                     // t2: second DMA half issued
    // j = 1
    mfma(1, 1, BUF); fence();
    read_slice(BUF, 2, 0); keep(1); fence();
    // j = 2
    mfma(2, 0, BUF); fence();
    read_slice(BUF, 3, 1); keep(0); fence();
    // j = 3
    mfma(3, 1, BUF); fence();
    cx.trace();                                                     // t3: M1..M3 issued (incl. operand waits)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    cx.trace();                                                     // t4: own DMA landed
    __builtin_amdgcn_s_barrier();
    cx.trace();                                                     // t5: barrier released
    fence();
    read_scales(BUF ^ 1, BUF ^ 1);   // (after the last stage these read stale LDS; the values are never used)
    read_slice(BUF ^ 1, 0, 0);
    keep(1); fence();
    asm volatile("" ::"v"(pf));                                      // previous warm-up load retired (covered by the vmcnt(0) above)
    if (dma_on) cx.issue_stage_part(kt + 2, BUF, 0, kt + 2 < KT);
    pf = cx.prefetch_stage(kt + 4, pf_on);
    fence();
    cx.trace();                                                     // t6 (= t0 of the next stage): R0' + first DMA half issued
  };

  // prologue: stage 0 -> buffer 0 (all of it), first half of stage 1 -> buffer 1
  cx.issue_stage_part(0, 0, 0);
  cx.issue_stage_part(0, 0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  fence();
  read_scales(0, 0);
  read_slice(0, 0, 0);
  fence();
  if (dma_on) cx.issue_stage_part(1, 1, 0, 1 < cx.KT);
  fence();

  int kt = 0;
  for (; kt + 1 < cx.KT; kt += 2) {
    stage(kt, std::integral_constant<int, 0>{});
    stage(kt + 1, std::integral_constant<int, 1>{});
  }
  if (kt < cx.KT) stage(kt, std::integral_constant<int, 0>{});
  fence();
  cx.epilogue();
  cx.trace_dump();
}

// -------------------------------------------------------------------------------------------------
// Schedule 4 ("simple"): the structure of tests/native/ubench.hip mode 11 -- ONE fragment set per wave,
// every k-slice is  R(j) ; M(j) ; a share of the LDS-DMA of stage kt+1 (3,3,2+scale,0 pieces), one
// vmcnt(0)+barrier hand-off per stage.  The WAR stall of R(j+1) behind M(j) makes each wave alternate
// read and MFMA phases, and the two waves of a SIMD fall into complementary phases by themselves.
// -------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void gemm_mx_simple(char* smem, const GemmParams& p) {
  constexpr int KSL = C::KSL;
  GemmCtx<C> cx(smem, p);
  const bool dma_on = !(C::ABL & ABL_NO_DMA);
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  auto slice = [&](int buf, int j) __attribute__((always_inline)) {
    cx.read_frags(buf, j);   // single fragment set: cx.fa[j]/fb[j] of different j never live together
    fence();
    cx.mfma_slice(j);
    fence();
  };
  cx.issue_stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  fence();
  for (int kt = 0; kt < cx.KT; ++kt) {
    const int buf = kt & 1;
    const bool nxt = kt + 1 < cx.KT;
    char* nb = smem + (buf ^ 1) * C::STAGE_BYTES;
    cx.read_scales(buf);
    if (KSL == 4) {
      const bool early = (p.pp_flags & 16) != 0;   // bench switch: whole DMA of stage kt+1 right after the first slice
      slice(buf, 0);
      if (dma_on) {
        cx.issue_pieces_range(C::NA, cx.rA, nb, kt + 1, nxt, 0, (C::NA * 3 + 3) / 4);
        if (early) {
          cx.issue_pieces_range(C::NA, cx.rA, nb, kt + 1, nxt, (C::NA * 3 + 3) / 4, C::NA);
          cx.issue_pieces_range(C::NB, cx.rB, nb + C::OFF_B, kt + 1, nxt, 0, C::NB);
          cx.issue_scales(kt + 1, nb, nxt);
        }
      }
      fence();
      slice(buf, 1);
      if (dma_on && !early) {
        cx.issue_pieces_range(C::NA, cx.rA, nb, kt + 1, nxt, (C::NA * 3 + 3) / 4, C::NA);
        cx.issue_pieces_range(C::NB, cx.rB, nb + C::OFF_B, kt + 1, nxt, 0, C::NB / 2);
      }
      fence();
      slice(buf, 2);
      if (dma_on && !early) {
        cx.issue_pieces_range(C::NB, cx.rB, nb + C::OFF_B, kt + 1, nxt, C::NB / 2, C::NB);
        cx.issue_scales(kt + 1, nb, nxt);
      }
      fence();
      slice(buf, 3 % KSL);
    } else {   // fp8: two slices of 8 x 64-cycle MFMAs
      slice(buf, 0);
      if (dma_on) {
        cx.issue_pieces_range(C::NA, cx.rA, nb, kt + 1, nxt, 0, C::NA);
        cx.issue_pieces_range(C::NB, cx.rB, nb + C::OFF_B, kt + 1, nxt, 0, C::NB);
        cx.issue_scales(kt + 1, nb, nxt);
      }
      fence();
      slice(buf, 1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fence();
  }
  cx.epilogue();
}

// -------------------------------------------------------------------------------------------------
// Schedule 5 ("deep"): ONE wave per SIMD, 128x128 wave tile (4 waves, MT = NT = 4).
//
// Why: LDS feeds ds_read_b128 at ~128 B/clk/CU (measured: fragment reads alone take 1500 cycles per
// 192 KiB, profiles/native_r1_nvfp4_ablation.log).  With 8 waves of 128x64 every k-slice reads
// 8 x 6 KiB = 48 KiB for 64 MFMAs; a stage is 192 KiB of reads + 16 KiB of scales + 72 KiB of DMA
// writes = 2200 LDS-cycles against 2048 MFMA-cycles: the 8-wave schedules are LDS-bandwidth bound.
// 128x128 wave tiles read 8 KiB per 16 MFMAs (32 KiB per slice per CU, -33 %).
//
// One wave per SIMD has no partner to cover its waits, so the wave pipelines itself: four fragment
// sets (one per k-slice), reads issued two slices ahead, and the stage hand-off (vmcnt + barrier) sits
// in the MIDDLE of the stage, between M(1) and M(2), when nothing it waits for is younger than a slice:
//     R(2) ; M(0)
//     R(3) ; M(1)
//     wait own DMA(kt+1) [issued one stage ago] + own reads ; BARRIER
//     scales' ; R'(0) ; M(2) interleaved with the DMA of stage kt+2 (one piece per MFMA)
//     R'(1) ; M(3)
// RAW  R'(0) reads stage kt+1 after the barrier every wave reaches after vmcnt(0) for its DMA(kt+1).
// WAR  DMA(kt+2) overwrites stage kt after the barrier every wave reaches after lgkmcnt(0) for R(3).
// -------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void gemm_mx_deep(char* smem, const GemmParams& p) {
  static_assert(C::EBITS == 4, "deep schedule is written for fp4 (4 k-slices of one 16-byte chunk)");
  constexpr int MT = C::MT, NT = C::NT;
  GemmCtx<C> cx(smem, p);
  v4i fa[4][MT] = {}, fb[4][NT] = {};
  int sa[2][MT], sb[2][NT];

  auto read_slice = [&](int buf, int j) __attribute__((always_inline)) {
    if (C::ABL & ABL_NO_READS) {
#pragma unroll
      for (int t = 0; t < MT; ++t) asm volatile("" : "+v"(fa[j][t]));
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(fb[j][t]));
      return;
    }
    const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[j][t] = *(const v4i*)(st + cx.rdA[j] + t * 32 * C::ROWB);
#pragma unroll
    for (int t = 0; t < NT; ++t) fb[j][t] = *(const v4i*)(st + cx.rdBd + cx.rdA[j] + t * 32 * C::ROWB);
  };
  auto read_scales = [&](int buf, int set) __attribute__((always_inline)) {
    const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < MT; ++t) sa[set][t] = *(const int*)(st + cx.rdSA[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) sb[set][t] = *(const int*)(st + cx.rdSB[t]);
  };
  auto mfma1 = [&](int j, int sset, int m, int n) __attribute__((always_inline)) {
    if (C::ABL & ABL_NO_MFMA) { asm volatile("" ::"v"(fa[j][m]), "v"(fb[j][n])); return; }
    const v4i a = fa[j][m], b = fb[j][n];
    const v8i A8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, B8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    if (j == 0) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 0, sb[sset][n], 0, sa[sset][m]);
    if (j == 1) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 1, sb[sset][n], 1, sa[sset][m]);
    if (j == 2) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 2, sb[sset][n], 2, sa[sset][m]);
    if (j == 3) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 3, sb[sset][n], 3, sa[sset][m]);
  };
  auto mfma = [&](int j, int sset) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) mfma1(j, sset, m, n);
  };
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  const bool dma_on = !(C::ABL & ABL_NO_DMA);
  constexpr int NPIECE = C::NA + C::NB;   // + 1 scale instruction

  // M(2) with the DMA of stage kt+2 threaded through it: one 1-KiB piece behind each MFMA
  auto mfma_dma = [&](int j, int sset, int kt2, int buf2) __attribute__((always_inline)) {
    char* st = smem + buf2 * C::STAGE_BYTES;
    const bool valid = kt2 < cx.KT;
    int idx = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        mfma1(j, sset, m, n);
        if (dma_on) {
          constexpr int PER = (NPIECE + MT * NT - 1) / (MT * NT);
#pragma unroll
          for (int e = 0; e < PER; ++e) {
            const int t = idx * PER + e;
            if (t < C::NA) cx.issue_pieces_range(C::NA, cx.rA, st, kt2, valid, t, t + 1);
            else if (t < NPIECE) cx.issue_pieces_range(C::NB, cx.rB, st + C::OFF_B, kt2, valid, t - C::NA, t - C::NA + 1);
          }
          if (idx == 0) cx.issue_scales(kt2, st, valid);
        }
        fence();
        ++idx;
      }
  };

  auto stage = [&](int kt, auto bufc) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    read_slice(BUF, 2); fence();
    mfma(0, BUF); fence();
    cx.trace();                                                     // t1: R(2) + M(0) issued
    read_slice(BUF, 3); fence();
    mfma(1, BUF); fence();
    cx.trace();                                                     // t2: R(3) + M(1) issued
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    cx.trace();                                                     // t3: own DMA(kt+1) + reads landed
    __builtin_amdgcn_s_barrier();
    cx.trace();                                                     // t4: barrier released
    fence();
    read_scales(BUF ^ 1, BUF ^ 1);   // (after the last stage these read stale LDS; never used)
    read_slice(BUF ^ 1, 0); fence();
    mfma_dma(2, BUF, kt + 2, BUF);
    cx.trace();                                                     // t5: R'(0) + M(2) + DMA(kt+2) issued
    read_slice(BUF ^ 1, 1); fence();
    mfma(3, BUF); fence();
    cx.trace();                                                     // t6 (= t0 of the next stage): R'(1) + M(3) issued
  };

  // prologue: stages 0 and 1 in flight; stage 0 landed -> first two slices into registers
  cx.issue_stage_part(0, 0, 0);
  cx.issue_stage_part(0, 0, 1);
  if (dma_on) {
    cx.issue_stage_part(1, 1, 0, 1 < cx.KT);
    cx.issue_stage_part(1, 1, 1, 1 < cx.KT);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE + 1) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  fence();
  read_scales(0, 0);
  read_slice(0, 0);
  read_slice(0, 1);
  fence();
  cx.trace();

  int kt = 0;
  for (; kt + 1 < cx.KT; kt += 2) {
    stage(kt, std::integral_constant<int, 0>{});
    stage(kt + 1, std::integral_constant<int, 1>{});
  }
  if (kt < cx.KT) stage(kt, std::integral_constant<int, 0>{});
  fence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // zero-fill DMA of the stages past K must not race the epilogue's LDS staging
  cx.epilogue();
  cx.trace_dump();
}

// -------------------------------------------------------------------------------------------------
// Schedule 5b ("deep", fp8): the same 4-wave 128x128 organisation for MXFP8.  A stage has two k-slices of two
// 16-byte chunks per fragment (register halves u = 0, 1; split layout: chunk 4j + 2u + g, op_sel 2j), so there
// are two fragment sets and the hand-off sits between the slices:
//     R(1) ; M(0)
//     wait own DMA(kt+1) + reads ; BARRIER
//     scales' ; R'(0) ; M(1) interleaved with the DMA of stage kt+2 (one piece per MFMA)
// An MFMA is 64 cycles here, a slice 1024: reads and DMA have twice the shadow they have in the fp4 kernel.
// -------------------------------------------------------------------------------------------------
template <class C, bool NN>
__device__ __forceinline__ void gemm_mx_deep8(char* smem, const GemmParams& p) {
  static_assert(C::EBITS == 8 && C::F8SPLIT && C::KSL == 2 && C::CPS == 2, "fp8 deep schedule: split register layout");
  constexpr int MT = C::MT, NT = C::NT;
  static_assert(!NN || (MT == 4 && C::BM == 256 && C::NWAVES == 4), "fused NN: 4 row fragments per lane, 8 A^T pieces per wave");
  GemmCtx<C> cx(smem, p);
  v8i fa[2][MT] = {}, fb[2][NT] = {};
  int sa[2][MT], sb[2][NT];

  // ---- fused NN (A handed over as (K, M), matmul_host_mxf8_bf16_nn, gemm.cu:388-434) --------------------------------
  // The A^T stage is DMAed as it lies in memory: [128 k][256 m] bytes, 256-byte rows (piece = 4 k-rows; 16-byte chunk
  // c of row k stored at chunk c ^ 8*((k>>4)&1) so that the two lane halves, which read k-chunks of opposite parity,
  // hit disjoint banks).  A lane then reads the DWORD (k, m = 4*i32 .. +3) for the 16 k of its chunk and transposes
  // 4x4 byte blocks in registers (v_perm_b32): the four bytes of a dword belong to FOUR DIFFERENT row fragments, i.e.
  // lane i32 of fragment t owns tile row 4*i32 + t -- a permutation of M inside the wave tile that only the scale
  // addressing and the epilogue need to know about.
  const __amdgpu_buffer_rsrc_t rAT = make_rsrc(p.A, p.a_bytes);
  int nn_voff[2], nn_rd = 0, nn_rdS[MT];
  if (NN) {
    cx.perm_rows = true;
    const int r = cx.lane >> 4, pc = cx.lane & 15;
#pragma unroll
    for (int par = 0; par < 2; ++par) {                       // parity of (piece >> 2) = (k >> 4) & 1
      const int lc = pc ^ (8 * par);
      nn_voff[par] = (cx.m0 + lc * 16 < p.M) ? r * p.M + cx.m0 + lc * 16 : 0x7f000000;   // columns past M read 0
    }
    const int lc = cx.wave_m * 8 + (cx.i32 >> 2);
    nn_rd = (lc << 4) + ((cx.i32 & 3) << 2);                  // + k*256, chunk ^ 8 for odd k-chunks (applied per read)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int row = cx.wave_m * C::WTM + 4 * cx.i32 + t;
      nn_rdS[t] = C::OFF_S + ((row >> 7) * C::SCT) * (1024 / C::PPW) + (row & 31) * 16 + ((row & 127) >> 5) * 4;
    }
  }
  auto issue_AT = [&](int kt, char* st, const int t0, const int t1) __attribute__((always_inline)) {
#pragma unroll
    for (int t = t0; t < t1; ++t) {
      const int q = cx.wave * 8 + t;                          // piece = k-rows 4q .. 4q+3 of the stage
      const int k = kt * 128 + 4 * q + (cx.lane >> 4);
      int v = (k < p.K) ? (((q >> 2) & 1) ? nn_voff[1] : nn_voff[0]) + (kt * 128 + 4 * q) * p.M : 0x7f000000;   // rows past K read 0
      asm volatile("" : "+v"(v));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rAT, (lds_ptr_t)(st + q * 1024), 16, v, 0, 0, QAMD_DMA_AUX);
    }
  };
  auto read_A_nn = [&](int buf, int j) __attribute__((always_inline)) {
    const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c16 = 4 * j + 2 * u + cx.g;                   // this lane's 16-byte k-chunk (split layout)
      const char* base = st + c16 * 16 * 256 + (nn_rd ^ ((c16 & 1) << 7));
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        uint32_t o[4];
        transpose4x4_u8(*(const uint32_t*)(base + (4 * a + 0) * 256), *(const uint32_t*)(base + (4 * a + 1) * 256),
                        *(const uint32_t*)(base + (4 * a + 2) * 256), *(const uint32_t*)(base + (4 * a + 3) * 256), o);
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[j][t][4 * u + a] = (int)o[t];
      }
    }
  };

  // the same in two halves, so that the loads of group gi+1 can be issued before the v_perms of group gi (one group =
  // 4 dwords = 4 k-rows x the lane's 4 row fragments; gi = 4u + a)
  uint32_t nn_d[2][4];
  auto nn_reads = [&](int buf, int j, int gi) __attribute__((always_inline)) {
    const int u = gi >> 2, a = gi & 3;
    const int c16 = 4 * j + 2 * u + cx.g;
    const char* base = smem + buf * C::STAGE_BYTES + c16 * 16 * 256 + (nn_rd ^ ((c16 & 1) << 7)) + 4 * a * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) nn_d[gi & 1][r] = *(const uint32_t*)(base + r * 256);
  };
  auto nn_perms = [&](int j, int gi) __attribute__((always_inline)) {
    uint32_t o[4];
    transpose4x4_u8(nn_d[gi & 1][0], nn_d[gi & 1][1], nn_d[gi & 1][2], nn_d[gi & 1][3], o);
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[j][t][gi] = (int)o[t];
  };
  auto read_B = [&](int buf, int j) __attribute__((always_inline)) {
    const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const v4i lo = *(const v4i*)(st + cx.rdBd + cx.rdA[2 * j] + t * 32 * C::ROWB);
      const v4i hi = *(const v4i*)(st + cx.rdBd + cx.rdA[2 * j + 1] + t * 32 * C::ROWB);
      fb[j][t] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  auto read_slice = [&](int buf, int j) __attribute__((always_inline)) {
    const char* st = smem + buf * C::STAGE_BYTES;
    if (NN) read_A_nn(buf, j);
    else {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const v4i lo = *(const v4i*)(st + cx.rdA[2 * j] + t * 32 * C::ROWB);
        const v4i hi = *(const v4i*)(st + cx.rdA[2 * j + 1] + t * 32 * C::ROWB);
        fa[j][t] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const v4i lo = *(const v4i*)(st + cx.rdBd + cx.rdA[2 * j] + t * 32 * C::ROWB);
      const v4i hi = *(const v4i*)(st + cx.rdBd + cx.rdA[2 * j + 1] + t * 32 * C::ROWB);
      fb[j][t] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  auto read_scales = [&](int buf, int set) __attribute__((always_inline)) {
    const char* st = smem + buf * C::STAGE_BYTES;
    const int shift = 8 * cx.g;   // split layout: lanes 0-31 carry K-block 2j, lanes 32-63 K-block 2j+1
#pragma unroll
    for (int t = 0; t < MT; ++t) sa[set][t] = (int)((unsigned)(*(const int*)(st + (NN ? nn_rdS[t] : cx.rdSA[t]))) >> shift);
#pragma unroll
    for (int t = 0; t < NT; ++t) sb[set][t] = (int)((unsigned)(*(const int*)(st + cx.rdSB[t])) >> shift);
  };
  auto mfma1 = [&](int j, int sset, int m, int n) __attribute__((always_inline)) {
    if (j == 0) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[0][n], fa[0][m], cx.acc[m][n], 0, C::AFMT, 0, sb[sset][n], 0, sa[sset][m]);
    if (j == 1) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[1][n], fa[1][m], cx.acc[m][n], 0, C::AFMT, 2, sb[sset][n], 2, sa[sset][m]);
  };
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  constexpr int NPIECE = C::NA + C::NB;

  auto stage = [&](int kt, auto bufc) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    if (NN) { read_B(BUF, 1); } else { read_slice(BUF, 1); }
    fence();
    {
      int k = 0;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          mfma1(0, BUF, m, n);
          if (NN) {   // A fragments of slice 1, one group per MFMA shadow, loads one group ahead of the permutes
            if (k < 8) nn_reads(BUF, 1, k);
            if (k >= 1 && k <= 8) nn_perms(1, k - 1);
            fence();
          }
          ++k;
        }
    }
    fence();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    read_scales(BUF ^ 1, BUF ^ 1);   // (after the last stage these read stale LDS; never used)
    if (NN) { read_B(BUF ^ 1, 0); } else { read_slice(BUF ^ 1, 0); }
    fence();
    char* st = smem + BUF * C::STAGE_BYTES;
    const bool valid = kt + 2 < cx.KT;
    int idx = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        mfma1(1, BUF, m, n);
        if (NN) {   // A fragments of slice 0 of the next stage
          if (idx < 8) nn_reads(BUF ^ 1, 0, idx);
          if (idx >= 1 && idx <= 8) nn_perms(0, idx - 1);
        }
        constexpr int PER = (NPIECE + MT * NT - 1) / (MT * NT);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
          const int t = idx * PER + e;
          if (t < C::NA) {
            if (NN) issue_AT(kt + 2, st, t, t + 1);
            else cx.issue_pieces_range(C::NA, cx.rA, st, kt + 2, valid, t, t + 1);
          }
          else if (t < NPIECE) cx.issue_pieces_range(C::NB, cx.rB, st + C::OFF_B, kt + 2, valid, t - C::NA, t - C::NA + 1);
        }
        if (idx == 0) cx.issue_scales(kt + 2, st, valid);
        fence();
        ++idx;
      }
  };

  if (NN) {
    issue_AT(0, smem, 0, C::NA);
    cx.issue_scales(0, smem, true);
    cx.issue_stage_part(0, 0, 1);
    issue_AT(1, smem + C::STAGE_BYTES, 0, C::NA);
    cx.issue_scales(1, smem + C::STAGE_BYTES, 1 < cx.KT);
    cx.issue_stage_part(1, 1, 1, 1 < cx.KT);
  } else {
    cx.issue_stage_part(0, 0, 0);
    cx.issue_stage_part(0, 0, 1);
    cx.issue_stage_part(1, 1, 0, 1 < cx.KT);
    cx.issue_stage_part(1, 1, 1, 1 < cx.KT);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE + 1) : "memory");
  __builtin_amdgcn_s_barrier();
  fence();
  read_scales(0, 0);
  read_slice(0, 0);
  fence();

  int kt = 0;
  for (; kt + 1 < cx.KT; kt += 2) {
    stage(kt, std::integral_constant<int, 0>{});
    stage(kt + 1, std::integral_constant<int, 1>{});
  }
  if (kt < cx.KT) stage(kt, std::integral_constant<int, 0>{});
  fence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  cx.epilogue();
}

// -------------------------------------------------------------------------------------------------
// Schedule 6 ("regstage"): the deep schedule with the HBM/L2 -> LDS copy staged through REGISTERS.
//
// A 2-deep LDS ring (2 x 68 KiB of 160) gives LDS-DMA exactly one stage to land, and the DMA of a stage
// can only be issued once the buffer it overwrites has been read: latency (~1 us) + streaming (68 KiB at
// the 64 B/clk the TA delivers into LDS) exceed the 2048 MFMA-cycles of a stage, and issuing a stage's 17
// pieces in one burst stalls the issuing wave -- and with it the wave's MFMAs -- on the full VMEM queue.
// Registers are the third buffer: every wave owns 16 one-KiB pieces of the tile (8 of A, 8 of B, 16 bytes
// per lane each) and keeps them in 64 VGPRs as ordinary global loads in flight for a whole stage.  Per
// stage, spread one piece per four MFMAs:  ds_write_b128 piece i (loaded one stage ago) into the LDS stage
// it belongs to, then reload register i for one stage later.  The loads need no LDS buffer to be free, so
// they are never bursty and have a full stage to land; LDS sees 68 KiB of plain writes per stage.
// Scales skip LDS altogether: the to_blocked line (r%32)*16 holds the scales of rows r, r+32, r+64, r+96,
// i.e. of the lane's four 32-row fragments -- one 16-byte global load per operand per stage.
// Arch-VGPR budget (the 256 accumulators live in AGPRs): 2 fragment sets 64 + staging 64 + scales 16.
//
//     stage kt, BUF = kt&1      reads threaded in     pieces written            reloaded for
//       M(0)                    slice 1               4..7   of stage kt+1      stage kt+2
//       M(1)                    slice 2               8..11  of stage kt+1      stage kt+2
//       M(2)                    slice 3               12..15 of stage kt+1      stage kt+2
//       lgkmcnt(0) ; BARRIER        (stage kt+1 complete in LDS[BUF^1]; LDS[BUF] no longer read)
//       M(3)                    slice 0 of kt+1       0..3   of stage kt+2      stage kt+3
// -------------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void gemm_mx_regstage(char* smem, const GemmParams& p) {
  static_assert(C::EBITS == 4 && C::NWAVES == 4 && C::MT == 4 && C::NT == 4 && C::BM == 256 && C::BN == 256,
                "regstage schedule: fp4, 256x256 tile, 4 waves of 128x128");
  constexpr int MT = C::MT, NT = C::NT, NA = C::NA, NB = C::NB, NP = NA + NB;
  static_assert(NP == 16, "one piece per four MFMAs");
  GemmCtx<C> cx(smem, p);
  v4i fa[2][MT] = {}, fb[2][NT] = {};   // fragment set j&1
  v4i sA[2], sB[2];     // scale words of the lane's 4 A / 4 B fragments, per stage parity
  v4i stg[NP];          // staging registers: pieces 0..NA-1 of A, NA..NP-1 of B

  const int lane = cx.lane, wave = cx.wave;
  // global side: piece q covers rows 8q..8q+7 of the operand tile; lane -> row 8q + lane/8, 16-byte chunk lane%8
  const int gl_off = (lane >> 3) * cx.rowbytes + ((lane & 7) << 4);
  const int tail_bytes = cx.rowbytes - (cx.KT - 1) * C::ROWB;                       // valid bytes of the last stage
  const int gl_tail = (((lane & 7) << 4) < tail_bytes) ? gl_off : 0x7f000000;       // K tail: chunks past K read 0
  // LDS side: physical chunk = chunk ^ ((row>>1)&7), row = 8q + lane/8 -> depends on the parity of q only; the wave's
  // first piece is folded into the per-lane base so every piece is base[parity] + a compile-time offset
  int wofs[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    wofs[par] = wave * (NA * 1024) + (lane >> 3) * C::ROWB + ((((lane & 7) ^ ((4 * par + (lane >> 4)) & 7))) << 4);
  // scales: 16-byte line of (row tile, column tile kt*2 + g), lane row i32
  const __amdgpu_buffer_rsrc_t rSA = make_rsrc(p.SFA, p.sfa_bytes), rSB = make_rsrc(p.SFB, p.sfb_bytes);
  const int sA_off = ((cx.m0 >> 7) + cx.wave_m) * cx.CB * 512 + cx.g * 512 + cx.i32 * 16;   // lane half g owns column tile g of the stage
  const int sB_off = ((cx.n0 >> 7) + cx.wave_n) * cx.CB * 512 + cx.g * 512 + cx.i32 * 16;

  auto load_piece = [&](int i, int kt) __attribute__((always_inline)) {
    const bool isB = i >= NA;
    const int t = isB ? i - NA : i;
    const int q = wave * NA + t;          // NA == NB
    int v = ((kt == cx.KT - 1) ? gl_tail : gl_off) + ((kt < cx.KT) ? 0 : 0x7f000000);
    asm volatile("" : "+v"(v));                                      // keep the K loop one basic block
    stg[i] = __builtin_amdgcn_raw_buffer_load_b128(isB ? cx.rB : cx.rA, v + q * cx.rstep, kt * C::ROWB, 0);
  };
  auto write_piece = [&](int i, int buf) __attribute__((always_inline)) {
    const bool isB = i >= NA;
    const int t = isB ? i - NA : i;       // parity of q = wave*NA + t is the parity of t (NA even)
    char* dst = smem + ((t & 1) ? wofs[1] : wofs[0]) + (buf * C::STAGE_BYTES + (isB ? C::OFF_B : 0) + t * 1024);
    *(v4i*)dst = stg[i];
  };
  auto load_scales = [&](int kt, int set) __attribute__((always_inline)) {
    int oob = (kt * C::SCT + cx.g < cx.CB) ? 0 : 0x7f000000;        // K tail / stages past K: no such column tile -> 0
    asm volatile("" : "+v"(oob));
    sA[set] = __builtin_amdgcn_raw_buffer_load_b128(rSA, sA_off + oob, kt * C::SCT * 512, 0);   // soffset must be wave-uniform
    sB[set] = __builtin_amdgcn_raw_buffer_load_b128(rSB, sB_off + oob, kt * C::SCT * 512, 0);
  };
  auto read_slice = [&](int buf, int j) __attribute__((always_inline)) {
    const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[j & 1][t] = *(const v4i*)(st + cx.rdA[j] + t * 32 * C::ROWB);
#pragma unroll
    for (int t = 0; t < NT; ++t) fb[j & 1][t] = *(const v4i*)(st + cx.rdBd + cx.rdA[j] + t * 32 * C::ROWB);
  };
  auto mfma1 = [&](int j, int sset, int m, int n) __attribute__((always_inline)) {
    if (C::ABL & ABL_NO_MFMA) { asm volatile("" ::"v"(fa[j & 1][m]), "v"(fb[j & 1][n])); return; }
    const v4i a = fa[j & 1][m], b = fb[j & 1][n];
    const v8i A8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, B8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    const int sa = sA[sset][m], sb = sB[sset][n];
    if (j == 0) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 0, sb, 0, sa);
    if (j == 1) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 1, sb, 1, sa);
    if (j == 2) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 2, sb, 2, sa);
    if (j == 3) cx.acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, cx.acc[m][n], 4, 4, 3, sb, 3, sa);
  };
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  const bool copy_on = !(C::ABL & ABL_NO_DMA);

  // One fragment read of slice j (set j&1): e = 0 -> B[0], 1 -> A[0], 2..4 -> B[1..3], 5..7 -> A[1..3]
  // (the order the m-major MFMA loop consumes them in).
  auto read_one = [&](int buf, int j, int e) __attribute__((always_inline)) {
    const char* st = smem + buf * C::STAGE_BYTES;
    const bool isA = (e == 1) || (e >= 5);
    const int t = (e == 0) ? 0 : (e == 1) ? 0 : (e <= 4) ? e - 1 : e - 4;
    if (isA) fa[j & 1][t] = *(const v4i*)(st + cx.rdA[j] + t * 32 * C::ROWB);
    else fb[j & 1][t] = *(const v4i*)(st + cx.rdBd + cx.rdA[j] + t * 32 * C::ROWB);
  };
  // 16 MFMAs of slice j with ONE other instruction in the shadow of each (a wave can queue only ~1 MFMA ahead,
  // so anything that does not hide behind the 32 cycles of the MFMA in front of it is lost matrix time):
  //   k = 0..7    fragment read e = k of slice rj in LDS[rbuf]      (next slice; complete long before k = 15)
  //   k = 8..11   ds_write_b128 of piece i0 + k-8 into LDS[wbuf]    (loaded one stage ago)
  //   k = 12..15  reload of that register for stage lkt
  auto slice = [&](int j, int sset, int rbuf, int rj, int i0, int wbuf, int lkt) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int k = m * NT + n;
        mfma1(j, sset, m, n);
        if (k < 8) {
          if (!(C::ABL & ABL_NO_READS)) read_one(rbuf, rj, k);
        } else if (copy_on) {
          if (k < 12) write_piece(i0 + k - 8, wbuf);
          else load_piece(i0 + k - 12, lkt);
        }
        fence();
      }
  };

  auto stage = [&](int kt, auto bufc) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    slice(0, BUF, BUF, 1, 4, BUF ^ 1, kt + 2);
    slice(1, BUF, BUF, 2, 8, BUF ^ 1, kt + 2);
    slice(2, BUF, BUF, 3, 12, BUF ^ 1, kt + 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    slice(3, BUF, BUF ^ 1, 0, 0, BUF, kt + 3);
    load_scales(kt + 2, BUF);      // set BUF is free once M(3) of this stage has been issued
    fence();
  };

  // ---- prologue: stage 0 by LDS-DMA (no registers), stage 1 into the staging registers in parallel ----------
  cx.issue_pieces(NA, cx.rA, smem, 0, true);
  cx.issue_pieces(NB, cx.rB, smem + C::OFF_B, 0, true);
  load_scales(0, 0);
  load_scales(1, 1);
#pragma unroll
  for (int i = 0; i < NP; ++i) load_piece(i, 1);
  fence();
#pragma unroll
  for (int i = 0; i < 4; ++i) {     // pieces 0..3 of stage 1 -> LDS[1]; their registers go on to stage 2
    write_piece(i, 1);
    load_piece(i, 2);
  }
  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // the DMA of stage 0 is older than everything still in flight (4 reloads)
  __builtin_amdgcn_s_barrier();
  fence();
  read_slice(0, 0);
  fence();

  int kt = 0;
  for (; kt + 1 < cx.KT; kt += 2) {
    stage(kt, std::integral_constant<int, 0>{});
    stage(kt + 1, std::integral_constant<int, 1>{});
  }
  if (kt < cx.KT) stage(kt, std::integral_constant<int, 0>{});
  fence();
  cx.epilogue();
}

// ================================================================================================
// Ring schedule for SMALL tiles (64x64 .. 128x128): the simple schedule keeps one stage in flight, so a K stage costs a
// full memory round trip (0.58 us measured, whatever the tile does: 64x64 tiles hold 4 MFMAs per wave per stage) and
// a problem with few tiles and a long K is latency bound (M = 64..256, N = 4096, K = 14336: 33 us flat).  Here the LDS
// ring is NSTAGE deep and NSTAGE-1 stages are in flight: at the top of stage kt a wave waits until its own pieces of
// stage kt have landed (vmcnt = (NSTAGE-2) x loads per stage: DMA loads retire in order), the barrier makes every
// wave's pieces visible and proves everyone is done with stage kt-1, whose slot then takes stage kt+NSTAGE-1.
// Stages past K are issued with out-of-range offsets (zero fill), which keeps the vmcnt arithmetic uniform.
// Same K order per output as every other schedule -> bit-identical results.
// ================================================================================================
// RM = true: SFA / SFB are the UN-swizzled row-major (rows, K/32) scale matrices of matmul_ada_mxf4_bf16_tn (64x64 fp4
// tiles only): per stage wave w fetches dword (w & 1) of the 8 scale bytes of the 64 rows of operand (w >> 1) -- lane =
// row, 4 bytes each -- into [operand][dword][row] in the stage's scale area, and a lane's scale dword for its K-blocks
// 4g .. 4g+3 is [operand][g][row].
template <class C, bool RM = false>
__device__ __forceinline__ void gemm_mx_ring(char* smem, const GemmParams& p) {
  constexpr int KSL = C::KSL, D = C::NSTAGE;
  constexpr int LPS = C::NA + C::NB + 1;            // DMA instructions per wave per stage
  static_assert(D >= 3 && (D - 2) * LPS <= 63, "vmcnt immediate");
  static_assert(!RM || (C::EBITS == 4 && C::BM == 64 && C::BN == 64 && C::NWAVES == 4), "row-major scales: 64x64 fp4 tiles");
  GemmCtx<C> cx(smem, p);
  __amdgpu_buffer_rsrc_t rSrm = cx.rS;
  int vSrm = 0x7fffffff;
  const int KBr = p.K >> 5;                         // scale bytes per row (row-major)
  if (RM) {
    const int opB = cx.wave >> 1, dw = cx.wave & 1;
    const uint32_t row0 = opB ? (uint32_t)cx.n0 : (uint32_t)cx.m0;
    const uint32_t total = opB ? p.sfb_bytes : p.sfa_bytes, off = row0 * (uint32_t)KBr;
    rSrm = make_rsrc((opB ? p.SFB : p.SFA) + off, total > off ? total - off : 0);   // rows past M / N fall off the end -> 0
    vSrm = cx.lane * KBr + dw * 4;
#pragma unroll
    for (int t = 0; t < C::MT; ++t) cx.rdSA[t] = C::OFF_S + cx.g * 256 + (cx.wave_m * C::WTM + 32 * t + cx.i32) * 4;
#pragma unroll
    for (int t = 0; t < C::NT; ++t) cx.rdSB[t] = C::OFF_S + 512 + cx.g * 256 + (cx.wave_n * C::WTN + 32 * t + cx.i32) * 4;
  }
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  // K range of this workgroup (split-K: grid.y = splits, every split non-empty by construction on the host)
  int kt0 = 0, kt1 = cx.KT;
  if (p.splits > 1) {
    const int per = (cx.KT + p.splits - 1) / p.splits;
    kt0 = uniform((int)blockIdx.y * per);
    kt1 = min(cx.KT, kt0 + per);
  }
  // With one wave per SIMD the stage cost is instruction issue, so everything that does not depend on the stage is
  // hoisted: per-piece source offsets (normal / K-tail flavour), and the ring is unrolled D times so that LDS slot
  // addresses are immediates.
  int vA[C::NA], vAT[C::NA], vB[C::NB], vBT[C::NB];
#pragma unroll
  for (int t = 0; t < C::NA; ++t) {
    const int q = cx.wave * C::NA + t;
    vA[t] = cx.voffAB[q & 1] + q * cx.rstep;
    vAT[t] = cx.voffT[q & 1] == 0x7fffffff ? 0x7fffffff : cx.voffT[q & 1] + q * cx.rstep;
  }
#pragma unroll
  for (int t = 0; t < C::NB; ++t) {
    const int q = cx.wave * C::NB + t;
    vB[t] = cx.voffAB[q & 1] + q * cx.rstep;
    vBT[t] = cx.voffT[q & 1] == 0x7fffffff ? 0x7fffffff : cx.voffT[q & 1] + q * cx.rstep;
  }
  // stages past the range re-load the last one into a free slot that is never read: keeps the vmcnt arithmetic uniform
  // without any out-of-range bookkeeping
  auto issue = [&](int kt, const int slot) __attribute__((always_inline)) {
    char* st = smem + slot * C::STAGE_BYTES;
    const int ktc = min(kt, kt1 - 1);
    const int soff = ktc * C::ROWB;
    int lastmask = (cx.ktail && ktc == cx.KT - 1) ? -1 : 0;
    asm volatile("" : "+v"(lastmask));
#pragma unroll
    for (int t = 0; t < C::NA; ++t) {
      const int v = (vAT[t] & lastmask) | (vA[t] & ~lastmask);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(cx.rA, (lds_ptr_t)(st + (cx.wave * C::NA + t) * 1024), 16, v, soff, 0, QAMD_DMA_AUX);
    }
#pragma unroll
    for (int t = 0; t < C::NB; ++t) {
      const int v = (vBT[t] & lastmask) | (vB[t] & ~lastmask);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(cx.rB, (lds_ptr_t)(st + C::OFF_B + (cx.wave * C::NB + t) * 1024), 16, v, soff, 0, QAMD_DMA_AUX);
    }
    if (RM) {
      const int vs = (ktc * 8 + (cx.wave & 1) * 4 < KBr) ? vSrm : 0x7fffffff;   // K tail: the stage's second scale dword does not exist
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rSrm, (lds_ptr_t)(st + C::OFF_S + cx.wave * 256), 4, vs, ktc * 8, 0, 0);
    } else {
      const int vs = (ktc * C::SCT + cx.colS < cx.CB) ? cx.voffS : 0x7fffffff;   // K tail: no such scale column tile
      __builtin_amdgcn_raw_ptr_buffer_load_lds(cx.rS, (lds_ptr_t)(st + C::OFF_S + cx.wave * 1024), 16, vs, ktc * C::SCT * 512, 0, 0);
    }
  };
  auto stage = [&](int kt, const int slot) __attribute__((always_inline)) {
    cx.trace();   // (ABL_TRACE builds only) 0: stage begin
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * LPS) : "memory");
    cx.trace();   // 1: own pieces landed
    __builtin_amdgcn_s_barrier();
    fence();
    cx.trace();   // 2: barrier passed
    // all fragments of the stage up front (<= 64 VGPRs): ONE LDS latency per stage instead of one per slice -- and the
    // reads go first, so the ~230 cycles of DMA issue below run while the LDS serves them (rtrace: reads 450 cycles)
    if (!(C::ABL & ABL_NO_READS)) {
      cx.read_scales(slot);
#pragma unroll
      for (int j = 0; j < KSL; ++j) cx.read_frags(slot, j);
    }
    fence();
    cx.trace();   // 3: reads issued
    if (!(C::ABL & ABL_NO_DMA)) issue(kt + D - 1, (slot + D - 1) % D);
    fence();
    if (C::ABL & ABL_TRACE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cx.trace();   // 4: DMA issued, fragments in registers
#pragma unroll
    for (int j = 0; j < KSL; ++j) cx.mfma_slice(j);
    fence();
    cx.trace();   // 5: MFMAs issued
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s)
    if (!(C::ABL & ABL_NO_DMA)) issue(kt0 + s, s);
  for (int kt = kt0; kt < kt1; kt += D) {
#pragma unroll
    for (int u = 0; u < D; ++u)
      if (u == 0 || kt + u < kt1) stage(kt + u, u);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing re-loads must land before the epilogue reuses the LDS
  cx.trace();
#if QAMD_BENCH
  if (p.splits > 1 && p.ctr) cx.epilogue_splitk_fused(blockIdx.y);
  else
#endif
  if (p.splits > 1) cx.epilogue_partial(blockIdx.y);
  else cx.epilogue();
  cx.trace();
  cx.trace_dump();
}

