// LAB COPY of the persistent MXFP4 / MXFP8 kernels (namespace qamd::labk; compiled into libqutlass_amd_bench.so only, -DQAMD_BENCH=1): the kernels of
// gemm_mx_deepp.hip.h WITH their laboratory arms -- stage traces (TRACE), result-changing timing ablations (LAB), the stream-K unit walk (SK), the round-4 RB2
// read-back schedule, the round-5 bf16-first retirement with block MFMA order / burst / per-part ablations (QAMD_DEEPP_RETIRE, QAMD_DEEPP_FS_IL, QAMD_FS_BURST,
// QAMD_FS_ABL), the NN-operand ablations of the fp8 twin.  The PRODUCT header carries none of this: an experiment edits this file and cannot change a byte of
// libqutlass_amd.so (tests/test_cabi_and_host.py::test_product_build_does_not_see_the_lab_sources).  History and measurements: docs/history.md, profiles/.
#pragma once
#include "../gemm_mx.hip.h"
#include "../streamk.hip.h"

namespace qamd {
namespace labk {

template <class C>
struct DeepPCfg {
  static constexpr int STAGE = C::STAGE_BYTES;
  // Epilogue scratch: none of its own.  In the last K stage of a tile, after the hand-off barrier nobody reads buffer 1 any
  // more and the only writer of rows [64 w, 64 w + 64) of its A area -- 8 KiB -- is wave w itself (its own LDS-DMA pieces of
  // the next tile's stage 1).  Wave w uses that slice as its private scratch and issues those 8 pieces when it is done.
  static constexpr int OFF_SCR = C::STAGE_BYTES;              // buffer 1, A area
  static constexpr int SCR_PER_WAVE = 8192;
  static constexpr int LDS_BYTES = 2 * C::STAGE_BYTES;
  static_assert(C::NA * 1024 == SCR_PER_WAVE, "a wave's A pieces of one stage are its scratch");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// post-hand-off MFMA index at which accumulator-tile PAIR P (tiles 2P, 2P + 1 in (m, n) row-major order) of the last stage is
// final: e(P) = 8 P + 3; inverse (-1: none)
constexpr int deepp_pair_done_at(int s) { return (s >= 3 && s <= 59 && (s - 3) % 8 == 0) ? (s - 3) / 8 : -1; }

// [r5] MFMA order of the last stage (final_stage16).  Rounds 2-4 ran it tile by tile: the four k-slices of an accumulator tile back to back, with the stage's other
// instructions threaded between them.  That is the one thing the matrix pipe does badly: an MFMA whose accumulator is the result of the MFMA right before it is fed
// from inside the pipe ONLY when it follows immediately; with anything in between it waits for the write-back -- ~110-130 cycles per MFMA instead of 32, whatever the
// retirement looked like (profiles/final_stage_contention_r4.txt and ..._r5a_*: ~8 400-9 000 cycles per stage with the fp32 retirement AND with the bf16-first one).
// Here the tiles go in BLOCKS of IL (row-major order), k-slice-major inside a block, so two MFMAs on one accumulator are IL slots apart -- the K loop's order on a
// smaller window.  Slices 0 and 1 of block 0 run before the hand-off (their fragments are in registers; they cover the LDS latency of slices 2, 3).
//   NS post-hand-off slot s -> k-slice fs_j, tile fs_T;  fs_F(T) = the slot of tile T's last MFMA
#ifndef QAMD_DEEPP_FS_IL
#define QAMD_DEEPP_FS_IL 4
#endif
// (lab, timing only -- the output is wrong: what the parts of the last stage cost, read off the stage trace of tools/final_stage_contention.py.
//  bit 0: no retirement at all; bit 1: a piece is its 4 accumulator reads only; bit 2: a piece without its ds_write_b64; bit 3: no read-back / stores;
//  bit 4: no next-tile DMA / fragment / scale reads inside the stage)
#ifndef QAMD_FS_ABL
#define QAMD_FS_ABL 0
#endif
#ifndef QAMD_FS_BURST
#define QAMD_FS_BURST 0
#endif
constexpr int fs_T(int IL, int s) { return s < 2 * IL ? s % IL : IL * (1 + (s - 2 * IL) / (4 * IL)) + (s - 2 * IL) % IL; }
constexpr int fs_j(int IL, int s) { return s < 2 * IL ? 2 + s / IL : ((s - 2 * IL) % (4 * IL)) / IL; }
constexpr int fs_F(int IL, int T) { return 4 * IL * (T / IL) + IL + T % IL; }
// the MXFP8 twin: two k-slices per tile, slice 0 of block 0 before the hand-off
constexpr int fs8_T(int IL, int s) { return s < IL ? s : IL * (1 + (s - IL) / (2 * IL)) + (s - IL) % IL; }
constexpr int fs8_j(int IL, int s) { return s < IL ? 1 : ((s - IL) % (2 * IL)) / IL; }
constexpr int fs8_F(int IL, int T) { return 2 * IL * (T / IL) + T % IL; }


// TRACE (lab build only): workgroup 0, wave 0 writes {shader cycles, 100 MHz wall ticks} pairs to p.dbg at: kernel entry,
// first stage landed, entry of the last stage of every tile, end of that stage, kernel exit (after the last store ack).
// ST_AUX: cache-policy bits of the output stores (buffer_store aux: 1 = sc0, 2 = nt, 16 = sc1; lab variants only, product = 0).
// bid / G: this workgroup's id among the G persistent workgroups; ntiles: the tiles they walk (tiles 0 .. ntiles-1 of the grouped
// raster).  A plain launch passes blockIdx.x / gridDim.x / all tiles; the heterogeneous launch (gemm_mx_hetero_kernel, end of this
// file) gives the persistent workgroups the full rounds and runs the residual tiles as 128x128 tiles on other workgroups.
// LAB (lab build only; 0 in the product): timing experiments that change the RESULT and exist to price an idea before it is built --
//   bit 0: no alpha multiply in the retirement (an alpha == 1 specialisation would save 4 v_pk_mul_f32 per store)
//   bit 2: the last stage retires only HALF of the tile (the pairs of m = 2, 3); the other accumulators are kept alive but never stored:
//          prices the store burst itself (time of the kernel with 16 MiB instead of 32 MiB of output at 4096^3)
//   bit 1: "spread" ablation (needs two copies of the stage code and spills -- 1.5 KB of scratch, 183 us: not usable, kept for the record): half of the tile's output stores (the pairs of m = 0, 1, zero data) are issued one stage EARLY, threaded
//          through the second half of stage KTe-2, and the last stage retires only the other half -- what a two-stage
//          accumulator-stationary window could gain at best by overlapping the 32 MiB store burst with more MFMA work
// SK ([r4], streamk.hip.h): the workgroup walks UNITS -- whole tiles, then its range of the stream-K region: at most one tile whose LAST K stages it
// computes first and parks (raw fp32 accumulators, accumulator layout, write-through stores + a tagged flag), whole tiles, and at most one tile
// whose FIRST K stages it computes last, with the accumulators initialised from the part its neighbour parked.  Ranges start and end on even
// stages, so a unit starts in LDS buffer 0 like a tile and the stage code is shared; the DMA stream runs on across unit boundaries exactly as
// across tiles.  Every output element is produced by one fixed summation order: deterministic; bit-identical to the single pass wherever the
// partial sums are exact (the reference's tests), one fp32 rounding apart otherwise.
#ifndef QAMD_DEEPP_SPLITB
#define QAMD_DEEPP_SPLITB 0
#endif
#ifndef QAMD_DEEPP_SOFF
#define QAMD_DEEPP_SOFF 1
#endif
#ifndef QAMD_KERNARG_EARLY
#define QAMD_KERNARG_EARLY 1
#endif
#ifndef QAMD_DEEPP_RB2
#define QAMD_DEEPP_RB2 0
#endif
#ifndef QAMD_DEEPP_PEEL
#define QAMD_DEEPP_PEEL 1
#endif
// [r5] retirement of the last stage: 0 = fp32 through the scratch (rounds 2-4), 1 = bf16 BEFORE the transposition (retire16 below),
// 2 = 1 + an alpha == 1 arm without the multiply (x * 1.0f == x: same bytes; spills -- kept for the record), 3 = 1 with v_pk_mul_f32 (lab A/B)
#ifndef QAMD_DEEPP_RETIRE
#define QAMD_DEEPP_RETIRE 0
#endif
#ifndef QAMD_DEEPP8_RETIRE   // the same for the MXFP8 twin (gemm_mx_deepp8): 0 = fp32 through the scratch, 1 = bf16 first
#define QAMD_DEEPP8_RETIRE 0
#endif
template <class C, bool TRACE = false, int ST_AUX = 0, int LAB = 0, bool SK = false, int DMA_SPREAD = 1>
__device__ __forceinline__ void gemm_mx_deepp(char* smem, const GemmParams& p, const int bid, const int G, const int ntiles) {
  static_assert(C::EBITS == 4 && C::BM == 256 && C::BN == 256 && C::WAVES_M == 2 && C::WAVES_N == 2 && C::NSTAGE == 2 && C::PPW == 2,
                "persistent deep schedule: fp4, 256x256 tiles, 4 waves of 128x128");
  constexpr int MT = 4, NT = 4;
  constexpr int STAGE = C::STAGE_BYTES, OFF_SCR = DeepPCfg<C>::OFF_SCR;
  GemmCtx<C> cx(smem, p);   // per-lane offsets / LDS addresses; its tile coordinates and descriptors are NOT used here
  const int lane = cx.lane, wave = cx.wave, i32 = cx.i32, g = cx.g;
  const int KT = cx.KT, KTe = (KT + 1) & ~1, CB = cx.CB, rowbytes = cx.rowbytes;
  const int wg = xcd_remap(bid, G);

  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  int trace_n = 0, ho_n = 0;
  auto trace = [&]() __attribute__((always_inline)) {
    if constexpr (TRACE) {
      if (blockIdx.x == 0 && wave == 0 && p.dbg && trace_n < 30) {
        const uint32_t c = (uint32_t)__builtin_readcyclecounter(), r = (uint32_t)__builtin_amdgcn_s_memrealtime();
        if (lane == 0) { p.dbg[2 + 2 * trace_n] = c; p.dbg[3 + 2 * trace_n] = r; }
      }
      ++trace_n;
    }
  };
  // (TRACE only) marks INSIDE the final stage: shader cycles at slot 0, 8, 16 ... 64 and behind the last slot of the first four final stages workgroup 0 runs,
  // dbg[1024 + 16 f + k] -- is the stage's time spread evenly (a throughput bound: LDS bytes, instruction issue) or does it pile up somewhere (tools/final_stage_contention.py)
  int fs_n = 0;
  auto trace_fs = [&](const int k) __attribute__((always_inline)) {
    if constexpr (TRACE) {
      if (blockIdx.x == 0 && wave == 0 && p.dbg && fs_n < 4) {
        const uint32_t c = (uint32_t)__builtin_readcyclecounter();
        if (lane == 0) p.dbg[1024 + 16 * fs_n + k] = c;
      }
    }
  };
  trace();
  uint32_t wg_t0 = 0;
  if constexpr (TRACE) wg_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();

  // ---- tile -> (m0, n0): rounds of G tiles, each round XCD-contiguous, grouped raster of 4 tile rows --------------------
  auto decode = [&](int t, int& m0, int& n0) __attribute__((always_inline)) {
    int tm, tn;
    raster_decode(t, p.tiles_m, p.tiles_n, p.raster_magic, tm, tn);
    m0 = uniform(tm * C::BM);
    n0 = uniform(tn * C::BN);
  };
  // SK: the stream-K form keeps THREE descriptors for the whole kernel (all of A, all of B, this wave's scale operand) and carries a tile as three
  // scalar byte offsets that ride in the soffset operand of the DMA (the range check of a raw buffer covers voffset + soffset on gfx950,
  // profiles/native_r2_soffset_probe.txt; "no tile" = offsets of 2^31).  36 scalar registers of per-tile descriptors (this tile, the next one, the
  // selected one) become 9: the unit walk's own scalars would otherwise push the kernel's scalar spills into a second VGPR -- and at 255 of 256
  // vector registers that one register spills a hundred others into the hand-scheduled stages.
  struct Desc { __amdgpu_buffer_rsrc_t a, b, s; uint32_t ao, bo, so; };
  const __amdgpu_buffer_rsrc_t skA = make_rsrc(p.A, SK ? p.a_bytes : 0u), skB = make_rsrc(p.B, SK ? p.b_bytes : 0u);
  const __amdgpu_buffer_rsrc_t skS = cx.sIsB ? make_rsrc(p.SFB, SK ? p.sfb_bytes : 0u) : make_rsrc(p.SFA, SK ? p.sfa_bytes : 0u);
  // operand descriptors of tile t (t >= ntiles: empty descriptors -> every DMA of that "tile" loads zeros)
  auto make_desc = [&](int t) __attribute__((always_inline)) {
    const bool valid = t < ntiles;
    int m0, n0;
    decode(valid ? t : ntiles - 1, m0, n0);
    const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
    const uint32_t sa_off = (uint32_t)(m0 >> 7) * CB * 512, sb_off = (uint32_t)(n0 >> 7) * CB * 512;
    Desc d;
    if constexpr (SK) {
      d.ao = valid ? a_off : 0x80000000u;
      d.bo = valid ? b_off : 0x80000000u;
      d.so = valid ? (cx.sIsB ? sb_off : sa_off) : 0x80000000u;
    } else {
      d.a = make_rsrc(p.A + a_off, valid ? p.a_bytes - a_off : 0u);
      d.b = make_rsrc(p.B + b_off, valid ? p.b_bytes - b_off : 0u);
      d.s = cx.sIsB ? make_rsrc(p.SFB + sb_off, valid ? p.sfb_bytes - sb_off : 0u) : make_rsrc(p.SFA + sa_off, valid ? p.sfa_bytes - sa_off : 0u);
    }
    return d;
  };

  // ---- registers -------------------------------------------------------------------------------------------------------
  v16f acc[MT][NT];
  v4i fa[4][MT] = {}, fb[4][NT] = {};
  int sa[2][MT], sb[2][NT];
  // output addressing of the current tile, mirrored for the spread ablation (set_out_tile keeps both in step)
  __amdgpu_buffer_rsrc_t est_rD = make_rsrc(p.D, 0);
  int est_lane = 0, est_colLim = 0;

  auto read_slice = [&](const int buf, const int j) __attribute__((always_inline)) {
    const char* st = smem + buf * STAGE;
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[j][t] = *(const v4i*)(st + cx.rdA[j] + t * 32 * C::ROWB);
#pragma unroll
    for (int t = 0; t < NT; ++t) fb[j][t] = *(const v4i*)(st + cx.rdBd + cx.rdA[j] + t * 32 * C::ROWB);
  };
  auto read_fa = [&](const int buf, const int j, const int t) __attribute__((always_inline)) {
    fa[j][t] = *(const v4i*)(smem + buf * STAGE + cx.rdA[j] + t * 32 * C::ROWB);
  };
  auto read_fb = [&](const int buf, const int j, const int t) __attribute__((always_inline)) {
    fb[j][t] = *(const v4i*)(smem + buf * STAGE + cx.rdBd + cx.rdA[j] + t * 32 * C::ROWB);
  };
  auto read_scales = [&](const int buf, const int set) __attribute__((always_inline)) {
    const char* st = smem + buf * STAGE;
#pragma unroll
    for (int t = 0; t < MT; ++t) sa[set][t] = *(const int*)(st + cx.rdSA[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) sb[set][t] = *(const int*)(st + cx.rdSB[t]);
  };
  // one scaled FP4 MFMA: acc[m][n] (+)= B-fragment n x A-fragment m of k-slice j (op_sel byte j of the scale dwords of set sset)
  auto mfma1 = [&](const int j, const int sset, const int m, const int n, const bool zero_c) __attribute__((always_inline)) {
    const v4i a = fa[j][m], b = fb[j][n];
    const v8i A8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, B8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    v16f c = acc[m][n];
    if (zero_c) c = v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j == 0) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, c, 4, 4, 0, sb[sset][n], 0, sa[sset][m]);
    if (j == 1) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, c, 4, 4, 1, sb[sset][n], 1, sa[sset][m]);
    if (j == 2) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, c, 4, 4, 2, sb[sset][n], 2, sa[sset][m]);
    if (j == 3) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, c, 4, 4, 3, sb[sset][n], 3, sa[sset][m]);
  };
  auto mfma_all = [&](const int j, const int sset, const bool zero_c) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) mfma1(j, sset, m, n, zero_c);
  };

  // ---- LDS-DMA of one stage, one instruction at a time (item 0..7: A pieces, 8..15: B pieces, 16: the scale piece) ------
  // vb[par]: per-lane source offset of an even / odd piece for THIS stage (K-tail flavour, out-of-range when the stage
  // does not exist), computed once per stage by dma_prep -- opaque to the optimiser so the selects stay arithmetic.
  int vb0 = 0, vb1 = 0, vbS = 0;
  auto dma_prep = [&](int kt, bool valid) __attribute__((always_inline)) {
    if constexpr (SK) {   // stream-K form: K is a whole, even number of stages (capi.hip) -- no tail flavour, no padding stage; "no tile" lives in the scalar offsets
      vb0 = cx.voffAB[0]; vb1 = cx.voffAB[1]; vbS = cx.voffS;
      asm volatile("" : "+v"(vb0), "+v"(vb1), "+v"(vbS));   // (opaque per stage: as loop invariants the 16 sums vb + q * rstep of dma_item would each claim a register)
      return;
    }
    int lastmask = (kt == KT - 1) ? -1 : 0;
    int oobm = (valid && kt < KT) ? 0 : -1;
    int oobs = (valid && kt * C::SCT + cx.colS < CB) ? 0 : -1;
    asm volatile("" : "+v"(lastmask), "+v"(oobm), "+v"(oobs));
    // out of range = 0x80000000: stays >= any descriptor range (< 2^31) after q * rstep (< 2^31) is added, never wraps
    vb0 = (((cx.voffT[0] & lastmask) | (cx.voffAB[0] & ~lastmask)) & ~oobm) | ((int)0x80000000 & oobm);
    vb1 = (((cx.voffT[1] & lastmask) | (cx.voffAB[1] & ~lastmask)) & ~oobm) | ((int)0x80000000 & oobm);
    vbS = (cx.voffS & ~oobs) | ((int)0x80000000 & oobs);
  };
  auto dma_item = [&](const Desc& d, int kt, const int buf, const int item) __attribute__((always_inline)) {
    char* st = smem + buf * STAGE;
    if (item < 16) {
      const int t = item & 7, q = wave * 8 + t;
      const int v = ((t & 1) ? vb1 : vb0) + q * cx.rstep;
      if constexpr (SK)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(item < 8 ? skA : skB, (lds_ptr_t)(st + (item < 8 ? 0 : C::OFF_B) + q * 1024), 16, v, (int)((item < 8 ? d.ao : d.bo) + (uint32_t)(kt * C::ROWB)), 0, QAMD_DMA_AUX);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(item < 8 ? d.a : d.b, (lds_ptr_t)(st + (item < 8 ? 0 : C::OFF_B) + q * 1024), 16, v, kt * C::ROWB, 0, QAMD_DMA_AUX);
    } else {
      if constexpr (SK) __builtin_amdgcn_raw_ptr_buffer_load_lds(skS, (lds_ptr_t)(st + C::OFF_S + wave * 1024), 16, vbS, (int)(d.so + (uint32_t)(kt * C::SCT * 512)), 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(d.s, (lds_ptr_t)(st + C::OFF_S + wave * 1024), 16, vbS, kt * C::SCT * 512, 0, 0);
    }
  };
  auto dma_stage = [&](const Desc& d, int kt, bool valid, const int buf) __attribute__((always_inline)) {
    dma_prep(kt, valid);
#pragma unroll
    for (int i = 0; i < 17; ++i) dma_item(d, kt, buf, i);
  };

  // The first stage of a tile sits outside the K loop; without a use of its results in its own block LLVM's MachineSink
  // moves all 64 MFMAs behind the DMA issue, into the loop pre-header.  pin_acc() "uses" the accumulators in place.
  auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" : "+a"(acc[m][n]));
  };

  // ---- one K stage (not the last of its tile).  Entry: fragment sets 0, 1 and scale set BUF hold slices 0, 1 of this
  //      stage; exit: the same for the next stage (other buffer).  The DMA threaded through M(2) is stage (d, ktl).
  // (defined below; the spread ablation calls it from the stage before the last)
  auto early_store = [&](const int m, const int h, const int pass) __attribute__((always_inline)) {
    const int off = est_lane + ((32 * m + 8 * pass) * p.ldd + 64 * h) * 2;
    __builtin_amdgcn_raw_buffer_store_b128(v4u{0u, 0u, 0u, 0u}, est_rD, (64 * h < est_colLim) ? off : (int)0x80000000, 0, ST_AUX);
  };
  auto stage = [&](auto bufc, auto firstc, const Desc& d, int ktl, bool dvalid, auto earlyc) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    constexpr bool EARLY = decltype(earlyc)::value;
    // [r3] ONE fragment read behind each MFMA instead of a burst of 8 in front of 16 MFMAs.  With one wave per SIMD the wave's own program order
    // is all that can put a read into an MFMA's shadow: after a run of MFMAs the matrix pipe drains while the 8 reads issue -- 82 cycles per 8
    // MFMAs (tests/native/ubench.hip "uinter": 8 MFMA + 6 reads in bursts 350 cycles, interleaved 276, MFMAs alone 268; on quantised-Gaussian
    // operands, where the clock is held back electrically, 187.5 -> 175.9 ns, and with the LDS-DMA in the mix 200.8 -> 189.3 ns = -6 %;
    // 200.8 ns x 8 = the 1.62 us this stage took).  QAMD_DEEPP_BURST restores the round-2 order (A/B).
    // (the slice's two base addresses are made opaque once: folded into every read, buffer offset + row-set offset exceed the 16-bit DS offset
    //  field and cost a v_add per read -- and, at 256 + 241 registers, spills)
    typedef __attribute__((address_space(3))) const v4i* lds_v4i_t;
    uint32_t rbA = 0, rbB = 0;   // 32-bit LDS addresses
    auto read_base = [&](const int buf, const int j) __attribute__((always_inline)) {
      rbA = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + buf * STAGE + cx.rdA[j]);
      rbB = rbA + (uint32_t)cx.rdBd;
      asm volatile("" : "+v"(rbA), "+v"(rbB));
    };
    auto read_frag = [&](const int j, const int i) __attribute__((always_inline)) {
      if (i < MT) fa[j][i] = *(lds_v4i_t)(uintptr_t)(rbA + (uint32_t)(i * 32 * C::ROWB));
      else fb[j][i - MT] = *(lds_v4i_t)(uintptr_t)(rbB + (uint32_t)((i - MT) * 32 * C::ROWB));
    };
    auto read_scale1 = [&](const int buf, const int set, const int i) __attribute__((always_inline)) {
      const char* st = smem + buf * STAGE;
      if (i < MT) sa[set][i] = *(const int*)(st + cx.rdSA[i]);
      else sb[set][i - MT] = *(const int*)(st + cx.rdSB[i - MT]);
    };
#ifndef QAMD_DEEPP_EARLYPREP
#define QAMD_DEEPP_EARLYPREP 1
#endif
#ifdef QAMD_DEEPP_BURST
    constexpr bool IL = false;
#else
    constexpr bool IL = true;
#endif
    static_assert(MT + NT <= MT * NT / 2, "a fragment read behind each of the first MT + NT MFMAs, a scale read behind each of the next");
    // group G: the MT x NT MFMAs of k-slice js, each followed by `extra(i)`
    auto group = [&](const int js, const bool zero_c, auto extra) __attribute__((always_inline)) {
      int i = 0;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          mfma1(js, BUF, m, n, zero_c);
          extra(i);
          fence();
          ++i;
        }
    };
#if QAMD_DEEPP_SPLITB   // [r5] two hand-offs per K stage (measured: 1 - 3 % slower, profiles/ab_lib_r5r_two_handoffs_per_stage_negative.txt)
    // [r5] TWO hand-offs per stage instead of one.  The single barrier of rounds 2-4 (behind group 1) served two purposes: "everybody is done READING this buffer,
    // its refill may start" and "everybody's pieces of the NEXT stage have landed, its fragments may be read".  Tied together, the refill of buffer BUF was issued in
    // slots 40 .. 63 and had to be in LDS by slot 32 of the next stage: 33 .. 56 slots = 1.2 - 2.1 kcycles of lead, and with the whole chip streaming that is not enough --
    // every wave sat ~300 - 440 cycles in `s_waitcnt vmcnt(0)` per stage (tools/handoff_trace.py, profiles/handoff_trace_r5q.txt: 8 cycles with 8 workgroups).
    //   * both remaining slices (2, 3) are read behind group 0 (their registers are free: slice 3's last use was group 3 of the previous stage);
    //   * hand-off A (slot 20): own reads back, barrier -> the refill starts NOW: 12 pieces behind group 1's MFMAs, 5 behind the first of group 2;
    //   * hand-off B (slot 32): `s_waitcnt vmcnt(12)` -- everything but the 12 pieces just issued, i.e. all of the next stage (loads return in order) -- and barrier.
    // Lead of a piece: 60 .. 76 slots.  Same products in the same order: bit-identical.
    read_base(BUF, 2);
    group(0, FIRST, [&](const int i) __attribute__((always_inline)) {
      if (i < MT + NT) read_frag(2, i);
      if (i == MT + NT - 1) read_base(BUF, 3);
      if (i >= MT + NT) read_frag(3, i - (MT + NT));
    });
    group(1, false, [&](const int i) __attribute__((always_inline)) {
      if (i == 0) dma_prep(ktl, dvalid);
      if (i == 2) read_base(BUF ^ 1, 0);
      if (i == 3) {   // hand-off A
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (i >= 4) dma_item(d, ktl, BUF, i - 4);
    });
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // hand-off B: own pieces of the next stage landed (the 12 just issued stay in flight)
    __builtin_amdgcn_s_barrier();
    fence();
    group(2, false, [&](const int i) __attribute__((always_inline)) {
      if (i < 5) dma_item(d, ktl, BUF, 12 + i);
      // the next stage's slice 0 (set 0 went dead with group 0) and its scale dwords
      if (i < MT + NT) read_frag(0, i);
      else if (i < 2 * (MT + NT)) read_scale1(BUF ^ 1, BUF ^ 1, i - (MT + NT));
    });
    read_base(BUF ^ 1, 1);
    group(3, false, [&](const int i) __attribute__((always_inline)) {
      if (i < MT + NT) read_frag(1, i);
    });
#else
    if (!IL) { read_slice(BUF, 2); fence(); } else read_base(BUF, 2);
    group(0, FIRST, [&](const int i) __attribute__((always_inline)) { if (IL && i < MT + NT) read_frag(2, i); });
    if (!IL) { read_slice(BUF, 3); fence(); } else read_base(BUF, 3);
    // [r4] EARLYPREP: the stage's DMA offsets and the next slice's base addresses are computed in the shadow of group 1's last MFMAs instead of between the
    // barrier and group 2's first MFMA (neither depends on the hand-off)
    group(1, false, [&](const int i) __attribute__((always_inline)) {
      if (IL && i < MT + NT) read_frag(3, i);
      if (QAMD_DEEPP_EARLYPREP && IL && i == 12) dma_prep(ktl, dvalid);
      if (QAMD_DEEPP_EARLYPREP && IL && i == 14) read_base(BUF ^ 1, 0);
    });
    // (TRACE, [r5]: what does the hand-off cost?  Shader clock before the two waits and behind each: dbg[3072 + 4 k] = before, + 1 = LDS reads back (lgkmcnt),
    //  + 2 = own DMA landed (vmcnt), + 3 = behind the barrier; workgroup 0, every wave: dbg[3072 + 1024 wave + ...], the first 60 hand-offs.  s_memtime answers through
    //  lgkmcnt, so the first mark is asked for early and costs nothing extra; the later ones add a few cycles each.)
    if constexpr (TRACE) {
      if (blockIdx.x == 0 && p.dbg && ho_n < 60) {
        const uint32_t t0 = (uint32_t)__builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t t1 = (uint32_t)__builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t2 = (uint32_t)__builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        const uint32_t t3 = (uint32_t)__builtin_readcyclecounter();
        if (lane == 0) { uint32_t* q = p.dbg + 3072 + 1024 * wave + 4 * ho_n; q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3; }
        ++ho_n;
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMA of the next stage landed; own reads of this buffer done
    __builtin_amdgcn_s_barrier();
    }
    fence();
    if (!IL) {
      read_scales(BUF ^ 1, BUF ^ 1);
      read_slice(BUF ^ 1, 0);
    }
    if (!(QAMD_DEEPP_EARLYPREP && IL)) {
      dma_prep(ktl, dvalid);
      if (IL) read_base(BUF ^ 1, 0);
      fence();
    }
    // [r4] the stage's 17 DMA items ride in the slots that have no fragment read: second half of group 2 (items 0 .. 7 + the scale piece) and second half
    // of group 3 (items 8 .. 15) -- all 17 behind the first 16 MFMAs after the hand-off put three auxiliary instructions into each of eight slots
    group(2, false, [&](const int i) __attribute__((always_inline)) {
      if (!DMA_SPREAD) { dma_item(d, ktl, BUF, i); if (i == 0) dma_item(d, ktl, BUF, 16); }
      else if (i >= 8) { dma_item(d, ktl, BUF, i - 8); if (i == 8) dma_item(d, ktl, BUF, 16); }
      if (IL) {   // the next stage's slice 0 (set 0 went dead with group 0) and its scale dwords
        if (i < MT + NT) read_frag(0, i);
        else if (i < 2 * (MT + NT)) read_scale1(BUF ^ 1, BUF ^ 1, i - (MT + NT));
      }
    });
    if (!IL) { read_slice(BUF ^ 1, 1); fence(); } else read_base(BUF ^ 1, 1);
    group(3, false, [&](const int i) __attribute__((always_inline)) {
      if (IL && i < MT + NT) read_frag(1, i);
      if (DMA_SPREAD == 2 && i == 8) dma_prep(ktl, dvalid);   // (2: the piece offsets are recomputed here instead of staying live through this group's fragment reads -- one register less,
                                                               //  which the heterogeneous kernel needs; costs the plain kernel ~1 %, profiles/ab_lib_gemm_r4at_reprep.txt)
      if (DMA_SPREAD && i >= 8) dma_item(d, ktl, BUF, i);
      if constexpr (EARLY) early_store(i / 8, (i / 4) % 2, i % 4);   // pairs (m = 0, 1) x (h = 0, 1), four passes each
    });
#endif
    if constexpr (FIRST) pin_acc();
  };

  // ---- epilogue pieces: a retired PAIR of accumulator tiles (m, 2h), (m, 2h + 1) = 32 rows x 64 columns through the
  //      wave-private scratch (fp32, 32 rows x 256 B; 16-byte chunk c = 8 n' + 2 q + g of row r stored at chunk
  //      (c & 8) | ((c & 7) ^ (r & 7)): the 8 lanes of a ds_write_b128 group hit 8 different chunks, the 16 lanes of a
  //      ds_read_b128 group 16 different ones).  Read-back is row-major: lane -> row 8 p + l / 8, columns 8 (l % 8) .. + 7, so
  //      8 lanes cover one whole 128-byte line of D and a wave instruction stores 8 rows x 128 B.
  char* scr = smem + OFF_SCR + wave * DeepPCfg<C>::SCR_PER_WAVE;
  const int scrW = i32 * 256 + ((((i32 & 6) << 4)) | ((g ^ (i32 & 1)) << 4));   // chunk (2q + g) ^ (row & 7) = this ^ (q << 5); + 128 n'
  const int rrl = lane >> 3, ccl = lane & 7;                                     // read-back: row rrl (+ 8 per pass), columns 8 ccl .. + 7
  const int scrR = rrl * 256 + (ccl >> 2) * 128 + ((((2 * ccl) & 7) ^ (rrl & 7)) << 4);   // chunk 2 ccl of that row; chunk 2 ccl + 1 = this ^ 16
  const float alpha = *p.alpha;
  __amdgpu_buffer_rsrc_t rD = make_rsrc(p.D, 0);
  __amdgpu_buffer_rsrc_t rP = make_rsrc(p.D, 0);          // SK: scratch slot of a parking unit (empty otherwise)
  const int pkLane = SK ? wave * 65536 + lane * 32 : 0;   // SK: this lane's 32 bytes of a parked pass
  int stLane = 0, colLim = 0;
  // output descriptor of the tile at (m0, n0): base = its first element, range = what is left of D from there (capped at
  // 2 GiB: a tile spans < 2^31 bytes, launch code rejects wider rows), so rows >= M fall out of range by themselves;
  // columns >= N are pushed out of range per lane.
  auto set_out_tile = [&](int m0, int n0) __attribute__((always_inline)) {
    const int64_t left = ((int64_t)(p.M - m0) * p.ldd - n0) * 2;
    rD = make_rsrc(p.D + ((int64_t)m0 * p.ldd + n0), (uint32_t)(left > 0x7fffffffll ? 0x7fffffffll : left));
    stLane = ((cx.wave_m * C::WTM + rrl) * p.ldd + cx.wave_n * C::WTN + 8 * ccl) * 2;
    colLim = p.N - n0 - cx.wave_n * C::WTN - 8 * ccl;   // column 64 h + 8 ccl of the wave tile exists iff 64 h < colLim
    if constexpr (LAB & 2) { est_rD = rD; est_lane = stLane; est_colLim = colLim; }
  };
  auto retire_write = [&](const int m, const int h) __attribute__((always_inline)) {
    if constexpr (QAMD_DEEPP_RB2 == 2 && !SK) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");   // (RB2 = 2: never more than 15 LDS operations outstanding -- the lgkmcnt field has 4 bits)
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(v4f*)(scr + (scrW ^ (q << 5)) + nn * 128) =
            v4f{acc[m][2 * h + nn][4 * q + 0], acc[m][2 * h + nn][4 * q + 1], acc[m][2 * h + nn][4 * q + 2], acc[m][2 * h + nn][4 * q + 3]};
  };
  // RB2 ([r4], behind QAMD_DEEPP_RB2 = 1 / 2 / 3, OFF: DO NOT ENABLE AS IT IS): a read-back register set per HALF of a pair (32 registers instead of 16; they exist since the
  // store offsets became scalar), so that a half is read SIX slots ahead of its stores instead of two.  Why: the stage trace (profiles/final_stage_contention_r4.txt)
  // puts the final stage at ~8 400 cycles on an idle chip, and the stores of a half wait for an LDS round trip that was issued 64 cycles earlier.  The one run it got
  // (the last 2.6 GPU seconds of round 4, profiles/ab_lib_rb2_r4bj.txt): 1 % SLOWER and the output differs from the product's -- the ISA reads correct (order, registers
  // and s_waitcnt values checked by hand for the first and the last pairs; the read -> store data flow of the two ISAs is identical).  Found afterwards, on the CPU:
  // RB2 = 1 has a VALU write of a store's data registers DIRECTLY behind the (scalar-offset, 16-byte) store -- a hazard the compiler only guards for stores without a
  // scalar offset; the product never gets closer than one instruction in between (tools/store_data_hazard.py, a CPU test now).  RB2 = 3 = the same schedule with
  // `s_nop 1` behind every store: the variant to try first with a GPU in hand.  (RB2 = 2: explicit lgkmcnt waits instead -- the static count of outstanding LDS
  // operations reaches 24 here and the field has 4 bits, but the validated K loop reaches 22 by the same count, tools/lgkm_pressure.py.)
  constexpr bool RB2 = QAMD_DEEPP_RB2 && !SK;
  v4f rb[RB2 ? 4 : 2][2];
  auto retire_read = [&](const int half) __attribute__((always_inline)) {   // rows 16 half .. + 15: passes 2 half, 2 half + 1
    if constexpr (QAMD_DEEPP_RB2 == 2 && !SK) asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory");
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      rb[RB2 ? 2 * half + ps : ps][0] = *(const v4f*)(scr + scrR + (2 * half + ps) * 2048);
      rb[RB2 ? 2 * half + ps : ps][1] = *(const v4f*)(scr + (scrR ^ 16) + (2 * half + ps) * 2048);
    }
  };
  auto retire_store = [&](const int m, const int h, const int pass) __attribute__((always_inline)) {
    const v4f lo = rb[RB2 ? pass : pass & 1][0], hi = rb[RB2 ? pass : pass & 1][1];
    v4i o;
    if constexpr (LAB & 1) {
      o[0] = (int)pack_bf16x2(lo[0], lo[1]); o[1] = (int)pack_bf16x2(lo[2], lo[3]); o[2] = (int)pack_bf16x2(hi[0], hi[1]); o[3] = (int)pack_bf16x2(hi[2], hi[3]);
    } else {
    o[0] = (int)pack_bf16x2(lo[0] * alpha, lo[1] * alpha);
    o[1] = (int)pack_bf16x2(lo[2] * alpha, lo[3] * alpha);
    o[2] = (int)pack_bf16x2(hi[0] * alpha, hi[1] * alpha);
    o[3] = (int)pack_bf16x2(hi[2] * alpha, hi[3] * alpha);
    }
    if constexpr (SK || QAMD_DEEPP_SOFF) {
      // (the wave-uniform part of the address rides in the scalar offset, recomputed per store -- as vector offsets the 32 sums stLane + k ldd are
      //  precomputed per tile and stay live across the K loop: ten spilled registers in the stream-K form, and in the plain kernel the 20 registers
      //  whose absence kept the scalar argument loads from being requested in one round, QAMD_KERNARG_EARLY; [r4] same speed by itself, profiles/ab_lib_gemm_r4bh_*)
      int ldd2 = p.ldd * 2;
      asm volatile("" : "+s"(ldd2));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), rD, (64 * h < colLim) ? stLane : (int)0x80000000, (32 * m + 8 * pass) * ldd2 + 128 * h, ST_AUX);
      // (RB2 = 3: two wait states behind the store before anything may overwrite its data registers -- the compiler guards that hazard only for stores WITHOUT a scalar
      //  offset, and RB2 = 1 had a packed multiply of the next pass directly behind the store: tools/store_data_hazard.py, the likely cause of its wrong output)
      if constexpr (QAMD_DEEPP_RB2 == 3 && !SK) asm volatile("s_nop 1" ::: "memory");
    } else {
    const int off = stLane + ((32 * m + 8 * pass) * p.ldd + 64 * h) * 2;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), rD, (64 * h < colLim) ? off : (int)0x80000000, 0, ST_AUX);
    }
    if constexpr (SK) {
      // stream-K: the same read-back registers, raw, to the scratch slot of a unit that PARKS its sums (rP is empty for every other unit: the stores
      // are dropped by the range check; a parking unit has an empty rD instead).  Row-major pairs: ((wave 8 + pair) 4 + pass) 2 KiB + lane 32 B.
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, lo), rP, pkLane, (((2 * m + h) * 4 + pass) * 2048), 17);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, hi), rP, pkLane, (((2 * m + h) * 4 + pass) * 2048) + 16, 17);
    }
  };

  // ---- the LAST stage of a tile (buffer 1), accumulator-stationary, with the tile's epilogue, the DMA of the next tile's
  //      stage 1 (d) and the fragment / scale reads of the next tile's stage 0 (buffer 0, scale set 0) threaded through.
  // MFMA sequence after the hand-off: tiles T = 0..15 in (m, n) row-major order, T0 and T1 with slices 2, 3 only (their
  // slices 0, 1 ran before the hand-off to cover the latency of R(2), R(3)), every later tile with slices 0..3.  Pair
  // P = 0..7 (tiles 2P, 2P + 1) is final after post-hand-off MFMA e = 8 P + 3; its retirement in the shadows of later MFMAs:
  //   write e+1 | read rows 0-15 e+3 | stores e+5, e+6 | read rows 16-31 e+7 | stores e+9, e+10
  // One scratch and one read-back register set per wave: write(P + 1) at e + 9 follows read(P) at e + 7, read(P + 1) at
  // e + 11 follows the last store of P at e + 10 (within a slot: stores, then write, then read).
  // ktn: the stage of the next tile (unit) whose DMA is threaded through here -- its second one: 1, or kb' + 1 of a stream-K unit
  auto final_stage = [&](const Desc& d, bool dvalid, const int ktn) __attribute__((always_inline)) {
    read_slice(1, 2);
    read_slice(1, 3);
    fence();
    mfma1(0, 1, 0, 0, false); mfma1(1, 1, 0, 0, false); mfma1(0, 1, 0, 1, false); mfma1(1, 1, 0, 1, false);
    fence();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next tile's stage 0 landed (buffer 0); all reads of buffer 1 done
    __builtin_amdgcn_s_barrier();
    fence();
    dma_prep(ktn, dvalid);
    fence();
    static_for<0, RB2 ? 73 : 71>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s < 60) {
        constexpr int T = s < 2 ? 0 : s < 4 ? 1 : 2 + (s - 4) / 4;
        constexpr int j = s < 4 ? 2 + (s & 1) : (s - 4) % 4;
        mfma1(j, 1, T / 4, T % 4, false);
      }
      // DMA of the next tile's stage 1 into buffer 1: the B pieces and the scale piece now, one instruction every third
      // slot; the wave's 8 A pieces land in its own scratch area, so they wait until the last read-back (after the loop)
      if constexpr (s % 3 == 0 && s / 3 < 9) dma_item(d, ktn, 1, 8 + s / 3);
      if constexpr (s == 1) read_scales(0, 0);
      // fragments of the next tile's stage 0, as their registers die: A rows of m after tile (m, 3), B rows of n after (3, n)
      if constexpr (QAMD_DEEPP_RB2 == 2 && !SK && (s == 12 || s == 28 || s == 44 || s == 48 || s == 52 || s == 56 || s == 60)) asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory");
      if constexpr (s == 12 || s == 28 || s == 44) { read_fa(0, 0, (s - 12) / 16); read_fa(0, 1, (s - 12) / 16); }
      if constexpr (s == 48 || s == 52 || s == 56) { read_fb(0, 0, (s - 48) / 4); read_fb(0, 1, (s - 48) / 4); }
      if constexpr (s == 60) { read_fa(0, 0, 3); read_fa(0, 1, 3); read_fb(0, 0, 3); read_fb(0, 1, 3); }
      // retirement items due in this slot (pair P final at e = 8 P + 3)
      // pair P (final at e = 8 P + 3):   write e+1 | read rows 0-15 e+3 | stores e+5, e+6 | read rows 16-31 e+7 | stores e+9, e+10
      // RB2:                            write e+1 | read rows 0-15 e+3 | read rows 16-31 e+5 | stores e+9, e+10 (set A) | stores e+11, e+12 (set B);
      //                                 pair P + 1 reads into set A at e+11 (after P's stores from it, same slot order: stores first) and into set B at e+13
      constexpr int d1 = s - 1, d3 = s - 3, d5 = s - (RB2 ? 9 : 5), d6 = s - (RB2 ? 10 : 6), d7 = s - (RB2 ? 5 : 7), d9 = s - (RB2 ? 11 : 9), d10 = s - (RB2 ? 12 : 10);
      constexpr int PMIN = (LAB & 6) ? 4 : 0;   // ablations: the pairs of m = 0, 1 are not retired here (bit 1: "stored" one stage earlier; bit 2: never)
      if constexpr ((LAB & 4) && deepp_pair_done_at(d1) >= 0 && deepp_pair_done_at(d1) < 4) {   // ... but their accumulators (and MFMAs) stay alive
        asm volatile("" ::"a"(acc[deepp_pair_done_at(d1) / 2][2 * (deepp_pair_done_at(d1) % 2)]), "a"(acc[deepp_pair_done_at(d1) / 2][2 * (deepp_pair_done_at(d1) % 2) + 1]));
      }
      if constexpr (deepp_pair_done_at(d5) >= PMIN) retire_store(deepp_pair_done_at(d5) / 2, deepp_pair_done_at(d5) % 2, 0);
      if constexpr (deepp_pair_done_at(d6) >= PMIN) retire_store(deepp_pair_done_at(d6) / 2, deepp_pair_done_at(d6) % 2, 1);
      if constexpr (deepp_pair_done_at(d9) >= PMIN) retire_store(deepp_pair_done_at(d9) / 2, deepp_pair_done_at(d9) % 2, 2);
      if constexpr (deepp_pair_done_at(d10) >= PMIN) retire_store(deepp_pair_done_at(d10) / 2, deepp_pair_done_at(d10) % 2, 3);
      if constexpr (deepp_pair_done_at(d1) >= PMIN) retire_write(deepp_pair_done_at(d1) / 2, deepp_pair_done_at(d1) % 2);
      if constexpr (deepp_pair_done_at(d3) >= PMIN) retire_read(0);
      if constexpr (deepp_pair_done_at(d7) >= PMIN) retire_read(1);
      if constexpr (TRACE && s % 8 == 0 && s <= 64 + 8 * QAMD_FS_BURST * 6) trace_fs(s / 8);
#ifdef QAMD_FS_TRACE_SLOTS   // (lab) a mark behind EVERY slot of the first two last stages: dbg[2048 + 128 f + s]
      if constexpr (TRACE) {
        if (blockIdx.x == 0 && wave == 0 && p.dbg && fs_n < 2) {
          const uint32_t c = (uint32_t)__builtin_readcyclecounter();
          if (lane == 0) p.dbg[2048 + 128 * fs_n + s] = c;
        }
      }
#endif
      fence();
    });
    // the wave's own A pieces of the next tile's stage 1 overwrite its scratch: its read-backs must have returned first
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (TRACE) { trace_fs(15); ++fs_n; }
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_item(d, ktn, 1, i);
    fence();
    if constexpr (LAB & 2) pin_acc();   // the accumulators the ablation does not retire must stay live, or their MFMAs are eliminated
  };

  // ---- [r5] the last stage with the bf16-first retirement ("retire16").  What the round-4 trace said about the fp32 form above (profiles/final_stage_contention_r4.txt):
  //      ~8 400 cycles per tile against 2 048 of MFMA, on an idle chip -- 512 KiB per tile through LDS as ds_write_b128 (13 cycles of store-data path each:
  //      3 300 cycles per tile by themselves) and three dependent LDS round trips per pair through ONE scratch and ONE read-back register set.  Here a lane
  //      converts its accumulator values BEFORE they go to LDS: 4 consecutive columns -> v_accvgpr_read x 4, (alpha) v_mul_f32 x 4, v_cvt_pk_bf16_f32 x 2 ->
  //      ONE ds_write_b64.  A pair of tiles is then 32 rows x 128 B = 4 KiB: the wave's 8-KiB slice holds TWO pairs (pair P in half P & 1), so the writes of
  //      pair P + 1 never wait for the read-back of pair P, a read-back is ONE ds_read_b128 per 8 rows x 128 B and goes to the store untouched (no VALU
  //      between LDS and the store, no read-back -> convert -> store chain), and the bytes through LDS halve (256 KiB per tile, ds_write_b64: 6 cycles each).
  //      The price is VALU issue: 10 vector instructions per piece, one piece per MFMA slot.
  //   piece pi = 4 T + q (tile T = 4 m + n in MFMA order, q = the lane's 4 columns 8 q + 4 g .. + 3 of the tile) rides in slot pi + 3 (tile T >= 2 is final
  //   after slot 4 T - 1, tiles 0 / 1 after slots 1 / 3); pair P = tiles 2 P, 2 P + 1 is read back in slots 8 P + 11 / + 12 and stored in 8 P + 15 .. + 18
  //   (the last pair: + 13 .. + 16); pair P + 2 writes the same half from slot 8 P + 19 on (LDS operations of a wave execute in order).
  //   LDS layout of a half: 16 row pairs x 256 B; row r, tile nn, quarter q, lane half g at
  //       (r >> 1) 256 + nn 128 + (r & 1) 64 + (q ^ ((r >> 1) & 3)) 16 + (g ^ ((r >> 3) & 1)) 8
  //   -- the 16 lanes of a ds_write_b64 group (rows r .. r + 15, same nn / q / g) hit 16 different 8-byte slots of the 128-byte bank window, and the 16 lanes of
  //   a ds_read_b128 group 16 different 16-byte chunks of the 256-byte window.  Rows with bit 3 set have their 8-byte halves swapped; a read-back pass covers rows
  //   8 p + lane / 8, so that is a property of the PASS: odd passes fetch their two halves separately, in the right register order.
  const int r16w = (i32 >> 1) * 256 + (i32 & 1) * 64 + ((g ^ ((i32 >> 3) & 1)) << 3);
  const int r16x = ((i32 >> 1) & 3) << 4;
  const int r16r = (rrl >> 1) * 256 + (ccl >> 2) * 128 + (rrl & 1) * 64 + (((ccl & 3) ^ (rrl >> 1)) << 4);
  v4i rbh[4];
  auto final_stage16 = [&](const Desc& d, bool dvalid, const int ktn, auto a1c) __attribute__((always_inline)) {
    constexpr bool A1 = decltype(a1c)::value;   // alpha == 1: no multiply
    // D: the slot of the first piece.  QAMD_FS_BURST: the whole retirement BEHIND the MFMAs instead of threaded through them (nothing issues in the shadow of this MFMA with one
    // wave per SIMD, and a v_accvgpr_read_b32 costs 9.5 cycles while MFMAs run against ~4 when none does: profiles/final_stage_ablation_r5c.txt)
    constexpr int IL = QAMD_DEEPP_FS_IL, NMF = 64 - 2 * IL, D = QAMD_FS_BURST ? NMF + 1 : IL + 3, NS = 70 + D;
    static_assert(IL == 1 || IL == 2 || IL == 4, "block of the last stage: 1, 2 or 4 tiles");
    read_slice(1, 2);
    read_slice(1, 3);
    fence();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < IL; ++t) mfma1(j, 1, t / 4, t % 4, false);
    fence();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next tile's stage 0 landed (buffer 0); all reads of buffer 1 done
    __builtin_amdgcn_s_barrier();
    fence();
    dma_prep(ktn, dvalid);
    int wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[q] = r16w + (r16x ^ (q << 4));
    // (odd passes: the second half through its own, opaque base address -- two adjacent 8-byte loads are otherwise merged into one ds_read_b128 + four v_mov_b32
    //  behind an s_waitcnt lgkmcnt(0) right where the read was issued)
    int r16r8 = r16r + 8;
    asm volatile("" : "+v"(r16r8));
    int ldd2 = p.ldd * 2;
    asm volatile("" : "+s"(ldd2));
    fence();
    auto piece = [&](const int pi) __attribute__((always_inline)) {
      const int T = pi >> 2, q = pi & 3, m = T >> 2, n = T & 3;
      float x0 = acc[m][n][4 * q + 0], x1 = acc[m][n][4 * q + 1], x2 = acc[m][n][4 * q + 2], x3 = acc[m][n][4 * q + 3];
      if constexpr ((QAMD_FS_ABL & 2) != 0) { asm volatile("" ::"v"(x0), "v"(x1), "v"(x2), "v"(x3)); return; }
      if constexpr (!A1 && QAMD_DEEPP_RETIRE == 3) {   // (lab A/B: what the compiler makes of it -- two v_pk_mul_f32)
        x0 *= alpha; x1 *= alpha; x2 *= alpha; x3 *= alpha;
      } else if constexpr (!A1) {   // (plain v_mul_f32: the compiler would pack these into v_pk_mul_f32)
        asm("v_mul_f32 %0, %1, %2" : "=v"(x0) : "s"(alpha), "v"(x0));
        asm("v_mul_f32 %0, %1, %2" : "=v"(x1) : "s"(alpha), "v"(x1));
        asm("v_mul_f32 %0, %1, %2" : "=v"(x2) : "s"(alpha), "v"(x2));
        asm("v_mul_f32 %0, %1, %2" : "=v"(x3) : "s"(alpha), "v"(x3));
      }
      if constexpr ((QAMD_FS_ABL & 4) != 0) { asm volatile("" ::"v"(pack_bf16x2(x0, x1)), "v"(pack_bf16x2(x2, x3))); return; }
      *(v2i*)(scr + ((T >> 1) & 1) * 4096 + (T & 1) * 128 + wq[q]) = v2i{(int)pack_bf16x2(x0, x1), (int)pack_bf16x2(x2, x3)};
    };
    auto readback = [&](const int P, const int pass) __attribute__((always_inline)) {
      const int o = (P & 1) * 4096 + pass * 1024;
      if (pass & 1) {
        const v2i lo = *(const v2i*)(scr + o + r16r8), hi = *(const v2i*)(scr + o + r16r);
        rbh[pass] = v4i{lo[0], lo[1], hi[0], hi[1]};
      } else {
        rbh[pass] = *(const v4i*)(scr + o + r16r);
      }
    };
    // (a wait state behind every store: on gfx950 a VALU write of the data registers of a 16-byte buffer store DIRECTLY behind it corrupts the store also when the store
    //  carries an SGPR offset -- tests/native/store_hazard_probe.hip, profiles/store_hazard_probe_r5.txt; the compiler guards only stores WITHOUT one.  The read-back
    //  registers are recycled as conversion temporaries, so this is not hypothetical here; tools/store_data_hazard.py scans the ISA, a CPU test)
    auto store16 = [&](const int P, const int pass) __attribute__((always_inline)) {
      const int m = P >> 1, h = P & 1;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, rbh[pass]), rD, (64 * h < colLim) ? stLane : (int)0x80000000, (32 * m + 8 * pass) * ldd2 + 128 * h, ST_AUX);
      asm volatile("s_nop 0" : "+v"(rbh[pass]) :: "memory");   // (the data registers stay claimed up to here: nothing else can be allocated into them, i.e. written, before the wait state)
    };
    static_for<0, NS>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s < NMF) {
        mfma1(fs_j(IL, s), 1, fs_T(IL, s) / 4, fs_T(IL, s) % 4, false);
        fence();   // the MFMA FIRST in its slot: left to the scheduler it drifts to the end of some slots and the start of others (3 .. 27 instructions between two MFMAs)
      }
      // DMA of the next tile's stage 1 into buffer 1 (B pieces + the scale piece; the wave's A pieces land in its scratch: after the loop), the next tile's
      // stage-0 scales, and its stage-0 fragments as their registers die: A rows of m behind the last MFMA of tile (m, 3), B rows of n behind that of tile (3, n)
      constexpr bool AUX = (QAMD_FS_ABL & 16) == 0, RETIRE = (QAMD_FS_ABL & 1) == 0, BACK = (QAMD_FS_ABL & 9) == 0;
      if constexpr (AUX && s % 3 == 0 && s / 3 < 9) dma_item(d, ktn, 1, 8 + s / 3);
      if constexpr (AUX && s == 1) read_scales(0, 0);
      static_for<0, 4>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        if constexpr (AUX && s == fs_F(IL, 4 * m + 3) + 1) { read_fa(0, 0, m); read_fa(0, 1, m); }
        if constexpr (AUX && s == fs_F(IL, 12 + m) + 1) { read_fb(0, 0, m); read_fb(0, 1, m); }
      });
      // pair P = tiles 2 P, 2 P + 1: pieces 8 P .. 8 P + 7 in slots 8 P + D .. + 7, read back in 8 P + D + 8 / + 9, stored in + 12 .. + 15 (the last pair: + 10 .. + 13)
      if constexpr (BACK && s >= D + 12 && (s - D - 12) / 8 < 7 && (s - D - 12) % 8 < 4) store16((s - D - 12) / 8, (s - D - 12) % 8);
      if constexpr (BACK && s >= D + 66) store16(7, s - D - 66);
      if constexpr (RETIRE && s >= D && s < D + 64) piece(s - D);
      if constexpr (BACK && s >= D + 8 && (s - D - 8) % 8 == 0 && (s - D - 8) / 8 < 8) { readback((s - D - 8) / 8, 0); readback((s - D - 8) / 8, 1); }
      if constexpr (BACK && s >= D + 9 && (s - D - 9) % 8 == 0 && (s - D - 9) / 8 < 8) { readback((s - D - 9) / 8, 2); readback((s - D - 9) / 8, 3); }
      if constexpr (TRACE && s % 8 == 0 && s <= 64 + 8 * QAMD_FS_BURST * 6) trace_fs(s / 8);
#ifdef QAMD_FS_TRACE_SLOTS   // (lab) a mark behind EVERY slot of the first two last stages: dbg[2048 + 128 f + s]
      if constexpr (TRACE) {
        if (blockIdx.x == 0 && wave == 0 && p.dbg && fs_n < 2) {
          const uint32_t c = (uint32_t)__builtin_readcyclecounter();
          if (lane == 0) p.dbg[2048 + 128 * fs_n + s] = c;
        }
      }
#endif
      fence();
    });
    // the wave's own A pieces of the next tile's stage 1 overwrite its scratch: its read-backs must have returned first
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (TRACE) { trace_fs(15); ++fs_n; }
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_item(d, ktn, 1, i);
    fence();
  };
  constexpr int RET16 = (SK || (LAB & 7)) ? 0 : QAMD_DEEPP_RETIRE;   // (LAB = 8: the plain entry routed to this copy, QAMD_ROUTE_LABK -- no ablation)
  // (RET16 == 2: the alpha == 1 arm is a second copy of the whole tile walk, entered once per workgroup -- a branch per tile around two copies of the last stage
  //  makes the register allocator join two hand-scheduled stages and spills 128 registers)
  auto last_stage = [&](const Desc& d, bool dvalid, const int ktn, auto a1c) __attribute__((always_inline)) {
    if constexpr (RET16 != 0) final_stage16(d, dvalid, ktn, a1c);
    else final_stage(d, dvalid, ktn);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using BT = std::integral_constant<bool, true>;
  using BF = std::integral_constant<bool, false>;
  if constexpr (SK) {
    // ---- stream-K walk (see the head of this function).  ONE instruction path for every kind of unit -- at 255 of 256 vector registers every join
    //      of two differently scheduled code paths (a skipped stage, a first stage with / without zero accumulators, a last stage with / without the
    //      epilogue) turned into register copies and 100+ spills inside the hand-scheduled stages; the kinds differ in DATA only:
    //        * the accumulators are set BEFORE the first stage -- 16 MFMAs on zero operands, or, for the unit that owns the FIRST K stages of a cut
    //          tile, 64 loads of the part its neighbour parked -- and the first stage accumulates like any other;
    //        * every unit ends with the accumulator-stationary last stage; it retires to D, and, through the same read-back registers, to the
    //          scratch slot rP, one of which is an empty descriptor (a parking unit writes no D, the others park nothing).
    SkWalk walk(wg, G, ntiles, p.sk_tiles, KTe, 2, 2);
    auto next_unit = [&]() __attribute__((always_inline)) {
      SkUnit u = walk.next();
      u.tile = uniform(u.tile); u.kb = uniform(u.kb); u.ke = uniform(u.ke); u.mode = uniform(u.mode); u.slot = uniform(u.slot);
      return u;
    };
    auto slot_rsrc = [&](const int slot, const bool on) __attribute__((always_inline)) {
      return make_rsrc((const char*)p.ws + (size_t)slot * SK_PART_BYTES, on ? (uint32_t)SK_PART_BYTES : 0u);
    };
    // accumulators of the unit that starts now: zero (one MFMA per 32x32 tile on zero operands: 16 instructions in the idle matrix pipe instead of
    // 256 v_accvgpr_write), or the parked part of the cut tile -- row-major pairs (retire_store's layout) gathered into the accumulator layout:
    // row i32 of pair (m, h), columns 32 nn + 8 q + 4 g  ->  ((wave 8 + pair) 4 + i32 / 8) 2 KiB + ((i32 % 8) 8 + 4 nn + q) 32 B + 16 g
    auto init_acc = [&](const bool from_slot, const int slot) __attribute__((always_inline)) {
      v8i z = {};
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          asm volatile("" : "+v"(z));   // (a fresh opaque value per tile: identical MFMAs are otherwise merged into one + 240 register copies)
          acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(z, z, v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 4, 4, 0, 0, 0, 0);
        }
      if (from_slot) {   // (parked at the START of its owner's walk: the flag is long set)
        const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (wave == 0 && elane == 0)
          while (__hip_atomic_load(p.ctr + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.tag) __builtin_amdgcn_s_sleep(4);
        __builtin_amdgcn_s_barrier();
        const __amdgpu_buffer_rsrc_t rW = slot_rsrc(slot, true);
        const int ei32 = elane & 31, eg = elane >> 5;
        const int ul = wave * 65536 + (ei32 >> 3) * 2048 + (ei32 & 7) * 256 + eg * 16;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rW, ul, ((2 * m + n / 2) * 4) * 2048 + (4 * (n & 1) + q) * 32, 17));
              acc[m][n][4 * q + 0] = v[0]; acc[m][n][4 * q + 1] = v[1]; acc[m][n][4 * q + 2] = v[2]; acc[m][n][4 * q + 3] = v[3];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0 && elane == 0) __hip_atomic_store(p.ctr + slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // a replayed graph (same tag) starts clean
      }
    };
    // (the walk runs one unit ahead; of a unit only what the stages need stays in scalar registers: its stage range and kind -- the tile went into
    //  the offsets / the output descriptors when the unit was fetched, the scratch slot follows from the kind)
    auto set_out_unit = [&](const int tile, const int mode) __attribute__((always_inline)) {
      int m0, n0;
      decode(tile, m0, n0);
      set_out_tile(m0, n0);
      if (mode == 1) rD = make_rsrc(p.D, 0);    // parking: nothing goes to D
      rP = slot_rsrc(wg, mode == 1);            // ... the raw sums go to this workgroup's slot
    };
    SkUnit u0 = next_unit();
    if (u0.mode < 0) return;
    Desc dcur = make_desc(u0.tile);
    int kb = u0.kb, ke = u0.ke, mode = u0.mode;
    set_out_unit(u0.tile, mode);
    SkUnit u1 = next_unit();
    Desc dnxt = make_desc(u1.mode >= 0 ? u1.tile : ntiles);
    int nkb = u1.kb, nke = u1.ke, nmode = u1.mode, ntile = u1.tile;
    dma_stage(dcur, kb, true, 0);
    dma_stage(dcur, kb + 1, true, 1);
    asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    read_scales(0, 0);
    read_slice(0, 0);
    read_slice(0, 1);
    fence();
    while (mode >= 0) {
      const bool nvalid = nmode >= 0;
      // the stage two ahead of stage s - 2 of this unit: stage s, or stage kb' + (s - ke) of the next unit
      auto ahead = [&](const int s, Desc& d, int& kt, bool& valid) __attribute__((always_inline)) {
        const bool tonext = s >= ke;
        d.ao = tonext ? dnxt.ao : dcur.ao; d.bo = tonext ? dnxt.bo : dcur.bo; d.so = tonext ? dnxt.so : dcur.so;
        kt = tonext ? nkb + (s - ke) : s;
        valid = tonext ? nvalid : true;
      };
      Desc d;
      int ktl;
      bool dv;
      init_acc(mode == 2, wg + 1);   // (the part of a cut tile is parked by the NEXT workgroup of the walk, in its own slot)
      ahead(kb + 2, d, ktl, dv);
      stage(I0{}, BF{}, d, ktl, dv, BF{});
      for (int kt = kb + 1; kt + 2 < ke; kt += 2) {
        ahead(kt + 2, d, ktl, dv);
        stage(I1{}, BF{}, d, ktl, dv, BF{});
        ahead(kt + 3, d, ktl, dv);
        stage(I0{}, BF{}, d, ktl, dv, BF{});
      }
      final_stage(dnxt, nvalid, nkb + 1);
      if (mode == 1) {   // parked: acknowledged by the coherence point, then the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (wave == 0 && elane == 0) __hip_atomic_store(p.ctr + wg, p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // the next unit becomes the current one; fetch the one after it
      kb = nkb; ke = nke; mode = nmode;
      dcur = dnxt;
      if (mode >= 0) set_out_unit(ntile, mode);
      u1 = next_unit();
      dnxt = make_desc(u1.mode >= 0 ? u1.tile : ntiles);
      nkb = u1.kb; nke = u1.ke; nmode = u1.mode; ntile = u1.tile;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- prologue: first tile's stages 0 and 1 in flight; stage 0 landed -> first two slices into registers ---------------
  int tile = wg;
  Desc cur = make_desc(tile);
  dma_stage(cur, 0, true, 0);
  dma_stage(cur, 1, true, 1);
  asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  fence();
  read_scales(0, 0);
  read_slice(0, 0);
  read_slice(0, 1);
  fence();
  trace();

  auto walk = [&](auto a1c) __attribute__((always_inline)) {
  while (tile < ntiles) {
    int m0, n0;
    decode(tile, m0, n0);
    set_out_tile(m0, n0);
    const int tnext = tile + G;
    const Desc nxt = make_desc(tnext);
    const bool nvalid = tnext < ntiles;
    // stage 0 (accumulators start from 0); its DMA is stage 2 of this tile, or stage 0 of the next tile when KTe == 2
    {
      const bool tonext = KTe == 2;
      Desc d;
      d.a = tonext ? nxt.a : cur.a; d.b = tonext ? nxt.b : cur.b; d.s = tonext ? nxt.s : cur.s;
      stage(I0{}, BT{}, d, tonext ? 0 : 2, tonext ? nvalid : true, BF{});
    }
    if constexpr ((LAB & 2) || !QAMD_DEEPP_PEEL) {
      for (int kt = 1; kt + 2 < KTe; kt += 2) {
        stage(I1{}, BF{}, cur, kt + 2, true, BF{});
        const bool tonext = kt + 3 == KTe;
        Desc d;
        d.a = tonext ? nxt.a : cur.a; d.b = tonext ? nxt.b : cur.b; d.s = tonext ? nxt.s : cur.s;
        if constexpr (LAB & 2) {
          if (tonext) stage(I0{}, BF{}, d, 0, nvalid, BT{});
          else stage(I0{}, BF{}, d, kt + 3, true, BF{});
        } else {
          stage(I0{}, BF{}, d, tonext ? 0 : kt + 3, tonext ? nvalid : true, BF{});
        }
      }
    } else {
      // [r4] only the LAST pair of stages issues DMA for the next tile: peeled, so that the loop body no longer selects three descriptors between its
      // two stages (12 s_cselect + compares = 21 scalar instructions in one MFMA slot, by the ISA's slot accounting)
      int kt = 1;
      for (; kt + 4 < KTe; kt += 2) {
        stage(I1{}, BF{}, cur, kt + 2, true, BF{});
        stage(I0{}, BF{}, cur, kt + 3, true, BF{});
      }
      if (kt + 2 < KTe) {
        stage(I1{}, BF{}, cur, kt + 2, true, BF{});
        stage(I0{}, BF{}, nxt, 0, nvalid, BF{});
      }
    }
    trace();
    last_stage(nxt, nvalid, 1, a1c);
    trace();
    cur = nxt;
    tile = tnext;
  }
  };
  if constexpr (RET16 == 2) {
    if (alpha == 1.0f) walk(std::true_type{});
    else walk(std::false_type{});
  } else {
    walk(std::false_type{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  trace();
  if constexpr (TRACE) {
    if (blockIdx.x == 0 && wave == 0 && lane == 0 && p.dbg) p.dbg[0] = (uint32_t)trace_n;
    // every workgroup: wall-clock entry / exit ticks (100 MHz) -> dispatch ramp and finish skew across the chip
    if (wave == 0 && lane == 0 && p.dbg && blockIdx.x < 256) {
      p.dbg[64 + 2 * blockIdx.x] = wg_t0;
      p.dbg[65 + 2 * blockIdx.x] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------------
// The same structure for MXFP8 (matmul_mxf8_bf16_tn; A e4m3 or e5m2 via C::AFMT, B e4m3).  A stage is 128 K-elements = two
// k-slices of two 16-byte chunks per fragment (split register layout of gemm_mx_deep8: chunk 4j + 2u + g, op_sel 2j), two
// fragment sets, an MFMA is 64 cycles:
//     R(1) ; M(0) ; hand-off ; scales' ; R'(0) ; M(1) with the DMA of stage kt + 2 threaded through
// Last stage of a tile: accumulator-stationary with 2 MFMAs per 32x32 tile; a pair of tiles is final after post-hand-off MFMA
// e = 4 P + 1 (30 MFMAs after the hand-off) and retires through the wave's own 8-KiB slice of buffer 1 exactly as in the fp4
// kernel:  write e+1 | read rows 0-15 e+2 | stores + read rows 16-31 e+4 | stores e+6.
// -------------------------------------------------------------------------------------------------------------------------
constexpr int deepp8_pair_done_at(int s) { return (s >= 1 && s <= 29 && (s - 1) % 4 == 0) ? (s - 1) / 4 : -1; }

template <class C, int ST_AUX = 0, bool NN = false, int NNABL = 0, bool SK = false>
__device__ __forceinline__ void gemm_mx_deepp8(char* smem, const GemmParams& p, const int bid, const int G, const int ntiles) {
  static_assert(!(SK && NN), "stream-K: TN only");
  static_assert(C::EBITS == 8 && C::F8SPLIT && C::BM == 256 && C::BN == 256 && C::WAVES_M == 2 && C::WAVES_N == 2 && C::NSTAGE == 2 && C::PPW == 1,
                "persistent deep schedule (fp8): 256x256 tiles, 4 waves of 128x128, split register layout");
  constexpr int MT = 4, NT = 4;
  constexpr int STAGE = C::STAGE_BYTES, OFF_SCR = DeepPCfg<C>::OFF_SCR;
  GemmCtx<C> cx(smem, p);   // per-lane offsets / LDS addresses; its tile coordinates and descriptors are NOT used here
  const int lane = cx.lane, wave = cx.wave, i32 = cx.i32, g = cx.g;
  const int KT = cx.KT, KTe = (KT + 1) & ~1, CB = cx.CB, rowbytes = cx.rowbytes;
  const int wg = xcd_remap(bid, G);
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

  auto decode = [&](int t, int& m0, int& n0) __attribute__((always_inline)) {
    int tm, tn;
    raster_decode(t, p.tiles_m, p.tiles_n, p.raster_magic, tm, tn);
    m0 = uniform(tm * C::BM);
    n0 = uniform(tn * C::BN);
  };
  // ---- NN: A handed over as (K, M) row-major (matmul_host_mxf8_bf16_nn, gemm.cu:388-434) ---------------------------------
  // The A stage is DMAed as it lies in memory: [128 k][256 m] bytes (piece q = k-rows 4q .. 4q+3 = 1 KiB, lane = row l/16,
  // 16-byte chunk l%16), and the row fragments are read with ds_read_b64_tr_b8: per 16 lanes it takes an [8 k][16 m] byte
  // block (lane i supplies the 8 bytes at row i/2, columns 8 (i%2) ..) and hands lane c column c -- 8 consecutive k of one
  // m, i.e. a quarter of a K-contiguous fragment row, in the natural row order (tests/native/tr_probe.hip,
  // profiles/native_r2_tr_probe.txt).  Four of them per fragment replace the two ds_read_b128 of the TN kernel at the same
  // LDS cycles.  Bank-conflict freedom: a 32-lane pass covers 8 k-rows x 32 bytes; 16-byte chunk c of row k is stored at
  // chunk c ^ 2 (k & 7), which spreads those 8 rows over all 64 banks.
  int nn_col0 = 0, nn_v[2] = {0, 0}, nnA0 = 0;
  if constexpr (NN) {
    const int kk = lane >> 4, pos = lane & 15;
#pragma unroll
    for (int par = 0; par < 2; ++par)   // par = piece parity: k & 7 = 4 par + kk; the parities' chunks differ by ^ 8 (128 bytes)
      nn_v[par] = kk * p.M + ((pos ^ (2 * (4 * par + kk))) << 4);
    nn_col0 = (pos ^ (2 * kk)) << 4;
    // fragment t of this lane: chunk (wave_m * 8 + 2 t + b) ^ 2 r = ((wave_m * 8 + b) ^ 2 r) ^ 2 t  ->  address nnA0 ^ 32 t
    const int idx = lane & 15, r = idx >> 1, b = (lane >> 4) & 1;
    nnA0 = (16 * g + r) * 256 + (((cx.wave_m * 8 + b) ^ (2 * r)) << 4) + 8 * (idx & 1);
  }
  // (SK: three kernel-wide descriptors + per-tile scalar offsets, as in the fp4 kernel)
  struct Desc { __amdgpu_buffer_rsrc_t a, b, s; int mrem; uint32_t ao, bo, so; };
  const __amdgpu_buffer_rsrc_t skA = make_rsrc(p.A, SK ? p.a_bytes : 0u), skB = make_rsrc(p.B, SK ? p.b_bytes : 0u);
  const __amdgpu_buffer_rsrc_t skS = cx.sIsB ? make_rsrc(p.SFB, SK ? p.sfb_bytes : 0u) : make_rsrc(p.SFA, SK ? p.sfa_bytes : 0u);
  auto make_desc = [&](int t) __attribute__((always_inline)) {
    const bool valid = t < ntiles;
    int m0, n0;
    decode(valid ? t : ntiles - 1, m0, n0);
    const uint32_t a_off = NN ? (uint32_t)m0 : (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
    const uint32_t sa_off = (uint32_t)(m0 >> 7) * CB * 512, sb_off = (uint32_t)(n0 >> 7) * CB * 512;
    Desc d;
    d.mrem = p.M - m0;   // NN: columns past M would read the next k-row: those lanes fetch out of range (zeros) instead
    if constexpr (SK) {
      d.ao = valid ? a_off : 0x80000000u;
      d.bo = valid ? b_off : 0x80000000u;
      d.so = valid ? (cx.sIsB ? sb_off : sa_off) : 0x80000000u;
    } else {
      d.a = make_rsrc(p.A + a_off, valid ? p.a_bytes - a_off : 0u);
      d.b = make_rsrc(p.B + b_off, valid ? p.b_bytes - b_off : 0u);
      d.s = cx.sIsB ? make_rsrc(p.SFB + sb_off, valid ? p.sfb_bytes - sb_off : 0u) : make_rsrc(p.SFA + sa_off, valid ? p.sfa_bytes - sa_off : 0u);
    }
    return d;
  };

  v16f acc[MT][NT];
  v8i fa[2][MT] = {}, fb[2][NT] = {};
  int sa[2][MT], sb[2][NT];

  auto read_fa = [&](const int buf, const int j, const int t) __attribute__((always_inline)) {
    const char* st = smem + buf * STAGE;
    if constexpr (NN && !(NNABL & 2)) {   // (lab: NNABL bit 1 = TN-style fragment reads, timing only)
      // Inline asm, not __builtin_amdgcn_ds_read_tr8_b64_v2i32: the builtin carries no memory operand, so the compiler's
      // wait-count pass assumes it may read what an outstanding LDS-DMA is writing and puts s_waitcnt vmcnt(0) in front of
      // every one of them -- the K loop then waits for the DMA of the NEXT stage before reading this one (65 us against
      // 58 us for 4096^3, profiles/native_r2_nn_steady.log).  The price: the compiler does not count these reads in
      // lgkmcnt, so every consumer sits behind an explicit s_waitcnt lgkmcnt(0) (nn_wait below; LDS returns in order, so the
      // compiler's own lgkmcnt(n) for its tracked reads can only wait longer than it needs, never shorter).
      const uint32_t a = (uint32_t)(uintptr_t)(lds_ptr_t)(st + (nnA0 ^ (32 * t)));
      v2i q[4];   // (u, h): k = 64 j + 32 u + 16 g + 8 h .. +7 of row wave_m * 128 + 32 t + i32
#pragma unroll
      for (int uh = 0; uh < 4; ++uh)
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(q[uh]) : "v"(a), "n"((64 * j + 32 * (uh >> 1) + 8 * (uh & 1)) * 256) : "memory");
      fa[j][t] = v8i{q[0][0], q[0][1], q[1][0], q[1][1], q[2][0], q[2][1], q[3][0], q[3][1]};
    } else {
      const v4i lo = *(const v4i*)(st + cx.rdA[2 * j] + t * 32 * C::ROWB);
      const v4i hi = *(const v4i*)(st + cx.rdA[2 * j + 1] + t * 32 * C::ROWB);
      fa[j][t] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  auto read_fb = [&](const int buf, const int j, const int t) __attribute__((always_inline)) {
    const char* st = smem + buf * STAGE;
    const v4i lo = *(const v4i*)(st + cx.rdBd + cx.rdA[2 * j] + t * 32 * C::ROWB);
    const v4i hi = *(const v4i*)(st + cx.rdBd + cx.rdA[2 * j + 1] + t * 32 * C::ROWB);
    fb[j][t] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto read_slice = [&](const int buf, const int j) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < MT; ++t) read_fa(buf, j, t);
#pragma unroll
    for (int t = 0; t < NT; ++t) read_fb(buf, j, t);
  };
  auto nn_wait = [&]() __attribute__((always_inline)) {   // the asm fragment reads above have landed (see read_fa)
    if constexpr (NN && !(NNABL & 2)) {
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0), as a builtin: the compiler's scoreboard is cleared with it
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);    // no consumer is scheduled above the wait
    }
  };
  // scale dwords of a stage: loaded raw (the four row fragments' dwords are consecutive: one 16-byte read per operand) and
  // shifted into place a few MFMAs later, so that no instruction waits on the load right after it was issued
  v4i sraw[2];
  auto scales_load = [&](const int buf) __attribute__((always_inline)) {
    const char* st = smem + buf * STAGE;
    sraw[0] = *(const v4i*)(st + cx.rdSA[0]);
    sraw[1] = *(const v4i*)(st + cx.rdSB[0]);
  };
  auto scales_fin = [&](const int set) __attribute__((always_inline)) {
    const int shift = 8 * g;   // split layout: lanes 0-31 carry K-block 2j, lanes 32-63 K-block 2j + 1
#pragma unroll
    for (int t = 0; t < MT; ++t) sa[set][t] = (int)((unsigned)sraw[0][t] >> shift);
#pragma unroll
    for (int t = 0; t < NT; ++t) sb[set][t] = (int)((unsigned)sraw[1][t] >> shift);
  };
  auto read_scales = [&](const int buf, const int set) __attribute__((always_inline)) {
    scales_load(buf);
    scales_fin(set);
  };
  auto mfma1 = [&](const int j, const int sset, const int m, const int n, const bool zero_c) __attribute__((always_inline)) {
    v16f c = acc[m][n];
    if (zero_c) c = v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j == 0) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[0][n], fa[0][m], c, 0, C::AFMT, 0, sb[sset][n], 0, sa[sset][m]);
    if (j == 1) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[1][n], fa[1][m], c, 0, C::AFMT, 2, sb[sset][n], 2, sa[sset][m]);
  };

  int vb0 = 0, vb1 = 0, vbS = 0, va0 = 0, va1 = 0;
  const int nn_rstep = 4 * p.M, nn_kstep = 128 * p.M;   // NN: bytes between consecutive A pieces / K stages
  auto dma_prep = [&](const Desc& d, int kt, bool valid) __attribute__((always_inline)) {
    if constexpr (SK) {   // (as in the fp4 kernel)
      vb0 = cx.voffAB[0]; vb1 = cx.voffAB[1]; vbS = cx.voffS;
      asm volatile("" : "+v"(vb0), "+v"(vb1), "+v"(vbS));
      return;
    }
    int lastmask = (kt == KT - 1) ? -1 : 0;
    int oobm = (valid && kt < KT) ? 0 : -1;
    int oobs = (valid && kt * C::SCT + cx.colS < CB) ? 0 : -1;
    asm volatile("" : "+v"(lastmask), "+v"(oobm), "+v"(oobs));
    vb0 = (((cx.voffT[0] & lastmask) | (cx.voffAB[0] & ~lastmask)) & ~oobm) | ((int)0x80000000 & oobm);
    vb1 = (((cx.voffT[1] & lastmask) | (cx.voffAB[1] & ~lastmask)) & ~oobm) | ((int)0x80000000 & oobm);
    vbS = (cx.voffS & ~oobs) | ((int)0x80000000 & oobs);
    if constexpr (NN) {
      const int o0 = oobm | (nn_col0 < d.mrem ? 0 : -1), o1 = oobm | ((nn_col0 ^ 128) < d.mrem ? 0 : -1);
      va0 = (nn_v[0] & ~o0) | ((int)0x80000000 & o0);
      va1 = (nn_v[1] & ~o1) | ((int)0x80000000 & o1);
    }
  };
  auto dma_item = [&](const Desc& d, int kt, const int buf, const int item) __attribute__((always_inline)) {
    char* st = smem + buf * STAGE;
    if (NN && !(NNABL & 1) && item < 8) {   // (lab: NNABL bit 0 = TN-style A addresses, timing only)
      const int q = wave * 8 + item;
      // k-rows past K (last stage of a K that is not a multiple of 128) lie past the end of the descriptor and read zeros: on gfx950
      // the range check of a raw buffer covers voffset + soffset (tests/native/soffset_probe.hip, profiles/native_r2_soffset_probe.txt;
      // tests/test_gpu_parity.py puts fp8 NaN bytes behind the operand)
      const int v = ((item & 1) ? va1 : va0) + q * nn_rstep;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d.a, (lds_ptr_t)(st + q * 1024), 16, v, kt * nn_kstep, 0, QAMD_DMA_AUX);
    } else if (item < 16) {
      const int t = item & 7, q = wave * 8 + t;
      const int v = ((t & 1) ? vb1 : vb0) + q * cx.rstep;
      if constexpr (SK)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(item < 8 ? skA : skB, (lds_ptr_t)(st + (item < 8 ? 0 : C::OFF_B) + q * 1024), 16, v, (int)((item < 8 ? d.ao : d.bo) + (uint32_t)(kt * C::ROWB)), 0, QAMD_DMA_AUX);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(item < 8 ? d.a : d.b, (lds_ptr_t)(st + (item < 8 ? 0 : C::OFF_B) + q * 1024), 16, v, kt * C::ROWB, 0, QAMD_DMA_AUX);
    } else {
      if constexpr (SK) __builtin_amdgcn_raw_ptr_buffer_load_lds(skS, (lds_ptr_t)(st + C::OFF_S + wave * 1024), 16, vbS, (int)(d.so + (uint32_t)(kt * C::SCT * 512)), 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(d.s, (lds_ptr_t)(st + C::OFF_S + wave * 1024), 16, vbS, kt * C::SCT * 512, 0, 0);
    }
  };
  auto dma_stage = [&](const Desc& d, int kt, bool valid, const int buf) __attribute__((always_inline)) {
    dma_prep(d, kt, valid);
#pragma unroll
    for (int i = 0; i < 17; ++i) dma_item(d, kt, buf, i);
  };
  auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" : "+a"(acc[m][n]));
  };

  // one K stage (not the last of its tile).  Entry: fragment set 0 and scale set BUF hold slice 0 of this stage.
  auto stage = [&](auto bufc, auto firstc, const Desc& d, int ktl, bool dvalid) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    // M(0) with R(1) threaded through (one fragment per MFMA: a burst of 16 -- NN: 24 -- LDS reads in front of the MFMAs
    // overflows the 4-bit lgkmcnt, and the compiler then has to wait for the burst itself before the first MFMA)
    int idx = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        mfma1(0, BUF, m, n, FIRST);
        if (idx < 4) read_fa(BUF, 1, idx);
        else if (idx < 8) read_fb(BUF, 1, idx - 4);
        if (idx == 12) dma_prep(d, ktl, dvalid);   // [r4] behind an MFMA, not between the barrier and the second group (it does not depend on the hand-off)
        fence();
        ++idx;
      }
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0), as a builtin: the compiler's wait-count scoreboard sees it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    scales_load(BUF ^ 1);
    fence();
    idx = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        mfma1(1, BUF, m, n, false);
        dma_item(d, ktl, BUF, idx);   // ([r4] two items in each of the eight slots without a fragment read, the fp4 kernel's gain, costs this one 3-6 %: 64-cycle MFMAs)
        if (idx == 0) dma_item(d, ktl, BUF, 16);
        if (idx < 4) read_fa(BUF ^ 1, 0, idx);
        else if (idx < 8) read_fb(BUF ^ 1, 0, idx - 4);
        if (idx == 8) scales_fin(BUF ^ 1);
        fence();
        ++idx;
      }
    nn_wait();   // R'(0) was issued 12+ MFMAs ago
    if constexpr (FIRST) pin_acc();
  };

  // epilogue pieces: identical to the fp4 kernel (pairs of 32x32 tiles through the wave's 8-KiB slice of buffer 1's A area)
  char* scr = smem + OFF_SCR + wave * DeepPCfg<C>::SCR_PER_WAVE;
  const int scrW = i32 * 256 + ((((i32 & 6) << 4)) | ((g ^ (i32 & 1)) << 4));
  const int rrl = lane >> 3, ccl = lane & 7;
  const int scrR = rrl * 256 + (ccl >> 2) * 128 + ((((2 * ccl) & 7) ^ (rrl & 7)) << 4);
  const float alpha = *p.alpha;
  __amdgpu_buffer_rsrc_t rD = make_rsrc(p.D, 0);
  __amdgpu_buffer_rsrc_t rP = make_rsrc(p.D, 0);          // SK: scratch slot of a parking unit (empty otherwise)
  const int pkLane = SK ? wave * 65536 + lane * 32 : 0;   // SK: this lane's 32 bytes of a parked pass
  int stLane = 0, colLim = 0;
  auto set_out_tile = [&](int m0, int n0) __attribute__((always_inline)) {
    const int64_t left = ((int64_t)(p.M - m0) * p.ldd - n0) * 2;
    rD = make_rsrc(p.D + ((int64_t)m0 * p.ldd + n0), (uint32_t)(left > 0x7fffffffll ? 0x7fffffffll : left));
    stLane = ((cx.wave_m * C::WTM + rrl) * p.ldd + cx.wave_n * C::WTN + 8 * ccl) * 2;
    // opaque: the 32 store offsets stLane + const * ldd are tile-invariant, and the compiler otherwise keeps all of them in
    // VGPRs across the tile loop (NN: 16 dwords of scratch spills); one v_add per store instead
    asm volatile("" : "+v"(stLane));
    colLim = p.N - n0 - cx.wave_n * C::WTN - 8 * ccl;
  };
  auto retire_write = [&](const int m, const int h) __attribute__((always_inline)) {
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(v4f*)(scr + (scrW ^ (q << 5)) + nn * 128) =
            v4f{acc[m][2 * h + nn][4 * q + 0], acc[m][2 * h + nn][4 * q + 1], acc[m][2 * h + nn][4 * q + 2], acc[m][2 * h + nn][4 * q + 3]};
  };
  v4f rb[2][2];
  auto retire_read = [&](const int half) __attribute__((always_inline)) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      rb[ps][0] = *(const v4f*)(scr + scrR + (2 * half + ps) * 2048);
      rb[ps][1] = *(const v4f*)(scr + (scrR ^ 16) + (2 * half + ps) * 2048);
    }
  };
  auto retire_store = [&](const int m, const int h, const int pass) __attribute__((always_inline)) {
    const v4f lo = rb[pass & 1][0], hi = rb[pass & 1][1];
    v4i o;
    o[0] = (int)pack_bf16x2(lo[0] * alpha, lo[1] * alpha);
    o[1] = (int)pack_bf16x2(lo[2] * alpha, lo[3] * alpha);
    o[2] = (int)pack_bf16x2(hi[0] * alpha, hi[1] * alpha);
    o[3] = (int)pack_bf16x2(hi[2] * alpha, hi[3] * alpha);
    if constexpr (SK) {   // (scalar offset per store, as in the fp4 kernel)
      int ldd2 = p.ldd * 2;
      asm volatile("" : "+s"(ldd2));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), rD, (64 * h < colLim) ? stLane : (int)0x80000000, (32 * m + 8 * pass) * ldd2 + 128 * h, ST_AUX);
    } else {
    const int off = stLane + ((32 * m + 8 * pass) * p.ldd + 64 * h) * 2;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), rD, (64 * h < colLim) ? off : (int)0x80000000, 0, ST_AUX);
    }
    if constexpr (SK) {   // stream-K: the raw read-back to the scratch slot of a parking unit (empty descriptor otherwise), as in the fp4 kernel
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, lo), rP, pkLane, (((2 * m + h) * 4 + pass) * 2048), 17);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, hi), rP, pkLane, (((2 * m + h) * 4 + pass) * 2048) + 16, 17);
    }
  };

  auto final_stage = [&](const Desc& d, bool dvalid, const int ktn) __attribute__((always_inline)) {   // ktn: as in the fp4 kernel
    read_slice(1, 1);
    fence();
    mfma1(0, 1, 0, 0, false); mfma1(0, 1, 0, 1, false);   // slice 0 of tiles 0, 1: covers the latency of R(1)
    fence();
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0), as a builtin: the compiler's wait-count scoreboard sees it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    dma_prep(d, ktn, dvalid);
    fence();
    static_for<0, 37>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s < 30) {
        constexpr int T = s < 2 ? s : 2 + (s - 2) / 2;
        constexpr int j = s < 2 ? 1 : (s - 2) % 2;
        mfma1(j, 1, T / 4, T % 4, false);
      }
      if constexpr (s % 2 == 0 && s / 2 < 9) dma_item(d, ktn, 1, 8 + s / 2);
      if constexpr (s == 1) scales_load(0);
      if constexpr (s == 4) scales_fin(0);
      // slice 0 of the next tile's stage 0, as the registers die: A rows of m after tile (m, 3) (MFMA 8 m + 5), B rows of n after (3, n) (MFMA 23 + 2 n)
      if constexpr (s == 6 || s == 14 || s == 22) read_fa(0, 0, (s - 6) / 8);
      if constexpr (s == 24 || s == 26 || s == 28) read_fb(0, 0, (s - 24) / 2);
      if constexpr (s == 30) { read_fa(0, 0, 3); read_fb(0, 0, 3); }
      constexpr int P4 = deepp8_pair_done_at(s - 4), P6 = deepp8_pair_done_at(s - 6), P1 = deepp8_pair_done_at(s - 1), P2 = deepp8_pair_done_at(s - 2);
      if constexpr (P6 >= 0) { retire_store(P6 / 2, P6 % 2, 2); retire_store(P6 / 2, P6 % 2, 3); }
      if constexpr (P4 >= 0) { retire_store(P4 / 2, P4 % 2, 0); retire_store(P4 / 2, P4 % 2, 1); retire_read(1); }
      if constexpr (P1 >= 0) retire_write(P1 / 2, P1 % 2);
      if constexpr (P2 >= 0) retire_read(0);
      fence();
    });
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the retirement reads of the scratch slice (and NN: the asm fragment reads)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_item(d, ktn, 1, i);
    fence();
  };

  // ---- [r5] the last stage with the bf16-first retirement (final_stage16 of the fp4 kernel has the layout and the reasons).  An fp8 MFMA is 64 cycles, so the
  //      64 pieces of a tile ride two per slot: piece pi = 4 T + q in slot pi / 2 + 1 (tile T >= 2 is final after slot 2 T - 1, tiles 0 / 1 after slots 0 / 1); pair P is
  //      read back in slots 4 P + 5 / + 6 and stored in 4 P + 7 / + 8; pair P + 2 writes the same half of the scratch from slot 4 P + 9 on.
  const int r16w = (i32 >> 1) * 256 + (i32 & 1) * 64 + ((g ^ ((i32 >> 3) & 1)) << 3);
  const int r16x = ((i32 >> 1) & 3) << 4;
  const int r16r = (rrl >> 1) * 256 + (ccl >> 2) * 128 + (rrl & 1) * 64 + (((ccl & 3) ^ (rrl >> 1)) << 4);
  v4i rbh[4];
  auto final_stage16 = [&](const Desc& d, bool dvalid, const int ktn) __attribute__((always_inline)) {
    // MFMA order: blocks of IL tiles, k-slice-major inside a block (fs8_* above; why: the comment at fs_T).  Two pieces per 64-cycle slot: piece pi = 4 T + q in slot
    // pi / 2 + D; pair P: pieces in slots 4 P + D .. + 3, read back in 4 P + D + 4 / + 5, stored in + 6 / + 7.
    constexpr int IL = QAMD_DEEPP_FS_IL, NMF = 32 - IL, D = IL / 2 + 2, NS = 36 + D;
    static_assert(IL == 1 || IL == 2 || IL == 4, "block of the last stage: 1, 2 or 4 tiles");
    read_slice(1, 1);
    fence();
#pragma unroll
    for (int t = 0; t < IL; ++t) mfma1(0, 1, t / 4, t % 4, false);   // slice 0 of block 0: covers the latency of R(1)
    fence();
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0), as a builtin: the compiler's wait-count scoreboard sees it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    dma_prep(d, ktn, dvalid);
    int wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[q] = r16w + (r16x ^ (q << 4));
    int r16r8 = r16r + 8;   // (odd passes fetch their swapped halves through a second, opaque base: two adjacent 8-byte loads would be merged into one ds_read_b128 + v_mov_b32 x 4)
    asm volatile("" : "+v"(r16r8));
    int ldd2 = p.ldd * 2;
    asm volatile("" : "+s"(ldd2));
    fence();
    auto piece = [&](const int pi) __attribute__((always_inline)) {
      const int T = pi >> 2, q = pi & 3, m = T >> 2, n = T & 3;
      float x0 = acc[m][n][4 * q + 0], x1 = acc[m][n][4 * q + 1], x2 = acc[m][n][4 * q + 2], x3 = acc[m][n][4 * q + 3];
      asm("v_mul_f32 %0, %1, %2" : "=v"(x0) : "s"(alpha), "v"(x0));
      asm("v_mul_f32 %0, %1, %2" : "=v"(x1) : "s"(alpha), "v"(x1));
      asm("v_mul_f32 %0, %1, %2" : "=v"(x2) : "s"(alpha), "v"(x2));
      asm("v_mul_f32 %0, %1, %2" : "=v"(x3) : "s"(alpha), "v"(x3));
      *(v2i*)(scr + ((T >> 1) & 1) * 4096 + (T & 1) * 128 + wq[q]) = v2i{(int)pack_bf16x2(x0, x1), (int)pack_bf16x2(x2, x3)};
    };
    auto readback = [&](const int P, const int pass) __attribute__((always_inline)) {
      const int o = (P & 1) * 4096 + pass * 1024;
      if (pass & 1) {
        const v2i lo = *(const v2i*)(scr + o + r16r8), hi = *(const v2i*)(scr + o + r16r);
        rbh[pass] = v4i{lo[0], lo[1], hi[0], hi[1]};
      } else {
        rbh[pass] = *(const v4i*)(scr + o + r16r);
      }
    };
    auto store16 = [&](const int P, const int pass) __attribute__((always_inline)) {   // (+ a wait state: see the fp4 kernel's store16)
      const int m = P >> 1, h = P & 1;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, rbh[pass]), rD, (64 * h < colLim) ? stLane : (int)0x80000000, (32 * m + 8 * pass) * ldd2 + 128 * h, ST_AUX);
      asm volatile("s_nop 0" : "+v"(rbh[pass]) :: "memory");   // (the data registers stay claimed up to here: nothing else can be allocated into them, i.e. written, before the wait state)
    };
    static_for<0, NS>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s < NMF) {
        mfma1(fs8_j(IL, s), 1, fs8_T(IL, s) / 4, fs8_T(IL, s) % 4, false);
        fence();
      }
      if constexpr (s % 2 == 0 && s / 2 < 9) dma_item(d, ktn, 1, 8 + s / 2);
      if constexpr (s == 1) scales_load(0);
      if constexpr (s == 4) scales_fin(0);
      // slice 0 of the next tile's stage 0, as the registers die: A rows of m behind the last MFMA of tile (m, 3), B rows of n behind that of tile (3, n)
      static_for<0, 4>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        if constexpr (s == fs8_F(IL, 4 * m + 3) + 1) read_fa(0, 0, m);
        if constexpr (s == fs8_F(IL, 12 + m) + 1) read_fb(0, 0, m);
      });
      if constexpr (s >= D + 6 && (s - D - 6) % 4 == 0 && (s - D - 6) / 4 < 8) { store16((s - D - 6) / 4, 0); store16((s - D - 6) / 4, 1); }
      if constexpr (s >= D + 7 && (s - D - 7) % 4 == 0 && (s - D - 7) / 4 < 8) { store16((s - D - 7) / 4, 2); store16((s - D - 7) / 4, 3); }
      if constexpr (s >= D && s < D + 32) { piece(2 * (s - D)); piece(2 * (s - D) + 1); }
      if constexpr (s >= D + 4 && (s - D - 4) % 4 == 0 && (s - D - 4) / 4 < 8) { readback((s - D - 4) / 4, 0); readback((s - D - 4) / 4, 1); }
      if constexpr (s >= D + 5 && (s - D - 5) % 4 == 0 && (s - D - 5) / 4 < 8) { readback((s - D - 5) / 4, 2); readback((s - D - 5) / 4, 3); }
      fence();
    });
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the retirement reads of the scratch slice (and NN: the asm fragment reads)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_item(d, ktn, 1, i);
    fence();
  };
  auto last_stage = [&](const Desc& d, bool dvalid, const int ktn) __attribute__((always_inline)) {
    if constexpr (!SK && QAMD_DEEPP8_RETIRE != 0) final_stage16(d, dvalid, ktn);
    else final_stage(d, dvalid, ktn);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using BT = std::integral_constant<bool, true>;
  using BF = std::integral_constant<bool, false>;
  if constexpr (SK) {
    // ---- stream-K walk (gemm_mx_deepp above has the comments).  ONE instruction path for every kind of unit -- at 255 of 256 vector registers every join
    //      of two differently scheduled code paths (a skipped stage, a first stage with / without zero accumulators, a last stage with / without the
    //      epilogue) turned into register copies and 100+ spills inside the hand-scheduled stages; the kinds differ in DATA only:
    //        * the accumulators are set BEFORE the first stage -- 16 MFMAs on zero operands, or, for the unit that owns the FIRST K stages of a cut
    //          tile, 64 loads of the part its neighbour parked -- and the first stage accumulates like any other;
    //        * every unit ends with the accumulator-stationary last stage; it retires to D, and, through the same read-back registers, to the
    //          scratch slot rP, one of which is an empty descriptor (a parking unit writes no D, the others park nothing).
    SkWalk walk(wg, G, ntiles, p.sk_tiles, KTe, 2, 2);
    auto next_unit = [&]() __attribute__((always_inline)) {
      SkUnit u = walk.next();
      u.tile = uniform(u.tile); u.kb = uniform(u.kb); u.ke = uniform(u.ke); u.mode = uniform(u.mode); u.slot = uniform(u.slot);
      return u;
    };
    auto slot_rsrc = [&](const int slot, const bool on) __attribute__((always_inline)) {
      return make_rsrc((const char*)p.ws + (size_t)slot * SK_PART_BYTES, on ? (uint32_t)SK_PART_BYTES : 0u);
    };
    // accumulators of the unit that starts now: zero (one MFMA per 32x32 tile on zero operands: 16 instructions in the idle matrix pipe instead of
    // 256 v_accvgpr_write), or the parked part of the cut tile -- row-major pairs (retire_store's layout) gathered into the accumulator layout:
    // row i32 of pair (m, h), columns 32 nn + 8 q + 4 g  ->  ((wave 8 + pair) 4 + i32 / 8) 2 KiB + ((i32 % 8) 8 + 4 nn + q) 32 B + 16 g
    auto init_acc = [&](const bool from_slot, const int slot) __attribute__((always_inline)) {
      v8i z = {};
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          asm volatile("" : "+v"(z));   // (a fresh opaque value per tile: identical MFMAs are otherwise merged into one + 240 register copies)
          acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(z, z, v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, C::AFMT, 0, 0, 0, 0);
        }
      if (from_slot) {   // (parked at the START of its owner's walk: the flag is long set)
        const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (wave == 0 && elane == 0)
          while (__hip_atomic_load(p.ctr + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.tag) __builtin_amdgcn_s_sleep(4);
        __builtin_amdgcn_s_barrier();
        const __amdgpu_buffer_rsrc_t rW = slot_rsrc(slot, true);
        const int ei32 = elane & 31, eg = elane >> 5;
        const int ul = wave * 65536 + (ei32 >> 3) * 2048 + (ei32 & 7) * 256 + eg * 16;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rW, ul, ((2 * m + n / 2) * 4) * 2048 + (4 * (n & 1) + q) * 32, 17));
              acc[m][n][4 * q + 0] = v[0]; acc[m][n][4 * q + 1] = v[1]; acc[m][n][4 * q + 2] = v[2]; acc[m][n][4 * q + 3] = v[3];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0 && elane == 0) __hip_atomic_store(p.ctr + slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // a replayed graph (same tag) starts clean
      }
    };
    // (the walk runs one unit ahead; of a unit only what the stages need stays in scalar registers: its stage range and kind -- the tile went into
    //  the offsets / the output descriptors when the unit was fetched, the scratch slot follows from the kind)
    auto set_out_unit = [&](const int tile, const int mode) __attribute__((always_inline)) {
      int m0, n0;
      decode(tile, m0, n0);
      set_out_tile(m0, n0);
      if (mode == 1) rD = make_rsrc(p.D, 0);    // parking: nothing goes to D
      rP = slot_rsrc(wg, mode == 1);            // ... the raw sums go to this workgroup's slot
    };
    SkUnit u0 = next_unit();
    if (u0.mode < 0) return;
    Desc dcur = make_desc(u0.tile);
    int kb = u0.kb, ke = u0.ke, mode = u0.mode;
    set_out_unit(u0.tile, mode);
    SkUnit u1 = next_unit();
    Desc dnxt = make_desc(u1.mode >= 0 ? u1.tile : ntiles);
    int nkb = u1.kb, nke = u1.ke, nmode = u1.mode, ntile = u1.tile;
    dma_stage(dcur, kb, true, 0);
    dma_stage(dcur, kb + 1, true, 1);
    asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    read_scales(0, 0);
    read_slice(0, 0);
    fence();
    while (mode >= 0) {
      const bool nvalid = nmode >= 0;
      // the stage two ahead of stage s - 2 of this unit: stage s, or stage kb' + (s - ke) of the next unit
      auto ahead = [&](const int s, Desc& d, int& kt, bool& valid) __attribute__((always_inline)) {
        const bool tonext = s >= ke;
        d.ao = tonext ? dnxt.ao : dcur.ao; d.bo = tonext ? dnxt.bo : dcur.bo; d.so = tonext ? dnxt.so : dcur.so;
        d.mrem = 0;
        kt = tonext ? nkb + (s - ke) : s;
        valid = tonext ? nvalid : true;
      };
      Desc d;
      int ktl;
      bool dv;
      init_acc(mode == 2, wg + 1);   // (the part of a cut tile is parked by the NEXT workgroup of the walk, in its own slot)
      ahead(kb + 2, d, ktl, dv);
      stage(I0{}, BF{}, d, ktl, dv);
      for (int kt = kb + 1; kt + 2 < ke; kt += 2) {
        ahead(kt + 2, d, ktl, dv);
        stage(I1{}, BF{}, d, ktl, dv);
        ahead(kt + 3, d, ktl, dv);
        stage(I0{}, BF{}, d, ktl, dv);
      }
      final_stage(dnxt, nvalid, nkb + 1);
      if (mode == 1) {   // parked: acknowledged by the coherence point, then the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (wave == 0 && elane == 0) __hip_atomic_store(p.ctr + wg, p.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // the next unit becomes the current one; fetch the one after it
      kb = nkb; ke = nke; mode = nmode;
      dcur = dnxt;
      if (mode >= 0) set_out_unit(ntile, mode);
      u1 = next_unit();
      dnxt = make_desc(u1.mode >= 0 ? u1.tile : ntiles);
      nkb = u1.kb; nke = u1.ke; nmode = u1.mode; ntile = u1.tile;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  int tile = wg;
  Desc cur = make_desc(tile);
  dma_stage(cur, 0, true, 0);
  dma_stage(cur, 1, true, 1);
  asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  fence();
  read_scales(0, 0);
  read_slice(0, 0);
  nn_wait();
  fence();

  while (tile < ntiles) {
    int m0, n0;
    decode(tile, m0, n0);
    set_out_tile(m0, n0);
    const int tnext = tile + G;
    const Desc nxt = make_desc(tnext);
    const bool nvalid = tnext < ntiles;
    {
      const bool tonext = KTe == 2;
      Desc d;
      d.a = tonext ? nxt.a : cur.a; d.b = tonext ? nxt.b : cur.b; d.s = tonext ? nxt.s : cur.s;
      d.mrem = tonext ? nxt.mrem : cur.mrem;
      stage(I0{}, BT{}, d, tonext ? 0 : 2, tonext ? nvalid : true);
    }
    for (int kt = 1; kt + 2 < KTe; kt += 2) {
      stage(I1{}, BF{}, cur, kt + 2, true);
      const bool tonext = kt + 3 == KTe;
      Desc d;
      d.a = tonext ? nxt.a : cur.a; d.b = tonext ? nxt.b : cur.b; d.s = tonext ? nxt.s : cur.s;
      d.mrem = tonext ? nxt.mrem : cur.mrem;
      stage(I0{}, BF{}, d, tonext ? 0 : kt + 3, tonext ? nvalid : true);
    }
    last_stage(nxt, nvalid, 1);
    cur = nxt;
    tile = tnext;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class C, int ST_AUX = 0, bool NN = false, int NNABL = 0, bool SK = false>
__global__ __launch_bounds__(C::THREADS) void gemm_mx_deepp8_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[DeepPCfg<C>::LDS_BYTES];
  gemm_mx_deepp8<C, ST_AUX, NN, NNABL, SK>(smem, p, (int)blockIdx.x, (int)gridDim.x, p.tiles_m * p.tiles_n);
}

template <class C, bool TRACE = false, int ST_AUX = 0, int LAB = 0, bool SK = false>
__global__ __launch_bounds__(C::THREADS) void gemm_mx_deepp_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[DeepPCfg<C>::LDS_BYTES];
#if QAMD_KERNARG_EARLY
  // every argument the prologue needs is asked for HERE: the scalar loads leave together and are waited for once (left alone they arrive in four dependent rounds)
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"((int)gridDim.x));
#endif
  labk::gemm_mx_deepp<C, TRACE, ST_AUX, LAB, SK>(smem, p, (int)blockIdx.x, (int)gridDim.x, p.tiles_m * p.tiles_n);
}

// -------------------------------------------------------------------------------------------------------------------------
// Heterogeneous launch ("residual-round scheduler"): ONE grid, two kinds of workgroups.
//   blockIdx.x <  g_big : the persistent 256x256 kernel above over the first t_main tiles of the grouped raster (t_main = a whole
//                         number of rounds of g_big tiles: every persistent workgroup walks the same number of tiles)
//   blockIdx.x >= g_big : one 128x128 tile each (pipelined ring schedule, gemm_mx_ringp) of the RESIDUAL 256x256 tiles
//                         t_main .. T-1, four per residual tile
// Every workgroup claims the kernel's whole static LDS (one per CU), so the residual workgroups are dispatched CU by CU as the
// persistent ones retire: the part-filled last round of a ragged tile count (320 tiles on 256 CUs = 1.25 rounds) turns into a short
// wave of quarter tiles that starts as soon as the first CUs are free -- no second launch (its launch gap and the serialisation
// behind the slowest persistent workgroup), no K split, no partial sums, no scratch: every output element is still computed by
// exactly one workgroup in the same K order, so the result is bit-identical to every other schedule.
// Reference counterpart: the M-bucketed tile choice + CUTLASS tile scheduler of qutlass/csrc/gemm.cu:195-222.
// -------------------------------------------------------------------------------------------------------------------------
template <class CB, class CT, int ST_AUX = 0>
__global__ __launch_bounds__(256) void gemm_mx_hetero_kernel(const GemmParams p, const int g_big, const int t_main) {
  static_assert(CB::THREADS == 256 && CT::THREADS == 256 && CT::BM == 128 && CT::BN == 128 && CB::EBITS == CT::EBITS && CB::AFMT == CT::AFMT, "tile pair");
  constexpr int LDS = DeepPCfg<CB>::LDS_BYTES > CT::LDS_BYTES ? DeepPCfg<CB>::LDS_BYTES : CT::LDS_BYTES;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  // (device pass only: the host pass has already instantiated the same gemm_mx_deepp specialisation for the plain kernel, and clang
  // marks a __device__ specialisation whose body holds target builtins as invalid for every later host-side reference)
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char smem[LDS];
#if QAMD_KERNARG_EARLY
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"(g_big), "s"(t_main));   // (all scalar argument loads in one round, as in gemm_mx_deepp_kernel)
#endif
  const int b = (int)blockIdx.x;
  if (b < g_big) {
    // (DMA_SPREAD = 1 since the output stores carry their row offsets as scalars (QAMD_DEEPP_SOFF): before that, with the residual-tile path in the same kernel,
    //  the spread order alone cost this kernel one spilled register and the piece offsets were recomputed mid-stage, DMA_SPREAD = 2)
    if constexpr (CB::EBITS == 4) gemm_mx_deepp<CB, false, ST_AUX, 0, false, 1>(smem, p, b, g_big, t_main);
    else gemm_mx_deepp8<CB, ST_AUX>(smem, p, b, g_big, t_main);
    return;
  }
  // residual quarter tile j: the dispatcher hands workgroup b to XCD b % 8, so ids are shifted by g_big % 8 (mod the count) before
  // the XCD-contiguous remap -- each XCD then works on a contiguous run of residual tiles (shared operand panels in its L2)
  const int nsmall = 4 * (p.tiles_m * p.tiles_n - t_main);
  int j = (b - g_big + (g_big & 7)) % nsmall;
  j = xcd_remap(j, nsmall);
  const int t = t_main + (j >> 2);
  int tm, tn;             // the grouped raster of the persistent kernel (decode)
  raster_decode(t, p.tiles_m, p.tiles_n, p.raster_magic, tm, tn);
  const int m0 = uniform(tm * CB::BM + ((j >> 1) & 1) * 128);
  const int n0 = uniform(tn * CB::BN + (j & 1) * 128);
  if (m0 >= p.M || n0 >= p.N) return;   // quarter of a partial edge tile that lies outside the output
  gemm_mx_ringp<CT>(smem, p, 0, m0, n0);
#endif
}

}  // namespace labk
}  // namespace qamd
