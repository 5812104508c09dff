// Persistent "deep" MXFP4 GEMM for gfx950: the 4-wave 128x128-per-wave schedule of gemm_mx_deep (gemm_mx.hip.h) as a
// tile LOOP with the epilogue folded into the last K stage.  Replaces the per-tile launch of
// qutlass/csrc/gemm.cu:174-248 (matmul_host_mxf4_bf16_tn) for outputs of >= 192 tiles of 256x256.
//
// What changes against gemm_mx_deep (DESIGN.md section 3.4b has the measurements):
//   * One workgroup per CU walks tiles t = wg, wg + G, ... (G = grid size).  The LDS-DMA stream never stops at a tile
//     boundary: stage kt of a tile issues the DMA of stage kt + 2, and for the last two stages of a tile that is stage
//     0 / 1 of the NEXT tile (descriptor selected with scalar selects -- the K loop stays branch-free).  The ~2 us a
//     fresh workgroup spends waiting for its first two stages is paid once per launch, not once per tile.
//   * The LAST K stage of a tile runs accumulator-stationary: all four k-slices are in registers, so the MFMA order is
//     (m, n)-major and an accumulator tile is final after 4 MFMAs.  Each retired 32x32 tile goes through a WAVE-PRIVATE
//     4-KiB LDS scratch (ds_write_b128 straight from the accumulator registers, 16-byte chunks XOR-swizzled by row) and
//     comes back row-major: 8 fp32 per lane -> alpha, v_cvt_pk_bf16_f32 -> one 16-byte store, 16 rows x 64 B per wave
//     instruction.  No workgroup barrier, no second pass over the tile: the stores of tile (m, n) issue in the MFMA
//     shadows of tile (m, n + 1), and the DMA of the next tile's stage 1 + the fragment reads of its stage 0 are threaded
//     through the same stage.
//   * The first stage of a tile starts its accumulators from the inline constant 0 (no zeroing pass).
//   * KT (K stages of 256 elements) is rounded up to even with an all-zero stage (out-of-range DMA offsets load zeros),
//     so every tile starts in LDS buffer 0 and the stage code is unrolled by parity exactly once.
// Same products, same K order per output as every other schedule: bit-identical results.
#pragma once
#include "gemm_mx.hip.h"

namespace qamd {

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

// ST_AUX: cache-policy bits of the output stores (buffer_store aux: 1 = sc0, 2 = nt, 16 = sc1; the product: 17 = write-through).
// bid / G: this workgroup's id among the G persistent workgroups; ntiles: the tiles they walk (tiles 0 .. ntiles-1 of the grouped
// raster).  A plain launch passes blockIdx.x / gridDim.x / all tiles; the heterogeneous launch (gemm_mx_hetero_kernel, end of this
// file) gives the persistent workgroups the full rounds and runs the residual tiles as 128x128 tiles on other workgroups.
// DMA_SPREAD: 1 = the product; 2 recomputes the piece offsets mid-stage (one register less; kept for the heterogeneous kernel should it need it again).
// (Stage traces, timing ablations, the stream-K walk and the retirement experiments of rounds 4-5 live in lab/gemm_mx_deepp_lab.hip.h, lab build only.)
// [r6] ODD: for an ODD number of K stages (>= 3) the tile walk runs without the empty stage -- a tile then starts in the buffer its predecessor's last stage did not use,
// so consecutive tiles of a workgroup alternate their starting buffer and the last stage exists for either buffer (the tile loop is unrolled twice).  Chosen by the host
// (K / 256 is known there): the even kernel is unchanged.  K = 11008 (43 stages, qutlass's own test list tests/mxfp4_test.py:194-199): one stage of 44 saved.
// [r6] ONETILE: every workgroup walks exactly one tile (grid == tile count: 4096^3 on 256 CUs) -- chosen by the host.  Nothing follows a tile then, so everything the
// walk does FOR THE NEXT TILE is compiled out: the 17 LDS-DMA items of "its stage 0" (stage KT - 2) and "its stage 1" (last stage) -- out-of-range descriptors, zero fill,
// but each still takes its issue slot and its pass through the address unit -- and the 34 LDS reads of "its first fragments" in the last stage.  (As a run-time arm of
// the one kernel the second copy of the last stage cost 107 spilled registers; as its own instantiation the kernel needs 176 + 256.)
template <class C, int ST_AUX = 0, int DMA_SPREAD = 1, bool ODD = false, bool ONETILE = false>
__device__ __forceinline__ void gemm_mx_deepp(char* smem, const GemmParams& p, const int bid, const int G, const int ntiles) {
  static_assert(C::EBITS == 4 && C::BM == 256 && C::BN == 256 && C::WAVES_M == 2 && C::WAVES_N == 2 && C::NSTAGE == 2 && C::PPW == 2,
                "persistent deep schedule: fp4, 256x256 tiles, 4 waves of 128x128");
  constexpr int MT = 4, NT = 4;
  constexpr int STAGE = C::STAGE_BYTES, OFF_SCR = DeepPCfg<C>::OFF_SCR;
  GemmCtx<C> cx(smem, p);   // per-lane offsets / LDS addresses; its tile coordinates and descriptors are NOT used here
  const int lane = cx.lane, wave = cx.wave, i32 = cx.i32, g = cx.g;
  const int KT = cx.KT, KTe = ODD ? KT : (KT + 1) & ~1, CB = cx.CB, rowbytes = cx.rowbytes;
  const int wg = xcd_remap(bid, G);

  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  // ---- tile -> (m0, n0): rounds of G tiles, each round XCD-contiguous, grouped raster of 4 tile rows --------------------
  auto decode = [&](int t, int& m0, int& n0) __attribute__((always_inline)) {
    int tm, tn;
    raster_decode(t, p.tiles_m, p.tiles_n, p.raster_magic, tm, tn);
    m0 = uniform(tm * C::BM);
    n0 = uniform(tn * C::BN);
  };
  struct Desc { __amdgpu_buffer_rsrc_t a, b, s; };
  // operand descriptors of tile t (t >= ntiles: empty descriptors -> every DMA of that "tile" loads zeros)
  auto make_desc = [&](int t) __attribute__((always_inline)) {
    const bool valid = t < ntiles;
    int m0, n0;
    decode(valid ? t : ntiles - 1, m0, n0);
    const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
    const uint32_t sa_off = (uint32_t)(m0 >> 7) * CB * 512, sb_off = (uint32_t)(n0 >> 7) * CB * 512;
    Desc d;
    d.a = make_rsrc(p.A + a_off, valid ? p.a_bytes - a_off : 0u);
    d.b = make_rsrc(p.B + b_off, valid ? p.b_bytes - b_off : 0u);
    d.s = cx.sIsB ? make_rsrc(p.SFB + sb_off, valid ? p.sfb_bytes - sb_off : 0u) : make_rsrc(p.SFA + sa_off, valid ? p.sfa_bytes - sa_off : 0u);
    return d;
  };

  // ---- registers -------------------------------------------------------------------------------------------------------
  v16f acc[MT][NT];
  v4i fa[4][MT] = {}, fb[4][NT] = {};
  // [r6] the scale dwords of a stage, one 16-byte LDS read per operand: a lane's four row fragments (rows 32 t + i32 of the wave's 128) have their dwords in ONE 16-byte
  // line of the to_blocked image (32 lanes x 16 B: conflict-free).  Rounds 3-5 read them as eight ds_read_b32 at a 16-byte lane stride -- a 4-way bank conflict each,
  // which is what SQ_LDS_BANK_CONFLICT counted in the K loop (192 cycles per stage and CU = 22 % of the LDS-active cycles, profiles/rocprof_pmc_gemm_r5a_mxfp4_4096.txt).
  static_assert(MT == 4 && NT == 4, "one ds_read_b128 = the scale dwords of the wave's four row fragments");
  v4i sa[2], sb[2];

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
    sa[set] = *(const v4i*)(st + cx.rdSA[0]);
    sb[set] = *(const v4i*)(st + cx.rdSB[0]);
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
      __builtin_amdgcn_raw_ptr_buffer_load_lds(item < 8 ? d.a : d.b, (lds_ptr_t)(st + (item < 8 ? 0 : C::OFF_B) + q * 1024), 16, v, kt * C::ROWB, 0, QAMD_DMA_AUX);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d.s, (lds_ptr_t)(st + C::OFF_S + wave * 1024), 16, vbS, kt * C::SCT * 512, 0, 0);
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
  auto stage = [&](auto bufc, auto firstc, const Desc& d, int ktl, bool dvalid, auto dmac) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    constexpr bool DMA = decltype(dmac)::value;   // false (ONETILE, stage KT - 2): the stage's DMA would be the next tile's stage 0
    // [r3] ONE fragment read behind each MFMA instead of a burst of 8 in front of 16 MFMAs.  With one wave per SIMD the wave's own program order
    // is all that can put a read into an MFMA's shadow: after a run of MFMAs the matrix pipe drains while the 8 reads issue -- 82 cycles per 8
    // MFMAs (tests/native/ubench.hip "uinter": 8 MFMA + 6 reads in bursts 350 cycles, interleaved 276, MFMAs alone 268; on quantised-Gaussian
    // operands, where the clock is held back electrically, 187.5 -> 175.9 ns, and with the LDS-DMA in the mix 200.8 -> 189.3 ns = -6 %;
    // 200.8 ns x 8 = the 1.62 us this stage took).
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
    auto read_scale1 = [&](const int buf, const int set, const int i) __attribute__((always_inline)) {   // i = 0: the A operand's dwords, 1: B's
      const char* st = smem + buf * STAGE;
      if (i == 0) sa[set] = *(const v4i*)(st + cx.rdSA[0]);
      else sb[set] = *(const v4i*)(st + cx.rdSB[0]);
    };
    static_assert(MT + NT <= MT * NT / 2, "a fragment read behind each of the first MT + NT MFMAs, the two scale reads behind the next two");
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
    read_base(BUF, 2);
    group(0, FIRST, [&](const int i) __attribute__((always_inline)) { if (i < MT + NT) read_frag(2, i); });
    read_base(BUF, 3);
    // [r4] EARLYPREP: the stage's DMA offsets and the next slice's base addresses are computed in the shadow of group 1's last MFMAs instead of between the
    // barrier and group 2's first MFMA (neither depends on the hand-off)
    group(1, false, [&](const int i) __attribute__((always_inline)) {
      if (i < MT + NT) read_frag(3, i);
      if (DMA && i == 12) dma_prep(ktl, dvalid);
      if (i == 14) read_base(BUF ^ 1, 0);
    });
    // ([r5] With the whole chip streaming every wave waits ~300 - 440 cycles here for its own pieces (tools/handoff_trace.py, profiles/handoff_trace_r5q.txt; 8 cycles with 8
    //  workgroups).  Splitting the hand-off in two -- refill issued 12 slots earlier behind its own barrier, `vmcnt(12)` here -- removes that wait and makes the kernel 1 - 3 %
    //  SLOWER, same bytes: the launch sits at the socket power limit, and cycles the matrix pipe idles come back as clock (profiles/ab_lib_r5r_two_handoffs_per_stage_negative.txt;
    //  the variant lives in the lab copy, QAMD_DEEPP_SPLITB).)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMA of the next stage landed; own reads of this buffer done
    __builtin_amdgcn_s_barrier();
    fence();
    // [r4] the stage's 17 DMA items ride in the slots that have no fragment read: second half of group 2 (items 0 .. 7 + the scale piece) and second half
    // of group 3 (items 8 .. 15) -- all 17 behind the first 16 MFMAs after the hand-off put three auxiliary instructions into each of eight slots
    group(2, false, [&](const int i) __attribute__((always_inline)) {
      if (!DMA) {}
      else if (!DMA_SPREAD) { dma_item(d, ktl, BUF, i); if (i == 0) dma_item(d, ktl, BUF, 16); }
      else if (i >= 8) { dma_item(d, ktl, BUF, i - 8); if (i == 8) dma_item(d, ktl, BUF, 16); }
      // the next stage's slice 0 (set 0 went dead with group 0) and its scale dwords
      if (i < MT + NT) read_frag(0, i);
      else if (i < MT + NT + 2) read_scale1(BUF ^ 1, BUF ^ 1, i - (MT + NT));
    });
    read_base(BUF ^ 1, 1);
    group(3, false, [&](const int i) __attribute__((always_inline)) {
      if (i < MT + NT) read_frag(1, i);
      if (DMA && DMA_SPREAD == 2 && i == 8) dma_prep(ktl, dvalid);   // (2: the piece offsets are recomputed here instead of staying live through this group's fragment reads -- one register less,
                                                               //  which the heterogeneous kernel needs; costs the plain kernel ~1 %, profiles/ab_lib_gemm_r4at_reprep.txt)
      if (DMA && DMA_SPREAD && i >= 8) dma_item(d, ktl, BUF, i);
    });
    if constexpr (FIRST) pin_acc();
  };

  // ---- epilogue pieces: a retired PAIR of accumulator tiles (m, 2h), (m, 2h + 1) = 32 rows x 64 columns through the
  //      wave-private scratch (fp32, 32 rows x 256 B; 16-byte chunk c = 8 n' + 2 q + g of row r stored at chunk
  //      (c & 8) | ((c & 7) ^ (r & 7)): the 8 lanes of a ds_write_b128 group hit 8 different chunks, the 16 lanes of a
  //      ds_read_b128 group 16 different ones).  Read-back is row-major: lane -> row 8 p + l / 8, columns 8 (l % 8) .. + 7, so
  //      8 lanes cover one whole 128-byte line of D and a wave instruction stores 8 rows x 128 B.
  char* scr = smem + OFF_SCR + wave * DeepPCfg<C>::SCR_PER_WAVE;   // (the last stage's buffer; ODD: re-pointed per tile by final_stage)
  const int scrW = i32 * 256 + ((((i32 & 6) << 4)) | ((g ^ (i32 & 1)) << 4));   // chunk (2q + g) ^ (row & 7) = this ^ (q << 5); + 128 n'
  const int rrl = lane >> 3, ccl = lane & 7;                                     // read-back: row rrl (+ 8 per pass), columns 8 ccl .. + 7
  const int scrR = rrl * 256 + (ccl >> 2) * 128 + ((((2 * ccl) & 7) ^ (rrl & 7)) << 4);   // chunk 2 ccl of that row; chunk 2 ccl + 1 = this ^ 16
  const float alpha = *p.alpha;
  __amdgpu_buffer_rsrc_t rD = make_rsrc(p.D, 0);
  int stLane = 0, colLim = 0;
  // output descriptor of the tile at (m0, n0): base = its first element, range = what is left of D from there (capped at
  // 2 GiB: a tile spans < 2^31 bytes, launch code rejects wider rows), so rows >= M fall out of range by themselves;
  // columns >= N are pushed out of range per lane.
  auto set_out_tile = [&](int m0, int n0) __attribute__((always_inline)) {
    const int64_t left = ((int64_t)(p.M - m0) * p.ldd - n0) * 2;
    rD = make_rsrc(p.D + ((int64_t)m0 * p.ldd + n0), (uint32_t)(left > 0x7fffffffll ? 0x7fffffffll : left));
    stLane = ((cx.wave_m * C::WTM + rrl) * p.ldd + cx.wave_n * C::WTN + 8 * ccl) * 2;
    colLim = p.N - n0 - cx.wave_n * C::WTN - 8 * ccl;   // column 64 h + 8 ccl of the wave tile exists iff 64 h < colLim
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
  auto retire_read = [&](const int half) __attribute__((always_inline)) {   // rows 16 half .. + 15: passes 2 half, 2 half + 1
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      rb[ps][0] = *(const v4f*)(scr + scrR + (2 * half + ps) * 2048);
      rb[ps][1] = *(const v4f*)(scr + (scrR ^ 16) + (2 * half + ps) * 2048);
    }
  };
  // (a VALU write of a store's data registers must not sit DIRECTLY behind the store: on gfx950 that corrupts a 16-byte buffer store also when it carries an SGPR offset,
  //  which the compiler does not guard -- tests/native/store_hazard_probe.hip, profiles/store_hazard_probe_r5.txt; tools/store_data_hazard.py scans the ISA, a CPU test)
  auto retire_store = [&](const int m, const int h, const int pass) __attribute__((always_inline)) {
    const v4f lo = rb[pass & 1][0], hi = rb[pass & 1][1];
    // ([r6] measured and not adopted: eight plain v_mul_f32 kept apart from the converts instead of the v_pk_mul_f32 pairs the vectoriser makes of these -- same time on
    //  zero and random operands, profiles/ab_lib_r6k_onetile_and_scalar_multiplies.txt)
    v4i o;
    o[0] = (int)pack_bf16x2(lo[0] * alpha, lo[1] * alpha);
    o[1] = (int)pack_bf16x2(lo[2] * alpha, lo[3] * alpha);
    o[2] = (int)pack_bf16x2(hi[0] * alpha, hi[1] * alpha);
    o[3] = (int)pack_bf16x2(hi[2] * alpha, hi[3] * alpha);
    // (the wave-uniform part of the address rides in the scalar offset, recomputed per store -- as vector offsets the 32 sums stLane + k ldd are precomputed per
    //  tile and stay live across the K loop: the 20 registers whose absence kept the scalar argument loads from being requested in one round, profiles/ab_lib_gemm_r4bh_*)
    int ldd2 = p.ldd * 2;
    asm volatile("" : "+s"(ldd2));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), rD, (64 * h < colLim) ? stLane : (int)0x80000000, (32 * m + 8 * pass) * ldd2 + 128 * h, ST_AUX);
  };

  // ---- the LAST stage of a tile (buffer 1), accumulator-stationary, with the tile's epilogue, the DMA of the next tile's
  //      stage 1 (d) and the fragment / scale reads of the next tile's stage 0 (buffer 0, scale set 0) threaded through.
  // MFMA sequence after the hand-off: tiles T = 0..15 in (m, n) row-major order, T0 and T1 with slices 2, 3 only (their
  // slices 0, 1 ran before the hand-off to cover the latency of R(2), R(3)), every later tile with slices 0..3.  Pair
  // P = 0..7 (tiles 2P, 2P + 1) is final after post-hand-off MFMA e = 8 P + 3; its retirement in the shadows of later MFMAs:
  //   write e+1 | read rows 0-15 e+3 | stores e+5, e+6 | read rows 16-31 e+7 | stores e+9, e+10
  // One scratch and one read-back register set per wave: write(P + 1) at e + 9 follows read(P) at e + 7, read(P + 1) at
  // e + 11 follows the last store of P at e + 10 (within a slot: stores, then write, then read).
  // ktn: the stage of the next tile whose DMA is threaded through here -- its second one.
  // WHAT THIS STAGE COSTS ([r5], profiles/final_stage_ablation_r5c.txt, issue_ubench*_r5*.txt): ~8 400 - 9 000 cycles per tile against 2 048 of MFMA.  With one wave per SIMD at most
  // ~5 other instructions hide behind a 32-cycle MFMA; the retirement adds ~7 per slot, its ds_write_b128 from the accumulator registers cost 52 cycles of LDS store path
  // each with four waves writing, and with 256 workgroups the 32 MiB store burst back-pressures the stage by another ~2 us.  Measured and NOT faster (lab copy of this
  // file): bf16 before the transposition (twice the vector instructions: +3 %), block-of-4 MFMA order (no change), the whole retirement behind the MFMAs (+4.5 %).
  // NEXT = false (ONETILE kernels): no next tile -- its stage-1 DMA and the reads of its first fragments are left out.
  auto final_stage = [&](auto fbc, auto nextc, const Desc& d, bool dvalid, const int ktn) __attribute__((always_inline)) {
    constexpr int FB = decltype(fbc)::value;   // the buffer this stage lives in (1 unless ODD); the next tile starts in the other one
    constexpr bool NEXT = decltype(nextc)::value;
    if constexpr (ODD) scr = smem + FB * STAGE + wave * DeepPCfg<C>::SCR_PER_WAVE;
    read_slice(FB, 2);
    read_slice(FB, 3);
    fence();
    mfma1(0, FB, 0, 0, false); mfma1(1, FB, 0, 0, false); mfma1(0, FB, 0, 1, false); mfma1(1, FB, 0, 1, false);
    fence();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next tile's stage 0 landed (buffer 0); all reads of buffer 1 done
    __builtin_amdgcn_s_barrier();
    fence();
    if constexpr (NEXT) dma_prep(ktn, dvalid);
    fence();
    static_for<0, 71>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s < 60) {
        constexpr int T = s < 2 ? 0 : s < 4 ? 1 : 2 + (s - 4) / 4;
        constexpr int j = s < 4 ? 2 + (s & 1) : (s - 4) % 4;
        mfma1(j, FB, T / 4, T % 4, false);
      }
      // DMA of the next tile's stage 1 into buffer 1: the B pieces and the scale piece now, one instruction every third
      // slot; the wave's 8 A pieces land in its own scratch area, so they wait until the last read-back (after the loop)
      if constexpr (NEXT && s % 3 == 0 && s / 3 < 9) dma_item(d, ktn, FB, 8 + s / 3);
      if constexpr (NEXT && s == 1) read_scales(FB ^ 1, FB ^ 1);
      // fragments of the next tile's stage 0, as their registers die: A rows of m after tile (m, 3), B rows of n after (3, n)
      if constexpr (NEXT && (s == 12 || s == 28 || s == 44)) { read_fa(FB ^ 1, 0, (s - 12) / 16); read_fa(FB ^ 1, 1, (s - 12) / 16); }
      if constexpr (NEXT && (s == 48 || s == 52 || s == 56)) { read_fb(FB ^ 1, 0, (s - 48) / 4); read_fb(FB ^ 1, 1, (s - 48) / 4); }
      if constexpr (NEXT && s == 60) { read_fa(FB ^ 1, 0, 3); read_fa(FB ^ 1, 1, 3); read_fb(FB ^ 1, 0, 3); read_fb(FB ^ 1, 1, 3); }
      // retirement items due in this slot; pair P (final at e = 8 P + 3):   write e+1 | read rows 0-15 e+3 | stores e+5, e+6 | read rows 16-31 e+7 | stores e+9, e+10
      constexpr int d1 = s - 1, d3 = s - 3, d5 = s - 5, d6 = s - 6, d7 = s - 7, d9 = s - 9, d10 = s - 10;
      if constexpr (deepp_pair_done_at(d5) >= 0) retire_store(deepp_pair_done_at(d5) / 2, deepp_pair_done_at(d5) % 2, 0);
      if constexpr (deepp_pair_done_at(d6) >= 0) retire_store(deepp_pair_done_at(d6) / 2, deepp_pair_done_at(d6) % 2, 1);
      if constexpr (deepp_pair_done_at(d9) >= 0) retire_store(deepp_pair_done_at(d9) / 2, deepp_pair_done_at(d9) % 2, 2);
      if constexpr (deepp_pair_done_at(d10) >= 0) retire_store(deepp_pair_done_at(d10) / 2, deepp_pair_done_at(d10) % 2, 3);
      if constexpr (deepp_pair_done_at(d1) >= 0) retire_write(deepp_pair_done_at(d1) / 2, deepp_pair_done_at(d1) % 2);
      if constexpr (deepp_pair_done_at(d3) >= 0) retire_read(0);
      if constexpr (deepp_pair_done_at(d7) >= 0) retire_read(1);
      fence();
    });
    // the wave's own A pieces of the next tile's stage 1 overwrite its scratch: its read-backs must have returned first
    if constexpr (NEXT) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) dma_item(d, ktn, FB, i);
    }
    fence();
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using BT = std::integral_constant<bool, true>;
  using BF = std::integral_constant<bool, false>;
  using NXT = std::integral_constant<bool, !ONETILE>;   // is there a next tile to prefetch for?

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

  if constexpr (!ODD) {
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
      stage(I0{}, BT{}, d, tonext ? 0 : 2, tonext ? nvalid : true, BT{});   // (ONETILE with K <= 512: a zero-fill DMA, harmless)
    }
    // [r4] only the LAST pair of stages issues DMA for the next tile: peeled, so that the loop body does not select three descriptors between its
    // two stages (12 s_cselect + compares = 21 scalar instructions in one MFMA slot, by the ISA's slot accounting)
    int kt = 1;
    for (; kt + 4 < KTe; kt += 2) {
      stage(I1{}, BF{}, cur, kt + 2, true, BT{});
      stage(I0{}, BF{}, cur, kt + 3, true, BT{});
    }
    if (kt + 2 < KTe) {
      stage(I1{}, BF{}, cur, kt + 2, true, BT{});
      stage(I0{}, BF{}, nxt, 0, nvalid, NXT{});
    }
    final_stage(I1{}, NXT{}, nxt, nvalid, 1);
    cur = nxt;
    tile = tnext;
  }
  } else {
    // ODD (KT = 3, 5, 7, ...): a tile whose first stage sits in buffer P ends in buffer P (its last stage has an even index); the next tile starts in P ^ 1.
    // Stage kt of a tile issues the DMA of stage kt + 2 -- stage KT - 2 that of the next tile's stage 0, the last stage that of its stage 1, each into its own buffer.
    auto tile_body = [&](auto pc) __attribute__((always_inline)) {
      using P0 = decltype(pc);
      using P1 = std::integral_constant<int, P0::value ^ 1>;
      int m0, n0;
      decode(tile, m0, n0);
      set_out_tile(m0, n0);
      const int tnext = tile + G;
      const Desc nxt = make_desc(tnext);
      const bool nvalid = tnext < ntiles;
      stage(P0{}, BT{}, cur, 2, true, BT{});
      int kt = 1;
      for (; kt + 2 <= KT - 2; kt += 2) {
        stage(P1{}, BF{}, cur, kt + 2, true, BT{});
        stage(P0{}, BF{}, cur, kt + 3, true, BT{});
      }
      stage(P1{}, BF{}, nxt, 0, nvalid, NXT{});   // stage KT - 2
      final_stage(P0{}, NXT{}, nxt, nvalid, 1);
      cur = nxt;
      tile = tnext;
    };
    while (tile < ntiles) {
      tile_body(I0{});
      if (tile >= ntiles) break;
      tile_body(I1{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// [r6] ONETILE (chosen by the host when every workgroup walks exactly one tile, 4096^3): the last stage leaves out what it does for a next tile -- 17 LDS-DMA items and
// the reads of its first fragments -- as in the fp4 kernel.
template <class C, int ST_AUX = 0, bool NN = false, bool ONETILE = false>
__device__ __forceinline__ void gemm_mx_deepp8(char* smem, const GemmParams& p, const int bid, const int G, const int ntiles) {
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
  struct Desc { __amdgpu_buffer_rsrc_t a, b, s; int mrem; };
  auto make_desc = [&](int t) __attribute__((always_inline)) {
    const bool valid = t < ntiles;
    int m0, n0;
    decode(valid ? t : ntiles - 1, m0, n0);
    const uint32_t a_off = NN ? (uint32_t)m0 : (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
    const uint32_t sa_off = (uint32_t)(m0 >> 7) * CB * 512, sb_off = (uint32_t)(n0 >> 7) * CB * 512;
    Desc d;
    d.mrem = p.M - m0;   // NN: columns past M would read the next k-row: those lanes fetch out of range (zeros) instead
    d.a = make_rsrc(p.A + a_off, valid ? p.a_bytes - a_off : 0u);
    d.b = make_rsrc(p.B + b_off, valid ? p.b_bytes - b_off : 0u);
    d.s = cx.sIsB ? make_rsrc(p.SFB + sb_off, valid ? p.sfb_bytes - sb_off : 0u) : make_rsrc(p.SFA + sa_off, valid ? p.sfa_bytes - sa_off : 0u);
    return d;
  };

  v16f acc[MT][NT];
  v8i fa[2][MT] = {}, fb[2][NT] = {};
  int sa[2][MT], sb[2][NT];

  auto read_fa = [&](const int buf, const int j, const int t) __attribute__((always_inline)) {
    const char* st = smem + buf * STAGE;
    if constexpr (NN) {
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
    if constexpr (NN) {
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
    if (NN && item < 8) {
      const int q = wave * 8 + item;
      // k-rows past K (last stage of a K that is not a multiple of 128) lie past the end of the descriptor and read zeros: on gfx950
      // the range check of a raw buffer covers voffset + soffset (tests/native/soffset_probe.hip, profiles/native_r2_soffset_probe.txt;
      // tests/test_gpu_parity.py puts fp8 NaN bytes behind the operand)
      const int v = ((item & 1) ? va1 : va0) + q * nn_rstep;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d.a, (lds_ptr_t)(st + q * 1024), 16, v, kt * nn_kstep, 0, QAMD_DMA_AUX);
    } else if (item < 16) {
      const int t = item & 7, q = wave * 8 + t;
      const int v = ((t & 1) ? vb1 : vb0) + q * cx.rstep;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(item < 8 ? d.a : d.b, (lds_ptr_t)(st + (item < 8 ? 0 : C::OFF_B) + q * 1024), 16, v, kt * C::ROWB, 0, QAMD_DMA_AUX);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d.s, (lds_ptr_t)(st + C::OFF_S + wave * 1024), 16, vbS, kt * C::SCT * 512, 0, 0);
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
    const int off = stLane + ((32 * m + 8 * pass) * p.ldd + 64 * h) * 2;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), rD, (64 * h < colLim) ? off : (int)0x80000000, 0, ST_AUX);
  };

  auto final_stage = [&](const Desc& d, bool dvalid, const int ktn) __attribute__((always_inline)) {   // ktn: as in the fp4 kernel (and the same cost structure)
    read_slice(1, 1);
    fence();
    mfma1(0, 1, 0, 0, false); mfma1(0, 1, 0, 1, false);   // slice 0 of tiles 0, 1: covers the latency of R(1)
    fence();
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0), as a builtin: the compiler's wait-count scoreboard sees it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    if constexpr (!ONETILE) dma_prep(d, ktn, dvalid);
    fence();
    static_for<0, 37>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s < 30) {
        constexpr int T = s < 2 ? s : 2 + (s - 2) / 2;
        constexpr int j = s < 2 ? 1 : (s - 2) % 2;
        mfma1(j, 1, T / 4, T % 4, false);
      }
      if constexpr (!ONETILE && s % 2 == 0 && s / 2 < 9) dma_item(d, ktn, 1, 8 + s / 2);
      if constexpr (!ONETILE && s == 1) scales_load(0);
      if constexpr (!ONETILE && s == 4) scales_fin(0);
      // slice 0 of the next tile's stage 0, as the registers die: A rows of m after tile (m, 3) (MFMA 8 m + 5), B rows of n after (3, n) (MFMA 23 + 2 n)
      if constexpr (!ONETILE && (s == 6 || s == 14 || s == 22)) read_fa(0, 0, (s - 6) / 8);
      if constexpr (!ONETILE && (s == 24 || s == 26 || s == 28)) read_fb(0, 0, (s - 24) / 2);
      if constexpr (!ONETILE && s == 30) { read_fa(0, 0, 3); read_fb(0, 0, 3); }
      constexpr int P4 = deepp8_pair_done_at(s - 4), P6 = deepp8_pair_done_at(s - 6), P1 = deepp8_pair_done_at(s - 1), P2 = deepp8_pair_done_at(s - 2);
      if constexpr (P6 >= 0) { retire_store(P6 / 2, P6 % 2, 2); retire_store(P6 / 2, P6 % 2, 3); }
      if constexpr (P4 >= 0) { retire_store(P4 / 2, P4 % 2, 0); retire_store(P4 / 2, P4 % 2, 1); retire_read(1); }
      if constexpr (P1 >= 0) retire_write(P1 / 2, P1 % 2);
      if constexpr (P2 >= 0) retire_read(0);
      fence();
    });
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the retirement reads of the scratch slice (and NN: the asm fragment reads)
    asm volatile("" ::: "memory");
    if constexpr (!ONETILE) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dma_item(d, ktn, 1, i);
    }
    fence();
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using BT = std::integral_constant<bool, true>;
  using BF = std::integral_constant<bool, false>;

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
    final_stage(nxt, nvalid, 1);
    cur = nxt;
    tile = tnext;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class C, int ST_AUX = 0, bool NN = false, bool ONETILE = false>
__global__ __launch_bounds__(C::THREADS) void gemm_mx_deepp8_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[DeepPCfg<C>::LDS_BYTES];
  gemm_mx_deepp8<C, ST_AUX, NN, ONETILE>(smem, p, (int)blockIdx.x, (int)gridDim.x, p.tiles_m * p.tiles_n);
}

template <class C, int ST_AUX = 0, bool ODD = false, bool ONETILE = false>
__global__ __launch_bounds__(C::THREADS) void gemm_mx_deepp_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[DeepPCfg<C>::LDS_BYTES];
  // every argument the prologue needs is asked for HERE: the scalar loads leave together and are waited for once (left alone they arrive in four dependent rounds)
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"((int)gridDim.x));
  gemm_mx_deepp<C, ST_AUX, 1, ODD, ONETILE>(smem, p, (int)blockIdx.x, (int)gridDim.x, p.tiles_m * p.tiles_n);
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
template <class CB, class CT, int ST_AUX = 0, bool ODD = false>   // ODD: the persistent workgroups' form for an odd number of K stages (fp4 only)
__global__ __launch_bounds__(256) void gemm_mx_hetero_kernel(const GemmParams p, const int g_big, const int t_main) {
  static_assert(CB::THREADS == 256 && CT::THREADS == 256 && CT::BM == 128 && CT::BN == 128 && CB::EBITS == CT::EBITS && CB::AFMT == CT::AFMT, "tile pair");
  constexpr int LDS = DeepPCfg<CB>::LDS_BYTES > CT::LDS_BYTES ? DeepPCfg<CB>::LDS_BYTES : CT::LDS_BYTES;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  // (device pass only: the host pass has already instantiated the same gemm_mx_deepp specialisation for the plain kernel, and clang
  // marks a __device__ specialisation whose body holds target builtins as invalid for every later host-side reference)
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char smem[LDS];
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"(g_big), "s"(t_main));   // (all scalar argument loads in one round, as in gemm_mx_deepp_kernel)
  const int b = (int)blockIdx.x;
  if (b < g_big) {
    if constexpr (CB::EBITS == 4) gemm_mx_deepp<CB, ST_AUX, 1, ODD>(smem, p, b, g_big, t_main);
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

}  // namespace qamd
