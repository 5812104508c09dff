// Persistent MXFP4 GEMM for gfx950 on EIGHT waves -- two per SIMD -- of 128x64 each: the 256x256 tile loop of gemm_mx_deepp (gemm_mx_deepp.hip.h)
// re-cut so that every SIMD holds two instruction streams.  Replaces the per-tile launch + tile scheduler of qutlass/csrc/gemm.cu:174-248, :40-88
// (matmul_host_mxf4_bf16_tn) for outputs of >= 192 tiles of 256x256, like the 4-wave kernel it is measured against.
//
// Why two waves per SIMD ([r6]).  With ONE wave per SIMD (the 4-wave kernel: 128x128 per wave, 256 accumulator registers) a wave's own program order
// is all that can put work into the shadow of its MFMAs, and almost nothing fits: a stage costs 64 x 32 cycles + the issue time of everything else in it (K loop 2 475
// cycles per stage against 2 048), and the tile's retirement -- ~550 LDS / vector / store instructions per wave -- runs at ~5.5 cycles per instruction IN SERIES with the
// last stage's MFMAs (8 400 - 9 000 cycles against 2 048; profiles/final_stage_ablation_r5c.txt, issue_ubench2_r5g.txt).  Three in-wave re-designs of that retirement
// did not beat it.  Here the wave tile is 128x64 (128 accumulator registers), so two waves fit a SIMD: while one waits for its LDS round trip, its
// fragment reads, the hand-off or a store, the other issues -- MFMAs beside vector work is what the hardware overlaps (matrix and vector pipes are separate), not
// MFMAs beside the same wave's next instruction.
// Price: 6 fragment reads per 8 MFMAs instead of 8 per 16 (192 KiB of LDS reads per stage and CU instead of 128) and 8 KiB more LDS for scale slots (144 KiB).
//
// Structure (same data path as gemm_mx_deepp: LDS-DMA stages of 128 B per row, XOR-swizzled 16-byte chunks, to_blocked scales fetched as 512-byte pieces,
// buffer-descriptor range checks for the edges, XCD-contiguous grouped raster, balanced rounds of tiles per workgroup):
//   * K stage = 4 k-slices x 8 MFMAs per wave, m-major.  Two fragment sets; a fragment register is re-read for the slice two ahead as soon as its last MFMA
//     of this slice has issued (A row-fragment m after MFMA (m, 1), B row-fragment n after (3, n)).  One hand-off (own DMA landed + barrier) per stage,
//     between slices 1 and 2; the stage's 9 DMA items per wave and the 6 scale dwords of the next stage ride behind the MFMAs of slices 2 and 3.
//   * The last stage of a tile is a K stage whose DMA (stage 1 of the next tile, into buffer 1) is held back: after its hand-off nobody reads buffer 1, and a
//     wave's own DMA pieces of it -- 4 KiB of the A area + 4 KiB of the B area -- are that wave's private retirement scratch.  The four 32x64 pairs of the wave
//     tile go through it as fp32 (ds_write_b128 straight from the accumulator registers, row-major ds_read_b128 back), alpha, v_cvt_pk_bf16_f32, whole-line
//     16-byte stores; then the wave issues the held-back pieces.  The two waves of a SIMD retire side by side, and the next tile's first stage starts on
//     whichever is done first.
// Same products, same K order per output as every other schedule: bit-identical results (tests/test_gpu_round6.py, tools/full_compare.py).
#pragma once
#include "../gemm_mx.hip.h"

namespace qamd {

template <class C>
struct DuoCfg {
  static constexpr int LDS_BYTES = 2 * C::STAGE_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(C::NA * 1024 == 4096 && C::NB * 1024 == 4096, "a wave's own A and B pieces of one stage are the two halves of its 8-KiB retirement scratch");
};

// RET: 0 = retirement after the last stage's MFMAs (burst); 1 = pair m starts behind MFMA (m, 1) of the last k-slice
template <class C, int ST_AUX = 0, int RET = 0, bool TRACE = false>
__device__ __forceinline__ void gemm_mx_duo(char* smem, const GemmParams& p, const int bid, const int G, const int ntiles) {
  static_assert(C::EBITS == 4 && C::BM == 256 && C::BN == 256 && C::WAVES_M == 2 && C::WAVES_N == 4 && C::NSTAGE == 2 && C::PPW == 1,
                "8-wave persistent schedule: fp4, 256x256 tiles, 2 x 4 waves of 128x64");
  constexpr int MT = 4, NT = 2;
  constexpr int STAGE = C::STAGE_BYTES;
  GemmCtx<C> cx(smem, p);   // per-lane offsets / LDS addresses; its tile coordinates and descriptors are NOT used here
  const int lane = cx.lane, wave = cx.wave, i32 = cx.i32, g = cx.g;
  const int KT = cx.KT, KTe = (KT + 1) & ~1, CB = cx.CB, rowbytes = cx.rowbytes;
  const int wg = xcd_remap(bid, G);
  // Per-lane state that lives across the K loop, kept to what cannot be rebuilt from an immediate: ONE fragment address (slice j: ^ 16 j -- chunk 4 g + j = 4 g ^ j and the
  // row swizzle is an XOR too), ONE scale-dword address per operand (row fragment t: + 4 t).  128 of the wave's 256 registers are accumulators.
  int rdA0 = cx.rdA[0], rdSA0 = cx.rdSA[0], rdSB0 = cx.rdSB[0];
  asm volatile("" : "+v"(rdA0), "+v"(rdSA0), "+v"(rdSB0));

  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  uint32_t tr_n = 0;
  auto mark = [&]() __attribute__((always_inline)) {   // TRACE: shader-clock stamps of workgroup 0, wave 0 -> p.dbg (lab only)
    if constexpr (TRACE) {
      if (bid == 0 && wave == 0 && tr_n < 64 && p.dbg) {
        const uint32_t t = (uint32_t)__builtin_readcyclecounter();
        if (lane == 0) p.dbg[tr_n] = t;
      }
      ++tr_n;
    }
  };
  auto decode = [&](int t, int& m0, int& n0) __attribute__((always_inline)) {
    int tm, tn;
    raster_decode(t, p.tiles_m, p.tiles_n, p.raster_magic, tm, tn);
    m0 = uniform(tm * C::BM);
    n0 = uniform(tn * C::BN);
  };
  struct Desc { __amdgpu_buffer_rsrc_t a, b, s; };
  auto make_desc = [&](int t) __attribute__((always_inline)) {   // t >= ntiles: empty descriptors -> every DMA of that "tile" loads zeros
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

  // ---- registers: 128 accumulators + 2 fragment sets (48) + 2 scale sets (12) ------------------------------------------------------------------
  v16f acc[MT][NT];
  v4i fa[2][MT] = {}, fb[2][NT] = {};
  v4i sa[2];   // scale dwords of the four A row fragments: one 16-byte line of the to_blocked image per lane (conflict-free ds_read_b128)
  v2i sb[2];   // ... of the two B row fragments (ds_read_b64)

  // one scaled FP4 MFMA: acc[m][n] (+)= B-fragment n x A-fragment m of k-slice j (fragment set j & 1, op_sel byte j of the scale dwords of set sset)
  auto mfma1 = [&](const int j, const int sset, const int m, const int n, const bool zero_c) __attribute__((always_inline)) {
    const v4i a = fa[j & 1][m], b = fb[j & 1][n];
    const v8i A8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, B8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    v16f c = acc[m][n];
    if (zero_c) c = v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j == 0) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, c, 4, 4, 0, sb[sset][n], 0, sa[sset][m]);
    if (j == 1) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, c, 4, 4, 1, sb[sset][n], 1, sa[sset][m]);
    if (j == 2) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, c, 4, 4, 2, sb[sset][n], 2, sa[sset][m]);
    if (j == 3) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, c, 4, 4, 3, sb[sset][n], 3, sa[sset][m]);
  };

  // ---- LDS-DMA of one stage: items 0..3 this wave's A pieces (rows 32 wave .. + 31), 4..7 its B pieces, 8 its scale piece ------------------------
  int vb0 = 0, vb1 = 0, vbS = 0;
  auto dma_prep = [&](int kt, bool valid) __attribute__((always_inline)) {
    // the last stage of a K that is not a multiple of 256: 16-byte chunks past the row's end are out of range.  The chunk a lane fetches in a piece of parity par is
    // rebuilt from the lane id here (once per stage) instead of living in two more registers (GemmCtx::voffT): ch = (l & 7) ^ ((l >> 4) + 4 par) & 7
    int tail = (kt == KT - 1) ? rowbytes - (KT - 1) * C::ROWB : C::ROWB;   // bytes of this stage per row
    int oobm = (valid && kt < KT) ? 0 : -1;
    int oobs = (valid && kt * C::SCT + cx.colS < CB) ? 0 : -1;
    int l = lane;
    asm volatile("" : "+v"(tail), "+v"(oobm), "+v"(oobs), "+v"(l));
    const int ch0 = (l & 7) ^ ((l >> 4) & 7), ch1 = ch0 ^ 4;
    const int o0 = oobm | ((ch0 << 4) < tail ? 0 : -1), o1 = oobm | ((ch1 << 4) < tail ? 0 : -1);
    vb0 = (cx.voffAB[0] & ~o0) | ((int)0x80000000 & o0);
    vb1 = (cx.voffAB[1] & ~o1) | ((int)0x80000000 & o1);
    vbS = (cx.voffS & ~oobs) | ((int)0x80000000 & oobs);
  };
  auto dma_item = [&](const Desc& d, int kt, const int buf, const int item) __attribute__((always_inline)) {
    char* st = smem + buf * STAGE;
    if (item < 8) {
      const int t = item & 3, q = wave * 4 + t;
      const int v = ((t & 1) ? vb1 : vb0) + q * cx.rstep;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(item < 4 ? d.a : d.b, (lds_ptr_t)(st + (item < 4 ? 0 : C::OFF_B) + q * 1024), 16, v, kt * C::ROWB, 0, QAMD_DMA_AUX);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d.s, (lds_ptr_t)(st + C::OFF_S + wave * 1024), 16, vbS, kt * C::SCT * 512, 0, 0);
    }
  };
  auto dma_stage = [&](const Desc& d, int kt, bool valid, const int buf) __attribute__((always_inline)) {
    dma_prep(kt, valid);
#pragma unroll
    for (int i = 0; i < 9; ++i) dma_item(d, kt, buf, i);
  };
  auto pin_acc = [&]() __attribute__((always_inline)) {   // (as in gemm_mx_deepp: keeps the first stage's MFMAs in their own block)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" : "+a"(acc[m][n]));
  };

  // ---- epilogue pieces: a PAIR of accumulator tiles (m, 0), (m, 1) = 32 rows x 64 columns through the wave's scratch as fp32 (32 rows x 256 B; rows 0-15 in the wave's
  //      own A pieces of buffer 1, rows 16-31 in its own B pieces; 16-byte chunk c of row r at chunk (c & 8) | ((c & 7) ^ (r & 7)): conflict-free ds_write_b128 and
  //      ds_read_b128, as in gemm_mx_deepp).  Read-back is row-major: lane -> row 8 pass + l / 8, columns 8 (l % 8) .. + 7: a wave instruction stores 8 rows x 128 B.
  // (every per-lane address of the retirement is rebuilt from the lane id once per tile, behind an opaque copy of it: hoisted out of the tile loop -- they are tile-invariant --
  //  they would sit in ~20 registers across the K loop, which this kernel does not have)
  char* scr = smem + STAGE + wave * 4096;
  int scrW = 0, scrR = 0, stLane = 0;
  const float alpha = *p.alpha;
  __amdgpu_buffer_rsrc_t rD = make_rsrc(p.D, 0);
  auto set_out_tile = [&](int m0, int n0) __attribute__((always_inline)) {
    const int64_t left = ((int64_t)(p.M - m0) * p.ldd - n0) * 2;
    rD = make_rsrc(p.D + ((int64_t)m0 * p.ldd + n0), (uint32_t)(left > 0x7fffffffll ? 0x7fffffffll : left));
  };
  auto retire_setup = [&](int n0) __attribute__((always_inline)) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int li32 = l & 31, lg = l >> 5, rrl = l >> 3, ccl = l & 7;
    scrW = (li32 & 15) * 256 + (li32 >> 4) * C::OFF_B + ((((li32 & 6) << 4)) | ((lg ^ (li32 & 1)) << 4));
    scrR = rrl * 256 + (ccl >> 2) * 128 + ((((2 * ccl) & 7) ^ (rrl & 7)) << 4);
    const int col = cx.wave_n * C::WTN + 8 * ccl;
    stLane = (n0 + col < p.N) ? ((cx.wave_m * C::WTM + rrl) * p.ldd + col) * 2 : (int)0x80000000;   // columns >= N: out of range per lane
  };
  auto retire_write = [&](const int m) __attribute__((always_inline)) {
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(v4f*)(scr + (scrW ^ (q << 5)) + nn * 128) = v4f{acc[m][nn][4 * q + 0], acc[m][nn][4 * q + 1], acc[m][nn][4 * q + 2], acc[m][nn][4 * q + 3]};
  };
  v4f rb[2][2];   // read-back registers of one half pair (16 rows)
  auto retire_read = [&](const int pass) __attribute__((always_inline)) {   // rows 8 pass .. + 7
    const char* base = scr + (pass >> 1) * C::OFF_B + (pass & 1) * 2048;
    rb[pass & 1][0] = *(const v4f*)(base + scrR);
    rb[pass & 1][1] = *(const v4f*)(base + (scrR ^ 16));
  };
  // (gfx950: a VALU write of a 16-byte buffer store's data registers must not sit DIRECTLY behind the store when it carries an SGPR offset -- the compiler does not guard that
  //  case; tests/native/store_hazard_probe.hip, tools/store_data_hazard.py.  So both halves' conversions come first, then the two stores back to back, then LDS traffic.)
  v4i pk[2];
  auto retire_pack = [&](const int pass) __attribute__((always_inline)) {
    const v4f lo = rb[pass & 1][0], hi = rb[pass & 1][1];
    pk[pass & 1][0] = (int)pack_bf16x2(lo[0] * alpha, lo[1] * alpha);
    pk[pass & 1][1] = (int)pack_bf16x2(lo[2] * alpha, lo[3] * alpha);
    pk[pass & 1][2] = (int)pack_bf16x2(hi[0] * alpha, hi[1] * alpha);
    pk[pass & 1][3] = (int)pack_bf16x2(hi[2] * alpha, hi[3] * alpha);
  };
  auto retire_store = [&](const int m, const int pass) __attribute__((always_inline)) {
    int ldd2 = p.ldd * 2;
    asm volatile("" : "+s"(ldd2));   // (the row part of the address rides in the scalar offset, recomputed per store: no per-tile offset registers live across the K loop)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, pk[pass & 1]), rD, stLane, (32 * m + 8 * pass) * ldd2, ST_AUX);
  };
  auto retire_pair = [&](const int m) __attribute__((always_inline)) {
    retire_write(m);
    fence();
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      retire_read(2 * hf);
      retire_read(2 * hf + 1);
      fence();
      retire_pack(2 * hf);
      retire_pack(2 * hf + 1);
      fence();
      retire_store(m, 2 * hf);
      retire_store(m, 2 * hf + 1);
      fence();
    }
  };

  // ---- one K stage.  Entry: fragment sets 0, 1 and scale set BUF hold k-slices 0, 1 of this stage; exit: the same for the next stage (other buffer).
  //      d / ktl / dvalid: the stage whose DMA is threaded through slices 2, 3 (into THIS buffer, free after the hand-off); LAST: that DMA is held back (retirement scratch).
  auto stage = [&](auto bufc, auto firstc, auto lastc, const Desc& d, int ktl, bool dvalid) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr bool FIRST = decltype(firstc)::value, LAST = decltype(lastc)::value;
    typedef __attribute__((address_space(3))) const v4i* lds_v4i_t;
    uint32_t rbA = 0, rbB = 0;   // 32-bit LDS addresses of the slice being read (made opaque once: folded into every read they exceed the DS offset field)
    auto read_base = [&](const int buf, const int j) __attribute__((always_inline)) {
      int r0 = rdA0;
      asm volatile("" : "+v"(r0));   // (opaque BEFORE the arithmetic: the eight buffer x slice addresses are loop invariants and would be hoisted into eight registers each for A and B)
      rbA = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + buf * STAGE) + (uint32_t)(r0 ^ (j << 4));
      rbB = rbA + (uint32_t)cx.rdBd;
      asm volatile("" : "+v"(rbA), "+v"(rbB));
    };
    // the re-read of a fragment register for slice js + 2 behind MFMA i = 2 m + n of slice js: A row-fragment m is dead after (m, 1), B row-fragment n after (3, n)
    auto recycle = [&](const int set, const int i) __attribute__((always_inline)) {
      if (i & 1) fa[set][i >> 1] = *(lds_v4i_t)(uintptr_t)(rbA + (uint32_t)((i >> 1) * 32 * C::ROWB));
      if (i >= 6) fb[set][i - 6] = *(lds_v4i_t)(uintptr_t)(rbB + (uint32_t)((i - 6) * 32 * C::ROWB));
    };
    auto read_scale1 = [&](const int buf, const int set, const int k) __attribute__((always_inline)) {   // k = 0: the A operand's dwords, 1: B's
      const char* st = smem + buf * STAGE;
      if (k == 0) sa[set] = *(const v4i*)(st + rdSA0);
      else sb[set] = *(const v2i*)(st + rdSB0);
    };
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
    group(0, FIRST, [&](const int i) __attribute__((always_inline)) { recycle(0, i); });
    read_base(BUF, 3);
    group(1, false, [&](const int i) __attribute__((always_inline)) {
      recycle(1, i);
      if (i == 2 && !LAST) dma_prep(ktl, dvalid);
    });
    read_base(BUF ^ 1, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMA of the next stage landed; own reads of this buffer done
    __builtin_amdgcn_s_barrier();
    fence();
    group(2, false, [&](const int i) __attribute__((always_inline)) {
      if (i == 0 || i == 2) read_scale1(BUF ^ 1, BUF ^ 1, i >> 1);   // (slots without a fragment read)
      recycle(0, i);
      if (!LAST && i < 5) dma_item(d, ktl, BUF, i);
    });
    read_base(BUF ^ 1, 1);
    group(3, false, [&](const int i) __attribute__((always_inline)) {
      recycle(1, i);
      if (!LAST && !(i & 1)) dma_item(d, ktl, BUF, 5 + (i >> 1));   // items 5 .. 8 behind MFMAs 0, 2, 4, 6
      if (LAST && RET == 1 && (i & 1)) retire_pair(i >> 1);
    });
    if constexpr (FIRST) pin_acc();
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using BT = std::integral_constant<bool, true>;
  using BF = std::integral_constant<bool, false>;

  // ---- prologue: first tile's stages 0 and 1 in flight; stage 0 landed -> first two slices into registers -------------------------------------
  int tile = wg;
  Desc cur = make_desc(tile);
  dma_stage(cur, 0, true, 0);
  dma_stage(cur, 1, true, 1);
  asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  fence();
  {
    const char* st = smem;
    sa[0] = *(const v4i*)(st + rdSA0);
    sb[0] = *(const v2i*)(st + rdSB0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int t = 0; t < MT; ++t) fa[j][t] = *(const v4i*)(st + (rdA0 ^ (j << 4)) + t * 32 * C::ROWB);
#pragma unroll
      for (int t = 0; t < NT; ++t) fb[j][t] = *(const v4i*)(st + cx.rdBd + (rdA0 ^ (j << 4)) + t * 32 * C::ROWB);
    }
  }
  fence();
  mark();

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
      stage(I0{}, BT{}, BF{}, d, tonext ? 0 : 2, tonext ? nvalid : true);
    }
    int kt = 1;
    for (; kt + 4 < KTe; kt += 2) {
      stage(I1{}, BF{}, BF{}, cur, kt + 2, true);
      stage(I0{}, BF{}, BF{}, cur, kt + 3, true);
    }
    if (kt + 2 < KTe) {
      stage(I1{}, BF{}, BF{}, cur, kt + 2, true);
      stage(I0{}, BF{}, BF{}, nxt, 0, nvalid);
    }
    mark();
    if constexpr (RET == 1) retire_setup(n0);
    stage(I1{}, BF{}, BT{}, nxt, 1, nvalid);
    mark();
    if constexpr (RET == 0) {
      retire_setup(n0);
#pragma unroll
      for (int m = 0; m < MT; ++m) retire_pair(m);
    }
    // the held-back DMA of the next tile's stage 1: the wave's pieces overwrite its scratch, whose read-backs have returned (their values were stored)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dma_prep(1, nvalid);
#pragma unroll
    for (int i = 0; i < 9; ++i) dma_item(nxt, 1, 1, i);
    fence();
    mark();
    cur = nxt;
    tile = tnext;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class C, int ST_AUX = 0, int RET = 0, bool TRACE = false>
__global__ __launch_bounds__(C::THREADS) void gemm_mx_duo_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[DuoCfg<C>::LDS_BYTES];
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"((int)gridDim.x));   // (all scalar argument loads leave in one round, as in gemm_mx_deepp_kernel)
  gemm_mx_duo<C, ST_AUX, RET, TRACE>(smem, p, (int)blockIdx.x, (int)gridDim.x, p.tiles_m * p.tiles_n);
}

}  // namespace qamd
