// Block-scaled MX GEMM for gfx950:  D[M,N] (bf16) = alpha * (A . SFA) (B . SFB)^T
//
//   A: (M, K) row-major, B: (N, K) row-major ("TN"), elements e2m1 (EBITS=4, two per byte, MXFP4)
//   or e4m3 (EBITS=8, MXFP8); SFA/SFB: e8m0 per 32 K-elements in the to_blocked 128x4 tiling.
//   Replaces qutlass/csrc/gemm.cu:174-248 (matmul_host_mxf4_bf16_tn) and :328-386
//   (matmul_host_mxf8_bf16_tn) of the reference, whose arithmetic lives in CUTLASS collectives.
//
// CDNA4 mapping:
//   * v_mfma_scale_f32_32x32x64_f8f6f4.  FP4: one lane holds 32 consecutive K elements of one row,
//     i.e. exactly one MX scale group, and the hardware applies the lane's e8m0 byte (selected from
//     a 32-bit scale register by op_sel) to it (device-verified, tests/native/probe.hip P1).  The
//     to_blocked layout puts the four K-block scales of a row in one dword and the four 32-row
//     slabs of a 128-row tile in one 16-byte line, so a lane fetches all scales of its MT
//     row-fragments x 4 K-blocks with ONE ds_read_b128.
//   * K is walked in stages of 128 BYTES per row (256 fp4 / 128 fp8 elements).  Within a stage the
//     two 32-lane halves of the wave take disjoint K-blocks (fp4: half g owns 16-byte chunks
//     4g..4g+3), so the k-slice index j of an MFMA is both the chunk offset and the op_sel byte.
//   * Stages are copied HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), each wave
//     instruction moving 8 rows x 128 B (full cache lines).  The LDS image is lane-linear, so the
//     16-byte-chunk XOR swizzle (chunk ^= (row>>1)&7) that makes the ds_read_b128 fragment reads
//     bank-conflict free is applied to the per-lane SOURCE address and to the read address.
//   * Operand roles are swapped in the MFMA (srcA = B fragment, srcB = A fragment) so a lane ends
//     up with 4 consecutive N for one M row: the bf16 epilogue packs 8-byte pieces, stages the
//     tile through LDS and writes whole 128-byte lines.
//   * blockIdx -> tile: XCD-contiguous remap + grouped raster so one XCD's L2 sees a compact
//     rectangle of tiles.
//   * Schedules (all share the tiling / DMA / epilogue code in GemmCtx; DESIGN.md section 3.3-3.4 has the measurements):
//       PRODUCT   gemm_mx_ringp (this file)        4 waves, 64x64 .. 256x128 tiles, 2- to 4-deep LDS ring, fragments of the next stage read during
//                                                  this stage's MFMAs, optional split-K / row-major scales; also the residual tiles of the
//                                                  heterogeneous launch
//                 gemm_mx_deepp / gemm_mx_deepp8   (gemm_mx_deepp.hip.h) persistent 256x256 kernels, fp4 / fp8 (+ the (K, M) operand of matmul_mxf8_bf16_nn)
//                 gemm_mx_skinny_kernel            (gemm_mx_skinny.hip.h) LDS-free split-K kernel for M <= 32
//       LAB ONLY  lab/gemm_mx_lab.hip.h (-DQAMD_BENCH=1): lockstep / ping-pong / queue / simple / per-tile deep / regstage / un-pipelined ring -- kept
//                                                  selectable ("gemm_variant") because their measurements are part of the design record
#pragma once
#include <type_traits>

#include "common.hip.h"
#include "transpose_u8.hip.h"

// cache policy of the operand LDS-DMA loads (buffer_load ... lds aux bits: 1 = sc0, 2 = nt, 16 = sc1); build-time
// switch for experiments (tools: hipcc -DQAMD_DMA_AUX=2 ...)
#ifndef QAMD_DMA_AUX
#define QAMD_DMA_AUX 0
#endif
#ifndef QAMD_BENCH
#define QAMD_BENCH 0   // 1: lab library (libqutlass_amd_bench.so) -- experimental code paths that lost their measurement stay compiled there
#endif   // transpose4x4_u8 (fused NN operand path)

namespace qamd {

#ifndef QAMD_CTX_MAGIC_DECODE
#define QAMD_CTX_MAGIC_DECODE 0
#endif
#ifndef QAMD_RING_KERNARG_EARLY
#define QAMD_RING_KERNARG_EARLY 0
#endif
struct GemmParams {
  const uint8_t* A;
  const uint8_t* B;
  const uint8_t* SFA;
  const uint8_t* SFB;
  const float* alpha;
  uint16_t* D;
  int M, N, K;           // K in elements
  int ldd;               // row stride of D in elements (= N unless the launch covers a column range of a wider D)
  int tiles_m, tiles_n;  // grid = tiles_m * tiles_n
  uint32_t raster_magic; // persistent kernels: raster_magic(tiles_n) (common.hip.h), set by their launchers
  uint32_t a_bytes, b_bytes, sfa_bytes, sfb_bytes;
  int pp_shift;          // ping-pong: wave group = (wave >> pp_shift) & 1
  int pp_flags;          // bit0: s_setprio around MFMA blocks
  uint32_t* dbg;         // ABL_TRACE builds only: per-wave timestamp dump of workgroup 0
  float* ws;             // split-K (ring schedule only): fp32 partial tiles, [splits][M][N]; NULL = no split
  int splits;            // grid.y; split z covers K stages [z * ceil(KT / splits), ...)
  // fused split-K reduction (ctr != NULL): one 64-bit arrival slot per output tile, {launch tag : 56, count : 8}; the last
  // split to arrive for a tile sums the partials and writes D.  tag = a per-launch number from the host, so slots need no
  // initialisation (whatever the memory held counts as "0 arrivals" unless it carries this launch's 56-bit tag).
  unsigned long long* ctr;
  unsigned long long tag;    // (launch number & (2^56 - 1)) << 8
  // [r4] persistent kernels, stream-K form (gemm_mx_deepp.hip.h, streamk.hip.h): the last sk_tiles tiles of the raster are walked as one stream of K stages
  // cut into equal ranges; ws = parked fp32 tiles (one 256 KiB slot per range boundary), ctr = one arrival flag per slot (== tag: parked), tag = launch number
  int sk_tiles;
};

// ablation bits (bench-only instantiations; 0 in the product path)
enum { ABL_NO_DMA = 1, ABL_NO_MFMA = 2, ABL_NO_STORE = 4, ABL_NO_EPILOGUE = 8, ABL_TRACE = 16, ABL_NO_READS = 32, ABL_CLOCK = 64,
       ABL_READS_FIRST = 128 };   // (not an ablation: pipelined ring with the next stage's fragment reads packed into the first MFMAs of the stage; lab A/B)

// AFMT_: element format of the A operand of an MXFP8 GEMM, 0 = e4m3 (the reference's only format), 1 = e5m2 (extension:
// gradient operand of BASELINE.json configs[4]; the scaled MFMA takes the format per operand in cbsz / blgp).  B is e4m3.
template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, int EBITS_, bool F8SPLIT_ = false, int ABL_ = 0, int NSTAGE_ = 2, int AFMT_ = 0>
struct GemmCfg {
  static constexpr int AFMT = AFMT_;
  static_assert(AFMT_ == 0 || (AFMT_ == 1 && EBITS_ == 8), "A format: 0 = e4m3 / e2m1, 1 = e5m2 (MXFP8 only)");
  static constexpr int NSTAGE = NSTAGE_;           // depth of the LDS stage ring (2 except for the ring schedule)
  static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, EBITS = EBITS_;
  static constexpr bool F8SPLIT = F8SPLIT_;
  static constexpr int ABL = ABL_;
  static constexpr int NWAVES = WAVES_M * WAVES_N, THREADS = NWAVES * 64;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, MT = WTM / 32, NT = WTN / 32;
  static constexpr int ROWB = 128;                 // bytes of K per row per stage
  static constexpr int BK = ROWB * 8 / EBITS;      // K elements per stage
  static constexpr int KSL = BK / 64;              // MFMA k-slices per stage (4 fp4 / 2 fp8)
  static constexpr int CPS = 16 * EBITS / 64;      // 16-byte chunks per lane per slice (1 / 2)
  static constexpr int SCT = BK / 128;             // scale column tiles (4 K-blocks) per stage
  static constexpr int SA_TILES = (BM + 127) / 128, SB_TILES = (BN + 127) / 128;
  static constexpr int PA = SA_TILES * SCT, PB = SB_TILES * SCT;   // 512-byte scale pieces (128 rows x 4 K-blocks)
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  // scale piece i is fetched by wave i into its own 1-KiB LDS slot (lanes 0-31 carry the 512 bytes,
  // lanes 32-63 load out-of-range zeros into the slot's pad half): every wave issues the same
  // instruction sequence, so the K loop needs no wave-dependent branch.
  // With more pieces than waves (4-wave configurations) a wave carries two pieces of the SAME tensor, one per lane half.
  static constexpr int PPW = (PA + PB > NWAVES) ? 2 : 1;   // pieces per wave instruction
  static constexpr int OFF_B = A_BYTES, OFF_S = A_BYTES + B_BYTES, S_BYTES = NWAVES * 1024;
  static constexpr int STAGE_BYTES = OFF_S + S_BYTES;
  static constexpr int NA = BM / 8 / NWAVES, NB = BN / 8 / NWAVES;  // 1-KiB DMA pieces per wave
  static constexpr int SROW = BN * 2;   // epilogue staging row stride (8-byte granules XOR-swizzled by row)
  static constexpr int LDS_MAIN = (NSTAGE * STAGE_BYTES > BM * SROW) ? NSTAGE * STAGE_BYTES : BM * SROW;
  static constexpr int TRACE_SLOTS = 96;
  static constexpr int LDS_BYTES = LDS_MAIN + ((ABL_ & 16) ? NWAVES * TRACE_SLOTS * 4 : 0);
  static_assert(BM % (8 * NWAVES) == 0 && BN % (8 * NWAVES) == 0, "DMA split");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  static_assert(PA + PB <= NWAVES * PPW && (PPW == 1 || (PA % 2 == 0 && PB % 2 == 0)), "scale pieces per wave");
  static_assert(THREADS % (BN / 8) == 0, "epilogue split");
};

// Per-thread state + the building blocks shared by both schedules.
template <class C>
struct GemmCtx {
  static constexpr int MT = C::MT, NT = C::NT, KSL = C::KSL, CPS = C::CPS;
  char* smem;
  const GemmParams& p;
  int tid, lane, wave, wave_m, wave_n, i32, g;
  int m0, n0, rowbytes, KT, CB;
  bool ktail;
  __amdgpu_buffer_rsrc_t rA, rB, rS;   // rS: the scale tensor (A's or B's) this wave fetches from
  int voffAB[2];   // per-lane source offset of a 1-KiB DMA piece, for even / odd piece index (rows 8q..8q+7)
  int voffT[2];    // the same for the last stage when K is not a multiple of the stage (chunks past K: out of range)
  int rstep;       // 8 * rowbytes: byte distance between consecutive pieces
  int voffS, colS; // scale piece of this wave
  bool sIsB;
  int rdA[KSL * CPS];   // fragment read address of chunk slot (j, u) for the A rows of this lane
  int rdBd;             // wave-uniform: B fragment address = rdA[..] + rdBd
  int rdSA[MT], rdSB[NT];
  v16f acc[MT][NT];
  v8i fa[KSL][MT], fb[KSL][NT];   // only the slices a schedule keeps live are materialised
  int sa[MT], sb[NT];
  float alpha_k;   // [r4] alpha[0], fetched by the constructor: loaded where the epilogue starts, its memory round trip sat on the critical path of every
                   // per-tile workgroup (a 5-20 us kernel paid ~0.5 us for one scalar)

  // bid: linear tile id (the workgroup id of a plain launch).  fm0 / fn0 >= 0: the tile's origin is given instead (residual
  // tiles of a heterogeneous launch, gemm_mx_deepp.hip.h: any 128-aligned origin inside the output).
  __device__ __forceinline__ GemmCtx(char* smem_, const GemmParams& p_, int bid = (int)blockIdx.x, int fm0 = -1, int fn0 = -1) : smem(smem_), p(p_) {
    alpha_k = *p.alpha;     // (issued here: the ISA keeps the s_load in the prologue, tools/kernel_resources.py-style check in tests)
    tid = threadIdx.x;
    lane = tid & 63;
    wave = uniform(tid >> 6);
    wave_m = wave / C::WAVES_N;
    wave_n = wave % C::WAVES_N;
    i32 = lane & 31;
    g = lane >> 5;

    // ---- tile coordinates (XCD-contiguous, grouped raster) ----------------------------------
    int tile_m, tile_n;
    {
      const int nb = p.tiles_m * p.tiles_n;
      const int b2 = xcd_remap(bid, nb);
#if QAMD_CTX_MAGIC_DECODE
      // the division-free decode of the persistent kernels (common.hip.h; p.raster_magic is set by every launcher of a GemmCtx kernel).  NOT the product's choice yet:
      // prepared at the end of round 4 (no GPU minutes left to validate it); ISA: two 32-bit divisions less at the top of every per-tile workgroup
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
    m0 = tile_m * C::BM;
    n0 = tile_n * C::BN;
    if (fm0 >= 0) { m0 = fm0; n0 = fn0; }
    rowbytes = (p.K * C::EBITS) >> 3;
    KT = (rowbytes + C::ROWB - 1) / C::ROWB;
    CB = (p.K / 32 + 3) >> 2;
    ktail = (rowbytes % C::ROWB) != 0;

    // ---- DMA descriptors and per-lane source offsets ----------------------------------------
    const uint32_t a_off = (uint32_t)m0 * rowbytes, b_off = (uint32_t)n0 * rowbytes;
    rA = make_rsrc(p.A + a_off, p.a_bytes - a_off);
    rB = make_rsrc(p.B + b_off, p.b_bytes - b_off);
    const uint32_t sa_off = (uint32_t)(m0 >> 7) * CB * 512, sb_off = (uint32_t)(n0 >> 7) * CB * 512;

    // piece q of an operand covers rows 8q..8q+7; lane L -> row 8q + (L>>3), physical chunk L&7,
    // logical chunk (L&7) ^ ((row>>1)&7) = (L&7) ^ ((L>>4) + 4(q&1)): only the parity of q matters, so
    // two per-lane offsets (relative to row 8q) serve every piece; the piece's row offset q*8*rowbytes
    // and the K offset of the stage travel in the scalar soffset operand.
    rstep = 8 * rowbytes;
#pragma unroll
    for (int par = 0; par < 2; ++par)
    {
      const int ch = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
      voffAB[par] = (lane >> 3) * rowbytes + (ch << 4);
      voffT[par] = (ch * 16 < rowbytes - (KT - 1) * C::ROWB) ? voffAB[par] : 0x7fffffff;
    }
    // scale piece of this lane: pieces 0..PA-1 belong to A, PA..PA+PB-1 to B; piece -> (tile row, tile col).
    // PPW == 1: piece = wave, carried by lanes 0-31 (lanes 32-63 load zeros into the slot's pad half);
    // PPW == 2: lanes 0-31 carry piece 2*wave, lanes 32-63 piece 2*wave+1 (same tensor: PA, PB even).
    {
      const int piece = (C::PPW == 2) ? 2 * wave + g : wave;
      sIsB = (C::PPW * wave) >= C::PA;
      rS = sIsB ? make_rsrc(p.SFB + sb_off, p.sfb_bytes - sb_off) : make_rsrc(p.SFA + sa_off, p.sfa_bytes - sa_off);
      const int idx = sIsB ? piece - C::PA : piece;
      colS = idx % C::SCT;
      const bool on = (piece < C::PA + C::PB) && (C::PPW == 2 || g == 0);
      voffS = on ? ((idx / C::SCT) * CB + colS) * 512 + i32 * 16 : 0x7fffffff;
    }

    // ---- LDS fragment read addresses --------------------------------------------------------
    // fragment row r = wave_row0 + 32t + i32 ; phys chunk = logical chunk ^ ((r>>1)&7)
    //   fp4              : slice j, lane half g        -> chunk 4g + j          (K-block 4g + j)
    //   fp8 (contiguous) : slice j, register half u    -> chunk 4g + 2j + u     (K-block 2g + j)
    //   fp8 (split)      : slice j, register half u    -> chunk 4j + 2u + g     (K-block 2j + u)
    const int sw = (i32 >> 1) & 7;
#pragma unroll
    for (int j = 0; j < KSL; ++j)
#pragma unroll
      for (int u = 0; u < CPS; ++u) {
        int c;
        if (C::EBITS == 4) c = 4 * g + j;
        else if (!C::F8SPLIT) c = 4 * g + 2 * j + u;
        else c = 4 * j + 2 * u + g;
        rdA[j * CPS + u] = (wave_m * C::WTM + i32) * C::ROWB + ((c ^ sw) << 4);
      }
    rdBd = C::OFF_B + (wave_n * C::WTN - wave_m * C::WTM) * C::ROWB;
    // scale dwords: row r_abs = (m0 & 127) + wave_row0 + 32t (m0 & 127 only matters when BM < 128)
    const int rbaseA = (C::BM >= 128) ? 0 : (m0 & 127), rbaseB = (C::BN >= 128) ? 0 : (n0 & 127);
    const int scol = (C::SCT == 2) ? g : 0;   // fp4: half g owns column tile g of the stage
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int r = rbaseA + wave_m * C::WTM + 32 * t;
      rdSA[t] = C::OFF_S + ((r >> 7) * C::SCT + scol) * (1024 / C::PPW) + i32 * 16 + ((r & 127) >> 5) * 4;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int r = rbaseB + wave_n * C::WTN + 32 * t;
      rdSB[t] = C::OFF_S + (C::PA + (r >> 7) * C::SCT + scol) * (1024 / C::PPW) + i32 * 16 + ((r & 127) >> 5) * 4;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  }

  int trace_n = 0;
  bool perm_rows = false;   // fused NN: lane i32 of fragment t owns tile row 4*i32 + t instead of 32*t + i32
  __device__ __forceinline__ void trace() {
    if (C::ABL & ABL_TRACE) {
      if (blockIdx.x == 0 && trace_n < C::TRACE_SLOTS) {
        const uint32_t t = (uint32_t)__builtin_readcyclecounter();
        if (lane == 0) ((uint32_t*)(smem + C::LDS_MAIN))[wave * C::TRACE_SLOTS + trace_n] = t;
      }
      ++trace_n;
    }
  }
  __device__ __forceinline__ void trace_dump() {
    if (C::ABL & ABL_TRACE) {
      __syncthreads();
      if (blockIdx.x == 0 && p.dbg)
        for (int i = tid; i < C::NWAVES * C::TRACE_SLOTS; i += C::THREADS) p.dbg[i] = ((uint32_t*)(smem + C::LDS_MAIN))[i];
    }
  }

  // One operand's share of a stage for this wave: NP pieces q = wave*NP + t.  Branch-free: `valid == false`
  // (no such stage) turns every lane's offset out of range, so the DMA writes zeros and the K loop stays one
  // basic block.  The whole row offset stays in the per-lane voffset (one v_add per piece) so rows past the end
  // of the tensor are out of range for the descriptor whatever the hardware does with soffset; soffset carries
  // only the K offset of the stage.
  __device__ __forceinline__ void issue_pieces(const int NP, __amdgpu_buffer_rsrc_t rsrc, char* dst, int kt, bool valid) {
    const int soff = kt * C::ROWB;
    // scalar masks made opaque to the optimiser: with plain selects LLVM threads the conditions into branches,
    // which splits the K loop into several basic blocks and lets MachineSink pile every MFMA up at its end
    int lastmask = (kt == KT - 1) ? -1 : 0;
    int oob = valid ? 0 : 0x7f000000;
    asm volatile("" : "+v"(lastmask), "+v"(oob));
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      const int q = wave * NP + t;
      const int par = q & 1;
      const int a = par ? voffAB[1] : voffAB[0], b = par ? voffT[1] : voffT[0];
      const int base = (b & lastmask) | (a & ~lastmask);
      const int v = base + q * rstep + oob;   // (unsigned) >= num_records when oob is set
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + q * 1024), 16, v, soff, 0, QAMD_DMA_AUX);
    }
  }
  __device__ __forceinline__ void issue_pieces_range(const int NP, __amdgpu_buffer_rsrc_t rsrc, char* dst, int kt, bool valid, const int t0, const int t1) {
    const int soff = kt * C::ROWB;
    int lastmask = (kt == KT - 1) ? -1 : 0;
    int oob = valid ? 0 : 0x7f000000;
    asm volatile("" : "+v"(lastmask), "+v"(oob));
#pragma unroll
    for (int t = t0; t < t1; ++t) {
      const int q = wave * NP + t;
      const int par = q & 1;
      const int a = par ? voffAB[1] : voffAB[0], b = par ? voffT[1] : voffT[0];
      const int v = ((b & lastmask) | (a & ~lastmask)) + q * rstep + oob;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + q * 1024), 16, v, soff, 0, QAMD_DMA_AUX);
    }
  }
  __device__ __forceinline__ void issue_scales(int kt, char* st, bool valid) {
    int oob = (valid && kt * C::SCT + colS < CB) ? 0 : 0x7f000000;   // K tail: no such scale column tile (per lane)
    asm volatile("" : "+v"(oob));
    const int ssoff = kt * C::SCT * 512;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, (lds_ptr_t)(st + C::OFF_S + wave * 1024), 16, voffS + oob, ssoff, 0, 0);
  }
  // L2 warm-up: one dword per 128-byte line of a future stage (rows of A for the first half of the waves, rows of B
  // for the second half; 64 lanes = 64 rows).  The loaded value is never used; it only makes the later LDS-DMA of
  // that stage an L2 hit instead of a MALL/HBM miss (the 2-deep LDS ring leaves the DMA a single stage to land).
  __device__ __forceinline__ int prefetch_stage(int kt, bool valid) {
    constexpr int HALFW = C::NWAVES / 2;
    const bool isB = wave >= HALFW;
    const int w = isB ? wave - HALFW : wave;
    constexpr int ROWS_PER_WAVE_A = C::BM / HALFW, ROWS_PER_WAVE_B = C::BN / HALFW;
    static_assert(ROWS_PER_WAVE_A <= 64 && ROWS_PER_WAVE_B <= 64, "one load covers a wave's rows (needs >= 8 waves for 256-row tiles)");
    const int rows = isB ? ROWS_PER_WAVE_B : ROWS_PER_WAVE_A;
    int oob = (valid && kt < KT) ? 0 : 0x7f000000;
    asm volatile("" : "+v"(oob));
    const int v = (w * rows + (lane % rows)) * rowbytes + oob;
    return __builtin_amdgcn_raw_buffer_load_b32(isB ? rB : rA, v, kt * C::ROWB, 0);
  }

  // half 0: the A pieces + the scale piece; half 1: the B pieces
  __device__ __forceinline__ void issue_stage_part(int kt, int buf, int half, bool valid = true) {
    char* st = smem + buf * C::STAGE_BYTES;
    if (half == 0) {
      issue_pieces(C::NA, rA, st, kt, valid);
      issue_scales(kt, st, valid);
    } else {
      issue_pieces(C::NB, rB, st + C::OFF_B, kt, valid);
    }
  }
  __device__ __forceinline__ void issue_stage(int kt, int buf) {
    issue_stage_part(kt, buf, 0);
    issue_stage_part(kt, buf, 1);
  }

  __device__ __forceinline__ void read_scales(int buf) {
    const char* st = smem + buf * C::STAGE_BYTES;
    // fp8: both halves read the same dword (4 K-blocks of the stage); bring "my" first byte down
    const int shift = (C::EBITS == 4) ? 0 : (C::F8SPLIT ? 8 * g : 16 * g);
#pragma unroll
    for (int t = 0; t < MT; ++t) sa[t] = (int)((unsigned)(*(const int*)(st + rdSA[t])) >> shift);
#pragma unroll
    for (int t = 0; t < NT; ++t) sb[t] = (int)((unsigned)(*(const int*)(st + rdSB[t])) >> shift);
  }

  __device__ __forceinline__ void read_frags(int buf, int j) {
    const char* st = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const v4i lo = *(const v4i*)(st + rdA[j * CPS] + t * 32 * C::ROWB);
      v4i hi = {0, 0, 0, 0};
      if (CPS == 2) hi = *(const v4i*)(st + rdA[j * CPS + CPS - 1] + t * 32 * C::ROWB);
      fa[j][t] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const v4i lo = *(const v4i*)(st + rdBd + rdA[j * CPS] + t * 32 * C::ROWB);
      v4i hi = {0, 0, 0, 0};
      if (CPS == 2) hi = *(const v4i*)(st + rdBd + rdA[j * CPS + CPS - 1] + t * 32 * C::ROWB);
      fb[j][t] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  }

  // op_sel byte of slice j:  fp4 -> j ; fp8 contiguous -> j ; fp8 split -> 2j.  `j` is always a literal
  // at the call site; the branch chain folds after inlining (op_sel must be an immediate).
  __device__ __forceinline__ void mfma_slice(const int j) {
    // cbsz = format of srcA (the B fragments), blgp = format of srcB (the A fragments): 4 = e2m1, 0 = e4m3, 1 = e5m2
    constexpr int FMT = (C::EBITS == 4) ? 4 : 0, FMTA = (C::EBITS == 4) ? 4 : C::AFMT;
    const int ops = (C::EBITS == 8 && C::F8SPLIT) ? 2 * j : j;
    if (C::ABL & ABL_NO_MFMA) {
#pragma unroll
      for (int t = 0; t < MT; ++t) asm volatile("" ::"v"(fa[j][t]));
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("" ::"v"(fb[j][t]));
      return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        // srcA = B fragment (rows of the MFMA = n), srcB = A fragment (cols = m)
        if (ops == 0) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j][n], fa[j][m], acc[m][n], FMT, FMTA, 0, sb[n], 0, sa[m]);
        if (ops == 1) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j][n], fa[j][m], acc[m][n], FMT, FMTA, 1, sb[n], 1, sa[m]);
        if (ops == 2) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j][n], fa[j][m], acc[m][n], FMT, FMTA, 2, sb[n], 2, sa[m]);
        if (ops == 3) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j][n], fa[j][m], acc[m][n], FMT, FMTA, 3, sb[n], 3, sa[m]);
      }
  }

  // ---- epilogue, register-direct: alpha, bf16, one v_permlane32_swap per dword to give each lane 8 contiguous
  //      columns (16 bytes) of its row, global_store_dwordx4.  A wave instruction writes 32 rows x 32 bytes; the
  //      other sectors of each 128-byte line follow from the same wave within a few instructions (L2 merges them).
  //      No LDS, no barrier: the stage buffers stay free for the next tile of a persistent loop.
  __device__ __forceinline__ void epilogue_direct() {
    const float alpha = alpha_k;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int grow = m0 + wave_m * C::WTM + 32 * m + i32;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int q0 = 2 * pr, q1 = 2 * pr + 1;
          uint32_t ax = pack_bf16x2(acc[m][n][4 * q0 + 0] * alpha, acc[m][n][4 * q0 + 1] * alpha);
          uint32_t ay = pack_bf16x2(acc[m][n][4 * q0 + 2] * alpha, acc[m][n][4 * q0 + 3] * alpha);
          uint32_t bx = pack_bf16x2(acc[m][n][4 * q1 + 0] * alpha, acc[m][n][4 * q1 + 1] * alpha);
          uint32_t by = pack_bf16x2(acc[m][n][4 * q1 + 2] * alpha, acc[m][n][4 * q1 + 3] * alpha);
          // lanes 32-63 of (ax, ay) <-> lanes 0-31 of (bx, by): lower half ends with columns +0..7, upper with +8..15
          auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
          auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
          const int gcol = n0 + wave_n * C::WTN + 32 * n + 16 * pr + 8 * g;
          if (grow < p.M && gcol < p.N) {
            const v4i v = {(int)rx[0], (int)ry[0], (int)rx[1], (int)ry[1]};
            if (C::ABL & ABL_NO_STORE) {
              if (v[0] == 0x12345678) p.D[(size_t)grow * p.ldd + gcol] = 1;
            } else {
              *(v4i*)(p.D + (size_t)grow * p.ldd + gcol) = v;
            }
          }
        }
    }
  }

  // ---- split-K partial: raw fp32 accumulators of this K range to ws[z][M][N] (alpha and the bf16 rounding happen in
  //      splitk_reduce_kernel); a lane owns 4 consecutive columns per q -> 16-byte stores
  __device__ __forceinline__ void epilogue_partial(int z) {
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
  }

#if QAMD_BENCH
  // LAB ONLY.  Measured on MI355X (profiles/native_r2_splitk_fused.log): 1.5 - 2.7 us SLOWER than the two-launch form at
  // M = 16 / 64 (10.8 / 13.6 vs 9.1 / 10.9 us at N = 4096, K = 14336) -- the reducing workgroup pays two memory round trips
  // (write-through acknowledgements, then the read-back across XCDs) plus the slot update in series, which costs more than
  // the ~3 us launch of a separate, fully parallel reduce kernel.  Kept selectable ("pp_flags" bit 9) so the number can be
  // reproduced; the product library does not contain it.
  // ---- split-K, ONE launch: every split stores its raw fp32 tile, the LAST split to arrive for the tile (arrival slot
  //      p.ctr[tile]) sums all S partials in z order -- deterministic, whichever split arrives last -- applies alpha, rounds
  //      to bf16 and writes D.  Cross-XCD visibility without cache flushes: the partials are stored write-through (sc0 sc1)
  //      and read back with sc0 sc1 loads, i.e. both sides meet at the memory-side coherence point; the only ordering
  //      needed is "my stores are acknowledged before I bump the slot" (s_waitcnt vmcnt(0) + workgroup barrier) -- the
  //      slot itself is a device-scope compare-and-swap.  The slot is reset to 0 by the reducing workgroup, so a replayed
  //      graph (same tag, same scratch) starts clean.
  __device__ __forceinline__ void epilogue_splitk_fused(int z) {
    const uint32_t plane = (uint32_t)p.M * (uint32_t)p.N * 4u;   // bytes of one partial matrix (< 2^28 by the host's tile limits)
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(p.ws, plane * (uint32_t)p.splits);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int grow = m0 + wave_m * C::WTM + 32 * m + i32;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gcol = n0 + wave_n * C::WTN + 32 * n + 8 * q + 4 * g;
          const v4f v = {acc[m][n][4 * q + 0], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
          const int off = (grow < p.M && gcol < p.N) ? (grow * p.N + gcol) * 4 : (int)0x80000000;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rW, off, (int)(plane * (uint32_t)z), 17);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial stores are acknowledged by the coherence point
    __syncthreads();                                    // ... and every wave's (also: nobody reads the stage buffers any more)
    if (tid == 0) {
      unsigned long long* slot = p.ctr + blockIdx.x;
      unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int cnt;
      for (;;) {
        cnt = ((cur & ~0xffull) == p.tag) ? (int)(cur & 0xffull) : 0;
        const unsigned long long want = p.tag | (unsigned long long)(cnt + 1);
        if (__hip_atomic_compare_exchange_strong(slot, &cur, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      *(volatile int*)smem = cnt + 1;
    }
    __syncthreads();
    const int arrived = uniform(*(volatile int*)smem);
    if (arrived != p.splits) return;
    // ---- last arrival: D tile = bf16(alpha * sum_z partial[z]) ----------------------------------------------------------
    const float alpha = alpha_k;
    constexpr int QPR = C::BN / 4;                 // float4 per tile row
    constexpr int RPP = C::THREADS / QPR;          // rows per pass
    constexpr int NPASS = C::BM / RPP;
    constexpr int GP = NPASS < 4 ? NPASS : 4;      // passes whose loads are in flight together (GP x 8 float4 per thread)
    const int c4 = (tid % QPR) * 4, r0 = tid / QPR;
#pragma unroll 1
    for (int pg = 0; pg < NPASS; pg += GP) {
      v4u t[GP][8];
#pragma unroll
      for (int u = 0; u < GP; ++u) {
        const int grow = m0 + (pg + u) * RPP + r0, gcol = n0 + c4;
        const int off = (grow < p.M && gcol < p.N) ? (grow * p.N + gcol) * 4 : (int)0x80000000;
#pragma unroll
        for (int zz = 0; zz < 8; ++zz)
          t[u][zz] = (zz < p.splits) ? __builtin_amdgcn_raw_buffer_load_b128(rW, off, (int)(plane * (uint32_t)zz), 17) : v4u{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < GP; ++u) {
        const int grow = m0 + (pg + u) * RPP + r0, gcol = n0 + c4;
        v4f sum = __builtin_bit_cast(v4f, t[u][0]);
#pragma unroll
        for (int zz = 1; zz < 8; ++zz)
          if (zz < p.splits) {
            const v4f x = __builtin_bit_cast(v4f, t[u][zz]);
            sum[0] += x[0]; sum[1] += x[1]; sum[2] += x[2]; sum[3] += x[3];
          }
        if (grow < p.M && gcol < p.N) {
          v2i o;
          o[0] = (int)pack_bf16x2(sum[0] * alpha, sum[1] * alpha);
          o[1] = (int)pack_bf16x2(sum[2] * alpha, sum[3] * alpha);
          *(v2i*)(p.D + (size_t)grow * p.ldd + gcol) = o;
        }
      }
    }
    if (tid == 0) __hip_atomic_store(p.ctr + blockIdx.x, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the slot clean
  }

#endif   // QAMD_BENCH

  // ---- epilogue: alpha, bf16, stage through LDS, whole-line stores ---------------------------
  __device__ __forceinline__ void epilogue() {
    if (p.pp_flags & 8) { epilogue_direct(); return; }
    if (C::ABL & ABL_NO_EPILOGUE) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) s += acc[m][n][r];
      if (s == 123456.789f) p.D[tid] = 1;
      return;
    }
    const float alpha = alpha_k;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = wave_m * C::WTM + (perm_rows ? 4 * i32 + m : 32 * m + i32);
          const int cg = (wave_n * C::WTN + 32 * n + 8 * q + 4 * g) >> 2;   // 8-byte granule in the row
          v2i w;
          w[0] = pack_bf16x2(acc[m][n][4 * q + 0] * alpha, acc[m][n][4 * q + 1] * alpha);
          w[1] = pack_bf16x2(acc[m][n][4 * q + 2] * alpha, acc[m][n][4 * q + 3] * alpha);
          // granule ^ (row & 15): the 16 lanes of a ds_write_b64 group (16 consecutive rows) hit 32 banks
          *(v2i*)(smem + row * C::SROW + ((cg ^ (row & 15)) << 3)) = w;
        }
    __syncthreads();
    constexpr int CPR = C::BN / 8;               // 16-byte chunks per tile row
    constexpr int RPP = C::THREADS / CPR;        // rows per pass
    const int chunk = tid % CPR, r0 = tid / CPR;
    const int gcol = n0 + chunk * 8;
#pragma unroll 4
    for (int pss = 0; pss < C::BM / RPP; ++pss) {
      const int row = pss * RPP + r0;
      const int grow = m0 + row;
      if (grow < p.M && gcol < p.N) {
        v4i v = *(const v4i*)(smem + row * C::SROW + ((((2 * chunk) ^ (row & 15)) & ~1) << 3));
        if (row & 1) v = v4i{v[2], v[3], v[0], v[1]};   // odd rows hold the granule pair swapped
        if (C::ABL & ABL_NO_STORE) {
          if (v[0] == 0x12345678) p.D[(size_t)grow * p.ldd + gcol] = 1;
        } else {
          if (p.pp_flags & 32) __builtin_nontemporal_store(v, (v4i*)(p.D + (size_t)grow * p.ldd + gcol)); else *(v4i*)(p.D + (size_t)grow * p.ldd + gcol) = v;
        }
      }
    }
  }
};

// The round-1 / alternative schedules (lockstep, ping-pong, queue, simple, per-tile deep, regstage, un-pipelined ring) live in
// lab/gemm_mx_lab.hip.h and exist in the lab build only; what follows is what the product library ships.
#if QAMD_BENCH
#include "lab/gemm_mx_lab.hip.h"
#endif

// ================================================================================================
// Pipelined ring schedule (product: ring variants 70..73 and the row-major-scale kernel of matmul_ada_mxf4_bf16_tn).
// Same LDS ring, same DMA stream, same K order -> bit-identical results.  What changes is WHEN a wave reads its fragments:
// gemm_mx_ring reads the whole stage after the barrier and waits for it before the first MFMA -- with one workgroup of
// four waves per CU (the ring fills the LDS) nothing else runs in the meantime: ~450 cycles of exposed LDS latency per
// stage against 128 (64x64 tiles) .. 512 (128x128) cycles of MFMAs.  Here the fragments live in TWO register sets: while
// the MFMAs of stage kt issue from one set, the reads of stage kt+1 into the other set and the DMA of stage kt+D are
// threaded between them, one or two per MFMA.  Per stage:
//     lgkmcnt(0) [stage kt in registers] ; own DMA of stage kt+1 landed ; BARRIER
//     MFMAs(kt) interleaved with  reads(kt+1 -> other set)  and  DMA(kt+D -> the slot of stage kt, free since the barrier)
// The ring is unrolled lcm(D, 2) times so that slot addresses are immediates and register sets alternate statically.
// ================================================================================================
// order of the auxiliary instructions of a pipelined-ring stage: position a of nr + nd -> read unit (>= 0) or DMA item (-1 - index);
// a DMA item after every second read unit, whatever is left of either kind at the end
// (reads_first: all read units before the first DMA item -- with ONE workgroup per CU nothing else covers the LDS latency of the last
//  reads at the top of the next stage, so they should not be the last instructions of this one)
constexpr int ringp_aux_item(int a, int nr, int nd, bool reads_first = false) {
  int r = 0, d = 0, last = 0;
  for (int pos = 0; pos <= a; ++pos) {
    const bool dma = (r >= nr) || (!reads_first && d < nd && pos % 3 == 2);
    if (dma) last = -1 - d++;
    else last = r++;
  }
  return last;
}

template <class C, bool RM = false>
__device__ __forceinline__ void gemm_mx_ringp(char* smem, const GemmParams& p, int bid = (int)blockIdx.x, int fm0 = -1, int fn0 = -1) {
  constexpr int KSL = C::KSL, D = C::NSTAGE, MT = C::MT, NT = C::NT, CPS = C::CPS;
  constexpr int LPS = C::NA + C::NB + 1;            // DMA instructions per wave per stage
  constexpr int U = (D % 2 == 0) ? D : 2 * D;       // unroll: slot = u % D, register set = u & 1
  static_assert(D >= 2 && (D - 2) * LPS <= 63, "vmcnt immediate");   // D = 2: one stage in flight, LDS of the simple schedule (two workgroups per CU)
  static_assert(!RM || (C::EBITS == 4 && C::BM == 64 && C::BN == 64 && C::NWAVES == 4), "row-major scales: 64x64 fp4 tiles");
  static_assert((C::ABL & ~ABL_READS_FIRST) == 0, "no ablation builds of this schedule");
  GemmCtx<C> cx(smem, p, bid, fm0, fn0);
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
    for (int t = 0; t < MT; ++t) cx.rdSA[t] = C::OFF_S + cx.g * 256 + (cx.wave_m * C::WTM + 32 * t + cx.i32) * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) cx.rdSB[t] = C::OFF_S + 512 + cx.g * 256 + (cx.wave_n * C::WTN + 32 * t + cx.i32) * 4;
  }
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  int kt0 = 0, kt1 = cx.KT;
  if (p.splits > 1) {
    const int per = (cx.KT + p.splits - 1) / p.splits;
    kt0 = uniform((int)blockIdx.y * per);
    kt1 = min(cx.KT, kt0 + per);
  }
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
  // one DMA instruction of stage kt (item 0 .. LPS-1: A pieces, B pieces, the scale piece); stages past the range re-load
  // the last one into a free slot that is never used: keeps the vmcnt arithmetic uniform
  int d_soff = 0, d_last = 0, d_ktc = 0;
  auto dma_prep = [&](int kt) __attribute__((always_inline)) {
    d_ktc = min(kt, kt1 - 1);
    d_soff = d_ktc * C::ROWB;
    d_last = (cx.ktail && d_ktc == cx.KT - 1) ? -1 : 0;
    asm volatile("" : "+v"(d_last));
  };
  auto dma_item = [&](const int slot, const int item) __attribute__((always_inline)) {
    char* st = smem + slot * C::STAGE_BYTES;
    if (item < C::NA) {
      const int t = item, v = (vAT[t] & d_last) | (vA[t] & ~d_last);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(cx.rA, (lds_ptr_t)(st + (cx.wave * C::NA + t) * 1024), 16, v, d_soff, 0, QAMD_DMA_AUX);
    } else if (item < C::NA + C::NB) {
      const int t = item - C::NA, v = (vBT[t] & d_last) | (vB[t] & ~d_last);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(cx.rB, (lds_ptr_t)(st + C::OFF_B + (cx.wave * C::NB + t) * 1024), 16, v, d_soff, 0, QAMD_DMA_AUX);
    } else if (RM) {
      const int vs = (d_ktc * 8 + (cx.wave & 1) * 4 < KBr) ? vSrm : 0x7fffffff;   // K tail: the stage's second scale dword does not exist
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rSrm, (lds_ptr_t)(st + C::OFF_S + cx.wave * 256), 4, vs, d_ktc * 8, 0, 0);
    } else {
      const int vs = (d_ktc * C::SCT + cx.colS < cx.CB) ? cx.voffS : 0x7fffffff;   // K tail: no such scale column tile
      __builtin_amdgcn_raw_ptr_buffer_load_lds(cx.rS, (lds_ptr_t)(st + C::OFF_S + cx.wave * 1024), 16, vs, d_ktc * C::SCT * 512, 0, 0);
    }
  };
  auto issue = [&](int kt, const int slot) __attribute__((always_inline)) {
    dma_prep(kt);
#pragma unroll
    for (int i = 0; i < LPS; ++i) dma_item(slot, i);
  };

  // two register sets: fragments of a whole stage + its raw scale dwords
  v8i fa[2][KSL][MT], fb[2][KSL][NT];
  int sa[2][MT], sb[2][NT];
  // read unit r of a stage: 0 = the A scale dwords, 1 = the B scale dwords, then per slice the MT A fragments and NT B fragments
  constexpr int NRU = 2 + KSL * (MT + NT);
  auto read_unit = [&](const int slot, const int set, const int r) __attribute__((always_inline)) {
    const char* st = smem + slot * C::STAGE_BYTES;
    if (r == 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t) sa[set][t] = *(const int*)(st + cx.rdSA[t]);
    } else if (r == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t) sb[set][t] = *(const int*)(st + cx.rdSB[t]);
    } else {
      const int j = (r - 2) / (MT + NT), t = (r - 2) % (MT + NT);
      const int boff = (t < MT) ? t * 32 * C::ROWB : cx.rdBd + (t - MT) * 32 * C::ROWB;
      const v4i lo = *(const v4i*)(st + boff + cx.rdA[j * CPS]);
      v4i hi = {0, 0, 0, 0};
      if (CPS == 2) hi = *(const v4i*)(st + boff + cx.rdA[j * CPS + CPS - 1]);
      const v8i v = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (t < MT) fa[set][j][t] = v; else fb[set][j][t - MT] = v;
    }
  };
  constexpr int FMT = (C::EBITS == 4) ? 4 : 0, FMTA = (C::EBITS == 4) ? 4 : C::AFMT;
  auto mfma1 = [&](const int set, const int j, const int m, const int n) __attribute__((always_inline)) {
    const int ops = (C::EBITS == 8 && C::F8SPLIT) ? 2 * j : j;
    v16f& c = cx.acc[m][n];
    if (ops == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[set][j][n], fa[set][j][m], c, FMT, FMTA, 0, sb[set][n], 0, sa[set][m]);
    if (ops == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[set][j][n], fa[set][j][m], c, FMT, FMTA, 1, sb[set][n], 1, sa[set][m]);
    if (ops == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[set][j][n], fa[set][j][m], c, FMT, FMTA, 2, sb[set][n], 2, sa[set][m]);
    if (ops == 3) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[set][j][n], fa[set][j][m], c, FMT, FMTA, 3, sb[set][n], 3, sa[set][m]);
  };
  const int sshift = (C::EBITS == 4) ? 0 : (C::F8SPLIT ? 8 * cx.g : 16 * cx.g);   // fp8: bring "my" first scale byte down
  auto top = [&](const int set) __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0), as a builtin (the compiler's scoreboard sees it): this stage's fragments are in registers
    if (C::EBITS == 8) {
#pragma unroll
      for (int t = 0; t < MT; ++t) sa[set][t] = (int)((unsigned)sa[set][t] >> sshift);
#pragma unroll
      for (int t = 0; t < NT; ++t) sb[set][t] = (int)((unsigned)sb[set][t] >> sshift);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * LPS) : "memory");   // own pieces of the NEXT stage landed (DMA retires in order)
    __builtin_amdgcn_s_barrier();
    fence();
  };
  // MFMAs of stage kt (set `set`) with the reads of stage kt+1 (slot `nslot` -> the other set) and the DMA of stage kt+D (-> `slot`)
  // threaded through; auxiliary item i of NAUX goes after MFMA floor(i * NM / NAUX)
  constexpr int NM = KSL * MT * NT, NAUX = NRU + LPS;
  auto stage = [&](int kt, auto slotc, auto setc) __attribute__((always_inline)) {
    constexpr int slot = decltype(slotc)::value, set = decltype(setc)::value, nslot = (slot + 1) % D;
    top(set);
    dma_prep(kt + D);
    fence();
    static_for<0, NM>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int j = i / (MT * NT), m = (i / NT) % MT, n = i % NT;
      mfma1(set, j, m, n);
      // auxiliary items [i * NAUX / NM, (i + 1) * NAUX / NM): reads first-come, a DMA item after every second read unit
      static_for<i * NAUX / NM, (i + 1) * NAUX / NM>([&](auto ac) __attribute__((always_inline)) {
        constexpr int a = decltype(ac)::value;
        constexpr int code = ringp_aux_item(a, NRU, LPS, (C::ABL & ABL_READS_FIRST) != 0);   // >= 0: read unit, < 0: DMA item -1 - code
        if constexpr (code >= 0) read_unit(nslot, set ^ 1, code);
        else dma_item(slot, -1 - code);
      });
      fence();
    });
  };

#pragma unroll
  for (int s = 0; s < D - 1; ++s) issue(kt0 + s, s);
  asm volatile("" :: "s"(cx.alpha_k));   // alpha is waited for HERE, behind the first stages' DMA (left alone, its load is sunk to the epilogue)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * LPS) : "memory");   // stage kt0 landed
  __builtin_amdgcn_s_barrier();
  fence();
#pragma unroll
  for (int r = 0; r < NRU; ++r) read_unit(0, 0, r);
  issue(kt0 + D - 1, D - 1);
  fence();
  for (int kt = kt0; kt < kt1; kt += U) {
    static_for<0, U>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if (u == 0 || kt + u < kt1) stage(kt + u, std::integral_constant<int, u % D>{}, std::integral_constant<int, u & 1>{});
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing re-loads must land before the epilogue reuses the LDS
  __builtin_amdgcn_s_waitcnt(0xc07f);                // and the look-ahead reads of the stage past the end
#if QAMD_BENCH
  if (p.splits > 1 && p.ctr) cx.epilogue_splitk_fused(blockIdx.y);
  else
#endif
  if (p.splits > 1) cx.epilogue_partial(blockIdx.y);
  else cx.epilogue();
}

// split-K second pass: D = bf16(alpha * sum_z ws[z]) in fixed z order (deterministic); 4 columns per thread.  S is a
// template parameter so that all S loads of a thread are in flight together (a runtime loop serialises S memory round
// trips: 4.8 us for a 1 MB output).
template <int S>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, uint16_t* __restrict__ D, const float* __restrict__ alpha_p,
                                                            int M, int N, int ldd) {
  const int64_t quads = (int64_t)M * (N >> 2);
  const float alpha = *alpha_p;
  const size_t plane = (size_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / (N >> 2)), c4 = (int)(i % (N >> 2)) * 4;
    const float* src = ws + (size_t)row * N + c4;
    v4f t[S];
#pragma unroll
    for (int z = 0; z < S; ++z) t[z] = __builtin_nontemporal_load((const v4f*)(src + z * plane));
    v4f s = t[0];
#pragma unroll
    for (int z = 1; z < S; ++z) { s[0] += t[z][0]; s[1] += t[z][1]; s[2] += t[z][2]; s[3] += t[z][3]; }
    v2i o;
    o[0] = (int)pack_bf16x2(s[0] * alpha, s[1] * alpha);
    o[1] = (int)pack_bf16x2(s[2] * alpha, s[3] * alpha);
    *(v2i*)(D + (size_t)row * ldd + c4) = o;
  }
}

// One __global__ entry per (config, schedule).
enum { SCHED_LOCKSTEP = 0, SCHED_PINGPONG = 1, SCHED_QUEUE = 2, SCHED_SIMPLE = 3, SCHED_DEEP = 4, SCHED_REGSTAGE = 5, SCHED_DEEP_NN = 6, SCHED_RING = 7, SCHED_RING_RM = 8, SCHED_RINGP = 9, SCHED_RINGP_RM = 10 };
// pipelined schedule on a 2-deep ring: sized for two workgroups per CU, i.e. (4-wave configurations) two waves per SIMD = 256 registers
template <class C, int SCHED>
constexpr int gemm_min_waves_per_eu() { return ((SCHED == 9 || SCHED == 10) && C::NSTAGE == 2 && C::NWAVES == 4 && C::BM * C::BN <= 128 * 128) ? 2 : 1; }
template <class C, int SCHED>
__global__ __launch_bounds__(C::THREADS, (gemm_min_waves_per_eu<C, SCHED>())) void gemm_mx_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
  // ABL_CLOCK builds only (qutlass_amd_debug_set_trace_buffer): workgroup 0 reports its shader-cycle and
  // 100 MHz wall-clock duration, i.e. the clock the chip actually ran at under this kernel's power draw
  const bool clk = (C::ABL & ABL_CLOCK) && p.dbg && blockIdx.x == 0 && threadIdx.x == 0;
  const uint64_t c0 = clk ? __builtin_readcyclecounter() : 0, r0 = clk ? __builtin_amdgcn_s_memrealtime() : 0;
#if QAMD_RING_KERNARG_EARLY
  // every scalar argument asked for at entry: ONE scalar-load round ahead of the first LDS-DMA instead of two (what gemm_mx_deepp_kernel does since round 4).
  // Prepared at the end of round 4, not measured yet -- build with tools/build_variant.py ... -DQAMD_RING_KERNARG_EARLY=1 and A/B the mid-size shapes.
  asm volatile("" :: "s"(p.A), "s"(p.D), "s"(p.K), "s"(p.b_bytes), "s"(p.ws), "s"(p.splits), "s"(p.ctr), "s"(p.tag));
#endif
  if constexpr (SCHED == SCHED_RINGP_RM) gemm_mx_ringp<C, true>(smem, p);
  else if constexpr (SCHED == SCHED_RINGP) gemm_mx_ringp<C>(smem, p);
#if QAMD_BENCH   // gemm_mx_lab.hip.h
  else if constexpr (SCHED == SCHED_RING_RM) gemm_mx_ring<C, true>(smem, p);
  else if constexpr (SCHED == SCHED_RING) gemm_mx_ring<C>(smem, p);
  else if constexpr (SCHED == SCHED_REGSTAGE) gemm_mx_regstage<C>(smem, p);
  else if constexpr (SCHED == SCHED_DEEP_NN) gemm_mx_deep8<C, true>(smem, p);
  else if constexpr (SCHED == SCHED_DEEP && C::EBITS == 8) gemm_mx_deep8<C, false>(smem, p);
  else if constexpr (SCHED == SCHED_DEEP) gemm_mx_deep<C>(smem, p);
  else if constexpr (SCHED == SCHED_SIMPLE) gemm_mx_simple<C>(smem, p);
  else if constexpr (SCHED == SCHED_QUEUE) gemm_mx_queue<C>(smem, p);
  else if constexpr (SCHED == SCHED_PINGPONG) gemm_mx_pingpong<C>(smem, p);
  else gemm_mx_lockstep<C>(smem, p);
#else
  else static_assert(SCHED == SCHED_RINGP || SCHED == SCHED_RINGP_RM, "the product library ships the pipelined ring schedule only (the others: gemm_mx_lab.hip.h, lab build)");
#endif
  if (clk) {
    p.dbg[0] = (uint32_t)(__builtin_readcyclecounter() - c0);
    p.dbg[1] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - r0);
  }
}

}  // namespace qamd
