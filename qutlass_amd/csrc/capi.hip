// C ABI (include/qutlass_amd.h) -> kernel launches.  No torch types, no allocation, no sync.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/qutlass_amd.h"
#include "gemm_mx.hip.h"
#include "gemm_mx_deepp.hip.h"
#if QAMD_BENCH   // csrc/lab/: sources of the LAB library only -- not read by the product build, not shipped by setup.py
#include "lab/gemm_mx_duo.hip.h"         // [r6] the 8-wave persistent kernel (built, bit-identical, slower under the power cap: DESIGN.md section 7)
#include "lab/gemm_mx_deepp_lab.hip.h"   // the lab copy (namespace qamd::labk): traces, ablations, stream-K, retirement experiments
#endif
#include "gemm_mx_skinny.hip.h"
#include "gemm_mx_ks.hip.h"
#include "gemm_mx_os.hip.h"
#include "gemm_mx_fusedq.hip.h"
#include "gemm_nvf4.hip.h"
#include "gemm_nvf4_pk.hip.h"
#include "gemm_nvf4_os.hip.h"
#include "quantize.hip.h"
#include "to_blocked.hip.h"
#include "transpose_u8.hip.h"
#include "quartet_bwd.hip.h"
#if QAMD_BENCH
#include "lab/quartet_bwd_lab.hip.h"     // backward_qt_bf16's whole-line panel kernel (built, bit-identical, not faster)
#endif

using namespace qamd;

// One source, several translation units, two libraries.
//   libqutlass_amd.so        the PRODUCT: the C ABI of include/qutlass_amd.h and exactly the kernels its dispatch rules can
//                            reach.  No process-wide switch can change which kernel runs or what it computes: qutlass_amd_set_option
//                            knows no key ([r4] the "hw_fp4_cvt" encoder switch of rounds 1-3 lives in the lab build only).
//   libqutlass_amd_bench.so  the LAB (-DQAMD_BENCH=1): the same entry points plus every schedule variant, ablation, trace
//                            and clock-probe instantiation the design record cites, selectable through
//                            qutlass_amd_set_option("gemm_variant" / "nvf4_variant" / "pp_flags" / ...).  Only
//                            tests/native, the sweeps under tools/ and the forced-tile parity tests load it.
// build.py compiles this file once per QAMD_TU value in parallel: every heavy template family is instantiated in exactly
// one unit (explicit instantiation) and only declared (`extern template`) in the others.  QAMD_TU = 0 (a plain
// `hipcc capi.hip`) still gives a whole library in one unit.
//   1  C entry points, dispatch rules, small kernels      2  MXFP4 tile / schedule variants      3  MXFP8 variants
//   4  NVFP4 kernels + MXFP8 with an e5m2 A operand        5  fused quantizers
//   6  MXFP4 ablations (100+, 200+, 300+; lab only)       7  NVFP4 v2 ablations (gemm_nvf4.hip.h; lab only)
//   8  the persistent NVFP4 kernel (gemm_nvf4_pk.hip.h)
#ifndef QAMD_TU
#define QAMD_TU 0
#endif
#define QAMD_DEF(n) (QAMD_TU == 0 || QAMD_TU == (n))
#ifndef QAMD_BENCH
#define QAMD_BENCH 0
#endif

namespace qamd_host {

#if QAMD_DEF(1)
thread_local char g_err[512] = "";
#if QAMD_BENCH
std::atomic<int> g_hw_fp4_cvt{1};   // lab: 0 = the software e2m1 encoder (device-verified bit-identical to v_cvt_scalef32_pk_fp4_f32, tests/native/probe.hip P2)
std::atomic<int> g_gemm_variant{0};
std::atomic<int> g_nvf4_variant{0};
std::atomic<int> g_splitk_wg{0};        // split-K: target workgroup count ("splitk_wg"; one per CU measured best, profiles/native_r1_splitk_wg.log)
std::atomic<int> g_splitk_min_kt{32};   // split-K: minimum number of 128-byte K stages ("splitk_min_kt"; 16 loses at K = 4096, 48 leaves K = 8192 .. 11008 unsplit)
std::atomic<int> g_transpose_nc{0};     // lab: kernel forcing of mxfp4_transpose_mxfp8 (128 / 256 / 3 / 4 / 2) and backward_bf16_square_double_mxfp8 (1 / 4 / 8); 0 = the product rules
std::atomic<int> g_pp_shift{2};
std::atomic<int> g_pp_flags{1};
std::atomic<int> g_quant_wg_per_cu{0};  // 0 = auto
std::atomic<int> g_deepp_grid{0};       // lab: workgroups of the persistent deep kernels (0 = the balanced-rounds rule)
std::atomic<int> g_bwd_variant{0};      // lab: 1 = the round-3 backward_t / backward_qt kernel (8 waves meeting at two barriers per tile) instead of the wave-owned-lines one
std::atomic<int> g_splitk_force{0};     // lab: K splits for a FORCED ring variant ("gemm_variant" 70..73); 0 = the plan's
std::atomic<uint32_t*> g_dbg{nullptr};
#endif

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QAMD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return QAMD_OK;
}

// Per-launch number of the fused split-K kernels (host state only; no device memory behind it): starts at a value drawn
// from the clock and the process, then counts up.  56 bits of it tag the arrival slots of one launch.
unsigned long long next_launch_tag() {
  static std::atomic<unsigned long long> tag{[] {
    unsigned long long x = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((unsigned long long)(uintptr_t)&g_err << 17);
    x ^= x >> 31; x *= 0x9e3779b97f4a7c15ull; x ^= x >> 29;
    return x | 1ull;
  }()};
  return tag.fetch_add(1, std::memory_order_relaxed) + 1;
}
#else
unsigned long long next_launch_tag();
#if QAMD_BENCH
extern std::atomic<int> g_hw_fp4_cvt, g_gemm_variant, g_nvf4_variant, g_splitk_wg, g_splitk_min_kt, g_transpose_nc, g_pp_shift, g_pp_flags, g_quant_wg_per_cu, g_splitk_force, g_deepp_grid, g_bwd_variant;
extern std::atomic<uint32_t*> g_dbg;
#endif
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
#endif

// Tuning state.  In the product library these are compile-time constants: nothing a caller (or another thread) does can
// change which kernel a shape gets.  The lab library reads them from qutlass_amd_set_option().
#if QAMD_BENCH
inline bool opt_hw_fp4() { return g_hw_fp4_cvt.load() != 0; }
inline int opt_gemm_variant() { return g_gemm_variant.load(); }
inline int opt_nvf4_variant() { return g_nvf4_variant.load(); }
inline int opt_splitk_wg() { return g_splitk_wg.load(); }   // 0 = one workgroup per CU
inline int opt_splitk_min_kt() { return g_splitk_min_kt.load(); }
inline int opt_transpose_nc() { return g_transpose_nc.load(); }
inline int opt_pp_shift() { return g_pp_shift.load(); }
inline int opt_pp_flags() { return g_pp_flags.load(); }
inline int opt_quant_wg_per_cu() { return g_quant_wg_per_cu.load(); }
inline int opt_splitk_force() { return g_splitk_force.load(); }
inline int opt_deepp_grid() { return g_deepp_grid.load(); }
inline int opt_bwd_variant() { return g_bwd_variant.load(); }
inline uint32_t* opt_dbg() { return g_dbg.load(); }
#else
constexpr bool opt_hw_fp4() { return true; }   // the product ships ONE e2m1 encoder: the hardware convert
constexpr int opt_gemm_variant() { return 0; }
constexpr int opt_nvf4_variant() { return 0; }
constexpr int opt_splitk_wg() { return 0; }
constexpr int opt_splitk_min_kt() { return 32; }
constexpr int opt_transpose_nc() { return 0; }
constexpr int opt_pp_shift() { return 2; }
constexpr int opt_pp_flags() { return 1; }
constexpr int opt_quant_wg_per_cu() { return 0; }
constexpr int opt_splitk_force() { return 0; }
constexpr int opt_deepp_grid() { return 0; }
constexpr int opt_bwd_variant() { return 0; }
constexpr uint32_t* opt_dbg() { return nullptr; }
#endif

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Dry run (qutlass_amd_debug_gemm_plan): the dispatch code below runs unchanged, but instead of launching it records
// which kernels it WOULD launch -- {variant, N of the launch, K splits} per launch -- so the auto rules are testable on a
// machine without a GPU (tests/test_cabi_and_host.py).
struct DryRun { bool on = false; int n = 0; int rec[8][3]; };
#if QAMD_DEF(1)
thread_local DryRun t_dry;
#else
extern thread_local DryRun t_dry;
#endif
// Compute units of the current device, asked once per device (hipDeviceAttributeMultiprocessorCount: 256 on an MI355X in SPX
// mode; fewer in a partitioned / CU-masked configuration).  Every grid size and occupancy threshold below derives from it.
// (The kernels' blockIdx -> XCD remap assumes the dispatcher's round-robin over 8 XCDs; it is a bijection for any grid, so
// on another XCD count it only costs L2 locality.)  The dry-run hook describes a full MI355X whatever the machine.
#if QAMD_DEF(1)
int chip_cus() {
  if (t_dry.on) return 256;
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v <= 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
    v = n;
  }
  return v;
}
#else
int chip_cus();
#endif

inline bool dry_record(int variant, int n_cols, int splits) {
  if (!t_dry.on) return false;
  if (t_dry.n < 8) { t_dry.rec[t_dry.n][0] = variant; t_dry.rec[t_dry.n][1] = n_cols; t_dry.rec[t_dry.n][2] = splits; }
  ++t_dry.n;
  return true;
}


template <class C, int PP>
int launch_gemm(GemmParams p, hipStream_t s) {
  p.tiles_m = (int)cdiv(p.M, C::BM);
  p.tiles_n = (int)cdiv(p.N, C::BN);
  p.raster_magic = raster_magic(p.tiles_n);
  if (PP != 7 && PP != 9) { p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0; }   // only the (blocked-scale) ring schedules know about split-K
  hipLaunchKernelGGL((gemm_mx_kernel<C, PP>), dim3(p.tiles_m * p.tiles_n, p.splits), dim3(C::THREADS), 0, s, p);
  return check_launch("gemm_mx_kernel");
}

// Workgroups of a persistent deep launch over T tiles of 256x256 on the 256 CUs of an MI355X (one 512-register workgroup per
// CU): BALANCED rounds -- R = ceil(T / 256) tiles per workgroup, G = ceil(T / R) workgroups rounded up to a multiple of 8 (one
// share per XCD) -- instead of 256 workgroups with a ragged last round.  The kernels run at the socket power limit, so the CUs
// a smaller grid leaves idle are not lost: the others clock higher (896 tiles: 224 workgroups x 4 tiles beat 256 workgroups
// x 3.5 rounds, profiles/native_r2_deepp_grid.log).
inline int deepp_grid(int tiles) {
  const int forced = opt_deepp_grid();
  if (forced > 0) return std::min(forced, tiles);
  const int cus = chip_cus();
  const int rounds = (tiles + cus - 1) / cus;
  const int g = ((tiles + rounds - 1) / rounds + 7) / 8 * 8;
  return std::min(std::min(g, cus), tiles);
}

// persistent deep schedule (gemm_mx_deepp.hip.h): one workgroup per CU walks the tiles; 136 KiB of static LDS
template <class C, bool TRACE = false, int ST_AUX = 0, int LAB = 0>
int launch_gemm_deepp(GemmParams p, hipStream_t s) {
  p.tiles_m = (int)cdiv(p.M, C::BM);
  p.tiles_n = (int)cdiv(p.N, C::BN);
  p.raster_magic = raster_magic(p.tiles_n);
  p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
  const int grid = deepp_grid(p.tiles_m * p.tiles_n);
#if QAMD_BENCH
#ifdef QAMD_ROUTE_LABK   // side builds for schedule A/Bs (tools/build_variant.py --lab ... -DQAMD_ROUTE_LABK): the plain entry runs the LAB copy of the kernel, no ablation
  if constexpr (!TRACE && LAB == 0) {
    hipLaunchKernelGGL((labk::gemm_mx_deepp_kernel<C, false, ST_AUX, 8>), dim3(grid), dim3(C::THREADS), 0, s, p);
    return check_launch("labk::gemm_mx_deepp_kernel");
  }
#endif
  if constexpr (TRACE || LAB != 0) {   // stage traces and result-changing ablations: the lab copy of the kernel
    hipLaunchKernelGGL((labk::gemm_mx_deepp_kernel<C, TRACE, ST_AUX, LAB>), dim3(grid), dim3(C::THREADS), 0, s, p);
    return check_launch("labk::gemm_mx_deepp_kernel");
  }
#else
  static_assert(!TRACE && LAB == 0, "traces / ablations: lab build only");
#endif
  // [r6] two host-side choices of the kernel form (gemm_mx_deepp.hip.h): ODD -- an odd number (>= 3) of K stages runs without the empty stage; ONETILE -- every workgroup
  // walks exactly one tile (grid == tile count), so nothing is prefetched for a next tile
  const int64_t kt = cdiv((int64_t)p.K * C::EBITS / 8, 128);
  const bool odd = (kt & 1) && kt >= 3, one = grid == p.tiles_m * p.tiles_n;
  if (odd && one) hipLaunchKernelGGL((gemm_mx_deepp_kernel<C, ST_AUX, true, true>), dim3(grid), dim3(C::THREADS), 0, s, p);
  else if (odd) hipLaunchKernelGGL((gemm_mx_deepp_kernel<C, ST_AUX, true, false>), dim3(grid), dim3(C::THREADS), 0, s, p);
  else if (one) hipLaunchKernelGGL((gemm_mx_deepp_kernel<C, ST_AUX, false, true>), dim3(grid), dim3(C::THREADS), 0, s, p);
  else hipLaunchKernelGGL((gemm_mx_deepp_kernel<C, ST_AUX>), dim3(grid), dim3(C::THREADS), 0, s, p);
  return check_launch("gemm_mx_deepp_kernel");
}

#if QAMD_BENCH
// [r6] the 8-wave persistent schedule (lab/gemm_mx_duo.hip.h): same grid rule, 144 KiB of static LDS.  RET: retirement placement (0 burst, 1 behind the last k-slice)
template <class C, int ST_AUX = 17, int RET = 0, bool TRACE = false>
int launch_gemm_duo(GemmParams p, hipStream_t s) {
  p.tiles_m = (int)cdiv(p.M, C::BM);
  p.tiles_n = (int)cdiv(p.N, C::BN);
  p.raster_magic = raster_magic(p.tiles_n);
  p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
  const int grid = deepp_grid(p.tiles_m * p.tiles_n);
  hipLaunchKernelGGL((gemm_mx_duo_kernel<C, ST_AUX, RET, TRACE>), dim3(grid), dim3(C::THREADS), 0, s, p);
  return check_launch("gemm_mx_duo_kernel");
}
#endif

// [r6] small-batch kernel with the K split inside the workgroup (gemm_mx_ks.hip.h): one 32x32 / 32x64 / 64x32 tile per workgroup of four waves
template <int TM, int TN, int D = 8>
int launch_gemm_ks(GemmParams p, hipStream_t s) {
  using C = KsCfg<TM, TN, D>;
  p.tiles_m = (int)cdiv(p.M, TM);
  p.tiles_n = (int)cdiv(p.N, TN);
  p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
  hipLaunchKernelGGL((gemm_mx_ks_kernel<C>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
  return check_launch("gemm_mx_ks_kernel");
}

// [r6] small-batch kernel whose waves own K stages (gemm_mx_os.hip.h): wave w owns stages w, w + 4, ...; the tile's whole K extent requested up front while it fits the
// LDS (16 stages of 128 bytes per row for a 32-row tile, 12 for a 64-row tile), wave-owned rings beyond
// (RM: row-major scale operands -- matmul_ada_mxf4_bf16_tn; TN: columns per workgroup, 32 or 16; EBITS 8: MXFP8, AFMT 1: e5m2 A operand; TM: rows per workgroup, 32 or 64)
template <bool RM = false, int TN = 32, int EBITS = 4, int TM = 32, int AFMT = 0>
int launch_gemm_os(GemmParams p, hipStream_t s) {
  p.tiles_m = (int)cdiv(p.M, TM);
  p.tiles_n = (int)cdiv(p.N, TN);
  p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
  const int64_t KT = cdiv((int64_t)p.K * EBITS / 8, 128);
  const dim3 grid(p.tiles_m * p.tiles_n), block(256);
  constexpr int SMAX = TM == 32 ? 4 : 3;   // slots per wave that fit the LDS
  if (KT <= 4) hipLaunchKernelGGL((gemm_mx_os_kernel<OsCfg<1, TN, EBITS, TM, AFMT>, RM>), grid, block, 0, s, p);
  else if (KT <= 8) hipLaunchKernelGGL((gemm_mx_os_kernel<OsCfg<2, TN, EBITS, TM, AFMT>, RM>), grid, block, 0, s, p);
  else if (KT <= 12) hipLaunchKernelGGL((gemm_mx_os_kernel<OsCfg<3, TN, EBITS, TM, AFMT>, RM>), grid, block, 0, s, p);
  else if (KT <= 4 * SMAX) hipLaunchKernelGGL((gemm_mx_os_kernel<OsCfg<SMAX, TN, EBITS, TM, AFMT>, RM>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_mx_os_kernel<OsCfg<SMAX, TN, EBITS, TM, AFMT>, RM, true>), grid, block, 0, s, p);   // wave-owned rings of SMAX slots
  return check_launch("gemm_mx_os_kernel");
}
// [r6] its decode form (gemm_mx_os16_kernel: 16 x TN tiles on the 16x16x128 MFMA, rows in tiles of 16): one shot while the tile's K extent fits the LDS (SMAX slots per wave),
// wave-owned rings of SMAX slots beyond
template <int EBITS = 4, int AFMT = 0, bool RM = false, int TN = 16>
int launch_gemm_os16(GemmParams p, hipStream_t s) {
  p.tiles_m = (int)cdiv(p.M, 16);
  p.tiles_n = (int)cdiv(p.N, TN);
  p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
  const int64_t KT = cdiv((int64_t)p.K * EBITS / 8, 128);
  const dim3 grid(p.tiles_m * p.tiles_n), block(256);
  constexpr int STAGE = Os16Cfg<1, EBITS, AFMT, TN>::STAGE;
  constexpr int SFIT = 160 * 1024 / (4 * STAGE), SMAX = SFIT >= 8 ? 8 : SFIT >= 4 ? 4 : SFIT;   // slots per wave that fit the LDS: 8 (TN = 16), 4 (32 ... 56), 3 (64)
  static_assert(SMAX >= 3, "LDS budget");
  if (KT <= 4) hipLaunchKernelGGL((gemm_mx_os16_kernel<Os16Cfg<1, EBITS, AFMT, TN>, false, RM>), grid, block, 0, s, p);
  else if (KT <= 8) hipLaunchKernelGGL((gemm_mx_os16_kernel<Os16Cfg<2, EBITS, AFMT, TN>, false, RM>), grid, block, 0, s, p);
  else if (KT <= 16 && SMAX >= 4) hipLaunchKernelGGL((gemm_mx_os16_kernel<Os16Cfg<(SMAX >= 4 ? 4 : SMAX), EBITS, AFMT, TN>, false, RM>), grid, block, 0, s, p);
  else if (KT <= 4 * SMAX) hipLaunchKernelGGL((gemm_mx_os16_kernel<Os16Cfg<SMAX, EBITS, AFMT, TN>, false, RM>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_mx_os16_kernel<Os16Cfg<SMAX, EBITS, AFMT, TN>, true, RM>), grid, block, 0, s, p);
  return check_launch("gemm_mx_os16_kernel");
}
// columns per workgroup of the decode form for M <= 16: the narrowest of 16 / 32 / 48 / 56 / 64 that leaves at most one workgroup per CU; 0 = none does
inline int os16_tn(int64_t N) {
  const int64_t cus = chip_cus();
  for (int tn : {16, 32, 48, 56, 64})
    if (cdiv(N, tn) <= cus) return tn;
  return 0;
}
// [r6] M <= 16 against a weight too wide for 16-column workgroups: the decode form with 32 / 48 / 56 / 64 columns per workgroup (variants 572 ... 575) where it was measured ahead
// (tools/calib_os2.py, profiles/calib_os16w_r7.txt; K in stages of 128 bytes per row).  MXFP4: N = 11008 / 12288 x K = 4096 5.0-5.2 -> 3.8-4.0 us (48 columns: 64 rows per
// stage where the 32x64 K-split ring kernel fetched 96 and met at a barrier per stage), 14336 x 4096 5.9-6.1 -> 4.6-4.9 (56 columns = exactly 256 workgroups), 12288 x 5120
// 7.5-8.0 -> 5.6-6.1, 16384 x 4096 -7 %; N = 5120 ... 8192 (32 columns against the 32x32 form's 64 rows per stage) -3 ... -7 % at K = 3072 ... 8192, +3 % at K = 2048.
// Past the one-shot range the wider forms lose (14336 x 8192: +3 %).  MXFP8: 32 columns -6 ... -12 % at K = 2048 ... 8192, 48 columns -9 % at 11008 x 4096; 14336 x 4096 loses.
// Returns the variant or 0.
inline int os16_wide_plan(int ebits, int64_t N, int64_t K) {
  const int tn = os16_tn(N);
  const int64_t KT = cdiv(K * ebits / 8, 128);
  if (ebits == 4) {
    if (tn == 32) return (KT >= 12 && KT <= 32) ? 572 : 0;
    if (tn == 48) return (KT >= 8 && KT <= 24) ? 573 : 0;
    if (tn == 56) return (KT >= 8 && KT <= 16) ? 574 : 0;
    if (tn == 64) return (KT >= 8 && KT <= 16) ? 575 : 0;
    return 0;
  }
  if (tn == 32) return (KT >= 16 && KT <= 64) ? 572 : 0;
  if (tn == 48) return (KT >= 16 && KT <= 32) ? 573 : 0;
  return 0;
}
template <int EBITS, int AFMT, bool RM>
int launch_gemm_os16_tn(int tn, const GemmParams& p, hipStream_t s) {
  switch (tn) {
    case 16: return launch_gemm_os16<EBITS, AFMT, RM, 16>(p, s);
    case 32: return launch_gemm_os16<EBITS, AFMT, RM, 32>(p, s);
    case 48: return launch_gemm_os16<EBITS, AFMT, RM, 48>(p, s);
    case 56: return launch_gemm_os16<EBITS, AFMT, RM, 56>(p, s);
    case 64: return launch_gemm_os16<EBITS, AFMT, RM, 64>(p, s);
  }
  return fail(QAMD_ERR_INVALID, "gemm_mx_os16: no column tile of %d", tn);
}
// Does the one-shot kernel take the shape, and with how many columns per workgroup?  Returns 0 (no), 32 or 16.  32x32 tiles, one per CU at most; 16 columns per
// workgroup whenever that still leaves one workgroup per CU (N = 4096, M <= 32: 256 workgroups pulling half the bytes each -- 3.34 -> 3.22 us at K = 4096, 7.8 -> 6.9 at
// K = 14336).  K <= 16 stages of 256: always.  Longer K (wave-owned rings): up to 32 stages when the tiles fill a quarter of the chip, up to 64 stages (K = 16384) when
// they fill half of it (N = 1024, K = 14336: 32 workgroups take 7.4 us where the split-K plans take 5.7-6.9); with 32 columns per workgroup the blocked-scale op also keeps
// M < 8 against more than 32 stages on the LDS-free split-K kernel (4096 x 11008, M = 1: 6.07 vs 6.40 us).  ada: matmul_ada_mxf4_bf16_tn (row-major scales: its other
// kernels are 25-45 % slower on every such shape).  profiles/calib_os_r6q.txt, calib_osring_r6q.txt, calib_ada_r6q.txt, calib_os16_r6s.txt
inline int os_plan(int64_t M, int64_t N, int64_t K, bool ada = false) {
  const int64_t cus = chip_cus(), T32 = cdiv(M, 32) * cdiv(N, 32), T16 = cdiv(M, 32) * cdiv(N, 16), KT = cdiv(K, 256);
  if (T32 > cus) return 0;
  const int tn = T16 <= cus ? 16 : 32;
  if (KT <= 16) return tn;
  if (KT <= 32) return 4 * T32 >= cus ? tn : 0;
  if (KT > 64 || 2 * T32 < cus) return 0;   // (measured with 128 and 256 tiles only: fewer tiles against K > 8192 stay with the split plans)
  return (ada || M >= 8 || tn == 16) ? tn : 0;
}

// [r6, third session] ... and its 64-row form (variant 570: two m-tiles per stage owner) for MXFP4 batches whose 32x32 tiles no longer fit one per CU while 64x32 tiles do
// (M = 65 ... 128 at N = 4096, M = 33 ... 64 at N = 8192).  A 64x32 tile's K extent fits the LDS up to 12 stages (K <= 3072) -- there it is the one-shot kernel and wins
// 8-13 % (4096 x 2048, M = 96 / 128: 4.10 / 4.14 -> 3.57 / 3.63 us; 2048^2, M = 160 ... 256: 4.07-4.22 -> 3.51-3.70).  Beyond, the wave-owned rings hold three stages per
// wave and a fourth costs a second memory round trip: K = 4096 ... 6144 lose 0-8 % against the 64x64 ring tiles (N = K = 4096, M = 128: 5.20 vs 5.60 us) and stay there; from
// ~40 stages on the round trips overlap and it wins again (4096 x 11008, M = 96 / 128: 12.05 / 12.59 -> 9.19 / 9.73; x 14336: 13.5 / 14.1 -> 11.7 / 12.4; 8192^2, M = 64,
// 32 stages on every CU: 11.65 -> 9.66).  profiles/calib_os2_fp4_r7.txt
// ada: matmul_ada_mxf4_bf16_tn has no scratch argument, so no split-K plan competes -- every measured K wins there (4096 x 4096, M = 96 / 128: 6.5 / 6.7 -> 6.0 / 6.3 us;
// x 8192: 10.1 -> 8.9; x 14336: 20.9 / 22.6 -> 14.4 / 15.3; profiles/calib_ada_570_r7.txt)
inline bool os64_plan(int64_t M, int64_t N, int64_t K, bool ada = false) {
  const int64_t cus = chip_cus(), T32 = cdiv(M, 32) * cdiv(N, 32), T64 = cdiv(M, 64) * cdiv(N, 32), KT = cdiv(K, 256);
  if (T32 <= cus || T64 > cus) return false;
  if (ada) return KT <= 64;
  return KT <= 12 || (KT >= 40 && KT <= 64) || (KT >= 32 && KT < 40 && T64 == cus && M <= 64);   // (the last: 8192^2, M = 64; 4096 x 8192, M = 128 -- two tile rows -- loses 3 %)
}
// [r6, third session] The same kernel on MXFP8 operands (EBITS = 8: a stage is 128 elements, so K = 4096 is 32 stages and the one-shot form ends at K = 2048): returns the
// variant (568 = 32x32 tiles, 569 = 32x16, 570 = 64x32) or 0.  Measured against the plans below -- 64x64 ring tiles, split-K with caller scratch -- on M = 1 ... 256 x 21
// (N, K) (tools/calib_os2.py, profiles/calib_os2_fp8_r7.txt): N = K = 4096, M = 1 ... 32 5.9-7.0 -> 4.8-5.4 us, M = 64 8.6 -> 6.4; 8192 x 4096, M <= 32 10.3-10.7 -> 6.0-7.0;
// N = K = 2048, M <= 128 4.8-5.1 -> 3.1-3.5; 4096 x 14336, M <= 32 13.2-15.0 -> 10.9-12.6.
//   * 32-row tiles, one per CU at most (G workgroups; 16 columns per workgroup when that still fits): always up to 40 stages (K <= 5120); up to 64 stages when they fill
//     3/4 of the chip (N = 1024 / 2048 against K = 8192 on 64 / 128 workgroups: 7-20 % behind the split-K plans); up to 128 stages for one m-tile or a weight of N >= 4096
//     (2048 x 14336, M = 64: a tie); up to 256 stages (K = 32768) for one m-tile.
//   * 64-row tiles where the 32-row ones overflow the chip: 17 ... 128 stages (4096 x 8192, M = 96 / 128: 15.6 / 14.8 -> 12.0 / 12.5 us; 8192 x 4096, M = 64: 11.3 -> 9.3;
//     K <= 2048 ties, K = 28672 loses 13 %).
inline int os8_plan(int64_t M, int64_t N, int64_t K) {
  const int64_t cus = chip_cus(), T32 = cdiv(M, 32) * cdiv(N, 32), T16 = cdiv(M, 32) * cdiv(N, 16), T64 = cdiv(M, 64) * cdiv(N, 32), KT = cdiv(K, 128);
  if (M <= 16) {   // the decode form with 32 / 48 columns per workgroup
    if (const int v = os16_wide_plan(8, N, K)) return v;
  }
  if (T32 <= cus) {
    const int v = T16 <= cus ? 569 : 568;
    const int64_t G = T16 <= cus ? T16 : T32;
    // decode form (16x16 tiles on the 16x16x128 MFMA) where those fit one per CU, up to 64 stages (N = K = 4096, M <= 16: 4.85-4.98 -> 3.65-3.82 us; K = 14336: +2 ... 4 %)
    const bool d16 = cdiv(M, 16) * cdiv(N, 16) <= cus;
    if (KT <= 40) return d16 ? 571 : v;
    if (4 * G < 3 * cus) return 0;
    if (KT <= 64) return d16 ? 571 : v;
    if (KT <= 128) return (M <= 32 || N >= 4096) ? v : 0;
    return (KT <= 256 && M <= 32) ? v : 0;
  }
  return (T64 <= cus && KT > 16 && KT <= 128) ? 570 : 0;
}

// [r4] stream-K form of the two persistent kernels (lab variant 89): one workgroup per CU; p.ws / p.ctr / p.tag / p.sk_tiles set by gemm_mx
#if QAMD_BENCH
template <class C>
int launch_gemm_deepp_sk(GemmParams p, hipStream_t s) {
  p.tiles_m = (int)cdiv(p.M, C::BM);
  p.tiles_n = (int)cdiv(p.N, C::BN);
  p.raster_magic = raster_magic(p.tiles_n);
  p.splits = 1;
  const int grid = chip_cus();
  if constexpr (C::EBITS == 4) hipLaunchKernelGGL((labk::gemm_mx_deepp_kernel<C, false, 17, 0, true>), dim3(grid), dim3(C::THREADS), 0, s, p);
  else hipLaunchKernelGGL((labk::gemm_mx_deepp8_kernel<C, 17, false, 0, true>), dim3(grid), dim3(C::THREADS), 0, s, p);
  return check_launch("gemm_mx_deepp_kernel (stream-K)");
}
#endif

template <class C, bool NN = false, int NNABL = 0>
int launch_gemm_deepp8(GemmParams p, hipStream_t s) {   // the fp8 twin (gemm_mx_deepp8), write-through output stores; NN: A is (K, M)
  p.tiles_m = (int)cdiv(p.M, C::BM);
  p.tiles_n = (int)cdiv(p.N, C::BN);
  p.raster_magic = raster_magic(p.tiles_n);
  p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
  const int grid = deepp_grid(p.tiles_m * p.tiles_n);
#if QAMD_BENCH
  if constexpr (NNABL != 0) {   // the NN-operand ablations: the lab copy of the kernel
    hipLaunchKernelGGL((labk::gemm_mx_deepp8_kernel<C, 17, NN, NNABL>), dim3(grid), dim3(C::THREADS), 0, s, p);
    return check_launch("labk::gemm_mx_deepp8_kernel");
  }
#else
  static_assert(NNABL == 0, "ablations: lab build only");
#endif
  if (grid == p.tiles_m * p.tiles_n) hipLaunchKernelGGL((gemm_mx_deepp8_kernel<C, 17, NN, true>), dim3(grid), dim3(C::THREADS), 0, s, p);   // [r6] one tile per workgroup
  else hipLaunchKernelGGL((gemm_mx_deepp8_kernel<C, 17, NN>), dim3(grid), dim3(C::THREADS), 0, s, p);
  return check_launch("gemm_mx_deepp8_kernel");
}

// Heterogeneous launch (gemm_mx_hetero_kernel): the persistent 256x256 kernel over the full rounds of `cus` tiles + the residual
// tiles as 128x128 quarter tiles on extra workgroups of the SAME grid, dispatched CU by CU as the persistent workgroups retire.
template <class CB, class CT, int ST_AUX>
int launch_gemm_hetero(GemmParams p, hipStream_t s) {
  p.tiles_m = (int)cdiv(p.M, CB::BM);
  p.tiles_n = (int)cdiv(p.N, CB::BN);
  p.raster_magic = raster_magic(p.tiles_n);
  p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
  const int T = p.tiles_m * p.tiles_n;
  const int forced = opt_deepp_grid();
  const int g_big = std::min(forced > 0 ? forced : chip_cus(), T);
  const int t_main = T / g_big * g_big;
  const int nsmall = 4 * (T - t_main);
  if (nsmall == 0) {   // a whole number of rounds: nothing residual, the plain persistent kernel
    if constexpr (CB::EBITS == 4) return launch_gemm_deepp<CB, false, ST_AUX>(p, s);
    else return launch_gemm_deepp8<CB>(p, s);
  }
  if constexpr (CB::EBITS == 4) {   // [r6] odd number of K stages: the persistent workgroups run without the empty stage
    if (const int64_t kt = cdiv((int64_t)p.K, 256); (kt & 1) && kt >= 3) {
      hipLaunchKernelGGL((gemm_mx_hetero_kernel<CB, CT, ST_AUX, true>), dim3(g_big + nsmall), dim3(256), 0, s, p, g_big, t_main);
      return check_launch("gemm_mx_hetero_kernel (odd stage count)");
    }
  }
  hipLaunchKernelGGL((gemm_mx_hetero_kernel<CB, CT, ST_AUX>), dim3(g_big + nsmall), dim3(256), 0, s, p, g_big, t_main);
  return check_launch("gemm_mx_hetero_kernel");
}

// Persistent launch or heterogeneous launch for T tiles of 256x256?  Cost in units of one full-chip round of `cus` tiles, fitted to the
// steady-state A/B of 18 shapes, K = 1024 .. 14336 (tests/native/qamd_check heterobench, profiles/native_r3_heterobench.log):
//   balanced rounds   R = ceil(T / cus) rounds at occupancy o = T / (R cus).  The kernels run at the socket power limit, so a round that
//                     leaves CUs idle is cheaper -- but never cheaper than the clock ceiling allows: f(o) = max(0.45 + 0.55 o, 0.78)
//                     (measured: 272 / 320 / 384 / 448 / 576 tiles = 1.56 / 1.62 / 1.77 / 1.90 / 2.62 rounds)
//   heterogeneous     the full rounds + waves of 128x128 quarter tiles: 0.05 once, then 0.25 + 0.13 x fill per wave of `cus` quarter tiles
//                     (a quarter tile walks K at ~0.6 us per stage against 1.79 us for the 256x256 tile, plus its own prologue /
//                     epilogue; measured: 272 / 288 / 320 / 384 / 448 / 576 tiles = 1.30 / 1.32 / 1.43 / 1.87 / 2.25 / 2.43 rounds)
// 4096 x 5120 x 4096: 52.3 us balanced, 49.9 us as two launches (round 2), 46.8 us heterogeneous; 3072 x 6144: 51.5 / 47.4 / 43.3.
inline bool hetero_wins(int64_t T, int cus) {
  const int64_t r = T / cus, rem = T % cus;
  if (r < 1 || rem == 0) return false;
  const int64_t R = r + 1;
  const double a = (double)R * std::max(0.45 + 0.55 * (double)T / (double)(R * cus), 0.78);
  const int64_t nsmall = 4 * rem, waves = nsmall / cus;
  const double last = (double)(nsmall % cus) / (double)cus;
  const double b = (double)r + 0.05 + 0.38 * (double)waves + (last > 0 ? 0.25 + 0.13 * last : 0.0);
  return b < a;
}

// Tile/schedule variants (0 = auto; the lab library can force one through the "gemm_variant" option):
//   PRODUCT (what the auto rules below can pick):
//     90  persistent deep schedule, 256x256 (fp4: gemm_mx_deepp, fp8: gemm_mx_deepp8)      lab: 30 = the per-tile deep schedule of round 1
//   LAB only, [r4]:  89  = 90 as a stream-K walk (tiles of a part-filled round cut along K, fp32 parts parked in caller scratch).  Correct and
//     deterministic (tests/test_gpu_round4.py) and NOT adopted: a cut tile moves 256 KiB of fp32 sums through memory twice, and a stream over G + T % G
//     tiles cuts ~G - gcd tiles -- 128 ... 255 parked tiles = 64 ... 128 MB per launch at ~3.7 TB/s: +18 us at 384 tiles, +34 us at 320 (MXFP4,
//     K = 4096: 74.8 / 86.1 us against 57.0 balanced / 48.2 heterogeneous); break-even only beyond K ~ 14336, where the heterogeneous launch still
//     matches or beats it (MXFP8 3072 x 8192 x 28672: 612 us stream-K, 601 heterogeneous, 666 balanced).  profiles/ab_mxsk_r4c.txt
//     98  heterogeneous launch: 90 over the full rounds + the residual tiles as 128x128 tiles in the same grid     lab: 99 = 3-deep ring for those
//     24 / 25 / 27 / 28 / 29  pipelined schedule on a 2-deep ring: 128x128, 256x128 (8 waves), 128x64, 64x128, 64x64      58  256x128 on four waves, 3-deep ring
//     70..73  ring schedule 64x64, 128x64, 64x128, 128x128 (+ split-K)          60  skinny split-K kernel (fp4, M <= 32)
//   LAB only:
//   1  256x256 ping-pong     5  256x256 lockstep     6..9  queue schedule (256x256, 128x128, 256x128, 128x256)
//   20 / 24 / 25 / 26  simple schedule (256x256, 128x128, 256x128, 128x256)
//   60  skinny split-K kernel (fp4, M <= 32 per tile, no LDS staging): auto for M <= 32
//   30  deep schedule (fp4, 256x256, 4 waves of 128x128, LDS-DMA)      40  regstage (same tiling, copy through registers)
//   31..36, 41..43, 50..56  ablations / traces / clock probes of those (bench only)
//   2  128x128 lockstep (small M or N)                     3  256x128 lockstep    4  128x256 lockstep
//   100+b / 200+b  ablations of variants 1 / 5 (bench only), b = OR of ABL_* bits
// bench-only ablations of the 8-wave 256x256 MXFP4 schedules: 100 + b ping-pong, 200 + b lockstep, 300 + b queue, b = OR
// of ABL_* bits.  Returns -1 for any other variant.
#if QAMD_BENCH
#if QAMD_DEF(6)
int dispatch_ablation_mx4(int v, const GemmParams& p, hipStream_t s) {
  switch (v) {
#define QAMD_ABL(b) \
    case 100 + b: return launch_gemm<GemmCfg<256, 256, 2, 4, 4, false, b>, 1>(p, s); \
    case 200 + b: return launch_gemm<GemmCfg<256, 256, 2, 4, 4, false, b>, 0>(p, s); \
    case 300 + b: return launch_gemm<GemmCfg<256, 256, 2, 4, 4, false, b>, 2>(p, s);
    QAMD_ABL(1) QAMD_ABL(2) QAMD_ABL(3) QAMD_ABL(4) QAMD_ABL(8) QAMD_ABL(9) QAMD_ABL(10) QAMD_ABL(11) QAMD_ABL(16) QAMD_ABL(17) QAMD_ABL(18) QAMD_ABL(41) QAMD_ABL(40) QAMD_ABL(42)
#undef QAMD_ABL
  }
  return -1;
}
#else
int dispatch_ablation_mx4(int v, const GemmParams& p, hipStream_t s);
#endif
#endif   // QAMD_BENCH

template <int EBITS, bool SPLIT>
int dispatch_variant(int v, const GemmParams& p, hipStream_t s, const char* name) {
#if QAMD_BENCH
  if (p.pp_flags & 4096) {   // lab: "pp_flags" bit 12 = the round-1 ring / simple schedules wherever their pipelined successors are picked
    if (v >= 70 && v <= 73) v += 100;
    else if (v == 24 || v == 25 || (v >= 27 && v <= 29)) v += 200;
  }
#endif
  if (dry_record(v, p.N, ((v >= 70 && v <= 78) || (v >= 170 && v <= 173)) ? p.splits : 1)) return 0;
  switch (v) {
    // pipelined schedule on a 2-deep LDS ring (gemm_mx_ringp with NSTAGE = 2: the LDS footprint of the round-1 "simple" schedule,
    // two workgroups per CU, but the next stage's fragments are read during this stage's MFMAs): -3 .. -13 % against the
    // simple schedule on every mid-size shape measured (profiles/native_r2_midtile2.log)
    case 24: return launch_gemm<GemmCfg<128, 128, 2, 2, EBITS, SPLIT, 0, 2>, 9>(p, s);
    case 25: return launch_gemm<GemmCfg<256, 128, 4, 2, EBITS, SPLIT, 0, 2>, 9>(p, s);
    case 27: return launch_gemm<GemmCfg<128, 64, 2, 2, EBITS, SPLIT, 0, 2>, 9>(p, s);    // mid-size problems: more, smaller tiles
    // [r3] 256x128 on FOUR waves of 128x64, 3-deep ring, one workgroup per CU: 0.75 KiB of LDS fragment reads per MFMA against 1 KiB for the 64x64 wave
    // tiles of 24 / 25 -- half-chip outputs with a long K (2048 x 4096 x 8192: 36.9 -> 35.4 us, x 14336: 60.6 -> 57.1; profiles/native_r3_halftile.log)
    case 58: return launch_gemm<GemmCfg<256, 128, 2, 2, EBITS, SPLIT, 0, 3>, 9>(p, s);
    case 28: return launch_gemm<GemmCfg<64, 128, 2, 2, EBITS, SPLIT, 0, 2>, 9>(p, s);
    case 29: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 0, 2>, 9>(p, s);
    case 70: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 0, 3>, 9>(p, s);     // pipelined ring schedule: NSTAGE-deep LDS ring, NSTAGE-1 stages in flight,
    case 71: return launch_gemm<GemmCfg<128, 64, 2, 2, EBITS, SPLIT, 0, 3>, 9>(p, s);    //   fragments of the next stage read during this stage's MFMAs
    case 72: return launch_gemm<GemmCfg<64, 128, 2, 2, EBITS, SPLIT, 0, 3>, 9>(p, s);
    case 73: return launch_gemm<GemmCfg<128, 128, 2, 2, EBITS, SPLIT, 0, 3>, 9>(p, s);
#if QAMD_BENCH
    case 273: return launch_gemm<GemmCfg<128, 128, 2, 2, EBITS, SPLIT, 128, 3>, 9>(p, s);   // lab: 73 with the next stage's reads packed into the first MFMAs (ABL_READS_FIRST)
    case 272: return launch_gemm<GemmCfg<64, 128, 2, 2, EBITS, SPLIT, 128, 3>, 9>(p, s);
    case 270: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 128, 3>, 9>(p, s);
    case 224 + 1000: return launch_gemm<GemmCfg<128, 128, 2, 2, EBITS, SPLIT, 128, 2>, 9>(p, s);   //      24 (2-deep, two workgroups per CU) likewise
    case 373: return launch_gemm<GemmCfg<128, 128, 2, 4, EBITS, SPLIT, 0, 3>, 9>(p, s);    //      128x128 on EIGHT waves of 64x32 (two per SIMD cover each other's stage-top bubble)
    case 374: return launch_gemm<GemmCfg<128, 128, 4, 2, EBITS, SPLIT, 0, 3>, 9>(p, s);    //      ... of 32x64
    case 325: return launch_gemm<GemmCfg<256, 128, 2, 2, EBITS, SPLIT, 0, 3>, 9>(p, s);   //      = product variant 58
    case 326: return launch_gemm<GemmCfg<128, 256, 2, 2, EBITS, SPLIT, 0, 3>, 9>(p, s);   //      128x256 on four waves of 64x128
    case 327: return launch_gemm<GemmCfg<256, 128, 2, 2, EBITS, SPLIT, 0, 2>, 9>(p, s);   //      ... 2-deep
    case 76: return launch_gemm<GemmCfg<128, 128, 2, 2, EBITS, SPLIT, 0, 4>, 9>(p, s);   // lab: 73 with a 4-deep ring (144 KiB of LDS: one workgroup per CU, which 73's regime is anyway)
    case 79: return launch_gemm<GemmCfg<64, 128, 2, 2, EBITS, SPLIT, 0, 4>, 9>(p, s);    //      72 with a 4-deep ring
    case 224: return launch_gemm<GemmCfg<128, 128, 2, 2, EBITS, SPLIT>, 3>(p, s);   // the round-1 simple schedule (2 stages, reads after the barrier, then MFMAs)
    case 225: return launch_gemm<GemmCfg<256, 128, 4, 2, EBITS, SPLIT>, 3>(p, s);
    case 227: return launch_gemm<GemmCfg<128, 64, 2, 2, EBITS, SPLIT>, 3>(p, s);
    case 228: return launch_gemm<GemmCfg<64, 128, 2, 2, EBITS, SPLIT>, 3>(p, s);
    case 229: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT>, 3>(p, s);
    case 170: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 0, 3>, 7>(p, s);    // the round-1 ring schedule (whole-stage reads after the barrier, then the MFMAs)
    case 171: return launch_gemm<GemmCfg<128, 64, 2, 2, EBITS, SPLIT, 0, 3>, 7>(p, s);
    case 172: return launch_gemm<GemmCfg<64, 128, 2, 2, EBITS, SPLIT, 0, 3>, 7>(p, s);
    case 173: return launch_gemm<GemmCfg<128, 128, 2, 2, EBITS, SPLIT, 0, 3>, 7>(p, s);
    case 1: return launch_gemm<GemmCfg<256, 256, 2, 4, EBITS, SPLIT>, 1>(p, s);
    case 2: return launch_gemm<GemmCfg<128, 128, 2, 2, EBITS, SPLIT>, 0>(p, s);
    case 3: return launch_gemm<GemmCfg<256, 128, 4, 2, EBITS, SPLIT>, 0>(p, s);
    case 4: return launch_gemm<GemmCfg<128, 256, 2, 4, EBITS, SPLIT>, 0>(p, s);
    case 5: return launch_gemm<GemmCfg<256, 256, 2, 4, EBITS, SPLIT>, 0>(p, s);
    case 20: return launch_gemm<GemmCfg<256, 256, 2, 4, EBITS, SPLIT>, 3>(p, s);   // simple schedule, 256x256
    case 26: return launch_gemm<GemmCfg<128, 256, 2, 4, EBITS, SPLIT>, 3>(p, s);
    case 74: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 0, 4>, 7>(p, s);
    case 75: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 0, 6>, 7>(p, s);
    case 80: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 1, 3>, 7>(p, s);     //   ring ablations: no DMA
    case 81: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 2, 3>, 7>(p, s);     //   no MFMA
    case 82: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 32, 3>, 7>(p, s);    //   no fragment reads
    case 83: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 34, 3>, 7>(p, s);    //   DMA + barriers only
    case 84: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 35, 3>, 7>(p, s);    //   barriers only
    case 78: return launch_gemm<GemmCfg<64, 64, 2, 2, EBITS, SPLIT, 16, 3>, 7>(p, s);    //   per-wave timeline of workgroup 0
#endif
  }
  if constexpr (EBITS == 8) {
#if QAMD_BENCH
    if (v == 89) return launch_gemm_deepp_sk<GemmCfg<256, 256, 2, 2, 8, true>>(p, s);
#endif
    if (v == 90) return launch_gemm_deepp8<GemmCfg<256, 256, 2, 2, 8, true>>(p, s);   // persistent deep schedule, fp8
    // [r6] small batches on wave-owned K stages (gemm_mx_os.hip.h, EBITS = 8): 32x32 / 32x16 / 64x32 tiles
    if (v == 568) return launch_gemm_os<false, 32, 8>(p, s);
    if (v == 569) return launch_gemm_os<false, 16, 8>(p, s);
    if (v == 570) return launch_gemm_os<false, 32, 8, 64>(p, s);
    if (v >= 571 && v <= 575) return launch_gemm_os16_tn<8, 0, false>(v == 571 ? 16 : v == 572 ? 32 : v == 573 ? 48 : v == 574 ? 56 : 64, p, s);
    if (v == 98) return launch_gemm_hetero<GemmCfg<256, 256, 2, 2, 8, true>, GemmCfg<128, 128, 2, 2, 8, true, 0, 4>, 17>(p, s);
#if QAMD_BENCH
    if (v == 30) return launch_gemm<GemmCfg<256, 256, 2, 2, 8, true>, 4>(p, s);   // per-tile deep schedule (round 1)
#endif
  }
  if constexpr (EBITS == 4) {
    // persistent deep schedule; output stores write through (sc0 sc1): nothing dirty is left for the end-of-kernel L2
    // write-back (4096^3: 34.6 -> 33.4 .. 34.4 us, never slower; profiles/native_r2_store_policy.log)
#if QAMD_BENCH
    if (v == 89) return launch_gemm_deepp_sk<GemmCfg<256, 256, 2, 2, 4, false>>(p, s);
#endif
    if (v == 90) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 17>(p, s);
    if (v == 98) return launch_gemm_hetero<GemmCfg<256, 256, 2, 2, 4, false>, GemmCfg<128, 128, 2, 2, 4, false, 0, 4>, 17>(p, s);
    // [r6] K split inside the workgroup (gemm_mx_ks.hip.h): 561 = 32x32 tiles, 562 = 32x64.  Ring depth by the K stage count: a short K pays for a deep ring's prologue
    // (4096^2 weights, M <= 64: 4.16 us 4-deep, 4.47 us 8-deep), a long one needs the stages in flight (8192^2: 9.4 us 4-deep, 7.4 us 6-deep; profiles/calib_ks_r6*.txt)
    if (v == 561 || v == 562) {
      const bool deep = cdiv(p.K, 256) > 24;
      if (v == 561) return deep ? launch_gemm_ks<32, 32, 6>(p, s) : launch_gemm_ks<32, 32, 4>(p, s);
      return deep ? launch_gemm_ks<32, 64, 6>(p, s) : launch_gemm_ks<32, 64, 4>(p, s);
    }
    if (v == 568) return launch_gemm_os<false>(p, s);       // [r6] 32x32 tiles on wave-owned K stages (gemm_mx_os.hip.h): one shot up to 16 stages, wave-owned rings beyond
    if (v == 569) return launch_gemm_os<false, 16>(p, s);   // [r6] the same with 16 columns per workgroup
    if (v == 570) return launch_gemm_os<false, 32, 4, 64>(p, s);   // [r6] 64x32 tiles (two m-tiles per stage owner)
    if (v >= 571 && v <= 575) return launch_gemm_os16_tn<4, 0, false>(v == 571 ? 16 : v == 572 ? 32 : v == 573 ? 48 : v == 574 ? 56 : 64, p, s);   // [r6] decode form: 16 x (16 / 32 / 48 / 56 / 64) tiles on the 16x16x128 MFMA
#if QAMD_BENCH
    if (v == 99) return launch_gemm_hetero<GemmCfg<256, 256, 2, 2, 4, false>, GemmCfg<128, 128, 2, 2, 4, false, 0, 3>, 17>(p, s);
    // [r6] lab: the other tiles / ring depths of the in-workgroup K-split kernel (563 = 64x32, 564 = 64x64; 565 - 567 = 32x32 with a 4 / 8 / 6-deep ring)
    if (v == 563 || v == 564) {
      const bool deep = cdiv(p.K, 256) > 24;
      if (v == 563) return deep ? launch_gemm_ks<64, 32, 6>(p, s) : launch_gemm_ks<64, 32, 4>(p, s);
      return deep ? launch_gemm_ks<64, 64, 6>(p, s) : launch_gemm_ks<64, 64, 4>(p, s);
    }
    if (v == 565) return launch_gemm_ks<32, 32, 4>(p, s);
    if (v == 566) return launch_gemm_ks<32, 32, 8>(p, s);
    if (v == 567) return launch_gemm_ks<32, 32, 6>(p, s);
    // [r6] 8-wave persistent schedule (gemm_mx_duo.hip.h): 88 burst retirement, 87 retirement behind the last k-slice, 86 = 88 with stage stamps
    if (v == 88) return launch_gemm_duo<GemmCfg<256, 256, 2, 4, 4, false>, 17, 0>(p, s);
    if (v == 87) return launch_gemm_duo<GemmCfg<256, 256, 2, 4, 4, false>, 17, 1>(p, s);
    if (v == 86) return launch_gemm_duo<GemmCfg<256, 256, 2, 4, 4, false>, 17, 0, true>(p, s);
    if (v == 298) return launch_gemm_hetero<GemmCfg<256, 256, 2, 2, 4, false>, GemmCfg<128, 128, 2, 2, 4, false, 128, 4>, 17>(p, s);   // residual tiles with ABL_READS_FIRST
    if (v == 91) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, true, 17>(p, s);   //   + phase timestamps of workgroup 0 (qutlass_amd_debug_set_trace_buffer)
    if (v == 92) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 2>(p, s);       //   output stores nt
    if (v == 93) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 16>(p, s);      //   sc1
    if (v == 94) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 0>(p, s);       //   default (write-back) policy
    if (v == 95) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 19>(p, s);      //   sc0 sc1 nt
    if (v == 96) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 1>(p, s);       //   sc0
    // timing experiments that change the result (gemm_mx_deepp.hip.h LAB): 97 = no alpha multiply, 197 = half of the stores one stage early, 297 = both
    if (v == 97) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 17, 1>(p, s);
    if (v == 197) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 17, 2>(p, s);
    if (v == 297) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 17, 3>(p, s);
    if (v == 397) return launch_gemm_deepp<GemmCfg<256, 256, 2, 2, 4, false>, false, 17, 4>(p, s);   //   half of the tile is never stored (accumulators kept alive)
    switch (v) {
      case 6: return launch_gemm<GemmCfg<256, 256, 2, 4, 4, false>, 2>(p, s);
      case 7: return launch_gemm<GemmCfg<128, 128, 2, 2, 4, false>, 2>(p, s);
      case 8: return launch_gemm<GemmCfg<256, 128, 4, 2, 4, false>, 2>(p, s);
      case 9: return launch_gemm<GemmCfg<128, 256, 2, 4, 4, false>, 2>(p, s);
      case 30: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false>, 4>(p, s);      // deep schedule: 4 waves x 128x128, one tile per workgroup
      case 31: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 8>, 4>(p, s);   //   no epilogue
      case 32: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 9>, 4>(p, s);   //   no DMA, no epilogue
      case 33: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 10>, 4>(p, s);  //   no MFMA, no epilogue
      case 34: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 40>, 4>(p, s);  //   no reads, no epilogue
      case 40: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false>, 5>(p, s);      // regstage schedule: 4 waves x 128x128, copy through registers
      case 41: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 8>, 5>(p, s);   //   no epilogue
      case 42: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 9>, 5>(p, s);   //   no copy, no epilogue
      case 43: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 10>, 5>(p, s);  //   no MFMA, no epilogue
      case 50: return launch_gemm<GemmCfg<256, 256, 2, 4, 4, false, 64>, 3>(p, s);      // clock probes of 20 / 30 / 40 and their no-epilogue / no-copy ablations
      case 51: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 64>, 4>(p, s);
      case 52: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 64>, 5>(p, s);
      case 53: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 72>, 5>(p, s);
      case 54: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 73>, 5>(p, s);
      case 55: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 74>, 5>(p, s);
      case 56: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 104>, 5>(p, s);
      case 35: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 16>, 4>(p, s);  //   trace
      case 36: return launch_gemm<GemmCfg<256, 256, 2, 2, 4, false, 17>, 4>(p, s);  //   trace, no DMA
      case 21: return launch_gemm<GemmCfg<256, 256, 2, 4, 4, false, 8>, 3>(p, s);   //   no epilogue
      case 22: return launch_gemm<GemmCfg<256, 256, 2, 4, 4, false, 9>, 3>(p, s);   //   no DMA, no epilogue
      case 23: return launch_gemm<GemmCfg<256, 256, 2, 4, 4, false, 10>, 3>(p, s);  //   no MFMA, no epilogue
    }
#endif
  }
#if QAMD_BENCH
  if constexpr (EBITS == 4 && !SPLIT) {
    const int rc = dispatch_ablation_mx4(v, p, s);
    if (rc >= 0) return rc;
  }
#endif
  return fail(QAMD_ERR_INVALID, "%s: unknown gemm_variant %d", name, v);
}

// MXFP8 with an e5m2 A operand (extension, include/qutlass_amd.h qutlass_amd_matmul_mxf8_bf16_{tn,nn}_fmt): the variants the
// auto rules can pick, nothing else.
int dispatch_variant_a5(int v, const GemmParams& p, hipStream_t s, const char* name)
#if QAMD_DEF(4)   // (the e5m2-operand MXFP8 kernels ride with the NVFP4 unit: balances the parallel build)
{
  if (dry_record(v, p.N, (v >= 70 && v <= 78) ? p.splits : 1)) return 0;
  switch (v) {
    case 24: return launch_gemm<GemmCfg<128, 128, 2, 2, 8, true, 0, 2, 1>, 9>(p, s);
    case 25: return launch_gemm<GemmCfg<256, 128, 4, 2, 8, true, 0, 2, 1>, 9>(p, s);
    case 27: return launch_gemm<GemmCfg<128, 64, 2, 2, 8, true, 0, 2, 1>, 9>(p, s);
    case 58: return launch_gemm<GemmCfg<256, 128, 2, 2, 8, true, 0, 3, 1>, 9>(p, s);
    case 28: return launch_gemm<GemmCfg<64, 128, 2, 2, 8, true, 0, 2, 1>, 9>(p, s);
    case 29: return launch_gemm<GemmCfg<64, 64, 2, 2, 8, true, 0, 2, 1>, 9>(p, s);
    case 90: return launch_gemm_deepp8<GemmCfg<256, 256, 2, 2, 8, true, 0, 2, 1>>(p, s);
    case 98: return launch_gemm_hetero<GemmCfg<256, 256, 2, 2, 8, true, 0, 2, 1>, GemmCfg<128, 128, 2, 2, 8, true, 0, 4, 1>, 17>(p, s);
    case 70: return launch_gemm<GemmCfg<64, 64, 2, 2, 8, true, 0, 3, 1>, 9>(p, s);
    case 71: return launch_gemm<GemmCfg<128, 64, 2, 2, 8, true, 0, 3, 1>, 9>(p, s);
    case 72: return launch_gemm<GemmCfg<64, 128, 2, 2, 8, true, 0, 3, 1>, 9>(p, s);
    case 73: return launch_gemm<GemmCfg<128, 128, 2, 2, 8, true, 0, 3, 1>, 9>(p, s);
    case 568: return launch_gemm_os<false, 32, 8, 32, 1>(p, s);   // [r6] small batches on wave-owned K stages, e5m2 A
    case 569: return launch_gemm_os<false, 16, 8, 32, 1>(p, s);
    case 570: return launch_gemm_os<false, 32, 8, 64, 1>(p, s);
    case 571: case 572: case 573: case 574: case 575: return launch_gemm_os16_tn<8, 1, false>(v == 571 ? 16 : v == 572 ? 32 : v == 573 ? 48 : v == 574 ? 56 : 64, p, s);
  }
  return fail(QAMD_ERR_INVALID, "%s: gemm_variant %d has no e5m2-operand instantiation", name, v);
}
int launch_nn_fused_a5(const GemmParams& p, hipStream_t s, bool per_tile) {
#if QAMD_BENCH
  if (per_tile) return launch_gemm<GemmCfg<256, 256, 2, 2, 8, true, 0, 2, 1>, 6>(p, s);
#endif
  (void)per_tile;
  return launch_gemm_deepp8<GemmCfg<256, 256, 2, 2, 8, true, 0, 2, 1>, true>(p, s);
}
#else
;
int launch_nn_fused_a5(const GemmParams& p, hipStream_t s, bool per_tile);
#endif

#if QAMD_TU != 0
#if QAMD_TU == 2
template int dispatch_variant<4, false>(int, const GemmParams&, hipStream_t, const char*);
#else
extern template int dispatch_variant<4, false>(int, const GemmParams&, hipStream_t, const char*);
#endif
#if QAMD_TU == 3
template int dispatch_variant<8, true>(int, const GemmParams&, hipStream_t, const char*);
#if QAMD_BENCH
template int launch_gemm<GemmCfg<256, 256, 2, 2, 8, true>, 6>(GemmParams, hipStream_t);   // lab: per-tile kernel on the (K, M) operand (round 1)
#endif
#else
extern template int dispatch_variant<8, true>(int, const GemmParams&, hipStream_t, const char*);
#if QAMD_BENCH
extern template int launch_gemm<GemmCfg<256, 256, 2, 2, 8, true>, 6>(GemmParams, hipStream_t);
#endif
#endif
#endif

// NVFP4 launches live in their own unit
#if QAMD_DEF(4)
int launch_nvf4_host(const NvGemmParams& p, hipStream_t s, int variant, int* splits_out) {
  return launch_nvf4_gemm(p, s, variant, chip_cus(), splits_out) == hipSuccess ? 0 : 1;
}
#else
int launch_nvf4_host(const NvGemmParams& p, hipStream_t s, int variant, int* splits_out);
#endif

#if QAMD_DEF(1)
// EBITS: 4 = MXFP4, 8 = MXFP8 (TN)
// Small-output regime: ring schedule (one workgroup per CU with several stages in flight beats the 2-stage simple schedule
// whenever the tiles do not fill the chip twice), plus split-K over grid.y when 64x64 tiles leave CUs idle and K is long
// enough to pay for the second pass (>= 32 stages: fp4 K >= 8192).  One function so that the launcher and qutlass_amd_gemm_splitk_workspace_bytes
// agree.  Measured: profiles/native_r1_ring.log.
struct SmallPlan { int variant; int splits; };   // variant 0: not this regime
// tile of a ring variant (70: 64x64, 71: 128x64, 72: 64x128, 73: 128x128)
inline void ring_tile(int variant, int& bm, int& bn) {
  bm = (variant == 71 || variant == 73) ? 128 : 64;
  bn = (variant == 72 || variant == 73) ? 128 : 64;
}
// split-K scratch: [splits][M][N] fp32 partials (lab library: then, 256-byte aligned, one 64-bit arrival slot per output tile
// for the fused-reduction experiment)
inline int64_t splitk_partial_bytes(int64_t M, int64_t N, int splits) { return (int64_t)splits * M * N * 4; }
inline int64_t splitk_ctr_offset(int64_t M, int64_t N, int splits) { return (splitk_partial_bytes(M, N, splits) + 255) / 256 * 256; }
inline int64_t splitk_ws_bytes(int variant, int64_t M, int64_t N, int splits) {
#if QAMD_BENCH
  int bm, bn;
  ring_tile(variant, bm, bn);
  return splitk_ctr_offset(M, N, splits) + cdiv(M, bm) * cdiv(N, bn) * 8;
#else
  (void)variant;
  return splitk_partial_bytes(M, N, splits);
#endif
}
// the round-1/2 rule: thresholds on the tile counts (fitted on sweeps of the in-stream C++ harness, profiles/native_r1_ring.log, native_r2_tilesplit.log)
template <int EBITS>
SmallPlan plan_small_rule(int64_t M, int64_t N, int64_t K, bool may_split) {
  const int64_t cus = chip_cus();
  const int64_t T64 = cdiv(M, 64) * cdiv(N, 64), T128 = cdiv(M, 128) * cdiv(N, 128);
  if (T64 <= cus) {
    const int64_t KT = cdiv(K * EBITS / 8, 128);
    int64_t S = 1;
    if (may_split && T64 < cus && KT >= opt_splitk_min_kt()) {                          // shorter K: the reduce pass costs more than it saves
      const int64_t wg = opt_splitk_wg() > 0 ? opt_splitk_wg() : cus;
      S = std::min<int64_t>(std::min<int64_t>(8, wg / T64), KT / 8);   // up to one workgroup per CU, >= 8 stages per split
      if (S < 1) S = 1;
      const int64_t per = cdiv(KT, S);
      S = cdiv(KT, per);                                                // every split non-empty
      // a two-way split (more than 64 tiles) halves the K loop at best and pays a second launch + the reduce pass (~3.5 us): it
      // needs ~48 stages to win (M = 128, N = 4096: K = 8192 7.6 us unsplit vs 9.8 us split; K = 14336 17.3 vs 14.0 us;
      // profiles/native_r2_tilesplit.log)
      if (S == 2 && KT < 48) S = 1;
    }
    return {70, (int)S};
  }
  if (T64 <= 2 * cus) return {N >= M ? 72 : 71, 1};
  if (M > 64 && N > 64 && T128 <= cus) return {73, 1};
  return {0, 1};
}

// [r3] ... corrected by a fitted cost model where that rule leaves a long K to too few, too small tiles: it only ever splits 64x64 tiles, so 96 x 5120 x 25600
// ran 160 tiles of 64x64 unsplit (27.5 us; MXFP8 50.3) where 40 tiles of 128x128 in 4 K ranges take 19.5 (30.4).  tools/calib_mx_small.py timed the ring schedule
// on 64x64 / 64x128 / 128x128 tiles with 1 / 2 / 4 / 8 K ranges on 132 shapes per format (profiles/calib_mx_small_r3.txt); least squares on log time (rms 14 % /
// 18 %, measurements >= 13 us only -- below that the Python caller of the calibration is the bound) give, per tile kernel,
//     a + g b kt / 16 [+ r0 + (S + 1) M N 4 bytes / bw for the reduce pass]      kt = 128-byte K stages per workgroup, g = 1 while workgroups <= CUs, else
//                                                                                ceil(workgroups / CUs) e
// The model is a CORRECTION, not the rule: it replaces the rule's choice only where its own prediction for that choice is more than 8 % above its best candidate
// and the best candidate takes >= 12 us (K ranges of >= 4 stages).  Against the calibration: 23 of 264 shapes change, MXFP4 96 / 128 x 8192 x 28672 32.4 / 34.1 ->
// 28.2 / 30.5 us, x 5120 x 25600 27.5 / 28.0 -> 19.5 / 21.1; MXFP8 192 / 256 x 4096 x 14336 30.9 / 31.5 -> 23.0 / 25.3, 96 ... 384 x 5120 x 25600 50 ... 68 -> 30 ... 54,
// 96 ... 256 x 8192 x 28672 59 ... 84 -> 44 ... 70; three shapes lose 2 - 4 % (MXFP8 16 / 32 x 8192 x 8192, MXFP4 64 x 8192 x 28672).
template <int EBITS>
SmallPlan plan_small(int64_t M, int64_t N, int64_t K, bool may_split = true) {
  const SmallPlan cur = plan_small_rule<EBITS>(M, N, K, may_split);
  if (cur.variant != 70 && cur.variant != 72 && cur.variant != 73) return cur;      // (71: M > N, not calibrated)
  const int64_t cus = chip_cus();
  //                                   64x64  64x128 128x128
  static constexpr double A4[3] = {6.07, 6.04, 6.30}, B4[3] = {3.41, 4.80, 6.58}, E4[3] = {0.99, 1.2, 1.2};
  static constexpr double A8[3] = {3.31, 3.79, 2.91}, B8[3] = {3.87, 4.83, 6.41}, E8[3] = {0.90, 1.2, 1.2};
  const double* A = EBITS == 4 ? A4 : A8;
  const double* B = EBITS == 4 ? B4 : B8;
  const double* E = EBITS == 4 ? E4 : E8;
  const double r0 = EBITS == 4 ? 1.39 : 3.86, bw = EBITS == 4 ? 2.63e6 : 2.79e6;
  static constexpr int VAR[3] = {70, 72, 73}, BM[3] = {64, 64, 128}, BN[3] = {64, 128, 128};
  const int64_t KT = cdiv(K * EBITS / 8, 128);
  auto price = [&](int c, int S, int* s2) {
    const int64_t per = cdiv(KT, S), S2 = cdiv(KT, per);
    const double n = (double)(cdiv(M, BM[c]) * cdiv(N, BN[c]) * S2) / (double)cus;
    double t = A[c] + (n <= 1.0 ? 1.0 : std::ceil(n) * E[c]) * B[c] * (double)per / 16.0;
    if (S2 > 1) t += r0 + (double)(S2 + 1) * (double)M * (double)N * 4.0 / bw;
    *s2 = (int)S2;
    return t;
  };
  int s2;
  const int ccur = cur.variant == 70 ? 0 : cur.variant == 72 ? 1 : 2;
  const double t_cur = price(ccur, cur.splits, &s2);
  SmallPlan best = cur;
  double t_best = t_cur;
  for (int c = 0; c < 3; ++c)
    for (int S = 1; S <= (may_split && N % 4 == 0 ? 8 : 1); S *= 2) {
      if (S > 1 && cdiv(KT, S) < 4) continue;
      const double t = price(c, S, &s2);
      if (S > 1 && s2 < 2) continue;
      if (t < t_best) { t_best = t; best = {VAR[c], s2}; }
    }
  SmallPlan res = (t_best >= 12.0 && t_best < 0.92 * t_cur) ? best : cur;
  // A long K (>= 96 stages) on 64x128 tiles, single pass: 128x128 tiles in two K ranges while those still fit one per CU -- the guard above leaves these alone (the fit
  // is coarse there), the GPU-only measurement does not: all ten such shapes of the calibration grid gain 8 ... 25 % (MXFP8 384 / 512 x 4096 x 14336 38.6 / 43.0 -> 32.1 /
  // 36.1 us, MXFP4 256 x 8192 x 28672 46.1 -> 42.2, 384 x 5120 x 25600 41.0 -> 36.2), every shape with fewer stages ties or loses
  // (profiles/calib_mx_small_r3_graph_m192_1024.txt, profiles/instream_mx_plan_check_r3.txt)
  if (may_split && N % 4 == 0 && res.variant == 72 && res.splits == 1 && KT >= 96 && 2 * cdiv(M, 128) * cdiv(N, 128) <= cus) res = {73, 2};
  return res;
}

// [r6] When does the in-workgroup K-split kernel (gemm_mx_ks.hip.h, one 32x32 tile per workgroup of four waves) take an MXFP4 shape?  Measured against the plans below on
// M = 1 ... 256 x 14 (N, K) (tools/calib_ks.py, profiles/calib_ks_r6g.txt; GPU-only timing, both sides with caller scratch):
//   * the tiles must fit one per CU (two workgroups on a CU share its LDS-DMA path: +40 ... +100 %), and
//   * with a long K (> 24 stages of 256) they must also fill more than half the chip: below that the split-K plans, which spread K over more CUs, are ahead
//     (N = 4096, K = 14336: M <= 16 8.1-9.1 us split against 9.2-9.4; M = 64 11.4 against 9.8).
// Where it applies it is 11 ... 30 % faster (N = K = 4096: M <= 64 4.9-5.3 -> 4.1-4.5 us; 8192^2: M <= 32 8.8-11.1 -> 7.4-7.8 us), M = 1 ... 8 included (the LDS-free
// split-K kernel: 4.55-4.92 us at N = K = 4096).  32x64 tiles: only where 32x32 tiles just overflow the chip and 32x64 nearly fill it (N = 14336: -6 %).
// Returns the variant (568 / 569 / 570 / 571 / 561 / 562) or 0.
inline int ks_plan(int64_t M, int64_t N, int64_t K) {
  const int64_t cus = chip_cus(), KT = cdiv(K, 256);
  const int64_t T32 = cdiv(M, 32) * cdiv(N, 32);
  // [r6] K <= 4096: the tile's whole K extent fits the LDS -- the one-shot kernel (gemm_mx_os.hip.h), no ring and no barrier in the K walk: N = K = 4096, M = 1 ... 64
  // 4.05-4.39 -> 3.34-3.67 us, N = K = 2048 3.15-3.26 -> 2.62-2.77 (profiles/calib_os_r6q.txt); its wave-owned-ring form for longer K and 16 columns per workgroup where os_plan says so
  // (4096 x 8192, M <= 32: 5.8-7.1 -> 5.5-5.7 us); past one tile per CU the ring plans below keep the shape
  if (M <= 16) {   // [r6] the decode form with wider column tiles (os16_wide_plan)
    if (const int v = os16_wide_plan(4, N, K)) return v;
  }
  if (const int tn = os_plan(M, N, K)) {
    // [r6] decode form (gemm_mx_os16_kernel, 16x16 tiles on the 16x16x128 MFMA) wherever those fit one per CU: a third fewer bytes through each CU's LDS-DMA path
    // (N = K = 4096, M <= 16: 3.25-3.31 -> 2.86-2.92 us; K = 8192 5.0-5.2 -> 3.7-4.1; K = 14336 6.8-7.0 -> 6.1-6.5; two per CU (N = 8192) lose 9 %; profiles/calib_os16_r7.txt)
    if (cdiv(M, 16) * cdiv(N, 16) <= cus) return 571;
    return tn == 16 ? 569 : 568;
  }
  if (os64_plan(M, N, K)) return 570;   // [r6] 64x32 tiles on wave-owned K stages where 32x32 tiles overflow the chip
  if (T32 <= cus && (KT <= 24 || (2 * T32 > cus && KT <= 64))) return 561;   // (K > 16384 was not calibrated, and a split-K plan on larger tiles moves fewer bytes per CU there)
  const int64_t T64 = cdiv(M, 32) * cdiv(N, 64);
  if (M <= 32 && T32 > cus && T64 <= cus && 8 * T64 >= 7 * cus && KT <= 24) return 562;
  return 0;
}

// a_fmt (MXFP8 only): QAMD_FP8_E4M3 / QAMD_FP8_E5M2 element format of A
template <int EBITS>
int gemm_mx(const char* name, const void* A, const void* B, const void* A_sf, const void* B_sf,
            const float* alpha, void* D, int64_t M, int64_t N, int64_t K, void* stream, void* ws = nullptr, int64_t ws_bytes = 0, int a_fmt = 0,
            int64_t ldd = 0) {   // ldd: row stride of D in elements when this call covers a column range of a wider output (0 = N)
  // one place decides which instantiation family a variant number is looked up in
  auto dispatch = [&](int v, const GemmParams& q, hipStream_t st) -> int {
    if (EBITS == 8 && a_fmt == 1) return dispatch_variant_a5(v, q, st, name);
    return dispatch_variant<EBITS, EBITS == 8>(v, q, st, name);
  };
  if (!A || !B || !A_sf || !B_sf || !alpha || !D) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  // [r5] operands are fetched as 16-byte LDS-DMA pieces and the output leaves as 16-byte stores (the reference's CUTLASS kernels ask for 128-bit alignment as well, via TMA)
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)A_sf | (uintptr_t)B_sf | (uintptr_t)D) % 16)
    return fail(QAMD_ERR_INVALID, "%s: A, B, the scale operands and D must be 16-byte aligned", name);
  if (M <= 0 || N <= 0) return fail(QAMD_ERR_INVALID, "%s: M and N must be positive (got M=%lld N=%lld)", name, (long long)M, (long long)N);
  const int kalign = (EBITS == 4) ? 128 : 32;
  if (K < 32 || K % kalign) return fail(QAMD_ERR_INVALID, "%s: K must be a positive multiple of %d (got %lld)", name, kalign, (long long)K);
  if (N % 8) return fail(QAMD_ERR_INVALID, "%s: N must be a multiple of 8 (got %lld)", name, (long long)N);
  const int64_t rowbytes = K * EBITS / 8;
  const int64_t CB = cdiv(K / 32, 4);
  const int64_t a_bytes = M * rowbytes, b_bytes = N * rowbytes;
  const int64_t sfa_bytes = cdiv(M, 128) * CB * 512, sfb_bytes = cdiv(N, 128) * CB * 512;
  if (M * N >= (1ll << 40) || M >= (1ll << 31) || N >= (1ll << 31)) return fail(QAMD_ERR_INVALID, "%s: an output of 2^40 elements is not supported", name);
  if (ldd == 0) ldd = N;
  if (b_bytes >= (1ll << 31)) {
    // B of >= 2 GiB (e.g. a 262400 x 16384 fp4 weight): column ranges of whole 256-column tiles, each writing its columns of the
    // same D (row stride ldd); A is shared.  Every output element is computed by exactly one launch, in the same K order.
    const int64_t cols = ((1ll << 31) - 1) / rowbytes / 256 * 256;
    if (cols < 256) return fail(QAMD_ERR_INVALID, "%s: K too large for a 256-column range of B to stay below 2 GiB", name);
    for (int64_t c0 = 0; c0 < N; c0 += cols) {
      const int64_t nc = std::min(cols, N - c0);
      if (int rc = gemm_mx<EBITS>(name, A, (const uint8_t*)B + c0 * rowbytes, A_sf, (const uint8_t*)B_sf + (c0 / 128) * CB * 512, alpha,
                                  (uint16_t*)D + c0, M, nc, K, stream, ws, ws_bytes, a_fmt, ldd))
        return rc;
    }
    return QAMD_OK;
  }
  if (a_bytes >= (1ll << 31)) {
    // The kernels address an operand through 32-bit buffer-descriptor offsets (< 2 GiB); the reference's CUTLASS kernels use
    // 64-bit strides.  A larger A (large batch x long K, e.g. 262144 x 16384 fp4) runs as row ranges of whole 256-row tiles:
    // rebased A / scale / D pointers, same B -- every output element is computed by exactly one launch, in the same K order.
    const int64_t rows = ((1ll << 31) - 1) / rowbytes / 256 * 256;
    if (rows < 256) return fail(QAMD_ERR_INVALID, "%s: K too large for a 256-row range of A to stay below 2 GiB", name);
    for (int64_t r0 = 0; r0 < M; r0 += rows) {
      const int64_t mc = std::min(rows, M - r0);
      if (int rc = gemm_mx<EBITS>(name, (const uint8_t*)A + r0 * rowbytes, B, (const uint8_t*)A_sf + (r0 / 128) * CB * 512, B_sf, alpha,
                                  (uint16_t*)D + r0 * ldd, mc, N, K, stream, ws, ws_bytes, a_fmt, ldd))
        return rc;
    }
    return QAMD_OK;
  }
  GemmParams p;
  p.A = (const uint8_t*)A; p.B = (const uint8_t*)B; p.SFA = (const uint8_t*)A_sf; p.SFB = (const uint8_t*)B_sf;
  p.alpha = alpha; p.D = (uint16_t*)D; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ldd = (int)ldd;
  p.a_bytes = (uint32_t)a_bytes; p.b_bytes = (uint32_t)b_bytes;
  p.sfa_bytes = (uint32_t)sfa_bytes; p.sfb_bytes = (uint32_t)sfb_bytes;
  p.pp_shift = opt_pp_shift();
  p.pp_flags = opt_pp_flags();
  p.dbg = opt_dbg();
  p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0; p.sk_tiles = 0;
  hipStream_t s = (hipStream_t)stream;
  int variant = opt_gemm_variant();
  if (variant >= 61 && variant <= 66) variant = 0;   // these select the NN operand path only (matmul_mxf8_bf16_nn)
  // ring schedule + optional split-K (needs caller scratch; "pp_flags" bit 7 turns split-K off, bit 8 the ring rule)
  // the plan with K ranges needs the caller's scratch; without it (or with too little) the best single-pass plan
  SmallPlan pl = plan_small<EBITS>(M, N, K, true);
  if (pl.variant && pl.splits > 1 && !(ws && ws_bytes >= splitk_ws_bytes(pl.variant, M, N, pl.splits) && !(opt_pp_flags() & 128))) pl = plan_small<EBITS>(M, N, K, false);
  auto ring_launch = [&](int v, int splits) -> int {
    {   // every split non-empty (a forced count may not divide the K stages)
      const int64_t KT = cdiv(K * EBITS / 8, 128);
      splits = (int)std::min<int64_t>(std::max(splits, 1), std::min<int64_t>(KT, 8));
      splits = (int)cdiv(KT, cdiv(KT, splits));
    }
    if (splits > 1 && ws && ws_bytes >= splitk_ws_bytes(v, M, N, splits) && !(p.pp_flags & 128)) {
      p.ws = (float*)ws; p.splits = splits; p.ctr = nullptr; p.tag = 0;
#if QAMD_BENCH
      if (p.pp_flags & 512) {   // lab: ONE launch, the last split to arrive for a tile reduces it (gemm_mx.hip.h epilogue_splitk_fused);
                                // measured slower than the reduce kernel below, see there
        p.ctr = (unsigned long long*)((char*)ws + splitk_ctr_offset(M, N, splits));
        p.tag = (next_launch_tag() & ((1ull << 56) - 1)) << 8;
        return dispatch(v, p, s);
      }
#endif
      // second launch: sum the partials in fixed z order, alpha, bf16 (deterministic)
      if (int rc = dispatch(v, p, s)) return rc;
      if (t_dry.on) return 0;
      const int64_t quads = M * (N / 4);
      const int grid = (int)std::min<int64_t>(cdiv(quads, 256), 2048);
      switch (splits) {
#define QAMD_RED(S_) case S_: hipLaunchKernelGGL(splitk_reduce_kernel<S_>, dim3(grid), dim3(256), 0, s, (const float*)ws, p.D, alpha, (int)M, (int)N, p.ldd); break;
        QAMD_RED(2) QAMD_RED(3) QAMD_RED(4) QAMD_RED(5) QAMD_RED(6) QAMD_RED(7) QAMD_RED(8)
#undef QAMD_RED
        default: return fail(QAMD_ERR_INVALID, "%s: unsupported split count %d", name, splits);
      }
      return check_launch("splitk_reduce_kernel");
    }
    p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
    return dispatch(v, p, s);
  };
  const bool can_split = pl.variant && pl.splits > 1 && ws && ws_bytes >= splitk_ws_bytes(pl.variant, M, N, pl.splits) && !(p.pp_flags & 128);
  if (variant == 77) return ring_launch(70, pl.variant == 70 ? pl.splits : 1);   // lab: 64x64 ring (+ split-K) whatever M
  if (variant >= 70 && variant <= 73 && opt_splitk_force() > 0) return ring_launch(variant, opt_splitk_force());   // lab: forced tile x forced split
  // small batch (M <= 32): weight-bandwidth bound.  With fewer than 128 64-row tiles (N < 8192) the split-K kernel without
  // LDS staging wins (gemm_mx_skinny.hip.h: N = K = 4096, M = 16: 5.9 us vs 7.1 us for the ring kernel on 64 CUs); from
  // 128 tiles on, the 64x64 ring kernel streams the weight through full-line LDS-DMA and wins (N = 14336, K = 4096: 6.9 us
  // vs 9.7 us; N = 57344, K = 8192: 36 us vs 58-74 us), as does ring + split-K over caller scratch for a long K
  // (N = 4096, K = 14336, M = 16: 11.6 us vs 14.8 us).  profiles/native_r1_skinny_shapes.log, native_r1_ring.log
  // [r3] re-measured GPU-only (HIP-graph replays, tools/calib_mx_small.py with CALIB_MS=1,4,8,16,24,32; profiles/calib_mx_decode_graph_r3.txt): the split-K kernel's time
  // grows with M (N = K = 4096: 4.65 us at M = 1, 5.4 at 16, 6.1 at 32) while the 3-deep ring stays at 5.0 -- it keeps M <= 8 (now including N = 8192: 8.9 us against
  // the ring's 10.8 at K = 8192) and M <= 24 only against small weights (N <= 2048: 3.7 against 3.9 us)
  // [r6] small batches against a weight that fills the chip with 32x32 tiles: the in-workgroup K-split kernel (gemm_mx_ks.hip.h; ks_plan above)
  if (EBITS == 4 && variant == 0 && !(p.pp_flags & 256)) {
    if (const int kv = ks_plan(M, N, K)) return dispatch(kv, p, s);
  }
  // [r6] MXFP8 small batches (e4m3 and e5m2 A): the wave-owned kernel where os8_plan says so
  if (EBITS == 8 && variant == 0 && !(p.pp_flags & 256)) {
    if (const int kv = os8_plan(M, N, K)) return dispatch(kv, p, s);
  }
  const bool skinny_auto = variant == 0 && !can_split && ((M <= 8 && cdiv(N, 64) <= chip_cus() / 2) || (M <= 24 && cdiv(N, 64) <= chip_cus() / 8));
  if (EBITS == 4 && (variant == 60 || (variant >= 44 && variant <= 49) || skinny_auto)) {
    if (dry_record(variant ? variant : 60, p.N, 1)) return 0;
    SkinnyParams q;
    q.A = p.A; q.B = p.B; q.SFA = p.SFA; q.SFB = p.SFB; q.alpha = alpha; q.D = p.D; q.M = p.M; q.N = p.N; q.K = p.K;
    q.a_bytes = p.a_bytes; q.b_bytes = p.b_bytes; q.sfa_bytes = p.sfa_bytes; q.sfb_bytes = p.sfb_bytes; q.ldd = p.ldd;
    switch (variant) {   // 44..49: lab-only shapes of the split-K kernel (waves, segments per trip, chunk mapping)
#if QAMD_BENCH
      case 44: launch_skinny<true, 8, 2, true>(q, s); break;
      case 45: launch_skinny<true, 4, 2, false>(q, s); break;
      case 46: launch_skinny<true, 4, 2, true>(q, s); break;
      case 47: launch_skinny<true, 8, 2, false>(q, s); break;
      case 48: launch_skinny<true, 8, 1, true>(q, s); break;
      case 49: launch_skinny<true, 4, 4, true>(q, s); break;
#endif
      default: launch_skinny<true, 8, 4, false>(q, s);   // 4 segments per wave per trip: 32 b128 loads in flight per wave
    }
    return check_launch("gemm_mx_skinny_kernel");
  }
  if (variant == 0 && !(p.pp_flags & 256) && pl.variant) return ring_launch(pl.variant, pl.splits);
  if (variant == 0) {
    // auto (measured, profiles/native_r1_schedules.log, profiles/bench_sweep_*.txt): the largest tile that still gives
    // every CU work -- 256x256 ("deep" schedule, 4 waves of 128x128), then 128x128, 128x64 / 64x128, 64x64 (simple
    // schedule, several workgroups per CU).  (fp4 with M <= 32 went to the split-K kernel above.)
    auto tiles = [&](int bm, int bn) { return cdiv(M, bm) * cdiv(N, bn); };
    const int cus = chip_cus();
    const int64_t want = cus * 3 / 4;   // 3/4 of the CUs
    // no point in tiles taller than the problem; 64x64 until 64x128 tiles fill the chip 1.5 times (weight-bandwidth
    // bound: N = 28672, K = 4096, M = 32: 12.6 us with 448 tiles of 64x64 vs 14.3 us with 224 of 64x128)
    if (M <= 64) variant = (tiles(64, 128) >= cus * 3 / 2) ? 28 : 29;
    else if (N <= 64) variant = (tiles(128, 64) >= want) ? 27 : 29;
    // [r3] at most 128 rows against a wide weight (more 128x128 tiles than CUs): the 256-row persistent tile would be half empty -- 128x128 tiles, two workgroups
    // per CU (96 x 57344 x 8192: 52.4 -> 42.7 us, 128 x 51200 x 5120: 32.7 -> 26.8, MXFP8 96 x 57344 x 8192 113.5 -> 100.3; profiles/calib_mx_small_r3.txt)
    else if (M <= 128 && tiles(128, 128) >= want) variant = 24;
    // [r3] ... or more than HALF of them with K >= 8 stages: the smaller tiles then no longer fit one round (128x128: two per CU, 256x128: one per CU --
    // both hold exactly cus / 2 tiles' worth of 256x256 output), and a second, part-filled round costs more than idle CUs do:
    // 2560 x 4096 x 4096 (160 tiles) 34.6 -> 28.8 us, 1536 x 6144 x 4096 33.2 -> 28.3, 2048 x 5120 x 5120 42.2 -> 35.3, MXFP8 2560 x 4096 x 4096
    // 49.3 -> 39.4, K = 14336 likewise (146.9 -> 120.9); at exactly half (2048 x 4096: 128 tiles) the small tiles tie or win
    // (tools/calib_tiles.py, profiles/calib_tiles_r3.txt; found by tools/dip_scan.py: a LARGER batch ran faster)
    else if (tiles(256, 256) >= want || (2 * tiles(256, 256) > cus && cdiv(K * EBITS / 8, 128) >= 8)) {
      // the persistent deep schedule (one workgroup per CU walks the tiles, epilogue folded into the last K stage), fp4 and
      // fp8; its epilogue addresses a tile with 32-bit byte offsets, so absurdly wide outputs stay with 256x128 simple tiles
      const int big = ldd < (1ll << 22) ? 90 : 25;
      variant = big;
      // Wave quantisation: T tiles on `cus` CUs run ceil(T / cus) rounds, the last one part-filled (4096 x 5120: 320 tiles =
      // 1.25 rounds).  [r3] The residual tiles run as 128x128 quarter tiles on extra workgroups of the SAME launch
      // (gemm_mx_hetero_kernel: dispatched CU by CU as the persistent workgroups retire) whenever the cost model says so;
      // otherwise ONE persistent launch with balanced rounds (deepp_grid).
      const int64_t tm = cdiv(M, 256), tn = cdiv(N, 256), T = tm * tn;
      bool hetero_ok = big == 90 && !(opt_pp_flags() & 64);   // (lab, "pp_flags" bit 6: balanced rounds only)
#if QAMD_BENCH
      // lab, "pp_flags" bit 13: the round-1/2 form of the same idea INSTEAD -- the trailing tile columns as a SECOND launch of smaller
      // tiles (kept for the A/B in profiles/native_r3_heterobench.log)
      const int64_t full = (T / cus) * cus, main_cols = (tm > 0) ? full / tm : 0;
      const bool two_launch = (opt_pp_flags() & 8192) != 0;
      if (two_launch && hetero_ok && full >= cus && full < 3 * cus && main_cols >= 1 && main_cols < tn && (T - main_cols * tm) <= 80) {
        const int64_t n1 = main_cols * 256;
        GemmParams pm = p;
        pm.N = (int)n1; pm.b_bytes = (uint32_t)(n1 * rowbytes); pm.sfb_bytes = (uint32_t)(cdiv(n1, 128) * CB * 512);
        if (int rc = dispatch(big, pm, s)) return rc;
        GemmParams pt = p;
        pt.N = (int)(N - n1);
        pt.B = p.B + n1 * rowbytes; pt.b_bytes = (uint32_t)((N - n1) * rowbytes);
        pt.SFB = p.SFB + (n1 / 128) * CB * 512; pt.sfb_bytes = (uint32_t)(cdiv(N - n1, 128) * CB * 512);
        pt.D = p.D + n1;
        const int64_t Nt = N - n1;
        auto tt = [&](int bm, int bn) { return cdiv(M, bm) * cdiv(Nt, bn); };
        const int vt = (tt(256, 128) >= want) ? 25 : (tt(128, 128) >= want) ? 24 : (tt(128, 64) >= want) ? 27 : 29;
        return dispatch(vt, pt, s);
      }
      if (two_launch) hetero_ok = false;
#endif
      if (hetero_ok && hetero_wins(T, cus)) variant = 98;
    }
    else if (tiles(128, 128) >= want) {
      // [r3] half-chip outputs (fewer than `want` tiles of 256x256): with a long K (>= 32 stages) the 256x128 tile on four waves wins 5-6 %
      // over 128x128 tiles on two workgroups per CU; with K = 4096 the two tie (22.97 vs 22.95 us at 2048 x 4096 x 4096) and 128x128 stays
      const int64_t KTs = cdiv(K * EBITS / 8, 128);
      // (MXFP8 likewise: 2048 x 4096 x 4096 34.9 -> 32.7 us, x 8192 63.2 -> 57.1 us; K = 2048 = 16 stages: -2 %, stays)
      variant = (M >= 256 && KTs >= 32 && tiles(256, 128) >= want) ? 58 : 24;
      // [r3] ... and, at the same K, whenever the 128x128 grid no longer fits one tile per CU while the 256x128 grid still does: the CUs that hold
      // two 128x128 tiles set the kernel's time.  MXFP8 768 x 6144 x 4096 28.3 -> 23.5 us, 1024 x 5120 x 5120 36.9 -> 29.6, 1024 x 5120 x 25600
      // 145.3 -> 120.8; MXFP4 1024 x 5120 x 25600 83.5 -> 75.2.  Not below 32 stages: MXFP4 2048 x 4096 x 4096 23.6 against 26.6 on the 256x128 tile
      // (profiles/calib_tiles_r3.txt)
      if (M >= 256 && KTs >= 32 && tiles(128, 128) > cus && tiles(256, 128) <= cus) variant = 58;
    }
    else if (tiles(128, 64) >= want || tiles(64, 128) >= want) variant = (N >= M) ? 27 : 28;
    else variant = 29;
  }
  if (variant == 89) {   // (lab: forced stream-K; falls back to the persistent kernel when the shape has nothing to cut or the scratch is missing)
    const int cus = chip_cus();
    const int64_t T = cdiv(M, 256) * cdiv(N, 256), KTe = (cdiv(K * EBITS / 8, 128) + 1) / 2 * 2;
    const bool ok = a_fmt == 0 && ws && ws_bytes >= sk_ws_bytes(cus) && (uintptr_t)ws % 16 == 0 && T % cus != 0 && T > cus && KTe >= 8 && (K * EBITS / 8) % 256 == 0 && ldd < (1ll << 22);
    if (!ok) variant = 90;
    else {
      p.sk_tiles = (int)(cus + T % cus);
      p.ws = (float*)ws;
      p.ctr = (unsigned long long*)((char*)ws + (int64_t)cus * SK_PART_BYTES);
      p.tag = next_launch_tag();
    }
  }
  return dispatch(variant, p, s);
}

#endif   // QAMD_DEF(1)

template <int R, bool NV, int METHOD, bool MASK, bool BLK>
int launch_quant(const QuantParams& p, hipStream_t s, int grid) {
#if QAMD_BENCH
  if (!opt_hw_fp4()) hipLaunchKernelGGL((fused_quantize_kernel<R, NV, METHOD, MASK, false, BLK>), dim3(grid), dim3(256), 0, s, p);
  else
#endif
  hipLaunchKernelGGL((fused_quantize_kernel<R, NV, METHOD, MASK, true, BLK>), dim3(grid), dim3(256), 0, s, p);
  return check_launch("fused_quantize_kernel");
}

// BLK: scales written in the to_blocked() layout (qutlass_amd_fused_quantize_{mx,nv}_blocked)
template <bool NV, int METHOD, bool MASK, bool BLK = false>
int dispatch_rot(int rot, const QuantParams& p, hipStream_t s, int grid, const char* name) {
  switch (rot) {
    case 16:
      if constexpr (NV) return launch_quant<16, NV, METHOD, false, BLK>(p, s, grid);
      break;
    case 32: return launch_quant<32, NV, METHOD, MASK, BLK>(p, s, grid);
    case 64:
      if constexpr (!MASK) return launch_quant<64, NV, METHOD, false, BLK>(p, s, grid);
      break;
    case 128:
      if constexpr (!MASK) return launch_quant<128, NV, METHOD, false, BLK>(p, s, grid);
      break;
  }
  if (MASK) return fail(QAMD_ERR_INVALID, "%s: Unsupported rotation size %d; expected 32.", name, rot);
  return fail(QAMD_ERR_INVALID, "%s: Unsupported rotation size %d; expected %s32, 64, or 128.", name, rot, NV ? "16, " : "");
}

#if QAMD_TU != 0
#if QAMD_TU == 5
#define QAMD_ROT_INST template
#else
#define QAMD_ROT_INST extern template
#endif
#define QAMD_ROT_BOTH(NV_, M_, K_) \
  QAMD_ROT_INST int dispatch_rot<NV_, M_, K_, false>(int, const QuantParams&, hipStream_t, int, const char*); \
  QAMD_ROT_INST int dispatch_rot<NV_, M_, K_, true>(int, const QuantParams&, hipStream_t, int, const char*);
QAMD_ROT_BOTH(false, METHOD_QUEST, true)
QAMD_ROT_BOTH(false, METHOD_QUEST, false)
QAMD_ROT_BOTH(false, METHOD_ABSMAX, false)
QAMD_ROT_BOTH(true, METHOD_QUEST, false)
QAMD_ROT_BOTH(true, METHOD_ABSMAX, false)
#undef QAMD_ROT_BOTH
#undef QAMD_ROT_INST
#endif

// backward_t_bf16 / backward_qt_bf16 kernels live in unit 5 with the other rotation quantizers (MFMA results straight in VGPRs: a
// v_accvgpr_read per accumulator register is 32 more VALU issues per tile)
//   which: 1 = the round-3 kernel (8 waves per unit of 8 groups x 64 m, two barriers per unit); 2 = wave-owned 64-byte segments (units of 4
//   groups, 12 - 16 waves per CU); 3 = wave-owned 128-byte lines (units of 8 groups, 8 waves per CU).  Product: QT 2, T 3 (bwd_kernel_choice).
int launch_bwd_quant(const BwdTParams& p, bool qt, int which, bool hw, int grid, hipStream_t s);
#if QAMD_DEF(5)
int launch_bwd_quant(const BwdTParams& p, bool qt, int which, bool hw, int grid, hipStream_t s) {
#define QAMD_BWD_GO(KERN, THREADS) hipLaunchKernelGGL((KERN), dim3(grid), dim3(THREADS), 0, s, p)
  if (which >= 5 && which <= 8) {   // [r5] QT only: the wave-owned kernel with the input fetched as whole lines into a ring shared by the workgroup's four m-tiles
#if QAMD_BENCH
    if (which == 6) { QAMD_BWD_GO((bwd_qt_ring_kernel<true, 8, 4, true>), 256); return check_launch("bwd_qt_ring_kernel"); }
    if (which == 7) { QAMD_BWD_GO((bwd_qt_ring_kernel<true, 8, 4, false>), 256); return check_launch("bwd_qt_ring_kernel"); }
    if (which == 8) { QAMD_BWD_GO((bwd_qt_ring_kernel<true, 4, 3, true>), 256); return check_launch("bwd_qt_ring_kernel"); }
    if (!hw) { QAMD_BWD_GO((bwd_qt_ring_kernel<false, 4, 3, false>), 256); return check_launch("bwd_qt_ring_kernel"); }
#endif
    QAMD_BWD_GO((bwd_qt_ring_kernel<true, 4, 3, false>), 256);
    return check_launch("bwd_qt_ring_kernel");
  }
#if QAMD_BENCH
  if (which == 4) {   // QT only: whole-line panels ([256 n][256 m] per workgroup), quartet_bwd_lab.hip.h
    if (!hw) { QAMD_BWD_GO((bwd_qt_panel_kernel<false>), 512); return check_launch("bwd_qt_panel_kernel"); }
    QAMD_BWD_GO((bwd_qt_panel_kernel<true>), 512);
    return check_launch("bwd_qt_panel_kernel");
  }
#endif
  if (which == 1) {
#if QAMD_BENCH
    if (!hw) { if (qt) QAMD_BWD_GO((bwd_quant_t_kernel<true, false>), 512); else QAMD_BWD_GO((bwd_quant_t_kernel<false, false>), 512); return check_launch("bwd_quant_t_kernel"); }
#endif
    if (qt) QAMD_BWD_GO((bwd_quant_t_kernel<true, true>), 512); else QAMD_BWD_GO((bwd_quant_t_kernel<false, true>), 512);
    return check_launch("bwd_quant_t_kernel");
  }
#if QAMD_BENCH
  if (!hw) {
    if (which == 3) { if (qt) QAMD_BWD_GO((bwd_quant_tw_kernel<true, false, 8>), 256); else QAMD_BWD_GO((bwd_quant_tw_kernel<false, false, 8>), 256); }
    else            { if (qt) QAMD_BWD_GO((bwd_quant_tw_kernel<true, false, 4>), 256); else QAMD_BWD_GO((bwd_quant_tw_kernel<false, false, 4>), 256); }
    return check_launch("bwd_quant_tw_kernel");
  }
  if (which == 9) {   // [r6] lab: units of 2 groups (32-byte output segments, twice the waves of variant 2) -- latency against write efficiency at small inputs
    if (qt) QAMD_BWD_GO((bwd_quant_tw_kernel<true, true, 2>), 256); else QAMD_BWD_GO((bwd_quant_tw_kernel<false, true, 2>), 256);
    return check_launch("bwd_quant_tw_kernel");
  }
  if (qt && which == 3) { QAMD_BWD_GO((bwd_quant_tw_kernel<true, true, 8>), 256); return check_launch("bwd_quant_tw_kernel"); }
  if (!qt && which == 2) { QAMD_BWD_GO((bwd_quant_tw_kernel<false, true, 4>), 256); return check_launch("bwd_quant_tw_kernel"); }
#endif
  if (qt) QAMD_BWD_GO((bwd_quant_tw_kernel<true, true, 4>), 256); else QAMD_BWD_GO((bwd_quant_tw_kernel<false, true, 8>), 256);
#undef QAMD_BWD_GO
  return check_launch("bwd_quant_tw_kernel");
}
#endif

#if QAMD_DEF(1)
int quant_grid(int ntiles, int rot) {
  // 4 waves per workgroup, one 32-row tile per wave per trip.  Small rotations are pure streaming: as many waves as
  // fit (8 workgroups per CU; 8192^2 NV runs at 6.26 TB/s).  R = 128 tiles are 8 KiB with 32 MFMAs each: fewer, longer
  // waves let the software pipeline (next tile's loads in flight during this tile's MFMAs) overlap (13.2 vs 14.6 us).
  // ([r4] One 12-wave workgroup per CU sharing one H image -- three waves per SIMD instead of two -- measured slower: 4096^2 8.9 -> 10.8 us warm.)
  int per_cu = opt_quant_wg_per_cu();
  if (per_cu <= 0) per_cu = rot >= 128 ? 2 : (rot >= 64 ? 4 : 8);
  int g = (ntiles + 3) / 4;
  const int cap = chip_cus() * per_cu;
  return g < 1 ? 1 : (g > cap ? cap : g);
}

// blocked-scale quantizers: the kernel also zero-fills the padding of the blocked layout (up to 127 rows x all column tiles); a one-row
// input would leave that to a single workgroup (M = 1, K = 4096: 14.4 us for the 2-launch linear layer against 9.1 us with 3 launches)
int blocked_pad_grid(int grid, int sf_rows, int sf_cols) {
  const int64_t prow = cdiv(sf_rows, 128) * 128, cb = cdiv(sf_cols, 4);
  const int64_t stores = (prow - sf_rows) * cb + (int64_t)sf_rows * (cb * 4 - sf_cols);
  const int64_t want = std::min<int64_t>(cdiv(stores, 256), chip_cus());
  return (int)std::max<int64_t>(grid, want);
}

#endif   // QAMD_DEF(1)

}  // namespace qamd_host

using namespace qamd_host;

#if QAMD_DEF(1)
// the library is built with -fvisibility=hidden: only the C ABI below is exported
#pragma GCC visibility push(default)
extern "C" {

int qutlass_amd_matmul_mxf4_bf16_tn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                    const float* alpha, void* D, int64_t M, int64_t N, int64_t K, void* stream) {
  return gemm_mx<4>("matmul_mxf4_bf16_tn", A, B, A_sf, B_sf, alpha, D, M, N, K, stream);
}

int64_t qutlass_amd_gemm_splitk_workspace_bytes(int ebits, int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || (ebits != 4 && ebits != 8)) return 0;
  const SmallPlan pl = (ebits == 4) ? plan_small<4>(M, N, K) : plan_small<8>(M, N, K);
  int64_t need = (pl.variant && pl.splits > 1) ? splitk_ws_bytes(pl.variant, M, N, pl.splits) : 0;
  if (ebits == 4 && opt_gemm_variant() == 0 && !(opt_pp_flags() & 256) && ks_plan(M, N, K)) need = 0;   // [r6] the in-workgroup K-split kernel takes the shape: no scratch
  if (ebits == 8 && opt_gemm_variant() == 0 && !(opt_pp_flags() & 256) && os8_plan(M, N, K)) need = 0;  // [r6] ... the wave-owned kernel an MXFP8 one
#if QAMD_BENCH
  if (opt_gemm_variant() == 89) need = std::max<int64_t>(need, sk_ws_bytes(chip_cus()));   // lab: forced stream-K
  if (opt_splitk_force() > 1) need = std::max<int64_t>(need, splitk_ws_bytes(70, M, N, 8));   // lab: room for any forced tile x split
#endif
  return need;
}

int qutlass_amd_matmul_mxf4_bf16_tn_ws(const void* A, const void* B, const void* A_sf, const void* B_sf, const float* alpha, void* D,
                                       int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes, void* stream) {
  if (workspace && (uintptr_t)workspace % 16) return fail(QAMD_ERR_INVALID, "matmul_mxf4_bf16_tn: the workspace must be 16-byte aligned");   // v4f partials
  return gemm_mx<4>("matmul_mxf4_bf16_tn", A, B, A_sf, B_sf, alpha, D, M, N, K, stream, workspace, workspace_bytes);
}

int qutlass_amd_matmul_mxf8_bf16_tn_ws(const void* A, const void* B, const void* A_sf, const void* B_sf, const float* alpha, void* D,
                                       int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes, void* stream) {
  if (workspace && (uintptr_t)workspace % 16) return fail(QAMD_ERR_INVALID, "matmul_mxf8_bf16_tn: the workspace must be 16-byte aligned");   // v4f partials
  return gemm_mx<8>("matmul_mxf8_bf16_tn", A, B, A_sf, B_sf, alpha, D, M, N, K, stream, workspace, workspace_bytes);
}

static int ada_impl(const void* A, const void* B, const void* A_sf, const void* B_sf, const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                    int64_t ldd, void* stream) {
  const char* name = "matmul_ada_mxf4_bf16_tn";
  if (!A || !B || !A_sf || !B_sf || !alpha || !D) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (M <= 0 || N <= 0) return fail(QAMD_ERR_INVALID, "%s: M and N must be positive (got M=%lld N=%lld)", name, (long long)M, (long long)N);
  if (K < 128 || K % 128) return fail(QAMD_ERR_INVALID, "%s: K must be a positive multiple of 128 (got %lld)", name, (long long)K);
  if (N % 8) return fail(QAMD_ERR_INVALID, "%s: N must be a multiple of 8 (got %lld)", name, (long long)N);
  if (M >= (1ll << 31) || N >= (1ll << 31) || M * N >= (1ll << 40)) return fail(QAMD_ERR_INVALID, "%s: an output of 2^40 elements is not supported", name);
  if (ldd == 0) ldd = N;
  const int64_t rowbytes = K / 2, KB = K / 32;
  // operands of >= 2 GiB (32-bit buffer-descriptor offsets; the reference hands 64-bit strides to CUTLASS): ranges of whole 64-row
  // tiles with rebased operand / row-major scale / D pointers -- every output element is computed by exactly one launch
  if (N * rowbytes >= (1ll << 31)) {
    const int64_t cols = ((1ll << 31) - 1) / rowbytes / 64 * 64;
    if (cols < 64) return fail(QAMD_ERR_INVALID, "%s: K too large for a 64-column range of B to stay below 2 GiB", name);
    for (int64_t c0 = 0; c0 < N; c0 += cols)
      if (int rc = ada_impl(A, (const uint8_t*)B + c0 * rowbytes, A_sf, (const uint8_t*)B_sf + c0 * KB, alpha, (uint16_t*)D + c0, M, std::min(cols, N - c0), K, ldd, stream)) return rc;
    return QAMD_OK;
  }
  if (M * rowbytes >= (1ll << 31)) {
    const int64_t rows = ((1ll << 31) - 1) / rowbytes / 64 * 64;
    if (rows < 64) return fail(QAMD_ERR_INVALID, "%s: K too large for a 64-row range of A to stay below 2 GiB", name);
    for (int64_t r0 = 0; r0 < M; r0 += rows)
      if (int rc = ada_impl((const uint8_t*)A + r0 * rowbytes, B, (const uint8_t*)A_sf + r0 * KB, B_sf, alpha, (uint16_t*)D + r0 * ldd, std::min(rows, M - r0), N, K, ldd, stream)) return rc;
    return QAMD_OK;
  }
  // Same regimes as matmul_mxf4_bf16_tn: the LDS-free split-K kernel while the weight has fewer than 128 64-row tiles;
  // from 128 tiles on (N >= 8192) the 64x64 ring kernel with row-major scale fetch streams the weight through full-line
  // LDS-DMA (M = 16: N = 14336, K = 4096 9.7 -> 6.9 us; N = 57344, K = 8192 62.6 -> 39.9 us), and any M > 32 goes there
  // too ("gemm_variant" 60 / 70 force either).
  const int forced = opt_gemm_variant();
  const int64_t T64 = cdiv(N, 64);   // (no split-K here -- the op has no scratch argument -- so a long K on few tiles stays with the split-K kernel:
                                     //  8 x 8192 x 28672: 27.9 us vs 34.2 us on 128 workgroups of the ring kernel)
  const int cus = chip_cus();
  const bool ring = forced == 70 || (forced != 60 && (M > 32 || T64 >= cus || (T64 >= cus / 2 && K < 16384)));
  // [r6] K <= 4096 and at most one 32x32 tile per CU: the one-shot kernel with row-major scale pieces (gemm_mx_os.hip.h; "gemm_variant" 568 forces it where it fits)
  const int os_tn = forced == 0 ? os_plan(M, N, K, true) : 0;
  const bool os16 = forced == 569 || os_tn == 16;   // 16 columns per workgroup
  // [r6] ... its decode form (16x16 tiles on the 16x16x128 MFMA) where those fit one per CU (matmul_mxf4_bf16_tn's rule, ks_plan)
  const int ada_tn16 = (forced >= 571 && forced <= 575) ? (forced == 571 ? 16 : forced == 572 ? 32 : forced == 573 ? 48 : forced == 574 ? 56 : 64)
                       : (os_tn != 0 && cdiv(M, 16) * cdiv(N, 16) <= cus) ? 16
                       : (forced == 0 && M <= 16 && os16_wide_plan(4, N, K)) ? os16_tn(N) : 0;   // (wider column tiles: matmul_mxf4_bf16_tn's rule)
  if (ada_tn16) {
    GemmParams p;
    p.A = (const uint8_t*)A; p.B = (const uint8_t*)B; p.SFA = (const uint8_t*)A_sf; p.SFB = (const uint8_t*)B_sf;
    p.alpha = alpha; p.D = (uint16_t*)D; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ldd = (int)ldd;
    p.a_bytes = (uint32_t)(M * rowbytes); p.b_bytes = (uint32_t)(N * rowbytes);
    p.sfa_bytes = (uint32_t)(M * KB); p.sfb_bytes = (uint32_t)(N * KB);
    p.pp_shift = opt_pp_shift(); p.pp_flags = opt_pp_flags(); p.dbg = opt_dbg();
    return launch_gemm_os16_tn<4, 0, true>(ada_tn16, p, (hipStream_t)stream);
  }
  const bool os64 = forced == 570 || (forced == 0 && os_tn == 0 && os64_plan(M, N, K, true));   // 64x32 tiles where the 32-row tiles overflow the chip (os64_plan)
  const bool oneshot = (forced >= 568 && forced <= 570) ? cdiv(M, 32) * cdiv(N, 32) <= 4 * cus : (os_tn != 0 || os64);
  if (ring || oneshot) {
    GemmParams p;
    p.A = (const uint8_t*)A; p.B = (const uint8_t*)B; p.SFA = (const uint8_t*)A_sf; p.SFB = (const uint8_t*)B_sf;
    p.alpha = alpha; p.D = (uint16_t*)D; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ldd = (int)ldd;
    p.a_bytes = (uint32_t)(M * rowbytes); p.b_bytes = (uint32_t)(N * rowbytes);
    p.sfa_bytes = (uint32_t)(M * KB); p.sfb_bytes = (uint32_t)(N * KB);   // row-major (rows, K/32), un-swizzled
    p.pp_shift = opt_pp_shift(); p.pp_flags = opt_pp_flags(); p.dbg = opt_dbg();
    p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
    if (oneshot && os64) return launch_gemm_os<true, 32, 4, 64>(p, (hipStream_t)stream);
    if (oneshot) return os16 ? launch_gemm_os<true, 16>(p, (hipStream_t)stream) : launch_gemm_os<true>(p, (hipStream_t)stream);
#if QAMD_BENCH
    if (opt_gemm_variant() == 178) return launch_gemm<GemmCfg<64, 64, 2, 2, 4, false, 0, 3>, 8>(p, (hipStream_t)stream);   // round-1 ring schedule
#endif
    return launch_gemm<GemmCfg<64, 64, 2, 2, 4, false, 0, 3>, 10>(p, (hipStream_t)stream);
  }
  SkinnyParams q;
  q.A = (const uint8_t*)A; q.B = (const uint8_t*)B; q.SFA = (const uint8_t*)A_sf; q.SFB = (const uint8_t*)B_sf;
  q.alpha = alpha; q.D = (uint16_t*)D; q.M = (int)M; q.N = (int)N; q.K = (int)K; q.ldd = (int)ldd;
  q.a_bytes = (uint32_t)(M * rowbytes); q.b_bytes = (uint32_t)(N * rowbytes);
  q.sfa_bytes = (uint32_t)(M * KB); q.sfb_bytes = (uint32_t)(N * KB);   // row-major (rows, K/32), un-swizzled
  launch_skinny<false, 8, 4, false>(q, (hipStream_t)stream);
  return check_launch("gemm_mx_skinny_kernel");
}

int qutlass_amd_matmul_ada_mxf4_bf16_tn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                        const float* alpha, void* D, int64_t M, int64_t N, int64_t K, void* stream) {
  return ada_impl(A, B, A_sf, B_sf, alpha, D, M, N, K, 0, stream);
}

int qutlass_amd_matmul_mxf8_bf16_tn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                    const float* alpha, void* D, int64_t M, int64_t N, int64_t K, void* stream) {
  return gemm_mx<8>("matmul_mxf8_bf16_tn", A, B, A_sf, B_sf, alpha, D, M, N, K, stream);
}

// one rule for the launcher and the workspace query: the persistent kernel on the (K, M) operand wherever the TN op would
// pick the persistent 256x256 kernel for the whole problem (gemm_mx auto rule); everything smaller goes through the
// byte-transpose pre-pass and the TN dispatch with its smaller tiles / split-K
// (operands of >= 2 GiB also take the pre-pass: the in-place path walks the (K, M) operand with 32-bit offsets k * M + m, which only a
//  per-K-chunk descriptor could extend; the (M, K) copy in the workspace then runs as row ranges of the TN dispatch, gemm_mx)
static bool mxf8_nn_is_fused(int64_t M, int64_t N, int64_t K) {
  return M > 64 && N > 64 && N < (1ll << 22) && cdiv(M, 256) * cdiv(N, 256) >= chip_cus() * 3 / 4 && M * K < (1ll << 31) && N * K < (1ll << 31);
}

int64_t qutlass_amd_mxf8_nn_workspace_bytes(int64_t M, int64_t K) { return (M > 0 && K > 0) ? M * K : 0; }
int64_t qutlass_amd_mxf8_nn_workspace_bytes_for(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (opt_gemm_variant() == 0 && mxf8_nn_is_fused(M, N, K)) ? 0 : M * K;
}

static int mxf8_nn_impl(const void* A, const void* B, const void* A_sf, const void* B_sf, const float* alpha, void* D, int64_t M, int64_t N,
                        int64_t K, int a_fmt, void* workspace, int64_t workspace_bytes, void* stream) {
  const char* name = "matmul_mxf8_bf16_nn";
  if (!A) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (M <= 0 || N <= 0) return fail(QAMD_ERR_INVALID, "%s: M and N must be positive (got M=%lld N=%lld)", name, (long long)M, (long long)N);
  if (K < 32 || K % 32) return fail(QAMD_ERR_INVALID, "%s: K must be a positive multiple of 32 (got %lld)", name, (long long)K);
  if (M % 16) return fail(QAMD_ERR_INVALID, "%s: M must be a multiple of 16 for the (K, M) operand (got %lld)", name, (long long)M);
  if (M >= (1ll << 31) || K >= (1ll << 31)) return fail(QAMD_ERR_INVALID, "%s: tensor too large", name);
  // large problems: the persistent kernel reads A^T directly (no pre-pass, no workspace).  Lab library only: "gemm_variant"
  // 63 forces it, 61 forces the per-tile fused kernel of round 1 (dword reads + v_perm byte transposes), 62 the pre-pass
  const int forced = opt_gemm_variant();
  const bool fused = (M * K < (1ll << 31) && N * K < (1ll << 31)) && (forced == 61 || (forced >= 63 && forced <= 66) || (forced == 0 && mxf8_nn_is_fused(M, N, K)));
  if (!fused) {
    if (!workspace) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
    if (workspace_bytes < M * K) return fail(QAMD_ERR_INVALID, "%s: workspace too small (%lld < %lld bytes)", name, (long long)workspace_bytes, (long long)(M * K));
  }
  if (fused) {
    if (!B || !A_sf || !B_sf || !alpha || !D) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
    if (N % 8) return fail(QAMD_ERR_INVALID, "%s: N must be a multiple of 8 (got %lld)", name, (long long)N);
    const int64_t CB = cdiv(K / 32, 4);
    GemmParams p;
    p.A = (const uint8_t*)A; p.B = (const uint8_t*)B; p.SFA = (const uint8_t*)A_sf; p.SFB = (const uint8_t*)B_sf;
    p.alpha = alpha; p.D = (uint16_t*)D; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ldd = (int)N;
    p.a_bytes = (uint32_t)(M * K); p.b_bytes = (uint32_t)(N * K);
    p.sfa_bytes = (uint32_t)(cdiv(M, 128) * CB * 512); p.sfb_bytes = (uint32_t)(cdiv(N, 128) * CB * 512);
    p.pp_shift = opt_pp_shift(); p.pp_flags = opt_pp_flags(); p.dbg = opt_dbg();
    p.ws = nullptr; p.splits = 1; p.ctr = nullptr; p.tag = 0;
    if (dry_record(forced == 61 ? 61 : 63, p.N, 1)) return 0;
    if (a_fmt == 1) return launch_nn_fused_a5(p, (hipStream_t)stream, forced == 61);
#if QAMD_BENCH
    if (forced == 61) return launch_gemm<GemmCfg<256, 256, 2, 2, 8, true>, 6>(p, (hipStream_t)stream);
    // timing-only ablations (wrong results): 64 = A fetched with the TN addresses, 65 = A fragments read the TN way, 66 = both
    if (forced == 64) return launch_gemm_deepp8<GemmCfg<256, 256, 2, 2, 8, true>, true, 1>(p, (hipStream_t)stream);
    if (forced == 65) return launch_gemm_deepp8<GemmCfg<256, 256, 2, 2, 8, true>, true, 2>(p, (hipStream_t)stream);
    if (forced == 66) return launch_gemm_deepp8<GemmCfg<256, 256, 2, 2, 8, true>, true, 3>(p, (hipStream_t)stream);
#endif
    return launch_gemm_deepp8<GemmCfg<256, 256, 2, 2, 8, true>, true>(p, (hipStream_t)stream);
  }
  TransposeParams t;
  t.in = (const uint8_t*)A; t.out = (uint8_t*)workspace; t.K = (int)K; t.M = (int)M;
  hipLaunchKernelGGL(transpose_u8_kernel<>, dim3((unsigned)cdiv(M, 128), (unsigned)cdiv(K, 128)), dim3(256), 0, (hipStream_t)stream, t);
  if (int rc = check_launch("transpose_u8_kernel")) return rc;
  return gemm_mx<8>(name, workspace, B, A_sf, B_sf, alpha, D, M, N, K, stream, nullptr, 0, a_fmt);
}

int qutlass_amd_matmul_mxf8_bf16_nn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                    const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  return mxf8_nn_impl(A, B, A_sf, B_sf, alpha, D, M, N, K, QAMD_FP8_E4M3, workspace, workspace_bytes, stream);
}

static int check_fp8_formats(const char* name, int a_format, int b_format) {
  if (a_format != QAMD_FP8_E4M3 && a_format != QAMD_FP8_E5M2) return fail(QAMD_ERR_INVALID, "%s: invalid a_format %d", name, a_format);
  if (b_format != QAMD_FP8_E4M3) return fail(QAMD_ERR_INVALID, "%s: b_format must be QAMD_FP8_E4M3 (got %d)", name, b_format);
  return QAMD_OK;
}

int qutlass_amd_matmul_mxf8_bf16_tn_fmt(const void* A, const void* B, const void* A_sf, const void* B_sf, const float* alpha, void* D, int64_t M,
                                        int64_t N, int64_t K, int a_format, int b_format, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_fp8_formats("matmul_mxf8_bf16_tn", a_format, b_format)) return rc;
  return gemm_mx<8>("matmul_mxf8_bf16_tn", A, B, A_sf, B_sf, alpha, D, M, N, K, stream, workspace, workspace_bytes, a_format);
}

int qutlass_amd_matmul_mxf8_bf16_nn_fmt(const void* A, const void* B, const void* A_sf, const void* B_sf, const float* alpha, void* D, int64_t M,
                                        int64_t N, int64_t K, int a_format, int b_format, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_fp8_formats("matmul_mxf8_bf16_nn", a_format, b_format)) return rc;
  return mxf8_nn_impl(A, B, A_sf, B_sf, alpha, D, M, N, K, a_format, workspace, workspace_bytes, stream);
}

// ldd: row stride of D in elements when the call covers a column range of a wider output (0 = N)
// [r3] split-K scratch of matmul_nvf4_bf16_tn: bytes for the fp32 partials ws[splits][M][N] of the planned split, 0 when the shape does not split
static int64_t nvf4_ws_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K < 32 || N % 4 || N * (K / 2) >= (1ll << 31) || M * (K / 2) >= (1ll << 31)) return 0;
#if QAMD_BENCH
  if (opt_nvf4_variant() >= 100) return 8 * M * N * 4;   // lab: room for any forced split
  if (opt_nvf4_variant() >= 42 && opt_nvf4_variant() <= 45) return qamd::nvpk_ws_bytes(chip_cus());   // lab: forced persistent kernel
#endif
  const qamd::NvPlan pl = qamd::nvf4_plan(M, N, K, chip_cus(), true);
  if (pl.splits > 1) return (int64_t)pl.splits * M * N * 4;
  return 0;   // (256x256 tiles run the persistent kernel in balanced whole-tile rounds: no scratch; its stream-K form is lab-only, gemm_nvf4_pk.hip.h)
}

static int nvf4_impl(const void* A, const void* B, const void* A_sf, const void* B_sf, const float* alpha, void* D, int64_t M, int64_t N,
                     int64_t K, int64_t ldd, void* stream, void* ws = nullptr, int64_t ws_bytes = 0) {
  const char* name = "matmul_nvf4_bf16_tn";
  if (!A || !B || !A_sf || !B_sf || !alpha || !D) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)A_sf | (uintptr_t)B_sf | (uintptr_t)D) % 16)   // [r5] as in gemm_mx
    return fail(QAMD_ERR_INVALID, "%s: A, B, the scale operands and D must be 16-byte aligned", name);
  if (M <= 0 || N <= 0) return fail(QAMD_ERR_INVALID, "%s: M and N must be positive", name);
  if (K < 16 || K % 32) return fail(QAMD_ERR_INVALID, "%s: K must be a positive multiple of 32 (got %lld)", name, (long long)K);
  if (N % 8) return fail(QAMD_ERR_INVALID, "%s: N must be a multiple of 8 (got %lld)", name, (long long)N);
  if (M >= (1ll << 31) || N >= (1ll << 31) || M * N >= (1ll << 40)) return fail(QAMD_ERR_INVALID, "%s: an output of 2^40 elements is not supported", name);
  if (ldd == 0) ldd = N;
  const int64_t rowbytes = K / 2, CB = cdiv(K / 16, 4);
  // The kernels address an operand through 32-bit buffer-descriptor offsets (< 2 GiB); the reference hands 64-bit strides to CUTLASS
  // (qutlass/csrc/gemm.cu:90-143).  A larger operand runs as ranges of whole 256-row tiles with rebased operand / scale / D pointers
  // (gemm_mx does the same): every output element is computed by exactly one launch, in the same K order.
  if (N * rowbytes >= (1ll << 31)) {
    const int64_t cols = ((1ll << 31) - 1) / rowbytes / 256 * 256;
    if (cols < 256) return fail(QAMD_ERR_INVALID, "%s: K too large for a 256-column range of B to stay below 2 GiB", name);
    for (int64_t c0 = 0; c0 < N; c0 += cols)
      if (int rc = nvf4_impl(A, (const uint8_t*)B + c0 * rowbytes, A_sf, (const uint8_t*)B_sf + (c0 / 128) * CB * 512, alpha, (uint16_t*)D + c0, M,
                             std::min(cols, N - c0), K, ldd, stream))
        return rc;
    return QAMD_OK;
  }
  if (M * rowbytes >= (1ll << 31)) {
    const int64_t rows = ((1ll << 31) - 1) / rowbytes / 256 * 256;
    if (rows < 256) return fail(QAMD_ERR_INVALID, "%s: K too large for a 256-row range of A to stay below 2 GiB", name);
    for (int64_t r0 = 0; r0 < M; r0 += rows)
      if (int rc = nvf4_impl((const uint8_t*)A + r0 * rowbytes, B, (const uint8_t*)A_sf + (r0 / 128) * CB * 512, B_sf, alpha, (uint16_t*)D + r0 * ldd,
                             std::min(rows, M - r0), N, K, ldd, stream))
        return rc;
    return QAMD_OK;
  }
  NvGemmParams p;
  p.A = (const uint8_t*)A; p.B = (const uint8_t*)B; p.SFA = (const uint8_t*)A_sf; p.SFB = (const uint8_t*)B_sf;
  p.alpha = alpha; p.D = (uint16_t*)D; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ldd = (int)ldd;
  p.a_bytes = (uint32_t)(M * rowbytes); p.b_bytes = (uint32_t)(N * rowbytes);
  p.sfa_bytes = (uint32_t)(cdiv(M, 128) * CB * 512); p.sfb_bytes = (uint32_t)(cdiv(N, 128) * CB * 512);
  p.dbg = opt_dbg();
  // split-K only with caller scratch of the size nvf4_ws_bytes reports for this very shape (a smaller or absent workspace runs the single pass)
  const int64_t need = ws ? nvf4_ws_bytes(M, N, K) : 0;
  p.ws = (need > 0 && ws_bytes >= need && ldd == N) ? (float*)ws : nullptr;
  p.splits = 1; p.kt_per = 0;
  p.sk_tiles = 0; p.sk_ws = nullptr; p.sk_flags = nullptr; p.sk_tag = next_launch_tag();
  int splits = 1;
  if (launch_nvf4_host(p, (hipStream_t)stream, opt_nvf4_variant(), &splits)) return fail(QAMD_ERR_INVALID, "%s: unknown nvf4_variant %d", name, opt_nvf4_variant());
  if (int rc = check_launch(name)) return rc;
  if (splits <= 1) return QAMD_OK;
  if ((int64_t)splits * M * N * 4 > ws_bytes) return fail(QAMD_ERR_INVALID, "%s: internal: split count %d exceeds the workspace", name, splits);
  // second launch: sum the K ranges in fixed z order, alpha, bf16 (deterministic; gemm_mx.hip.h)
  const int64_t quads = M * (N / 4);
  const int grid = (int)std::min<int64_t>(cdiv(quads, 256), 2048);
  switch (splits) {
#define QAMD_RED(S_) case S_: hipLaunchKernelGGL(splitk_reduce_kernel<S_>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)ws, (uint16_t*)D, alpha, (int)M, (int)N, (int)ldd); break;
    QAMD_RED(2) QAMD_RED(3) QAMD_RED(4) QAMD_RED(5) QAMD_RED(6) QAMD_RED(7) QAMD_RED(8)
#undef QAMD_RED
    default: return fail(QAMD_ERR_INVALID, "%s: unsupported split count %d", name, splits);
  }
  return check_launch("splitk_reduce_kernel");
}

int qutlass_amd_matmul_nvf4_bf16_tn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                    const float* alpha, void* D, int64_t M, int64_t N, int64_t K, void* stream) {
  return nvf4_impl(A, B, A_sf, B_sf, alpha, D, M, N, K, 0, stream);
}

int64_t qutlass_amd_nvf4_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) { return nvf4_ws_bytes(M, N, K); }

int qutlass_amd_matmul_nvf4_bf16_tn_ws(const void* A, const void* B, const void* A_sf, const void* B_sf, const float* alpha, void* D, int64_t M,
                                       int64_t N, int64_t K, void* workspace, int64_t workspace_bytes, void* stream) {
  if (workspace_bytes < 0 || (workspace_bytes > 0 && !workspace)) return fail(QAMD_ERR_INVALID, "matmul_nvf4_bf16_tn: invalid workspace");
  if ((uintptr_t)workspace % 16) return fail(QAMD_ERR_INVALID, "matmul_nvf4_bf16_tn: the workspace must be 16-byte aligned");
  return nvf4_impl(A, B, A_sf, B_sf, alpha, D, M, N, K, 0, stream, workspace, workspace_bytes);
}

// sf_rows / k: logical 2-D shape of x (rows of k elements) for the blocked-scale variants; k == 0: flat scales (the reference's contract)
static int fused_quantize_mx_impl(const char* name, const void* x, const void* h, int rot, int64_t numel, int64_t k, int method,
                                  void* out_e2m1, void* out_e8m0, void* out_mask, void* stream) {
  if (!x || !h || !out_e2m1 || !out_e8m0) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (rot != 32 && rot != 64 && rot != 128)
    return fail(QAMD_ERR_INVALID, "%s: Unsupported rotation size %d; expected %s.", name, rot, out_mask ? "32" : "32, 64, or 128");
  if (numel <= 0 || numel % rot) return fail(QAMD_ERR_INVALID, "%s: A must be divisible by %d", name, rot);
  if (numel * 2 >= (1ll << 32)) return fail(QAMD_ERR_INVALID, "%s: more than 2^31 elements is not supported", name);
  if (method != QAMD_METHOD_QUEST && method != QAMD_METHOD_ABSMAX) return fail(QAMD_ERR_INVALID, "%s: invalid method %d", name, method);
  if (out_mask && method != QAMD_METHOD_QUEST) return fail(QAMD_ERR_INVALID, "%s: the clip mask is only defined for method quest", name);
  if (k && (k % rot || numel % k)) return fail(QAMD_ERR_INVALID, "%s: the row length %lld must be a multiple of the rotation size %d and divide numel", name, (long long)k, rot);
  // (rot >= 64 stages H with 16-byte vector loads, quantize.hip.h: an offset view of a larger tensor may be 2-byte aligned only -- rejected rather than left to
  //  the device's unaligned-access mode; torch allocations are 256-byte aligned)
  if (rot >= 64 && (uintptr_t)h % 16) return fail(QAMD_ERR_INVALID, "%s: the rotation matrix must be 16-byte aligned for rotation sizes >= 64", name);
  QuantParams p;
  p.x = (const uint16_t*)x; p.h = (const uint16_t*)h; p.out = (uint8_t*)out_e2m1; p.out_sf = (uint8_t*)out_e8m0;
  p.out_mask = (uint32_t*)out_mask; p.global_scale = nullptr; p.numel = numel;
  p.ntiles = (int)cdiv(numel, (int64_t)rot * 32);
  p.sf_rows = k ? (int)(numel / k) : 0; p.sf_cols = k ? (int)(k / 32) : 0;
  int grid = quant_grid(p.ntiles, rot);
  if (k) grid = blocked_pad_grid(grid, p.sf_rows, p.sf_cols);
  hipStream_t s = (hipStream_t)stream;
  if (k) {
    if (out_mask) return dispatch_rot<false, METHOD_QUEST, true, true>(rot, p, s, grid, name);
    if (method == QAMD_METHOD_QUEST) return dispatch_rot<false, METHOD_QUEST, false, true>(rot, p, s, grid, name);
    return dispatch_rot<false, METHOD_ABSMAX, false, true>(rot, p, s, grid, name);
  }
  if (out_mask) return dispatch_rot<false, METHOD_QUEST, true>(rot, p, s, grid, name);
  if (method == QAMD_METHOD_QUEST) return dispatch_rot<false, METHOD_QUEST, false>(rot, p, s, grid, name);
  return dispatch_rot<false, METHOD_ABSMAX, false>(rot, p, s, grid, name);
}

int qutlass_amd_fused_quantize_mx(const void* x, const void* h, int rot, int64_t numel, int method,
                                  void* out_e2m1, void* out_e8m0, void* out_mask, void* stream) {
  return fused_quantize_mx_impl("fusedQuantizeMx", x, h, rot, numel, 0, method, out_e2m1, out_e8m0, out_mask, stream);
}

int qutlass_amd_fused_quantize_mx_blocked(const void* x, const void* h, int rot, int64_t rows, int64_t k, int method,
                                          void* out_e2m1, void* out_e8m0_blocked, void* out_mask, void* stream) {
  const char* name = "fusedQuantizeMxBlocked";
  if (rows <= 0 || k <= 0 || rows >= (1ll << 31) || k >= (1ll << 31)) return fail(QAMD_ERR_INVALID, "%s: bad shape (%lld, %lld)", name, (long long)rows, (long long)k);
  return fused_quantize_mx_impl(name, x, h, rot, rows * k, k, method, out_e2m1, out_e8m0_blocked, out_mask, stream);
}

static int fused_quantize_nv_impl(const char* name, const void* x, const void* h, int rot, int64_t numel, int64_t k, int method,
                                  const float* global_scale, void* out_e2m1, void* out_e4m3, void* stream) {
  if (!x || !h || !out_e2m1 || !out_e4m3 || !global_scale) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (rot != 16 && rot != 32 && rot != 64 && rot != 128)
    return fail(QAMD_ERR_INVALID, "%s: Unsupported rotation size %d; expected 16, 32, 64, or 128.", name, rot);
  if (numel <= 0 || numel % rot) return fail(QAMD_ERR_INVALID, "%s: A must be divisible by %d", name, rot);
  if (numel * 2 >= (1ll << 32)) return fail(QAMD_ERR_INVALID, "%s: more than 2^31 elements is not supported", name);
  if (method != QAMD_METHOD_QUEST && method != QAMD_METHOD_ABSMAX) return fail(QAMD_ERR_INVALID, "%s: invalid method %d", name, method);
  const int rp = rot < 32 ? 32 : rot;
  if (k && (k % rp || numel % k)) return fail(QAMD_ERR_INVALID, "%s: the row length %lld must be a multiple of %d and divide numel", name, (long long)k, rp);
  if (rot >= 64 && (uintptr_t)h % 16) return fail(QAMD_ERR_INVALID, "%s: the rotation matrix must be 16-byte aligned for rotation sizes >= 64", name);
  QuantParams p;
  p.x = (const uint16_t*)x; p.h = (const uint16_t*)h; p.out = (uint8_t*)out_e2m1; p.out_sf = (uint8_t*)out_e4m3;
  p.out_mask = nullptr; p.global_scale = global_scale; p.numel = numel;
  p.ntiles = (int)cdiv(numel, (int64_t)rp * 32);
  p.sf_rows = k ? (int)(numel / k) : 0; p.sf_cols = k ? (int)(k / 16) : 0;
  int grid = quant_grid(p.ntiles, rot);
  if (k) grid = blocked_pad_grid(grid, p.sf_rows, p.sf_cols);
  hipStream_t s = (hipStream_t)stream;
  if (k) {
    if (method == QAMD_METHOD_QUEST) return dispatch_rot<true, METHOD_QUEST, false, true>(rot, p, s, grid, name);
    return dispatch_rot<true, METHOD_ABSMAX, false, true>(rot, p, s, grid, name);
  }
  if (method == QAMD_METHOD_QUEST) return dispatch_rot<true, METHOD_QUEST, false>(rot, p, s, grid, name);
  return dispatch_rot<true, METHOD_ABSMAX, false>(rot, p, s, grid, name);
}

int qutlass_amd_fused_quantize_nv(const void* x, const void* h, int rot, int64_t numel, int method,
                                  const float* global_scale, void* out_e2m1, void* out_e4m3, void* stream) {
  return fused_quantize_nv_impl("fusedQuantizeNv", x, h, rot, numel, 0, method, global_scale, out_e2m1, out_e4m3, stream);
}

int qutlass_amd_fused_quantize_nv_blocked(const void* x, const void* h, int rot, int64_t rows, int64_t k, int method,
                                          const float* global_scale, void* out_e2m1, void* out_e4m3_blocked, void* stream) {
  const char* name = "fusedQuantizeNvBlocked";
  if (rows <= 0 || k <= 0 || rows >= (1ll << 31) || k >= (1ll << 31)) return fail(QAMD_ERR_INVALID, "%s: bad shape (%lld, %lld)", name, (long long)rows, (long long)k);
  return fused_quantize_nv_impl(name, x, h, rot, rows * k, k, method, global_scale, out_e2m1, out_e4m3_blocked, stream);
}

// How many launches should the activation path y = Q(x h) W^T of one linear layer take (the rule behind qutlass_amd.fused_quantize_matmul_mxf4_bf16_tn;
// reference flow: qutlass/__init__.py:149-180 -> qutlass/utils.py:160-193 -> qutlass/__init__.py:34-76 = three launches)?
//   1  ONE launch, qutlass_amd_fused_quantize_matmul_mxf4_bf16_tn (gemm_mx_fusedq.hip.h: the small-batch GEMM rotates and quantises its own A operand).  It repeats
//      the rotate + quantize chains of its K slices in every workgroup -- ceil(K / 2048) x (1, 2, 4 for M <= 4, 8, 16) chains per wave -- and wins while that stays
//      short: [r6] M <= 4: K <= 4096, M <= 8: K <= 2048 (see below), R = 32, and a weight the small-batch GEMM handles (N < 32 x CUs = 8192 on an MI355X).
//      M = 1 / 8 / 16 at N = K = 4096: 5.0 / 6.0 / 7.6 us against 7.3 / 7.9 / 8.2 us for two launches and 9.0 / 9.5 / 9.9 for three (GEMM alone 4.6 / 4.9 / 5.3);
//      M = 32 or K = 14336 lose (14.2 vs 9.0 us, 12.6 vs 10.7 us).
//   2  TWO launches everywhere else: qutlass_amd_fused_quantize_mx_blocked (scales written in the to_blocked layout) + the GEMM.  The blocked quantizer costs
//      0.2 ... 0.8 us more than the flat one and saves the 1.9 ... 3.2 us to_blocked launch: 2.2 vs 4.5 us at 256 x 4096, 7.3 vs 10.4 at 4096^2, 27.9 vs 31.4 at
//      8192^2, 23.4 vs 26.5 at 4096 x 14336 -- device time, every size measured (profiles/ab_blocked_quant_r3y.txt); with the GEMM behind it: 4096 x 14336 x 4096
//      123.2 vs 124.5 us (profiles/bench_r4c.json).  It never returns 3: the reference's three-launch flow stays available as fusedQuantizeMx + to_blocked + matmul.
// Thresholds scale with the CU count of the current device.  No GPU work; the dry-run hook describes a 256-CU part.
int qutlass_amd_activation_path_launches(int64_t M, int64_t N, int64_t K, int rot) {
  if (M <= 0 || N <= 0 || K <= 0) return 2;
  const int cus = chip_cus();
  if (M > 16 || rot != 32 || N >= 32ll * cus || K % 128) return 2;
  // [r6] re-taken after the decode forms made the GEMM of the two-launch path faster (N = K = 4096, M <= 16: 4.6-5.3 -> 2.85-3.0 us; tools/calib_actpath.py,
  // profiles/calib_actpath_r7.txt): two launches now take 5.2-5.7 us at N = K = 4096 where the one-launch kernel takes 5.1 / 5.3 / 6.2 / 8.1 (M = 1 / 4 / 8 / 16), 6.4 against
  // 8.4-8.8 at 4096 x 8192 -- one launch keeps M <= 4 with K <= 4096 (4096 x 6144: a tie) and M <= 8 with K <= 2048 (3.8-4.4 against 4.6-4.9); rounds 3-5: M <= 4: K <= 8192,
  // M <= 8: K <= 6144, M <= 16: K <= 4096
  return K <= (M <= 4 ? 4096 : M <= 8 ? 2048 : 0) ? 1 : 2;
}

// decode-time activation path in one launch (gemm_mx_fusedq.hip.h): D = alpha * Q(x . h) (B . SFB)^T for M <= 32
int qutlass_amd_fused_quantize_matmul_mxf4_bf16_tn(const void* x, const void* h, int rot, int method, const void* B, const void* B_sf,
                                                   const float* alpha, void* D, int64_t M, int64_t N, int64_t K, void* stream) {
  const char* name = "fusedQuantizeMatmulMxf4";
  if (!x || !h || !B || !B_sf || !alpha || !D) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (rot != 32) return fail(QAMD_ERR_INVALID, "%s: Unsupported rotation size %d; expected 32.", name, rot);
  if (method != QAMD_METHOD_QUEST && method != QAMD_METHOD_ABSMAX) return fail(QAMD_ERR_INVALID, "%s: invalid method %d", name, method);
  if (M <= 0 || M > 32) return fail(QAMD_ERR_INVALID, "%s: M must be in 1..32 (got %lld); larger batches take fusedQuantizeMxBlocked + matmul_mxf4_bf16_tn", name, (long long)M);
  if (N <= 0 || N % 8) return fail(QAMD_ERR_INVALID, "%s: N must be a positive multiple of 8 (got %lld)", name, (long long)N);
  if (K < 128 || K % 128) return fail(QAMD_ERR_INVALID, "%s: K must be a positive multiple of 128 (got %lld)", name, (long long)K);
  if (N * (K / 2) >= (1ll << 31) || M * K * 2 >= (1ll << 31) || cdiv(N, 32) >= (1ll << 31)) return fail(QAMD_ERR_INVALID, "%s: operand larger than 2 GiB is not supported", name);
#if QAMD_BENCH
  // [r6] lab ("gemm_variant" 580): the hand-off form of the one-launch layer (gemm_mx_os16_fq_kernel: the quantizer's body on the first workgroups, the decode form's K walk
  // behind one arrival counter) on a scratch buffer the LAB library allocates once -- an experiment's plumbing, not an API (the product would take caller scratch)
  if (opt_gemm_variant() == 580 && M <= 16 && K <= 8192 && os16_tn(N) != 0) {
    static void* scratch = nullptr;
    if (!scratch && hipMalloc(&scratch, 1 << 20) != hipSuccess) return fail(QAMD_ERR_HIP, "%s: lab scratch", name);
    const int64_t CB = cdiv(K / 32, 4);
    FqOsParams P;
    P.q.x = (const uint16_t*)x; P.q.h = (const uint16_t*)h; P.q.out = (uint8_t*)scratch; P.q.out_sf = (uint8_t*)scratch + (256 << 10); P.q.out_mask = nullptr; P.q.global_scale = nullptr;
    P.q.numel = M * K; P.q.ntiles = (int)cdiv(M * K, 1024); P.q.sf_rows = (int)M; P.q.sf_cols = (int)(K / 32);
    GemmParams& g = P.g;
    g.A = (const uint8_t*)scratch; g.B = (const uint8_t*)B; g.SFA = (const uint8_t*)scratch + (256 << 10); g.SFB = (const uint8_t*)B_sf; g.alpha = alpha; g.D = (uint16_t*)D;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.ldd = (int)N;
    g.a_bytes = (uint32_t)(M * (K / 2)); g.b_bytes = (uint32_t)(N * (K / 2)); g.sfa_bytes = (uint32_t)(CB * 512); g.sfb_bytes = (uint32_t)(cdiv(N, 128) * CB * 512);
    g.pp_shift = 0; g.pp_flags = 0; g.dbg = nullptr; g.ws = nullptr; g.splits = 1; g.ctr = nullptr; g.tag = 0; g.sk_tiles = 0;
    const int tn = os16_tn(N);
    g.tiles_m = 1; g.tiles_n = (int)cdiv(N, tn); g.raster_magic = 0;
    P.flag = (unsigned long long*)((uint8_t*)scratch + (512 << 10));
    P.tag = (next_launch_tag() & ((1ull << 56) - 1)) << 8;
    P.nq = (int)std::min<int64_t>(cdiv(P.q.ntiles, 4), g.tiles_n);
    const int64_t KT = cdiv(K, 256);
    const dim3 grid(g.tiles_n), block(256);
    hipStream_t st = (hipStream_t)stream;
#define QAMD_FQOS(TN_, SPW_)                                                                                                                     \
    do {                                                                                                                                         \
      if (method == QAMD_METHOD_ABSMAX) hipLaunchKernelGGL((gemm_mx_os16_fq_kernel<Os16Cfg<SPW_, 4, 0, TN_>, METHOD_ABSMAX>), grid, block, 0, st, P); \
      else hipLaunchKernelGGL((gemm_mx_os16_fq_kernel<Os16Cfg<SPW_, 4, 0, TN_>, METHOD_QUEST>), grid, block, 0, st, P);                              \
    } while (0)
    if (tn == 16) { if (KT <= 16) QAMD_FQOS(16, 4); else QAMD_FQOS(16, 8); }
    else if (tn == 32 && KT <= 16) QAMD_FQOS(32, 4);
    else if (tn == 48 && KT <= 16) QAMD_FQOS(48, 4);
    else if (tn == 56 && KT <= 16) QAMD_FQOS(56, 4);
    else return fail(QAMD_ERR_INVALID, "%s: lab variant 580 has no instantiation for this shape", name);
#undef QAMD_FQOS
    return check_launch("gemm_mx_os16_fq_kernel");
  }
#endif
  FusedQParams p;
  p.x = (const uint16_t*)x; p.h = (const uint16_t*)h; p.B = (const uint8_t*)B; p.SFB = (const uint8_t*)B_sf; p.alpha = alpha; p.D = (uint16_t*)D;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.x_bytes = (uint32_t)(M * K * 2); p.b_bytes = (uint32_t)(N * (K / 2)); p.sfb_bytes = (uint32_t)(cdiv(N, 128) * cdiv(K / 32, 4) * 512);
  const dim3 grid((unsigned)cdiv(N, 32), 1), block(512);
  hipStream_t s = (hipStream_t)stream;
  const bool hw = opt_hw_fp4();
  // rows per rotation tile: the smallest of 4 / 8 / 16 / 32 that holds the batch (fewer rows = more scale groups per tile = fewer
  // rotate + quantize chains per K segment, gemm_mx_fusedq.hip.h)
  const int vr = M <= 4 ? 4 : M <= 8 ? 8 : M <= 16 ? 16 : 32;
#define QAMD_FQ(METH_, HW_, VR_) hipLaunchKernelGGL((gemm_mx_fusedq_kernel<METH_, HW_, VR_>), grid, block, 0, s, p)
#define QAMD_FQ_VR(METH_, HW_) \
  switch (vr) { case 4: QAMD_FQ(METH_, HW_, 4); break; case 8: QAMD_FQ(METH_, HW_, 8); break; case 16: QAMD_FQ(METH_, HW_, 16); break; default: QAMD_FQ(METH_, HW_, 32); }
#if QAMD_BENCH
  if (!hw) {
    if (method == QAMD_METHOD_ABSMAX) { QAMD_FQ_VR(METHOD_ABSMAX, false) } else { QAMD_FQ_VR(METHOD_QUEST, false) }
  } else
#endif
  if (method == QAMD_METHOD_ABSMAX) { QAMD_FQ_VR(METHOD_ABSMAX, true) } else { QAMD_FQ_VR(METHOD_QUEST, true) }
  (void)hw;
#undef QAMD_FQ_VR
#undef QAMD_FQ
  return check_launch("gemm_mx_fusedq_kernel");
}

// which backward_t / backward_qt kernel (see launch_bwd_quant), from the A/B on one box (profiles/ab_bwd_r4k_variants.txt; nu = units of 8 groups x
// 64 m): the wave-owned kernels need a few units per wave to amortise their pipeline fill, below that the round-3 kernel (8 waves share a unit)
// stays.  QT: 64-byte segments with 16 waves per CU (4096^2 cold 10.2 -> 8.5 us, 8192^2 30.9 -> 26.6); T: whole lines with 8 waves per CU
// (8192^2 cold 38.0 -> 36.5, 2048 x 14336 17.7 -> 15.2; the 64-byte form ties with the round-3 kernel there).  In the product build the product
// kernels are the only instantiations: QT never takes 3, T never 2.  Lab: option "bwd_variant" (low 4 bits) forces 1 / 2 / 3.
// [r5] QT, large inputs with M % 128 == 0: 5 = the wave-owned kernel fed through a shared whole-line ring (bwd_qt_ring_kernel).  Cold it wins from ~12 units per CU
// (profiles/ab_bwd_r5x_ring_threshold.txt: 14336 x 4096 22.1 -> 20.6 us, 8192^2 26.1 -> 22.4, 8192 x 16384 55.1 -> 43.5; 11008 x 4096 ties, 6144^2 15.1 -> 15.5), warm
// (input in the MALL) the plain kernel stays ahead until ~32 per CU -- the rule goes by the cold numbers, as every streaming-op figure of the design record does.
static int bwd_kernel_rule(bool qt, int64_t nu, int64_t cu, bool ring_ok = false) {
  if (qt) return (ring_ok && nu >= 12 * cu) ? 5 : nu >= 3 * cu ? 2 : 1;
  return nu >= 6 * cu ? 3 : 1;
}
static int bwd_kernel_choice(bool qt, int64_t nu, bool ring_ok = false) {
  const int v = opt_bwd_variant() & 15;
  if (v >= 1 && v <= 9) return v;
  return bwd_kernel_rule(qt, nu, chip_cus(), ring_ok);
}
// column tiles (of 128) per workgroup of backward_bf16_square_double_mxfp8: 4 (16 waves, 16-byte row-scale pieces) when n allows and every CU still gets a workgroup
static int sq_column_tiles_rule(int64_t m_pad, int64_t n, int64_t cu) { return (n % 512 == 0 && (m_pad / 128) * (n / 512) >= cu) ? 4 : 1; }

int qutlass_amd_backward_t_bf16(const void* x, const void* h, int64_t B, int64_t N, int64_t M, void* out_e2m1,
                                void* out_e8m0, void* stream) {
  const char* name = "backward_t_bf16";
  if (!x || !h || !out_e2m1 || !out_e8m0) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (B <= 0 || N <= 0 || M <= 0 || N % 32 || M % 8)
    return fail(QAMD_ERR_INVALID, "%s: need N %% 32 == 0 and M %% 8 == 0 (got B=%lld N=%lld M=%lld)", name, (long long)B, (long long)N, (long long)M);
  if (B * N * M >= (1ll << 40) || M >= (1ll << 24) || N >= (1ll << 26)) return fail(QAMD_ERR_INVALID, "%s: tensor too large", name);   // 32 input rows / 64 output rows < 2^31 bytes
  BwdTParams p{};
  p.x = (const uint16_t*)x; p.h = (const uint16_t*)h; p.out = (uint8_t*)out_e2m1; p.out_sf = (uint8_t*)out_e8m0;
  p.B = (int)B; p.N = (int)N; p.M = (int)M; p.tiles_m = (int)cdiv(M, 64);
  if (B * (N / 32) * p.tiles_m >= (1ll << 31) - 65536) return fail(QAMD_ERR_INVALID, "%s: tensor too large", name);
  const int64_t ntw = B * p.tiles_m * cdiv(N / 32, 8);   // units: 8 scale groups (256 n) x 64 m = 64 whole output lines
  const int which0 = bwd_kernel_choice(false, ntw);
  const int which = (which0 >= 4 && which0 != 9) ? bwd_kernel_rule(false, ntw, chip_cus()) : which0;   // (4 .. 8 = QT-only kernels)
  p.abl = opt_bwd_variant() >> 4;
  const int64_t nuw = B * p.tiles_m * cdiv(N / 32, which == 3 ? 8 : which == 9 ? 2 : 4);   // wave units
  const int grid = which == 1 ? (int)std::min<int64_t>(ntw, chip_cus() * 2) : (int)std::min<int64_t>(cdiv(nuw, 4), chip_cus() * (which == 3 ? 2 : which == 9 ? 4 : 3));
  return launch_bwd_quant(p, false, which, opt_hw_fp4(), grid, (hipStream_t)stream);
}

int qutlass_amd_backward_qt_bf16(const void* x_e2m1, const void* x_e8m0, const void* h, const float* alpha, int64_t B,
                                 int64_t N, int64_t M, void* out_e2m1, void* out_e8m0, void* stream) {
  const char* name = "backward_qt_bf16";
  if (!x_e2m1 || !x_e8m0 || !h || !alpha || !out_e2m1 || !out_e8m0) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (B <= 0 || N <= 0 || M <= 0 || N % 32 || M % 32)
    return fail(QAMD_ERR_INVALID, "%s: need N %% 32 == 0 and M %% 32 == 0 (got B=%lld N=%lld M=%lld)", name, (long long)B, (long long)N, (long long)M);
  if (B * N * M >= (1ll << 40) || M >= (1ll << 24) || N >= (1ll << 26)) return fail(QAMD_ERR_INVALID, "%s: tensor too large", name);   // 32 input rows / 64 output rows < 2^31 bytes
  BwdTParams p{};
  p.xq = (const uint8_t*)x_e2m1; p.xs = (const uint8_t*)x_e8m0; p.h = (const uint16_t*)h; p.alpha = alpha;
  p.out = (uint8_t*)out_e2m1; p.out_sf = (uint8_t*)out_e8m0;
  p.B = (int)B; p.N = (int)N; p.M = (int)M; p.tiles_m = (int)cdiv(M, 64);
  if (B * (N / 32) * p.tiles_m >= (1ll << 31) - 65536) return fail(QAMD_ERR_INVALID, "%s: tensor too large", name);
  // virtual tiles: units of 4 sibling m-tiles (bwd_quant_t_kernel's tile order); the grid is a multiple of 32 workgroups (4 siblings x 8 XCDs)
  const int64_t ntw = B * cdiv(p.tiles_m, 4) * 4 * cdiv(N / 32, 8);
  // 49 KB of LDS per workgroup: three fit a CU.  A round of three per CU takes ~1.24x a round of two (8192^2: 6 rounds 20.7 us against 8 rounds
  // 22.2 us; 4096^2, 2 rounds either way: 9.3 against 8.5 us): three when that wins the round count
  const int64_t cu = chip_cus();
  const int per_cu = 1.24 * (double)cdiv(ntw, 3 * cu) < (double)cdiv(ntw, 2 * cu) ? 3 : 2;
  const int grid = (int)std::max<int64_t>(32, std::min<int64_t>(cdiv(ntw, 32) * 32, cu * per_cu / 32 * 32));
  // [r4] wave-owned output segments: the four waves of a workgroup take four consecutive m-tiles -- the siblings that share QT's 128-byte input lines
  const int64_t nu = B * p.tiles_m * cdiv(N / 32, 8);
  // (the ring kernel fetches by LDS-DMA: 16-byte pieces of the codes, dword pieces of the scale bytes -- an offset view that breaks either alignment takes the other kernels)
  const int which = bwd_kernel_choice(true, nu, M % 128 == 0 && (uintptr_t)x_e2m1 % 16 == 0 && (uintptr_t)x_e8m0 % 4 == 0);
  p.abl = opt_bwd_variant() >> 4;
  const int64_t nuw = B * p.tiles_m * cdiv(N / 32, which == 3 ? 8 : which == 9 ? 2 : 4);
  const int gridw = (int)std::min<int64_t>(cdiv(nuw, 4), cu * (which == 3 ? 2 : which == 9 ? 5 : 4));
  if (which >= 5 && which <= 8) {   // [r5] ring kernels: units of NG groups x 256 m
    if (M % 128) return fail(QAMD_ERR_INVALID, "%s: the ring kernel needs M %% 128 == 0", name);
    const int ng = (which == 6 || which == 7) ? 8 : 4;
    const int64_t ur = B * cdiv(M, 256) * cdiv(N / 32, ng);
    return launch_bwd_quant(p, true, which, opt_hw_fp4(), (int)std::min<int64_t>(ur, std::max<int64_t>(8, cu * (ng == 8 ? 2 : 3) / 8 * 8)), (hipStream_t)stream);   // (>= 8: a part with fewer than 3 visible CUs must not get an empty grid)
  }
  if (which == 4) {   // [r5] panel kernel: units of 8 groups x 256 m, two workgroups per CU (persistent grids step by a multiple of the 8 XCDs)
    if (M % 128) return fail(QAMD_ERR_INVALID, "%s: the panel kernel needs M %% 128 == 0", name);
    const int64_t up = B * cdiv(M, 256) * cdiv(N / 32, 8);
    return launch_bwd_quant(p, true, 4, opt_hw_fp4(), (int)std::min<int64_t>(up, std::max<int64_t>(8, cu * 2 / 8 * 8)), (hipStream_t)stream);
  }
  return launch_bwd_quant(p, true, which, opt_hw_fp4(), which == 1 ? grid : gridw, (hipStream_t)stream);
}

int qutlass_amd_backward_bf16_square_double_mxfp8_rows(const void* x, int64_t m, int64_t m_pad, int64_t n, void* y, void* row_scales,
                                                       void* col_scales, void* stream) {
  const char* name = "backward_bf16_square_double_mxfp8";
  if (!x || !y || !row_scales || !col_scales) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (m <= 0 || n <= 0 || n % 128 || m_pad < m || m_pad % 128)
    return fail(QAMD_ERR_INVALID, "%s: need m > 0, n a positive multiple of 128 and the output row extent a multiple of 128 >= m (got m=%lld m_pad=%lld n=%lld)", name,
                (long long)m, (long long)m_pad, (long long)n);
  if (m_pad >= (1ll << 31) || n >= (1ll << 31) || (m_pad / 128) * (n / 128) >= (1ll << 31)) return fail(QAMD_ERR_INVALID, "%s: tensor too large", name);
  SqParams p;
  p.x = (const uint16_t*)x; p.y = (uint8_t*)y; p.row_sf = (uint8_t*)row_scales; p.col_sf = (uint8_t*)col_scales;
  p.m = (int)m; p.n = (int)n; p.m_pad = (int)m_pad; p.abl = opt_bwd_variant() >> 4;
  // [r4] 512 columns per workgroup (16 waves: the row scales leave as 16-byte pieces) when that still gives every CU a workgroup; lab: option
  // "transpose_nc" = 1 forces the 4-wave form, 4 / 8 the 512- / 1024-column forms
  {
    const int64_t rb = m_pad / 128, cu = chip_cus();
    int ct = sq_column_tiles_rule(m_pad, n, cu);   // (1024 columns, two tiles per wave: slower again except warm at 8192^2 -- profiles/ab_sq_abl_r4ae.txt)
#if QAMD_BENCH
    if (opt_transpose_nc() == 1) ct = 1;
    if ((opt_transpose_nc() == 8 && n % 1024 == 0) || (opt_transpose_nc() == 4 && n % 512 == 0)) ct = opt_transpose_nc();
    if (ct == 8) { hipLaunchKernelGGL((bwd_square_double_mxfp8_kernel<4, 2>), dim3((unsigned)(rb * (n / 1024))), dim3(1024), 0, (hipStream_t)stream, p); return check_launch("bwd_square_double_mxfp8_kernel"); }
#endif
    if (ct == 4) hipLaunchKernelGGL((bwd_square_double_mxfp8_kernel<4, 1>), dim3((unsigned)(rb * (n / 512))), dim3(1024), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((bwd_square_double_mxfp8_kernel<1, 1>), dim3((unsigned)(rb * (n / 128))), dim3(256), 0, (hipStream_t)stream, p);
  }
  return check_launch("bwd_square_double_mxfp8_kernel");
}

int qutlass_amd_backward_bf16_square_double_mxfp8(const void* x, int64_t m, int64_t n, void* y, void* row_scales,
                                                  void* col_scales, void* stream) {
  if (m > 0 && m % 128) return fail(QAMD_ERR_INVALID, "backward_bf16_square_double_mxfp8: m and n must be positive multiples of 128 (got m=%lld n=%lld)", (long long)m, (long long)n);
  return qutlass_amd_backward_bf16_square_double_mxfp8_rows(x, m, m, n, y, row_scales, col_scales, stream);
}

int qutlass_amd_mxfp4_transpose_mxfp8_rows(const void* x_fp4, const void* scales, int64_t m, int64_t m_pad, int64_t n, void* y,
                                           void* out_e8m0, void* stream) {
  const char* name = "mxfp4_transpose_mxfp8";
  if (!x_fp4 || !scales || !y || !out_e8m0) return fail(QAMD_ERR_INVALID, "%s: null pointer argument", name);
  if (m <= 0 || n <= 0 || n % 256 || m_pad < m || m_pad % 128)
    return fail(QAMD_ERR_INVALID, "%s: need m > 0, n %% 256 == 0 and the output row extent a multiple of 128 >= m (got m=%lld m_pad=%lld n=%lld)", name, (long long)m,
                (long long)m_pad, (long long)n);
  if (m_pad >= (1ll << 31) || n >= (1ll << 31) || (m_pad / 128) * (n / 128) >= (1ll << 31)) return fail(QAMD_ERR_INVALID, "%s: tensor too large", name);
  TrParams p;
  p.xq = (const uint8_t*)x_fp4; p.xs = (const uint8_t*)scales; p.y = (uint8_t*)y; p.out_sf = (uint8_t*)out_e8m0;
  p.m = (int)m; p.n = (int)n; p.m_pad = (int)m_pad; p.abl = opt_bwd_variant() >> 4;
  // [r4] A wave-owned-lines form of this op (mxfp4_transpose_mxfp8_tw_kernel: persistent waves, no workgroup barrier, whole 128-byte lines or
  // 64-byte segments) is byte-identical and NOT faster (8192^2 cold 30.7 / 35.0 us against 28.4, warm 21.4 / 22.5 against 21.2;
  // profiles/ab_transpose_r4o.txt): the one-shot kernel stays the product, the other lives in the lab build (option "transpose_nc" = 4 / 2).
#if QAMD_BENCH
  const int64_t cu = chip_cus();
  const int tw = (m_pad < (1ll << 25) && (opt_transpose_nc() == 4 || opt_transpose_nc() == 2)) ? opt_transpose_nc() : 0;   // (64 output rows within one buffer window)
  if (opt_transpose_nc() == 256) {
    hipLaunchKernelGGL(mxfp4_transpose_mxfp8_kernel<256>, dim3((unsigned)((m_pad / 128) * (n / 256))), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("mxfp4_transpose_mxfp8_kernel");
  }
  if (tw == 4) {
    const int64_t units = (m_pad / 128) * (n / 64);
    hipLaunchKernelGGL(mxfp4_transpose_mxfp8_tw_kernel<4>, dim3((unsigned)std::min<int64_t>(cdiv(units, 4), cu * 2)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("mxfp4_transpose_mxfp8_tw_kernel");
  }
  if (tw == 2) {
    const int64_t units = (m_pad / 64) * (n / 64);
    hipLaunchKernelGGL(mxfp4_transpose_mxfp8_tw_kernel<2>, dim3((unsigned)std::min<int64_t>(cdiv(units, 4), cu * 3)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("mxfp4_transpose_mxfp8_tw_kernel");
  }
#endif
  // [r4] 256 rows per workgroup (8 waves, 8-byte scale pieces per output row instead of 4) is byte-identical and slower -- the two workgroup barriers then
  // wait for eight waves: 8192^2 cold 28.6 -> 31.5 us (profiles/ab_transpose_r4ak_8wave.txt); lab only: option "transpose_nc" = 3
#if QAMD_BENCH
  if (opt_transpose_nc() == 3 && m_pad % 256 == 0) {
    hipLaunchKernelGGL((mxfp4_transpose_mxfp8_kernel<128, 256>), dim3((unsigned)((m_pad / 256) * (n / 128))), dim3(512), 0, (hipStream_t)stream, p);
    return check_launch("mxfp4_transpose_mxfp8_kernel");
  }
#endif
#if QAMD_BENCH
  if (opt_transpose_nc() >= 5 && opt_transpose_nc() <= 8) {   // [r5] the persistent form with the next tile's rows in flight: 5 / 6 / 7 / 8 = 4 / 3 / 2 / 6 workgroups' worth of grid per CU
    const int64_t tiles = (m_pad / 128) * (n / 128);
    const int per_cu = opt_transpose_nc() == 5 ? 4 : opt_transpose_nc() == 6 ? 3 : opt_transpose_nc() == 7 ? 2 : 6;
    int64_t grid = std::min<int64_t>(tiles, chip_cus() * per_cu);
    if (grid < tiles) grid = grid / 16 * 16;
    hipLaunchKernelGGL((mxfp4_transpose_mxfp8_pp_kernel<128, 128>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("mxfp4_transpose_mxfp8_pp_kernel");
  }
#endif
  hipLaunchKernelGGL((mxfp4_transpose_mxfp8_kernel<128, 128>), dim3((unsigned)((m_pad / 128) * (n / 128))), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("mxfp4_transpose_mxfp8_kernel");
}

int qutlass_amd_mxfp4_transpose_mxfp8(const void* x_fp4, const void* scales, int64_t m, int64_t n, void* y,
                                      void* out_e8m0, void* stream) {
  if (m > 0 && m % 128) return fail(QAMD_ERR_INVALID, "mxfp4_transpose_mxfp8: need m %% 128 == 0 and n %% 256 == 0 (got m=%lld n=%lld)", (long long)m, (long long)n);
  return qutlass_amd_mxfp4_transpose_mxfp8_rows(x_fp4, scales, m, m, n, y, out_e8m0, stream);
}

int qutlass_amd_to_blocked(const void* in, int64_t rows, int64_t cols, void* out, void* stream) {
  if (!in || !out) return fail(QAMD_ERR_INVALID, "to_blocked: null pointer argument");
  if (rows <= 0 || cols <= 0 || rows >= (1ll << 31) || cols >= (1ll << 31))
    return fail(QAMD_ERR_INVALID, "to_blocked: bad shape (%lld, %lld)", (long long)rows, (long long)cols);
  BlockedParams p;
  p.in = (const uint8_t*)in; p.out = (uint8_t*)out; p.rows = (int)rows; p.cols = (int)cols;
  p.RB = (int)cdiv(rows, 128); p.CB = (int)cdiv(cols, 4);
  // columns per workgroup: the widest slab that still gives every CU a workgroup
  int tc = 128;
  while (tc > 16 && (int64_t)p.RB * cdiv(p.CB, tc / 4) < chip_cus()) tc >>= 1;
  const int64_t grid = (int64_t)p.RB * cdiv(p.CB, tc / 4);
  if (grid >= (1ll << 31)) return fail(QAMD_ERR_INVALID, "to_blocked: matrix too large");
  switch (tc) {
    case 128: hipLaunchKernelGGL(to_blocked_kernel<128>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p); break;
    case 64: hipLaunchKernelGGL(to_blocked_kernel<64>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p); break;
    case 32: hipLaunchKernelGGL(to_blocked_kernel<32>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p); break;
    default: hipLaunchKernelGGL(to_blocked_kernel<16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p); break;
  }
  return check_launch("to_blocked_kernel");
}

// debug only (not declared in the public header): which kernels would matmul_mx{f4,f8}_bf16_tn(_ws) launch for this shape?
// out[3 * i + {0, 1, 2}] = {gemm_variant, N of the launch, K splits} of launch i; returns the number of launches (the split-K
// reduce pass is implied by splits > 1), or -1 when the arguments are rejected.  No GPU is touched.
int qutlass_amd_debug_gemm_plan(int ebits, int64_t M, int64_t N, int64_t K, int64_t workspace_bytes, int* out, int cap) {
  alignas(16) static char dummy[16];
  t_dry = DryRun{};
  t_dry.on = true;
  void* ws = workspace_bytes > 0 ? (void*)dummy : nullptr;
  const int rc = (ebits == 4) ? gemm_mx<4>("debug_gemm_plan", dummy, dummy, dummy, dummy, (const float*)dummy, dummy, M, N, K, nullptr, ws, workspace_bytes)
               : (ebits == 8) ? gemm_mx<8>("debug_gemm_plan", dummy, dummy, dummy, dummy, (const float*)dummy, dummy, M, N, K, nullptr, ws, workspace_bytes)
                              : QAMD_ERR_INVALID;
  const int n = t_dry.n;
  for (int i = 0; i < n && i < 8 && i < cap; ++i) { out[3 * i] = t_dry.rec[i][0]; out[3 * i + 1] = t_dry.rec[i][1]; out[3 * i + 2] = t_dry.rec[i][2]; }
  t_dry = DryRun{};
  return rc == QAMD_OK ? n : -1;
}

// [r4] debug only (not declared in the public header): the kernel rules of the QAT-backward data-prep ops on a 256-CU part, no GPU touched.
//   op 0 / 1: backward_t_bf16 / backward_qt_bf16 (a, b, c) = (B, N, M) -> 1 = the round-3 kernel, 2 = wave-owned 64-byte segments, 3 = wave-owned 128-byte lines, 5 = [r5] QT: 2 fed through the shared whole-line ring
//   op 2: backward_bf16_square_double_mxfp8 (a, b) = (m_pad, n) -> column tiles per workgroup (1 or 4)
int qutlass_amd_debug_stream_plan(int op, int64_t a, int64_t b, int64_t c) {
  if (op == 0 || op == 1) {
    if (a <= 0 || b <= 0 || c <= 0 || b % 32) return -1;
    return bwd_kernel_rule(op == 1, a * cdiv(c, 64) * cdiv(b / 32, 8), 256, op == 1 && c % 128 == 0);
  }
  if (op == 2) return (a <= 0 || b <= 0 || a % 128 || b % 128) ? -1 : sq_column_tiles_rule(a, b, 256);
  return -1;
}

// [r4] debug only (not declared in the public header): the grouped-raster decode of the persistent kernels (common.hip.h raster_decode, the SAME function the
// device runs, with the host's raster_magic) for tiles [t0, t0 + n) of a tiles_m x tiles_n grid: out[2 i] = tile row, out[2 i + 1] = tile column.  No GPU touched.
int qutlass_amd_debug_raster_decode(int tiles_m, int tiles_n, int t0, int n, int* out) {
  if (!out || tiles_m <= 0 || tiles_n <= 0 || t0 < 0 || n < 0) return -1;
  const uint32_t magic = qamd::raster_magic(tiles_n);
  for (int i = 0; i < n; ++i) qamd::raster_decode(t0 + i, tiles_m, tiles_n, magic, out[2 * i], out[2 * i + 1]);
  return n;
}

// debug only (not declared in the public header): what matmul_nvf4_bf16_tn's rule picks for an M x N x K problem (gemm_nvf4.hip.h: nvf4_plan; 256 CUs assumed,
// no GPU touched): the tile configuration (-1 skinny, 0 256x256, 1 128x128, 2 128x64, 3 64x64, 4 256x128), plus 256 x the number of K ranges when may_split
// (= the caller passes a workspace) and the shape splits
int qutlass_amd_debug_nvf4_plan(int64_t M, int64_t N, int64_t K, int may_split) {
  if (M <= 0 || N <= 0 || K <= 0) return -2;
  const qamd::NvPlan pl = qamd::nvf4_plan(M, N, K, 256, may_split != 0);
  return pl.splits > 1 ? pl.cfg + 256 * pl.splits : pl.cfg;
}

// [r4] the persistent NVFP4 kernel's plan on a 256-CU part: out[0] = workgroups, out[1] = tiles in the stream-K region (0: whole tiles only),
// out[2] = K stages per tile; returns 0 when the shape does not run the persistent kernel (tile configuration other than 256x256, K % 256, K < 512)
int qutlass_amd_debug_nvf4_pk_plan(int64_t M, int64_t N, int64_t K, int may_sk, int* out) {
  if (!out || M <= 0 || N <= 0 || K < 32) return 0;
  const qamd::NvPlan pl = qamd::nvf4_plan(M, N, K, 256, may_sk != 0);
  if (pl.cfg != 0 || !qamd::nvpk_shape_ok(M, N, K)) return 0;
  const qamd::NvPkPlan pk = qamd::nvpk_plan(M, N, K, 256, may_sk != 0);   // may_sk: what the LAB's stream-K variant would walk (the product never cuts tiles)
  out[0] = pk.grid; out[1] = pk.sk_tiles; out[2] = (int)(K / 256);
  return 1;
}
// the units workgroup w of `grid` walks over T tiles (the last sk_tiles as a stream of K stages), KT stages per tile: 5 ints per unit
// {tile, first stage, end stage, mode, slot} into out[0 .. 5 cap); returns the number of units (the device kernel runs the same NvPkWalk)
// gran: stage granularity of a range boundary (1: NVFP4 kernel, 2: the MX kernels)
int qutlass_amd_debug_sk_units(int w, int grid, int T, int sk_tiles, int KT, int gran, int* out, int cap) {
  if (!out || grid <= 0 || w < 0 || w >= grid || KT < 2 || sk_tiles < 0 || sk_tiles > T) return -1;
  qamd::SkWalk walk(w, grid, T, sk_tiles, KT, 2, gran);
  int n = 0;
  for (;;) {
    const qamd::SkUnit u = walk.next();
    if (u.mode < 0) break;
    if (n < cap) { out[5 * n] = u.tile; out[5 * n + 1] = u.kb; out[5 * n + 2] = u.ke; out[5 * n + 3] = u.mode; out[5 * n + 4] = u.slot; }
    ++n;
  }
  return n;
}
int qutlass_amd_debug_nvf4_pk_units(int w, int grid, int T, int sk_tiles, int KT, int* out, int cap) { return qutlass_amd_debug_sk_units(w, grid, T, sk_tiles, KT, 1, out, cap); }

#if QAMD_BENCH
// lab library only: device buffer for ABL_TRACE / ABL_CLOCK builds
void qutlass_amd_debug_set_trace_buffer(void* p) { g_dbg.store((uint32_t*)p); }
#endif

const char* qutlass_amd_last_error(void) { return g_err; }
#if QAMD_BENCH
const char* qutlass_amd_version(void) { return "qutlass_amd 0.2.0 (gfx950, lab build)"; }
#else
const char* qutlass_amd_version(void) { return "qutlass_amd 0.2.0 (gfx950)"; }
#endif

// Product library: knows NO key (every call returns -1) -- nothing a caller or another thread does can change which kernel runs or what it computes.
// The lab library accepts the tuning / variant keys its benches use, and "hw_fp4_cvt" (0 = the software e2m1 encoder, bit-identical to the hardware
// convert the product ships: tests/test_gpu_parity.py runs the quantizer goldens under both through the lab build).
int qutlass_amd_set_option(const char* key, int value) {
  if (!key) return -1;
  (void)value;
#if QAMD_BENCH
  if (!strcmp(key, "hw_fp4_cvt")) return g_hw_fp4_cvt.exchange(value);
  if (!strcmp(key, "gemm_variant")) return g_gemm_variant.exchange(value);
  if (!strcmp(key, "nvf4_variant")) return g_nvf4_variant.exchange(value);
  if (!strcmp(key, "transpose_nc")) return g_transpose_nc.exchange(value);
  if (!strcmp(key, "splitk_wg")) return g_splitk_wg.exchange(value);
  if (!strcmp(key, "splitk_min_kt")) return g_splitk_min_kt.exchange(value);
  if (!strcmp(key, "splitk_force")) return g_splitk_force.exchange(value);
  if (!strcmp(key, "deepp_grid")) return g_deepp_grid.exchange(value);
  if (!strcmp(key, "bwd_variant")) return g_bwd_variant.exchange(value);
  if (!strcmp(key, "quant_wg_per_cu")) return g_quant_wg_per_cu.exchange(value);
  if (!strcmp(key, "pp_shift")) return g_pp_shift.exchange(value);
  if (!strcmp(key, "pp_flags")) return g_pp_flags.exchange(value);
#endif
  return -1;
}

}  // extern "C"
#pragma GCC visibility pop
#endif   // QAMD_DEF(1)
