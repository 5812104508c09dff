/*
 * qutlass_amd.h -- C ABI of the MI355X (gfx950) microscaled low-bit GEMM library.
 *
 * This is the drop-in boundary for the qutlass hot path: every entry point replaces one host
 * launcher that the reference's torch op bindings (qutlass/csrc/bindings.cpp) call, with plain
 * device pointers, sizes and a HIP stream instead of torch::stable::Tensor.  A maintainer of the
 * reference binds these from bindings.cpp (or from Python via ctypes) -- see INTEGRATION.md.
 *
 * Conventions
 *   - all pointers are DEVICE pointers on the current HIP device; `stream` is a hipStream_t
 *     (NULL = the legacy default stream).  Calls are asynchronous and never synchronise.
 *   - return value: 0 = launched; QAMD_ERR_INVALID = argument rejected (nothing launched);
 *     QAMD_ERR_HIP = the HIP runtime refused the launch.  qutlass_amd_last_error() returns a
 *     thread-local message for the last non-zero return.
 *   - no global state besides the one verification switch at the end of this file; re-entrant; the library never allocates
 *     (the reference cudaMalloc's a CUTLASS workspace per call, gemm.cu:160-162); ops that use
 *     scratch (mxf8 NN pre-pass, split-K of the *_ws GEMMs) take it from the caller.
 */
#ifndef QUTLASS_AMD_H_
#define QUTLASS_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QAMD_OK 0
#define QAMD_ERR_INVALID 1
#define QAMD_ERR_HIP 2

#define QAMD_METHOD_QUEST 0
#define QAMD_METHOD_ABSMAX 1

/* ---- block-scaled GEMMs:  D[M,N] (bf16, row-major) = alpha[0] * (A.SFA) (B.SFB)^T ------------- */

/*
 * MXFP4.  A: (M, K/2) bytes, B: (N, K/2) bytes, two e2m1 per byte (element 2j = low nibble), K % 128 == 0.
 * A_sf / B_sf: e8m0, one per 32 K-elements, in the to_blocked layout of a
 * (ceil(M/128)*128, ceil(K/128)*4) matrix (resp. N).  alpha: device fp32[1].  N % 8 == 0.
 * Replaces matmul_host_mxf4_bf16_tn (qutlass/csrc/gemm.cu:174-248; declared include/gemm.h:21-27;
 * called from bindings.cpp:32-66).
 */
int qutlass_amd_matmul_mxf4_bf16_tn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                    const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                    void* stream);

/*
 * MXFP4, small-batch variant: same operands, but A_sf / B_sf are the UN-swizzled row-major (M, K/32) / (N, K/32) e8m0
 * matrices (what fusedQuantizeMx writes, without to_blocked).  Any M is accepted; built for small batches: an LDS-free
 * split-K kernel for M <= 32 against a small weight, the 64x64 ring kernel with row-major scale fetch for N >= 8192 or
 * M > 32 (both weight-bandwidth bound).  K % 128 == 0, N % 8 == 0.
 * Replaces matmul_host_ada_mxf4_bf16_tn (qutlass/csrc/gemm_ada.cu:30-135; bindings.cpp:104-138; scale addressing
 * cutlass_extensions/gemm/threadblock/mx_mma_multistage.h:418-448).
 */
int qutlass_amd_matmul_ada_mxf4_bf16_tn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                        const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                        void* stream);

/*
 * NVFP4.  Same operand layout, scales are e4m3fn per 16 K-elements in the to_blocked layout of a
 * (ceil(M/128)*128, ceil(K/64)*4) matrix; K % 32 == 0.
 * Replaces matmul_host_nvf4_bf16_tn (gemm.cu:250-326; bindings.cpp:68-102).
 */
int qutlass_amd_matmul_nvf4_bf16_tn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                    const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                    void* stream);

/*
 * MXFP8 (e4m3fn data, e8m0 per 32).  TN: A (M,K), B (N,K) row-major.  K % 32 == 0.
 * Replaces matmul_host_mxf8_bf16_tn (gemm.cu:328-386; bindings.cpp:140-177).
 */
int qutlass_amd_matmul_mxf8_bf16_tn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                    const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                    void* stream);

/*
 * The same two GEMMs with caller-owned scratch, which unlocks split-K for small outputs with a long K (fewer than 256
 * tiles of 64x64 -- at most 128 of them -- and K >= 32 stages of 128 bytes: e.g. M = 64, N = 4096, K = 14336 runs 64
 * workgroups without it).
 * Each K split writes its fp32 partial to workspace[z][M][N]; a second kernel sums the splits in fixed order, applies
 * alpha and rounds to bf16 (deterministic; identical to the single-pass result whenever the fp32 partial sums are
 * exact, which is the regime the reference's equality tests run in).  qutlass_amd_gemm_splitk_workspace_bytes(ebits = 4
 * or 8, M, N, K) returns the bytes the split needs, 0 when the shape does not split; a NULL or smaller workspace
 * silently runs the single-pass kernel.  The workspace is used on `stream` for the duration of the call's kernels and must be
 * 16-byte aligned (the partials are written as 16-byte vectors; a misaligned pointer is rejected with QAMD_ERR_INVALID).
 * (The reference allocates and frees a CUTLASS workspace inside every call, gemm.cu:160-162.)
 *
 * Alignment.  A, B, both scale operands and D of the matmul_{mxf4,mxf8,nvf4}_bf16_* entries must be 16-byte aligned (QAMD_ERR_INVALID otherwise): operands are
 * fetched as 16-byte pieces, the output leaves as 16-byte stores.  Tensors that torch allocates, and row ranges of them, are.  (The reference's CUTLASS kernels
 * require 128-bit alignment through their TMA descriptors, gemm.cu:90-143.)
 *
 * Reproducibility.  Every entry point is deterministic: the same call on the same device returns the same bytes, run after run and
 * under HIP-graph replay.  It is NOT bit-stable ACROSS entry points or devices for general data: whether a shape splits K depends on
 * whether a workspace was passed and on the device's CU count (the plans scale with it), and a split sums the fp32 partials in a
 * different order than the single pass -- the bf16 output can differ in its last bit.  Exact-regime operands (every partial sum
 * exactly representable: the regime of the reference's own equality tests) give identical bytes on every path.  For one summation
 * order everywhere call the entries without `_ws`.
 */
int64_t qutlass_amd_gemm_splitk_workspace_bytes(int ebits, int64_t M, int64_t N, int64_t K);
int qutlass_amd_matmul_mxf4_bf16_tn_ws(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                       const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                       void* workspace, int64_t workspace_bytes, void* stream);
int qutlass_amd_matmul_mxf8_bf16_tn_ws(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                       const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                       void* workspace, int64_t workspace_bytes, void* stream);

/*
 * [r3] The NVFP4 GEMM with caller-owned scratch: split-K for outputs of a few dozen 128x128 tiles with a long K (e.g. M = 256, N = 4096,
 * K = 14336), where the single pass can only fill the chip with 64x64 tiles whose 32x32 wave tiles dequantise two operand fragments per MFMA.
 * Same contract as the MX entries above: fp32 partials in workspace[z][M][N], summed in fixed z order by a second kernel (alpha, bf16);
 * qutlass_amd_nvf4_splitk_workspace_bytes(M, N, K) returns the bytes the planned split needs, 0 when the shape does not split; a NULL or
 * smaller workspace silently runs the single-pass kernel (= qutlass_amd_matmul_nvf4_bf16_tn).
 */
int64_t qutlass_amd_nvf4_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
int qutlass_amd_matmul_nvf4_bf16_tn_ws(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                       const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                       void* workspace, int64_t workspace_bytes, void* stream);

/*
 * MXFP8 NN: A is stored (K, M) row-major (the reference's ColumnMajor A), B (N, K); A_sf is still the
 * to_blocked layout of the (M, K/32) scale matrix.  K % 32 == 0, M % 16 == 0 (the reference's
 * AlignmentA = 16 on the contiguous M axis, gemm.cu:400).
 * workspace: caller-owned device scratch, used on `stream` only for the duration of the call's kernels (the library
 * itself never allocates).  Small problems re-lay A as (M, K) with a byte-transpose pre-pass and then run the TN kernel:
 * they need M * K bytes.  Problems whose 256x256 tiles fill the chip read the (K, M) operand directly and need NONE.
 * qutlass_amd_mxf8_nn_workspace_bytes_for(M, N, K) returns what THIS shape needs (0 for the in-place path: workspace may be
 * NULL); qutlass_amd_mxf8_nn_workspace_bytes(M, K) is the shape-independent upper bound M * K.
 * Replaces matmul_host_mxf8_bf16_nn (gemm.cu:388-434; bindings.cpp:179-216).
 */
int64_t qutlass_amd_mxf8_nn_workspace_bytes(int64_t M, int64_t K);
int64_t qutlass_amd_mxf8_nn_workspace_bytes_for(int64_t M, int64_t N, int64_t K);
int qutlass_amd_matmul_mxf8_bf16_nn(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                    const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                    void* workspace, int64_t workspace_bytes, void* stream);

/*
 * EXTENSION (no reference counterpart): MXFP8 with an e5m2 A operand -- the gradient x activation GEMMs of a QAT
 * backward pass (BASELINE.json configs[4]).  The reference's entry points reject every element type but e4m3
 * (bindings.cpp:157-160, 196-199; gemm.cu:339-345, 399-403); CDNA4's scaled MFMA takes the format per operand.
 * a_format / b_format: QAMD_FP8_E4M3 or QAMD_FP8_E5M2; b_format must be QAMD_FP8_E4M3.  Everything else (layouts, scales,
 * alignment, workspace rules, return codes) as the entry of the same name without `_fmt`: the TN entry takes the optional
 * split-K scratch of qutlass_amd_matmul_mxf8_bf16_tn_ws (NULL / 0 = never split), the NN entry the re-layout scratch reported by
 * qutlass_amd_mxf8_nn_workspace_bytes_for(M, N, K) (may be NULL when that returns 0: the in-place operand path).  With both
 * formats E4M3 they ARE those entries.
 */
#define QAMD_FP8_E4M3 0
#define QAMD_FP8_E5M2 1
int qutlass_amd_matmul_mxf8_bf16_tn_fmt(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                        const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                        int a_format, int b_format, void* workspace, int64_t workspace_bytes, void* stream);
int qutlass_amd_matmul_mxf8_bf16_nn_fmt(const void* A, const void* B, const void* A_sf, const void* B_sf,
                                        const float* alpha, void* D, int64_t M, int64_t N, int64_t K,
                                        int a_format, int b_format, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- fused rotate + quantize ------------------------------------------------------------------ */

/*
 * x: bf16[numel] viewed as (numel/rot, rot); h: bf16[rot*rot] row-major (runtime matrix, y = x_g . h);
 * rot in {32, 64, 128}; numel % rot == 0.  method: QAMD_METHOD_QUEST / QAMD_METHOD_ABSMAX.
 * out_e2m1: numel/2 bytes; out_e8m0: numel/32 bytes written FLAT in group order (the caller's
 * (padded_rows, padded_cols) buffer keeps its padding untouched, as in the reference);
 * out_mask: NULL, or numel/8 bytes (one u32 per 32-group; quest + rot 32 only).
 * Alignment: h must be 16-byte aligned for rot >= 64 (QAMD_ERR_INVALID otherwise; the kernel stages it with 16-byte loads).
 * Non-finite activations follow the reference's arithmetic: NaN -> code 0x7 whatever its sign, +-inf -> +-6 or, under an
 * infinite group scale (e8m0 byte 255), 0 / NaN as x / inf gives (tests/test_gpu_round5.py).
 * Replaces fusedQuantizeMx{Quest,AbsMax}{,Had64,Had128}_host (fused_quantize_mx.cu:107-207) and
 * fusedQuantizeMxQuestWithMask_host (fused_quantize_mx_mask.cu:107-123); bindings.cpp:218-333.
 */
int qutlass_amd_fused_quantize_mx(const void* x, const void* h, int rot, int64_t numel, int method,
                                  void* out_e2m1, void* out_e8m0, void* out_mask, void* stream);

/*
 * NVFP4 variant: rot in {16, 32, 64, 128}; out_e4m3: numel/16 bytes (e4m3fn), flat;
 * global_scale: device fp32[1].
 * Replaces fusedQuantizeNv{Quest,AbsMax}{,Had32,Had64,Had128}_host (fused_quantize_nv.cu:109-252;
 * bindings.cpp:335-426).
 */
int qutlass_amd_fused_quantize_nv(const void* x, const void* h, int rot, int64_t numel, int method,
                                  const float* global_scale, void* out_e2m1, void* out_e4m3,
                                  void* stream);

/*
 * EXTENSION (no reference counterpart; the reference's activation path is fusedQuantizeMx -> to_blocked -> matmul, three launches,
 * qutlass/__init__.py:149-180 + qutlass/utils.py:160-193): the same quantizers, but the scale bytes are written DIRECTLY in the
 * to_blocked() layout the GEMMs consume -- one launch instead of two.  x is a (rows, k) bf16 matrix (k % max(rot, 32) == 0);
 * out_*_blocked: ceil(rows/128)*128 * ceil(k/gs/4)*4 bytes (gs = 32 MX / 16 NV), padding zero-filled; byte for byte what
 * qutlass_amd_to_blocked() makes of the flat scales of the entry above (tests/test_gpu_parity.py).  Everything else as above.
 */
int qutlass_amd_fused_quantize_mx_blocked(const void* x, const void* h, int rot, int64_t rows, int64_t k, int method,
                                          void* out_e2m1, void* out_e8m0_blocked, void* out_mask, void* stream);
int qutlass_amd_fused_quantize_nv_blocked(const void* x, const void* h, int rot, int64_t rows, int64_t k, int method,
                                          const float* global_scale, void* out_e2m1, void* out_e4m3_blocked,
                                          void* stream);

/*
 * EXTENSION: the measured launch-count rule of the activation path y = Q(x h) W^T of one linear layer (reference flow: qutlass/__init__.py:149-180 ->
 * qutlass/utils.py:160-193 -> qutlass/__init__.py:34-76, three launches): returns 1 where the one-launch decode kernel below wins (M <= 16, R = 32, short K,
 * a weight of fewer than 32 x CUs rows), else 2 (quantizer with GEMM-ready scales + GEMM).  Pure host arithmetic on the current device's CU count; what
 * qutlass_amd.fused_quantize_matmul_mxf4_bf16_tn (Python) asks before it launches, so that a C caller gets the same rule.
 */
int qutlass_amd_activation_path_launches(int64_t M, int64_t N, int64_t K, int rot);

/*
 * EXTENSION: the whole decode-time activation path in ONE launch, for batches of at most 32 rows:
 *     D[M,N] (bf16) = alpha[0] * Q(x . h) (B . SFB)^T
 * x: (M, K) bf16 activations, h: 32 x 32 bf16 rotation (rot must be 32), method as above; B / B_sf: the MXFP4 weight and its
 * to_blocked e8m0 scales exactly as qutlass_amd_matmul_mxf4_bf16_tn takes them.  K % 128 == 0, N % 8 == 0, 1 <= M <= 32.
 * Bit-identical to qutlass_amd_fused_quantize_mx + qutlass_amd_to_blocked + qutlass_amd_matmul_mxf4_bf16_tn on the same
 * inputs (the three launches of qutlass/__init__.py:149-180, qutlass/utils.py:160-193, qutlass/__init__.py:34-76).
 */
int qutlass_amd_fused_quantize_matmul_mxf4_bf16_tn(const void* x, const void* h, int rot, int method, const void* B,
                                                   const void* B_sf, const float* alpha, void* D, int64_t M, int64_t N,
                                                   int64_t K, void* stream);

/* ---- QAT-backward data preparation ------------------------------------- */

/*
 * Input contract of the four ops below (and of the forward quantizers): FINITE operands.  What the reference's kernels do with NaN / inf
 * follows from the order of its fmaxf chains and is not a documented behaviour; here
 *   - backward_t_bf16 / backward_qt_bf16 reproduce the reference's arithmetic for the cases its own data can reach (all-zero groups:
 *     3 / 0 = inf, 0 * inf = NaN -> code 7; scales outside the division-free range take the reference's divisions), tests pin them;
 *   - backward_bf16_square_double_mxfp8 and mxfp4_transpose_mxfp8 take the block maximum on sign-stripped bf16 bit patterns: a NaN or inf
 *     element (or an input e8m0 byte of 255) becomes the block maximum, where an fmaxf chain would have skipped a NaN;
 *   - an input e8m0 byte of 0 (2^-127: the dequantised bf16 operand is a denormal) is outside what the MFMA rotation reproduces exactly
 *     (the matrix core flushes bf16 denormals); the forward quantizers never emit it for a non-zero group.
 * All four are deterministic and bit-stable across devices (no plan depends on the CU count in a way that changes arithmetic order).
 */

/*
 * x: bf16 (B, N, M) row-major; h: bf16 32 x 32.  For every (b, m) and every 32-group g along N:
 * y = x[b, 32g..32g+31, m] . h, abs-max MXFP4 (no epsilon, x3 before rounding).  out_e2m1: (B, M, N/2) bytes,
 * out_e8m0: (B, M, N/32) bytes.  N % 32 == 0, M % 8 == 0.
 * Replaces backward_t_bf16_cuda (qutlass/csrc/quartet_bwd_sm120.cu:237-325,430-454; include/backward_host.h:17-26;
 * bindings.cpp:429-443).
 */
int qutlass_amd_backward_t_bf16(const void* x, const void* h, int64_t B, int64_t N, int64_t M, void* out_e2m1,
                                void* out_e8m0, void* stream);

/*
 * The same on an MXFP4 operand: x_e2m1 (B, N, M/2) bytes, x_e8m0 (B, N, M/32); alpha: device fp32[1] (enters the
 * scale only: e8m0 = floor_pow2(amax / alpha), q = y * 3 / (scale * alpha)).  N % 32 == 0, M % 32 == 0.
 * Replaces backward_qt_bf16_cuda (quartet_bwd_sm120.cu:327-428,457-483; backward_host.h:4-15; bindings.cpp:445-464).
 */
int qutlass_amd_backward_qt_bf16(const void* x_e2m1, const void* x_e8m0, const void* h, const float* alpha, int64_t B,
                                 int64_t N, int64_t M, void* out_e2m1, void* out_e8m0, void* stream);

/*
 * x: bf16 (m, n).  One shared exponent per 32 x 32 block (floor(log2 amax) - 7, 127 for an all-zero block);
 * y: e4m3 (m, n); row_scales: e8m0 (m, n/32); col_scales: e8m0 (n, m/32).  m % 128 == 0 and n % 128 == 0 (the
 * reference wrapper pads m to 128 and launches n/128 blocks, qutlass/__init__.py:288-297, quartet_bwd_sm120.cu:596).
 * Replaces backward_bf16_square_double_mxfp8_cuda (quartet_bwd_sm120.cu:511-621; bindings.cpp:466-479).
 */
int qutlass_amd_backward_bf16_square_double_mxfp8(const void* x, int64_t m, int64_t n, void* y, void* row_scales,
                                                  void* col_scales, void* stream);

/*
 * x_fp4: packed e2m1 (m, n/2), scales: e8m0 (m, n/32).  y: e4m3 (n, m) = requantised transpose with one shared
 * exponent per 32 along m; out_e8m0: (n, m/32).  m % 128 == 0, n % 256 == 0 (the reference pads m to 256 and launches
 * n/256 blocks, __init__.py:299-315, quartet_bwd_sm120.cu:716).
 * Replaces mxfp4_transpose_mxfp8_cuda (quartet_bwd_sm120.cu:628-733; bindings.cpp:481-494).
 */
int qutlass_amd_mxfp4_transpose_mxfp8(const void* x_fp4, const void* scales, int64_t m, int64_t n, void* y,
                                      void* out_e8m0, void* stream);

/*
 * The same two ops for ANY row count m: the outputs are laid out for m_pad rows (a multiple of 128 >= m; the reference's
 * wrappers use ceil(m / 128) * 128 resp. ceil(m / 256) * 256) and rows m .. m_pad-1 of the input are treated as zeros (zero codes
 * with unit scales for the MXFP4 input) INSIDE the kernel.  The reference materialises that padding with
 * torch.nn.functional.pad -- a full extra copy of the operand -- and, for mxfp4_transpose_mxfp8, by writing 1.0 into rows
 * m .. m_pad-1 of the CALLER's scale tensor (qutlass/__init__.py:288-307, its own "TODO: padding in kernel"); here x / x_fp4 /
 * scales are read-only and only need their m real rows.  With m_pad == m these ARE the entries above.
 * y: (m_pad, n) resp. (n, m_pad); row_scales (m_pad, n/32), col_scales (n, m_pad/32); out_e8m0 (n, m_pad/32).
 */
int qutlass_amd_backward_bf16_square_double_mxfp8_rows(const void* x, int64_t m, int64_t m_pad, int64_t n, void* y,
                                                       void* row_scales, void* col_scales, void* stream);
int qutlass_amd_mxfp4_transpose_mxfp8_rows(const void* x_fp4, const void* scales, int64_t m, int64_t m_pad, int64_t n,
                                           void* y, void* out_e8m0, void* stream);

/* ---- block-scale swizzle ------------------------------------------------------------------------ */

/*
 * in: (rows, cols) row-major 1-byte elements; out: ceil(rows/128)*128 * ceil(cols/4)*4 bytes in the
 * 128x4 tiled order, zero padded.  Replaces qutlass/utils.py:160-193 (to_blocked) incl. the Triton
 * kernel at :16-133.
 */
int qutlass_amd_to_blocked(const void* in, int64_t rows, int64_t cols, void* out, void* stream);

/* ---- misc ----------------------------------------------------------------------------------------- */

const char* qutlass_amd_last_error(void);
const char* qutlass_amd_version(void);

/*
 * libqutlass_amd.so has NO options: this entry returns -1 for every key -- nothing a caller or another thread does can change which
 * kernel a shape gets or what it computes.  (The lab build of the same sources, libqutlass_amd_bench.so -- test and bench
 * infrastructure, see INTEGRATION.md -- accepts "gemm_variant", "nvf4_variant", "pp_flags", "splitk_wg", "splitk_min_kt",
 * "splitk_force", "transpose_nc", "quant_wg_per_cu", "pp_shift", "deepp_grid" and "hw_fp4_cvt" (0 = the software e2m1 encoder instead of
 * v_cvt_scalef32_pk_fp4_f32: same bits, tests/test_gpu_parity.py) and returns the previous value.)
 */
int qutlass_amd_set_option(const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* QUTLASS_AMD_H_ */
