// PyTorch (ROCm) extension: the `_qutlass_C` operator library of the reference, bound to the C ABI of
// libqutlass_amd.so (include/qutlass_amd.h).  No device code here -- the kernels are the hand-written HIP in
// gemm_*.hip.h / quantize.hip.h / to_blocked.hip.h behind the C ABI.
//
// Mirrors qutlass/csrc/bindings.cpp of the reference op for op: same op names and schema strings
// (bindings.cpp:499-513), same argument validation order and messages (bindings.cpp:32-426,
// include/bindings_utils.h:67-136), same ownership (GEMMs allocate and return the bf16 result, quantizers fill
// caller-allocated outputs and return them), same stream rule (current stream of the tensor's device,
// include/common.h:40-45).  Written against the LibTorch stable ABI, like the reference, so one build serves
// every torch >= 2.10.  One extra op, qutlass_amd::to_blocked, replaces the Triton/torch swizzle of
// qutlass/utils.py:160-193.
//
// Built as qutlass/_CUDA.abi3.so: like the reference's op library it is a Python extension module (PyInit__CUDA,
// include/registration.h + bindings.cpp:537-540) whose import -- or a plain dlopen through torch.ops.load_library --
// runs the static registrations; `_qutlass_C` is opened as a FRAGMENT and implemented for the CUDA dispatch key only
// (bindings.cpp:498, :516), so a CPU tensor gets the dispatcher's own "not implemented for the CPU backend" error.
#include <Python.h>

#include <torch/csrc/inductor/aoti_torch/c/shim.h>
#include <torch/csrc/stable/accelerator.h>
#include <torch/csrc/stable/library.h>
#include <torch/csrc/stable/ops.h>
#include <torch/csrc/stable/tensor.h>
#include <torch/headeronly/core/ScalarType.h>
#include <torch/headeronly/util/Exception.h>

#include <initializer_list>
#include <string>
#include <tuple>

#include "../../include/qutlass_amd.h"

namespace {

using torch::headeronly::ScalarType;
using torch::stable::Tensor;

struct Named {
  const Tensor& t;
  const char* name;
};

std::string arg_desc(int pos, const char* name) { return "argument #" + std::to_string(pos) + " '" + name + "'"; }

// ---- the three generic checks of include/bindings_utils.h, table-driven ------------------------------------------
void require_contiguous(const char* op, std::initializer_list<Named> args) {
  int pos = 0;
  for (const Named& a : args) {
    STD_TORCH_CHECK(a.t.is_contiguous(), "Expected contiguous tensor, but got non-contiguous tensor for ", arg_desc(pos, a.name),
                    " (while checking arguments for ", op, ")");
    ++pos;
  }
}

void require_gpu(const char* op, std::initializer_list<Named> args) {
  for (const Named& a : args)
    STD_TORCH_CHECK(a.t.is_cuda(), "Expected tensor to have cuda DeviceType, but got tensor with ", a.t.is_cpu() ? "cpu" : "another",
                    " DeviceType (while checking arguments for ", op, ")");
}

void require_same_gpu(const char* op, std::initializer_list<Named> args) {
  const Named& first = *args.begin();
  int pos = 0;
  for (const Named& a : args) {
    if (pos > 0)
      STD_TORCH_CHECK(a.t.get_device_index() == first.t.get_device_index(), "Expected tensor for ", arg_desc(0, first.name),
                      " to have the same device as tensor for ", arg_desc(pos, a.name), "; but device ", first.t.get_device_index(),
                      " does not equal ", a.t.get_device_index(), " (while checking arguments for ", op, ")");
    ++pos;
  }
}

// dtype through the C shim: Tensor::scalar_type() goes through the stable IValue conversion, which in torch 2.10 does
// not know Float8_e8m0fnu yet ("Not yet supported ScalarType 44")
bool has_dtype(const Tensor& t, ScalarType want) {
  int32_t code = -1;
  TORCH_ERROR_CODE_CHECK(aoti_torch_get_dtype(t.get(), &code));
  return code == static_cast<int32_t>(want);
}

void* current_stream(const Tensor& t) {   // include/common.h:40-45
  void* s = nullptr;
  TORCH_ERROR_CODE_CHECK(aoti_torch_get_current_cuda_stream(t.get_device_index(), &s));
  return s;
}

void check_rc(int rc) { STD_TORCH_CHECK(rc == QAMD_OK, qutlass_amd_last_error()); }

// ---- block-scaled GEMMs --------------------------------------------------------------------------------------------
enum class Gemm { MXF4, NVF4, MXF8_TN, MXF8_NN, ADA_MXF4 };

template <Gemm G>
Tensor matmul(const Tensor& A, const Tensor& B, const Tensor& A_sf, const Tensor& B_sf, const Tensor& alpha) {
  constexpr bool fp8 = G == Gemm::MXF8_TN || G == Gemm::MXF8_NN;
  constexpr bool nn = G == Gemm::MXF8_NN;
  const char* op = G == Gemm::ADA_MXF4 ? "matmul_ada_mxf4_bf16_tn" : G == Gemm::MXF4 ? "matmul_mxf4_bf16_tn" : G == Gemm::NVF4 ? "matmul_nvf4_bf16_tn" : G == Gemm::MXF8_TN ? "matmul_mxf8_bf16_tn" : "matmul_mxf8_bf16_nn";
  if (fp8) require_contiguous(op, {{A, "A"}, {B, "B"}, {A_sf, "A_sf"}, {B_sf, "B_sf"}, {alpha, "alpha"}});
  else require_contiguous(op, {{A, "A"}, {B, "B"}, {A_sf, "A_sf"}, {B_sf, "B_sf"}});
  require_gpu(op, {{A, "A"}, {B, "B"}, {A_sf, "A_sf"}, {B_sf, "B_sf"}, {alpha, "alpha"}});
  require_same_gpu(op, {{A, "A"}, {B, "B"}, {A_sf, "A_sf"}, {B_sf, "B_sf"}, {alpha, "alpha"}});

  const ScalarType data_t = fp8 ? ScalarType::Float8_e4m3fn : ScalarType::Byte;
  const ScalarType sf_t = G == Gemm::NVF4 ? ScalarType::Float8_e4m3fn : ScalarType::Float8_e8m0fnu;
  const char* data_n = fp8 ? "float8_e4m3fn" : "uint8";
  const char* sf_n = G == Gemm::NVF4 ? "float8_e4m3fn" : "float8_e8m0fnu";
  // EXTENSION over the reference (which accepts e4m3 only, bindings.cpp:157-160, 196-199): the A operand of the MXFP8 GEMMs
  // may be float8_e5m2 -- the gradient operand of a QAT backward GEMM (BASELINE.json configs[4]); B stays e4m3.
  const bool a_e5m2 = fp8 && has_dtype(A, ScalarType::Float8_e5m2);
  STD_TORCH_CHECK(has_dtype(A, data_t) || a_e5m2, "A must be ", data_n, fp8 ? " (or float8_e5m2)" : "");
  STD_TORCH_CHECK(has_dtype(B, data_t), "B must be ", data_n);
  STD_TORCH_CHECK(has_dtype(A_sf, sf_t), "A_sf must be ", sf_n);
  STD_TORCH_CHECK(has_dtype(B_sf, sf_t), "B_sf must be ", sf_n);
  STD_TORCH_CHECK(A.dim() == 2 && B.dim() == 2, "A and B must be 2D");
  const int64_t kmin = G == Gemm::NVF4 ? 16 : 32;
  int64_t M;
  if (nn) {
    STD_TORCH_CHECK(A.size(0) == B.size(1), "Inner dimensions must match for A.T @ B.T");
    STD_TORCH_CHECK(A.size(0) >= kmin, "A K-dim must be >= ", kmin);
    M = A.size(1);
  } else {
    STD_TORCH_CHECK(A.size(1) == B.size(1), "Inner dimensions must match for A @ B.T");
    STD_TORCH_CHECK(A.size(1) >= kmin, "A K-dim must be >= ", kmin);
    M = A.size(0);
  }
  STD_TORCH_CHECK(B.size(1) >= kmin, "B K-dim must be >= ", kmin);
  const int64_t N = B.size(0), K = B.size(1) * (fp8 ? 1 : 2);
  // the kernels read alpha as one fp32 and the scale operands through descriptors sized from M / N / K: make sure the
  // tensors are at least that large (the reference leaves both to CUTLASS' can_implement / the caller); checked before the
  // empty-shape return so that a malformed argument fails whatever the batch size
  STD_TORCH_CHECK(has_dtype(alpha, ScalarType::Float) && alpha.numel() >= 1, "alpha must be a float32 tensor with at least one element");
  {
    const int64_t gs = G == Gemm::NVF4 ? 16 : 32, kb = K / gs;
    const bool row_major_sf = G == Gemm::ADA_MXF4;   // un-swizzled (rows, K/32); every other op: to_blocked layout of the padded matrix
    const int64_t need_a = row_major_sf ? M * kb : (M + 127) / 128 * 128 * ((kb + 3) / 4 * 4);
    const int64_t need_b = row_major_sf ? N * kb : (N + 127) / 128 * 128 * ((kb + 3) / 4 * 4);
    STD_TORCH_CHECK(A_sf.numel() >= need_a, "A_sf has ", A_sf.numel(), " elements, the ", row_major_sf ? "row-major" : "blocked", " scale layout of A needs ", need_a);
    STD_TORCH_CHECK(B_sf.numel() >= need_b, "B_sf has ", B_sf.numel(), " elements, the ", row_major_sf ? "row-major" : "blocked", " scale layout of B needs ", need_b);
  }
  Tensor out = torch::stable::new_empty(A, {M, N}, ScalarType::BFloat16);
  if (M == 0 || N == 0) return out;   // empty batch / empty weight: nothing to launch (the C ABI requires positive extents)

  const torch::stable::accelerator::DeviceGuard guard(A.get_device_index());
  const float* al = static_cast<const float*>(alpha.data_ptr());
  void* s = current_stream(A);
  int rc;
  if (G == Gemm::ADA_MXF4) rc = qutlass_amd_matmul_ada_mxf4_bf16_tn(A.data_ptr(), B.data_ptr(), A_sf.data_ptr(), B_sf.data_ptr(), al, out.data_ptr(), M, N, K, s);
  else if (G == Gemm::MXF4 || G == Gemm::MXF8_TN) {
    // small outputs with a long K split K over scratch from torch's stream-ordered caching allocator (the reference
    // allocates its CUTLASS workspace per call as well, gemm.cu:160-162); 0 bytes = the shape does not split
    const int64_t ws_bytes = qutlass_amd_gemm_splitk_workspace_bytes(fp8 ? 8 : 4, M, N, K);
    Tensor ws = ws_bytes > 0 ? torch::stable::new_empty(A, {ws_bytes}, ScalarType::Byte) : Tensor();
    void* wp = ws_bytes > 0 ? ws.data_ptr() : nullptr;
    rc = fp8 ? qutlass_amd_matmul_mxf8_bf16_tn_fmt(A.data_ptr(), B.data_ptr(), A_sf.data_ptr(), B_sf.data_ptr(), al, out.data_ptr(), M, N, K,
                                                   a_e5m2 ? QAMD_FP8_E5M2 : QAMD_FP8_E4M3, QAMD_FP8_E4M3, wp, ws_bytes, s)
             : qutlass_amd_matmul_mxf4_bf16_tn_ws(A.data_ptr(), B.data_ptr(), A_sf.data_ptr(), B_sf.data_ptr(), al, out.data_ptr(), M, N, K, wp, ws_bytes, s);
  }
  else if (G == Gemm::NVF4) {
    // [r3] the same for NVFP4: outputs of a few dozen 128x128 tiles with a long K split K (M = 256, N = 4096, K = 14336: 54.7 -> 39.2 us)
    const int64_t ws_bytes = qutlass_amd_nvf4_splitk_workspace_bytes(M, N, K);
    Tensor ws = ws_bytes > 0 ? torch::stable::new_empty(A, {ws_bytes}, ScalarType::Byte) : Tensor();
    rc = qutlass_amd_matmul_nvf4_bf16_tn_ws(A.data_ptr(), B.data_ptr(), A_sf.data_ptr(), B_sf.data_ptr(), al, out.data_ptr(), M, N, K,
                                            ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes, s);
  }
  else {
    // scratch for the (K, M) -> (M, K) re-layout of small problems, from torch's stream-ordered caching allocator; shapes
    // that read the (K, M) operand in place need none (0 bytes: nothing is allocated)
    const int64_t ws_bytes = qutlass_amd_mxf8_nn_workspace_bytes_for(M, N, K);
    Tensor ws = ws_bytes > 0 ? torch::stable::new_empty(A, {ws_bytes}, ScalarType::Byte) : Tensor();
    rc = qutlass_amd_matmul_mxf8_bf16_nn_fmt(A.data_ptr(), B.data_ptr(), A_sf.data_ptr(), B_sf.data_ptr(), al, out.data_ptr(), M, N, K,
                                             a_e5m2 ? QAMD_FP8_E5M2 : QAMD_FP8_E4M3, QAMD_FP8_E4M3, ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes, s);
  }
  check_rc(rc);
  return out;
}

Tensor matmul_mxf4_bf16_tn(const Tensor& A, const Tensor& B, const Tensor& A_sf, const Tensor& B_sf, const Tensor& alpha) { return matmul<Gemm::MXF4>(A, B, A_sf, B_sf, alpha); }
Tensor matmul_ada_mxf4_bf16_tn(const Tensor& A, const Tensor& B, const Tensor& A_sf, const Tensor& B_sf, const Tensor& alpha) { return matmul<Gemm::ADA_MXF4>(A, B, A_sf, B_sf, alpha); }
Tensor matmul_nvf4_bf16_tn(const Tensor& A, const Tensor& B, const Tensor& A_sf, const Tensor& B_sf, const Tensor& alpha) { return matmul<Gemm::NVF4>(A, B, A_sf, B_sf, alpha); }
Tensor matmul_mxf8_bf16_tn(const Tensor& A, const Tensor& B, const Tensor& A_sf, const Tensor& B_sf, const Tensor& alpha) { return matmul<Gemm::MXF8_TN>(A, B, A_sf, B_sf, alpha); }
Tensor matmul_mxf8_bf16_nn(const Tensor& A, const Tensor& B, const Tensor& A_sf, const Tensor& B_sf, const Tensor& alpha) { return matmul<Gemm::MXF8_NN>(A, B, A_sf, B_sf, alpha); }

// ---- fused rotate + quantize ---------------------------------------------------------------------------------------
int64_t nbytes(const Tensor& t) { return t.numel() * (int64_t)t.element_size(); }

void quant_prologue(const char* op, const Tensor& A, const Tensor& R) {
  STD_TORCH_CHECK(has_dtype(A, ScalarType::BFloat16), "A must be bf16");
  STD_TORCH_CHECK(has_dtype(R, ScalarType::BFloat16), "B must be bf16");
  (void)op;
}

// mask == nullptr: plain variant (rotation 32 / 64 / 128); mask != nullptr: Quest with clip mask (rotation 32 only)
void quantize_mx(const char* op, const Tensor& A, const Tensor& R, Tensor& OUT, Tensor& OUT_sf, Tensor* OUT_mask, int method) {
  if (OUT_mask) {
    require_contiguous(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}, {*OUT_mask, "OUT_mask"}});
    require_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}, {*OUT_mask, "OUT_mask"}});
    require_same_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}, {*OUT_mask, "OUT_mask"}});
  } else {
    require_contiguous(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}});
    require_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}});
    require_same_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}});
  }
  quant_prologue(op, A, R);
  STD_TORCH_CHECK(R.dim() == 2 && R.size(0) == R.size(1), "Rotation matrix must be square");
  const int64_t rot = R.size(0), numel = A.numel();
  STD_TORCH_CHECK(numel % rot == 0, "A must be divisible by", rot);
  if (OUT_mask) {
    STD_TORCH_CHECK(rot == 32, "Unsupported rotation size ", rot, "; expected 32.");
  } else {
    STD_TORCH_CHECK(rot == 32 || rot == 64 || rot == 128, "Unsupported rotation size ", rot, "; expected 32, 64, or 128.");
  }
  // the C ABI writes numel/2, numel/32 (and numel/8) bytes: the caller's buffers must hold them
  STD_TORCH_CHECK(nbytes(OUT) >= numel / 2, "OUT is too small");
  STD_TORCH_CHECK(nbytes(OUT_sf) >= numel / 32, "OUT_sf is too small");
  if (OUT_mask) {
    STD_TORCH_CHECK(nbytes(*OUT_mask) >= numel / 8, "OUT_mask is too small");
  }
  const torch::stable::accelerator::DeviceGuard guard(A.get_device_index());
  check_rc(qutlass_amd_fused_quantize_mx(A.data_ptr(), R.data_ptr(), (int)rot, numel, method, OUT.data_ptr(), OUT_sf.data_ptr(),
                                         OUT_mask ? OUT_mask->data_ptr() : nullptr, current_stream(A)));
}

std::tuple<Tensor, Tensor> fusedQuantizeMxQuest(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf) {
  quantize_mx("fusedQuantizeMxQuest", A, R, OUT, OUT_sf, nullptr, QAMD_METHOD_QUEST);
  return {OUT, OUT_sf};
}
std::tuple<Tensor, Tensor> fusedQuantizeMxAbsMax(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf) {
  quantize_mx("fusedQuantizeMxAbsMax", A, R, OUT, OUT_sf, nullptr, QAMD_METHOD_ABSMAX);
  return {OUT, OUT_sf};
}
std::tuple<Tensor, Tensor, Tensor> fusedQuantizeMxQuestWithMask(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf, Tensor OUT_mask) {
  quantize_mx("fusedQuantizeMxQuestWithMask", A, R, OUT, OUT_sf, &OUT_mask, QAMD_METHOD_QUEST);
  return {OUT, OUT_sf, OUT_mask};
}

void quantize_nv(const char* op, const Tensor& A, const Tensor& R, Tensor& OUT, Tensor& OUT_sf, const Tensor& gscale, int method) {
  require_contiguous(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}});
  require_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}, {gscale, "global_scale"}});
  require_same_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}, {gscale, "global_scale"}});
  quant_prologue(op, A, R);
  STD_TORCH_CHECK(has_dtype(gscale, ScalarType::Float), "global_scale must be float");
  STD_TORCH_CHECK(gscale.dim() == 1 && gscale.size(0) == 1, "global_scale must be a scalar");
  STD_TORCH_CHECK(R.dim() == 2 && R.size(0) == R.size(1), "Rotation matrix must be square");
  const int64_t rot = R.size(0), numel = A.numel();
  STD_TORCH_CHECK(numel % rot == 0, "A must be divisible by", rot);
  STD_TORCH_CHECK(rot == 16 || rot == 32 || rot == 64 || rot == 128, "Unsupported rotation size ", rot, "; expected 16, 32, 64, or 128.");
  STD_TORCH_CHECK(nbytes(OUT) >= numel / 2, "OUT is too small");
  STD_TORCH_CHECK(nbytes(OUT_sf) >= numel / 16, "OUT_sf is too small");
  const torch::stable::accelerator::DeviceGuard guard(A.get_device_index());
  check_rc(qutlass_amd_fused_quantize_nv(A.data_ptr(), R.data_ptr(), (int)rot, numel, method, static_cast<const float*>(gscale.data_ptr()),
                                         OUT.data_ptr(), OUT_sf.data_ptr(), current_stream(A)));
}

std::tuple<Tensor, Tensor> fusedQuantizeNvQuest(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf, const Tensor& global_scale) {
  quantize_nv("fusedQuantizeNvQuest", A, R, OUT, OUT_sf, global_scale, QAMD_METHOD_QUEST);
  return {OUT, OUT_sf};
}
std::tuple<Tensor, Tensor> fusedQuantizeNvAbsMax(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf, const Tensor& global_scale) {
  quantize_nv("fusedQuantizeNvAbsMax", A, R, OUT, OUT_sf, global_scale, QAMD_METHOD_ABSMAX);
  return {OUT, OUT_sf};
}

// ---- in-place twins of the reference's output-filling ops, schemas with the mutation declared (`Tensor(a!)`) ------------------------
// The reference's schemas (bindings.cpp:504-513, kept verbatim above) declare neither that the quantizers write OUT / OUT_sf and return
// aliases of them, nor that the QAT-backward ops fill their last arguments.  Eager dispatch does not care; AOTAutograd / inductor do: an
// undeclared write is dead code to them (the call is removed, or OUT's storage reused while the returned alias is live).  The Python
// wrappers (qutlass_amd/__init__.py) therefore call these twins -- same checks, same C-ABI call -- and hand the caller the tensors they
// allocated; the `_qutlass_C` ops stay for callers that use them directly (eager), WITHOUT fake kernels, so tracing them fails loudly.
void fusedQuantizeMx_(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf, int64_t method) {
  STD_TORCH_CHECK(method == QAMD_METHOD_QUEST || method == QAMD_METHOD_ABSMAX, "method must be 0 (quest) or 1 (abs_max)");
  quantize_mx(method == QAMD_METHOD_QUEST ? "fusedQuantizeMxQuest" : "fusedQuantizeMxAbsMax", A, R, OUT, OUT_sf, nullptr, (int)method);
}
void fusedQuantizeMxMask_(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf, Tensor OUT_mask) {
  quantize_mx("fusedQuantizeMxQuestWithMask", A, R, OUT, OUT_sf, &OUT_mask, QAMD_METHOD_QUEST);
}
void fusedQuantizeNv_(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf, const Tensor& global_scale, int64_t method) {
  STD_TORCH_CHECK(method == QAMD_METHOD_QUEST || method == QAMD_METHOD_ABSMAX, "method must be 0 (quest) or 1 (abs_max)");
  quantize_nv(method == QAMD_METHOD_QUEST ? "fusedQuantizeNvQuest" : "fusedQuantizeNvAbsMax", A, R, OUT, OUT_sf, global_scale, (int)method);
}

// ---- EXTENSION: quantizers that emit GEMM-ready (to_blocked-layout) scales: one launch instead of quantize + to_blocked ----------
// A is (.., K); OUT_sf must hold the padded blocked matrix of the (numel / K, K / gs) scales.  method: 0 quest, 1 abs_max.
void fusedQuantizeMxBlocked(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf, int64_t method) {
  const char* op = "fusedQuantizeMxBlocked";
  require_contiguous(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}});
  require_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}});
  require_same_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}});
  quant_prologue(op, A, R);
  STD_TORCH_CHECK(R.dim() == 2 && R.size(0) == R.size(1), "Rotation matrix must be square");
  STD_TORCH_CHECK(A.dim() >= 1 && A.numel() > 0, "A must be a non-empty tensor");
  const int64_t rot = R.size(0), numel = A.numel(), k = A.size(A.dim() - 1), rows = numel / k;
  STD_TORCH_CHECK(rot == 32 || rot == 64 || rot == 128, "Unsupported rotation size ", rot, "; expected 32, 64, or 128.");
  STD_TORCH_CHECK(k % rot == 0, "the last dimension of A must be divisible by", rot);
  STD_TORCH_CHECK(nbytes(OUT) >= numel / 2, "OUT is too small");
  STD_TORCH_CHECK(nbytes(OUT_sf) >= (rows + 127) / 128 * 128 * ((k / 32 + 3) / 4 * 4), "OUT_sf is too small for the blocked scale layout");
  const torch::stable::accelerator::DeviceGuard guard(A.get_device_index());
  check_rc(qutlass_amd_fused_quantize_mx_blocked(A.data_ptr(), R.data_ptr(), (int)rot, rows, k, (int)method, OUT.data_ptr(), OUT_sf.data_ptr(), nullptr,
                                                 current_stream(A)));
}

void fusedQuantizeNvBlocked(const Tensor& A, const Tensor& R, Tensor OUT, Tensor OUT_sf, const Tensor& gscale, int64_t method) {
  const char* op = "fusedQuantizeNvBlocked";
  require_contiguous(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}});
  require_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}, {gscale, "global_scale"}});
  require_same_gpu(op, {{A, "A"}, {R, "B"}, {OUT, "OUT"}, {OUT_sf, "OUT_sf"}, {gscale, "global_scale"}});
  quant_prologue(op, A, R);
  STD_TORCH_CHECK(has_dtype(gscale, ScalarType::Float), "global_scale must be float");
  STD_TORCH_CHECK(gscale.dim() == 1 && gscale.size(0) == 1, "global_scale must be a scalar");
  STD_TORCH_CHECK(R.dim() == 2 && R.size(0) == R.size(1), "Rotation matrix must be square");
  STD_TORCH_CHECK(A.dim() >= 1 && A.numel() > 0, "A must be a non-empty tensor");
  const int64_t rot = R.size(0), numel = A.numel(), k = A.size(A.dim() - 1), rows = numel / k;
  STD_TORCH_CHECK(rot == 16 || rot == 32 || rot == 64 || rot == 128, "Unsupported rotation size ", rot, "; expected 16, 32, 64, or 128.");
  STD_TORCH_CHECK(k % (rot < 32 ? 32 : rot) == 0, "the last dimension of A must be divisible by", rot < 32 ? 32 : rot);
  STD_TORCH_CHECK(nbytes(OUT) >= numel / 2, "OUT is too small");
  STD_TORCH_CHECK(nbytes(OUT_sf) >= (rows + 127) / 128 * 128 * ((k / 16 + 3) / 4 * 4), "OUT_sf is too small for the blocked scale layout");
  const torch::stable::accelerator::DeviceGuard guard(A.get_device_index());
  check_rc(qutlass_amd_fused_quantize_nv_blocked(A.data_ptr(), R.data_ptr(), (int)rot, rows, k, (int)method, static_cast<const float*>(gscale.data_ptr()),
                                                 OUT.data_ptr(), OUT_sf.data_ptr(), current_stream(A)));
}

// ---- EXTENSION: rotate + quantize + MXFP4 GEMM in one launch for decode batches (M <= 32) ---------------------------------------
Tensor fusedQuantizeMatmulMxf4(const Tensor& X, const Tensor& R, const Tensor& B, const Tensor& B_sf, const Tensor& alpha, int64_t method) {
  const char* op = "fusedQuantizeMatmulMxf4";
  require_contiguous(op, {{X, "X"}, {R, "R"}, {B, "B"}, {B_sf, "B_sf"}});
  require_gpu(op, {{X, "X"}, {R, "R"}, {B, "B"}, {B_sf, "B_sf"}, {alpha, "alpha"}});
  require_same_gpu(op, {{X, "X"}, {R, "R"}, {B, "B"}, {B_sf, "B_sf"}, {alpha, "alpha"}});
  STD_TORCH_CHECK(has_dtype(X, ScalarType::BFloat16) && has_dtype(R, ScalarType::BFloat16), "X and R must be bf16");
  STD_TORCH_CHECK(has_dtype(B, ScalarType::Byte), "B must be uint8");
  STD_TORCH_CHECK(has_dtype(B_sf, ScalarType::Float8_e8m0fnu), "B_sf must be float8_e8m0fnu");
  STD_TORCH_CHECK(has_dtype(alpha, ScalarType::Float) && alpha.numel() >= 1, "alpha must be a float32 tensor with at least one element");
  STD_TORCH_CHECK(X.dim() >= 1 && B.dim() == 2, "X must be at least 1-D and B 2-D");
  STD_TORCH_CHECK(R.dim() == 2 && R.size(0) == R.size(1), "Rotation matrix must be square");
  const int64_t K = X.size(X.dim() - 1), M = K > 0 ? X.numel() / K : 0, N = B.size(0);
  STD_TORCH_CHECK(B.size(1) * 2 == K, "Inner dimensions must match for Q(X) @ B.T");
  STD_TORCH_CHECK(B_sf.numel() >= (N + 127) / 128 * 128 * ((K / 32 + 3) / 4 * 4), "B_sf is too small for the blocked scale layout of B");
  Tensor out = torch::stable::new_empty(X, {M, N}, ScalarType::BFloat16);
  if (M == 0 || N == 0) return out;
  const torch::stable::accelerator::DeviceGuard guard(X.get_device_index());
  check_rc(qutlass_amd_fused_quantize_matmul_mxf4_bf16_tn(X.data_ptr(), R.data_ptr(), (int)R.size(0), (int)method, B.data_ptr(), B_sf.data_ptr(),
                                                          static_cast<const float*>(alpha.data_ptr()), out.data_ptr(), M, N, K, current_stream(X)));
  return out;
}

// ---- QAT-backward data preparation (bindings.cpp:429-494: no validation there; the Python wrappers assert dtypes and
//      contiguity, qutlass/__init__.py:206-315 -- the C ABI checks the shape constraints) --------------------------------
void backward_t_bf16(const Tensor& x, const Tensor& h, Tensor xh_e2m1, Tensor xh_e8m0) {
  const char* op = "backward_t_bf16";
  require_contiguous(op, {{x, "x"}, {h, "h"}, {xh_e2m1, "xh_e2m1"}, {xh_e8m0, "xh_e8m0"}});
  require_gpu(op, {{x, "x"}, {h, "h"}, {xh_e2m1, "xh_e2m1"}, {xh_e8m0, "xh_e8m0"}});
  STD_TORCH_CHECK(has_dtype(x, ScalarType::BFloat16) && has_dtype(h, ScalarType::BFloat16), "x and h must be bf16");
  STD_TORCH_CHECK(x.dim() >= 2 && h.numel() == 32 * 32, "x must be at least 2-D and h 32 x 32");
  const int64_t M = x.size(x.dim() - 1), N = x.size(x.dim() - 2), B = x.numel() / (M * N);
  STD_TORCH_CHECK(nbytes(xh_e2m1) >= B * M * N / 2 && nbytes(xh_e8m0) >= B * M * N / 32, "output tensors are too small");
  const torch::stable::accelerator::DeviceGuard guard(x.get_device_index());
  check_rc(qutlass_amd_backward_t_bf16(x.data_ptr(), h.data_ptr(), B, N, M, xh_e2m1.data_ptr(), xh_e8m0.data_ptr(), current_stream(x)));
}

void backward_qt_bf16(const Tensor& x_e2m1, const Tensor& x_e8m0, const Tensor& h, const Tensor& alpha, Tensor xh_e2m1, Tensor xh_e8m0) {
  const char* op = "backward_qt_bf16";
  require_contiguous(op, {{x_e2m1, "x_e2m1"}, {x_e8m0, "x_e8m0"}, {h, "h"}, {xh_e2m1, "xh_e2m1"}, {xh_e8m0, "xh_e8m0"}});
  require_gpu(op, {{x_e2m1, "x_e2m1"}, {x_e8m0, "x_e8m0"}, {h, "h"}, {alpha, "alpha"}, {xh_e2m1, "xh_e2m1"}, {xh_e8m0, "xh_e8m0"}});
  STD_TORCH_CHECK(has_dtype(h, ScalarType::BFloat16) && has_dtype(alpha, ScalarType::Float), "h must be bf16 and alpha float");
  STD_TORCH_CHECK(x_e2m1.dim() >= 2 && x_e2m1.element_size() == 1 && x_e8m0.element_size() == 1 && h.numel() == 32 * 32,
                  "x_e2m1 / x_e8m0 must be 1-byte tensors of at least 2 dimensions and h 32 x 32");
  const int64_t M = x_e2m1.size(x_e2m1.dim() - 1) * 2, N = x_e2m1.size(x_e2m1.dim() - 2), B = x_e2m1.numel() * 2 / (M * N);
  STD_TORCH_CHECK(x_e8m0.numel() == B * N * M / 32, "x_e8m0 must hold one scale per 32 elements of x_e2m1");
  STD_TORCH_CHECK(nbytes(xh_e2m1) >= B * M * N / 2 && nbytes(xh_e8m0) >= B * M * N / 32, "output tensors are too small");
  const torch::stable::accelerator::DeviceGuard guard(h.get_device_index());
  check_rc(qutlass_amd_backward_qt_bf16(x_e2m1.data_ptr(), x_e8m0.data_ptr(), h.data_ptr(), static_cast<const float*>(alpha.data_ptr()), B, N, M,
                                        xh_e2m1.data_ptr(), xh_e8m0.data_ptr(), current_stream(h)));
}

void backward_bf16_square_double_mxfp8(const Tensor& x_bf16, Tensor x_fp8, Tensor row_scales, Tensor column_scales) {
  const char* op = "backward_bf16_square_double_mxfp8";
  require_contiguous(op, {{x_bf16, "x_bf16"}, {x_fp8, "x_fp8"}, {row_scales, "row_scales"}, {column_scales, "column_scales"}});
  require_gpu(op, {{x_bf16, "x_bf16"}, {x_fp8, "x_fp8"}, {row_scales, "row_scales"}, {column_scales, "column_scales"}});
  STD_TORCH_CHECK(has_dtype(x_bf16, ScalarType::BFloat16) && x_bf16.dim() == 2, "x_bf16 must be a 2-D bf16 tensor");
  // x_bf16 may have any row count m: the outputs carry the padded extent m_pad = x_fp8.size(0) (a multiple of 128 >= m) and the kernel
  // treats the missing rows as zeros (the reference pads x_bf16 with a copy before the call, qutlass/__init__.py:288-290)
  STD_TORCH_CHECK(x_fp8.dim() == 2 && x_fp8.size(1) == x_bf16.size(1), "x_fp8 must be (m_pad, n)");
  const int64_t m = x_bf16.size(0), n = x_bf16.size(1), m_pad = x_fp8.size(0);
  STD_TORCH_CHECK(m_pad >= m && m_pad % 128 == 0, "x_fp8 must have a multiple of 128 rows, at least as many as x_bf16");
  STD_TORCH_CHECK(nbytes(row_scales) >= m_pad * n / 32 && nbytes(column_scales) >= m_pad * n / 32, "output tensors are too small");
  const torch::stable::accelerator::DeviceGuard guard(x_bf16.get_device_index());
  check_rc(qutlass_amd_backward_bf16_square_double_mxfp8_rows(x_bf16.data_ptr(), m, m_pad, n, x_fp8.data_ptr(), row_scales.data_ptr(), column_scales.data_ptr(),
                                                              current_stream(x_bf16)));
}

void mxfp4_transpose_mxfp8(const Tensor& x_fp4, const Tensor& scales, Tensor x_fp8, Tensor shared_exps) {
  const char* op = "mxfp4_transpose_mxfp8";
  require_contiguous(op, {{x_fp4, "x_fp4"}, {scales, "scales"}, {x_fp8, "x_fp8"}, {shared_exps, "shared_exps"}});
  require_gpu(op, {{x_fp4, "x_fp4"}, {scales, "scales"}, {x_fp8, "x_fp8"}, {shared_exps, "shared_exps"}});
  STD_TORCH_CHECK(x_fp4.dim() == 2 && x_fp4.element_size() == 1 && scales.element_size() == 1, "x_fp4 must be a 2-D 1-byte tensor, scales 1-byte");
  // any row count m: x_fp8 is (n, m_pad) with m_pad a multiple of 128 >= m; rows m .. m_pad-1 count as zero codes with unit scales
  // inside the kernel -- `scales` is read-only and needs only its m real rows (the reference pads x_fp4 with a copy and writes 1.0 into
  // the caller's scale tensor, qutlass/__init__.py:299-307)
  const int64_t m = x_fp4.size(0), n = x_fp4.size(1) * 2;
  STD_TORCH_CHECK(x_fp8.dim() == 2 && x_fp8.size(0) == n, "x_fp8 must be (n, m_pad)");
  const int64_t m_pad = x_fp8.size(1);
  STD_TORCH_CHECK(m_pad >= m && m_pad % 128 == 0, "x_fp8 must have a multiple of 128 columns, at least as many as x_fp4 has rows");
  STD_TORCH_CHECK(scales.numel() >= m * n / 32, "scales must hold one e8m0 per 32 elements");
  STD_TORCH_CHECK(nbytes(shared_exps) >= m_pad * n / 32, "output tensors are too small");
  const torch::stable::accelerator::DeviceGuard guard(x_fp4.get_device_index());
  check_rc(qutlass_amd_mxfp4_transpose_mxfp8_rows(x_fp4.data_ptr(), scales.data_ptr(), m, m_pad, n, x_fp8.data_ptr(), shared_exps.data_ptr(), current_stream(x_fp4)));
}

// ---- block-scale swizzle -------------------------------------------------------------------------------------------
Tensor to_blocked(const Tensor& in) {
  STD_TORCH_CHECK(in.dim() == 2, "to_blocked expects a 2-D matrix");
  STD_TORCH_CHECK(in.element_size() == 1, "Expected element size to be 1 byte (8 bits)");
  STD_TORCH_CHECK(in.is_contiguous(), "Input tensor must be contiguous");
  STD_TORCH_CHECK(in.is_cuda(), "to_blocked: expected a GPU tensor (no CPU path in qutlass_amd)");
  const int64_t rows = in.size(0), cols = in.size(1);
  const int64_t pr = (rows + 127) / 128 * 128, pc = (cols + 3) / 4 * 4;
  Tensor out = torch::stable::new_empty(in, {pr * pc});
  const torch::stable::accelerator::DeviceGuard guard(in.get_device_index());
  check_rc(qutlass_amd_to_blocked(in.data_ptr(), rows, cols, out.data_ptr(), current_stream(in)));
  return out;
}

}  // namespace

// Schema strings: bindings.cpp:499-513.  FRAGMENT, as in the reference (bindings.cpp:498): another library may add to `_qutlass_C`.
STABLE_TORCH_LIBRARY_FRAGMENT(_qutlass_C, m) {
  m.def("matmul_mxf4_bf16_tn(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor");
  m.def("matmul_nvf4_bf16_tn(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor");
  m.def("matmul_ada_mxf4_bf16_tn(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor");
  m.def("matmul_mxf8_bf16_tn(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor");
  m.def("matmul_mxf8_bf16_nn(Tensor A, Tensor B, Tensor A_sf, Tensor B_sf, Tensor alpha) -> Tensor");
  m.def("fusedQuantizeMxQuest(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf) -> (Tensor, Tensor)");
  m.def("fusedQuantizeMxAbsMax(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf) -> (Tensor, Tensor)");
  m.def("fusedQuantizeNvQuest(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf, Tensor global_scale) -> (Tensor, Tensor)");
  m.def("fusedQuantizeNvAbsMax(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf, Tensor global_scale) -> (Tensor, Tensor)");
#ifndef QUTLASS_MINIMAL_BUILD   // the reference's trimmed build (bindings.cpp:254, :428, :508): inference ops only -- no clip-mask quantizer, no QAT-backward data prep
  m.def("fusedQuantizeMxQuestWithMask(Tensor A, Tensor R, Tensor OUT, Tensor OUT_sf, Tensor OUT_mask) -> (Tensor, Tensor, Tensor)");
  m.def("backward_t_bf16(Tensor x, Tensor h, Tensor xh_e2m1, Tensor xh_e8m0) -> ()");
  m.def("backward_qt_bf16(Tensor x_e2m1, Tensor x_e8m0, Tensor h, Tensor alpha, Tensor xh_e2m1, Tensor xh_e8m0) -> ()");
  m.def("backward_bf16_square_double_mxfp8(Tensor x_bf16, Tensor x_fp8, Tensor row_scales, Tensor column_scales) -> ()");
  m.def("mxfp4_transpose_mxfp8(Tensor x_fp4, Tensor scales, Tensor x_fp8, Tensor shared_exps) -> ()");
#endif
}

STABLE_TORCH_LIBRARY_FRAGMENT(qutlass_amd, m) {
  m.def("to_blocked(Tensor input_matrix) -> Tensor");
  // every op that fills caller tensors says so: `Tensor(a!)`, no aliasing return (see the note above fusedQuantizeMx_)
  m.def("fusedQuantizeMxBlocked(Tensor A, Tensor R, Tensor(a!) OUT, Tensor(b!) OUT_sf, int method) -> ()");
  m.def("fusedQuantizeNvBlocked(Tensor A, Tensor R, Tensor(a!) OUT, Tensor(b!) OUT_sf, Tensor global_scale, int method) -> ()");
  m.def("fusedQuantizeMx_(Tensor A, Tensor R, Tensor(a!) OUT, Tensor(b!) OUT_sf, int method) -> ()");
  m.def("fusedQuantizeNv_(Tensor A, Tensor R, Tensor(a!) OUT, Tensor(b!) OUT_sf, Tensor global_scale, int method) -> ()");
#ifndef QUTLASS_MINIMAL_BUILD
  m.def("fusedQuantizeMxMask_(Tensor A, Tensor R, Tensor(a!) OUT, Tensor(b!) OUT_sf, Tensor(c!) OUT_mask) -> ()");
  m.def("backward_t_bf16_(Tensor x, Tensor h, Tensor(a!) xh_e2m1, Tensor(b!) xh_e8m0) -> ()");
  m.def("backward_qt_bf16_(Tensor x_e2m1, Tensor x_e8m0, Tensor h, Tensor alpha, Tensor(a!) xh_e2m1, Tensor(b!) xh_e8m0) -> ()");
  m.def("backward_bf16_square_double_mxfp8_(Tensor x_bf16, Tensor(a!) x_fp8, Tensor(b!) row_scales, Tensor(c!) column_scales) -> ()");
  m.def("mxfp4_transpose_mxfp8_(Tensor x_fp4, Tensor scales, Tensor(a!) x_fp8, Tensor(b!) shared_exps) -> ()");
#endif
  m.def("fusedQuantizeMatmulMxf4(Tensor X, Tensor R, Tensor B, Tensor B_sf, Tensor alpha, int method) -> Tensor");
}

// CUDA dispatch key only, as the reference (bindings.cpp:516-535); there is no CPU compute path.
STABLE_TORCH_LIBRARY_IMPL(_qutlass_C, CUDA, m) {
  m.impl("matmul_mxf4_bf16_tn", TORCH_BOX(&matmul_mxf4_bf16_tn));
  m.impl("matmul_nvf4_bf16_tn", TORCH_BOX(&matmul_nvf4_bf16_tn));
  m.impl("matmul_ada_mxf4_bf16_tn", TORCH_BOX(&matmul_ada_mxf4_bf16_tn));
  m.impl("matmul_mxf8_bf16_tn", TORCH_BOX(&matmul_mxf8_bf16_tn));
  m.impl("matmul_mxf8_bf16_nn", TORCH_BOX(&matmul_mxf8_bf16_nn));
  m.impl("fusedQuantizeMxQuest", TORCH_BOX(&fusedQuantizeMxQuest));
  m.impl("fusedQuantizeMxAbsMax", TORCH_BOX(&fusedQuantizeMxAbsMax));
  m.impl("fusedQuantizeNvQuest", TORCH_BOX(&fusedQuantizeNvQuest));
  m.impl("fusedQuantizeNvAbsMax", TORCH_BOX(&fusedQuantizeNvAbsMax));
#ifndef QUTLASS_MINIMAL_BUILD
  m.impl("fusedQuantizeMxQuestWithMask", TORCH_BOX(&fusedQuantizeMxQuestWithMask));
  m.impl("backward_t_bf16", TORCH_BOX(&backward_t_bf16));
  m.impl("backward_qt_bf16", TORCH_BOX(&backward_qt_bf16));
  m.impl("backward_bf16_square_double_mxfp8", TORCH_BOX(&backward_bf16_square_double_mxfp8));
  m.impl("mxfp4_transpose_mxfp8", TORCH_BOX(&mxfp4_transpose_mxfp8));
#endif
}
STABLE_TORCH_LIBRARY_IMPL(qutlass_amd, CUDA, m) {
  m.impl("to_blocked", TORCH_BOX(&to_blocked));
  m.impl("fusedQuantizeMxBlocked", TORCH_BOX(&fusedQuantizeMxBlocked));
  m.impl("fusedQuantizeNvBlocked", TORCH_BOX(&fusedQuantizeNvBlocked));
  m.impl("fusedQuantizeMx_", TORCH_BOX(&fusedQuantizeMx_));
  m.impl("fusedQuantizeNv_", TORCH_BOX(&fusedQuantizeNv_));
#ifndef QUTLASS_MINIMAL_BUILD
  m.impl("fusedQuantizeMxMask_", TORCH_BOX(&fusedQuantizeMxMask_));
  m.impl("backward_t_bf16_", TORCH_BOX(&backward_t_bf16));
  m.impl("backward_qt_bf16_", TORCH_BOX(&backward_qt_bf16));
  m.impl("backward_bf16_square_double_mxfp8_", TORCH_BOX(&backward_bf16_square_double_mxfp8));
  m.impl("mxfp4_transpose_mxfp8_", TORCH_BOX(&mxfp4_transpose_mxfp8));
#endif
  m.impl("fusedQuantizeMatmulMxf4", TORCH_BOX(&fusedQuantizeMatmulMxf4));
}

// `import qutlass._CUDA` (reference: include/registration.h REGISTER_EXTENSION(_CUDA), bindings.cpp:537-540): an empty module
// whose only purpose is that loading it runs the registrations above.
// (QUTLASS_MINIMAL_BUILD drops it with the reference, bindings.cpp:537-540: such a library is loaded with torch.ops.load_library, not imported.)
#ifndef QUTLASS_MINIMAL_BUILD
extern "C" __attribute__((visibility("default"))) PyObject* PyInit__CUDA(void) {
  static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_CUDA", nullptr, 0, nullptr};
  return PyModule_Create(&module);
}
#endif
