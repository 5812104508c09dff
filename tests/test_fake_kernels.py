"""Shape-only (fake / meta) kernels of the torch ops (qutlass_amd/ops.py `_register_fakes`): a caller of the operator surface traces under FakeTensorMode -- no GPU needed,
the fake tensors only CLAIM to live on one.  The reference's default `to_blocked` is plain torch meant to be compiled through (qutlass/utils.py:160-193); a custom op
needs a fake kernel to keep that property.  The GPU half (compiled == eager, bit for bit) is in tests/test_gpu_round5.py."""
import pytest
import torch
from torch._subclasses.fake_tensor import FakeTensorMode
from torch.fx.experimental.proxy_tensor import make_fx

import qutlass_amd as q
from qutlass_amd.utils import to_blocked

DEV = "cuda"


def _targets(gm):
    return [str(n.target) for n in gm.graph.nodes if n.op == "call_function"]


FILL_OPS = ("fusedQuantizeMxQuest", "fusedQuantizeMxAbsMax", "fusedQuantizeMxQuestWithMask", "fusedQuantizeNvQuest", "fusedQuantizeNvAbsMax",
            "backward_t_bf16", "backward_qt_bf16", "backward_bf16_square_double_mxfp8", "mxfp4_transpose_mxfp8")
TWINS = ("fusedQuantizeMx_", "fusedQuantizeNv_", "fusedQuantizeMxMask_", "fusedQuantizeMxBlocked", "fusedQuantizeNvBlocked",
         "backward_t_bf16_", "backward_qt_bf16_", "backward_bf16_square_double_mxfp8_", "mxfp4_transpose_mxfp8_")


def _fake_of(name):   # (torch.library.register_fake keeps its kernels in this registry; the fake-tensor machinery looks them up there)
    return torch._library.simple_registry.singleton.find(name).fake_impl.kernel


def test_fake_kernels_exist_exactly_where_the_schema_tells_the_truth():
    """[r6, ADVICE r5] The five GEMMs and every `qutlass_amd` op have a fake kernel.  The reference's nine output-filling `_qutlass_C` ops have NONE: their schemas
    (bindings.cpp:504-513, verbatim) hide the writes, so a traced graph would drop or mis-schedule the call -- tracing them must fail loudly.  Their twins declare
    every written argument `Tensor(a!)` and return nothing."""
    q.ops.register_torch_ops()
    for n in ("matmul_mxf4_bf16_tn", "matmul_nvf4_bf16_tn", "matmul_ada_mxf4_bf16_tn", "matmul_mxf8_bf16_tn", "matmul_mxf8_bf16_nn"):
        assert _fake_of(f"_qutlass_C::{n}") is not None, n
    for n in ("to_blocked", "fusedQuantizeMatmulMxf4") + TWINS:
        assert _fake_of(f"qutlass_amd::{n}") is not None, n
    for n in FILL_OPS:
        assert _fake_of(f"_qutlass_C::{n}") is None, n
        schema = getattr(torch.ops._qutlass_C, n).default._schema
        assert not any(a.alias_info is not None and a.alias_info.is_write for a in schema.arguments), n   # the reference's schema, untouched
    for n in TWINS:
        schema = getattr(torch.ops.qutlass_amd, n).default._schema
        written = [a.name for a in schema.arguments if a.alias_info is not None and a.alias_info.is_write]
        assert len(written) >= 2 and len(schema.returns) == 0, (n, str(schema))
    with FakeTensorMode():   # and tracing a hidden-write op raises instead of producing a wrong graph
        x = torch.empty(64, 128, dtype=torch.bfloat16, device=DEV)
        h = torch.empty(32, 32, dtype=torch.bfloat16, device=DEV)
        with pytest.raises(Exception):
            torch.ops._qutlass_C.fusedQuantizeMxAbsMax(x, h, torch.empty(64, 64, dtype=torch.uint8, device=DEV), torch.empty(128, 4, dtype=torch.float8_e8m0fnu, device=DEV))


def _aot_graphs(fn, *args):
    """Forward graphs AOTAutograd hands to a backend (functionalised, dead code eliminated) -- what `aot_eager` / inductor would run."""
    from torch._dynamo.backends.common import aot_autograd
    graphs = []

    def capture(gm, example_inputs):
        graphs.append(gm)
        return gm.forward

    torch._dynamo.reset()
    try:   # CPU tensors: tracing never looks at the device, and the run after it has no kernel to call (CUDA key only) -- the graphs exist by then
        torch.compile(fn, backend=aot_autograd(fw_compiler=capture), fullgraph=True)(*args)
    except (NotImplementedError, RuntimeError) as e:
        assert graphs and ("CPU" in str(e) or "backend" in str(e)), e
    return graphs


FUNCTIONAL = ("quantize_mx", "quantize_nv", "quantize_mx_blocked", "quantize_nv_blocked", "quantize_mx_mask", "backward_t", "backward_qt", "square_double_mxfp8", "transpose_mxfp8")


def test_functional_forms_exist_for_compiled_callers():
    q.ops.register_torch_ops()
    for n in FUNCTIONAL:
        schema = getattr(torch.ops.qutlass_amd, n).default._schema
        assert not any(a.alias_info is not None for a in schema.arguments) and len(schema.returns) >= 2, str(schema)


def test_output_filling_ops_survive_aot_functionalisation():
    """[r6, ADVICE r5 high x2] After AOTAutograd's functionalisation + dead-code elimination the quantizer / backward calls are still in the graph and the values
    returned to the caller come out of them -- with the round-5 fakes the backward calls were removed (the wrappers returned uninitialised torch.empty buffers) and the
    quantizers' outputs were not tied to the call.  Under torch.compile the wrappers call the FUNCTIONAL forms (ops.py `_define_functional_ops`)."""
    def fwd(x, h):
        xq, xs = q.fusedQuantizeMx(x, h, method="abs_max")
        return xq, to_blocked(xs)

    def bwd(x, h):
        return q.backward_t_bf16(x, h)

    def sq(x):
        return q.backward_bf16_square_double_mxfp8(x)

    if True:
        h = torch.zeros(32, 32, dtype=torch.bfloat16)
        for fn, args, twin in ((fwd, (torch.zeros(64, 128, dtype=torch.bfloat16), h), "quantize_mx"),
                               (bwd, (torch.zeros(2, 128, 256, dtype=torch.bfloat16), h), "backward_t"),
                               (sq, (torch.zeros(200, 256, dtype=torch.bfloat16),), "square_double_mxfp8")):
            (gm,) = _aot_graphs(fn, *args)
            calls = [n for n in gm.graph.nodes if n.op == "call_function" and twin in str(n.args[:1]) + str(n.target)]
            assert calls, (twin, gm.code)
            outs = [a for a in gm.graph.output_node().args[0] if a is not None]
            def feeds(node, seen=None):   # does `node` depend on the op call?
                seen = seen if seen is not None else set()
                if node in calls:
                    return True
                seen.add(node)
                return any(feeds(i, seen) for i in node.all_input_nodes if i not in seen)
            assert all(feeds(o) for o in outs), (twin, gm.code)


def test_to_blocked_traces_with_fullgraph():
    with FakeTensorMode():
        s = torch.empty(300, 10, dtype=torch.float8_e8m0fnu, device=DEV)
        out = torch.compile(lambda t: to_blocked(t), backend="eager", fullgraph=True)(s)
        assert out.shape == (384 * 12,) and out.dtype == torch.float8_e8m0fnu and out.device.type == "cuda"


def test_quantize_swizzle_gemm_traces_under_fake_tensors():
    def layer(x, h, wq, wsf, alpha):          # the forward path of a quantized linear layer: Q(x h) W^T
        xq, xs = q.fusedQuantizeMx(x, h, method="abs_max")
        return q.matmul_mxf4_bf16_tn(xq.view(-1, xq.size(-1)), wq, to_blocked(xs), wsf, alpha)

    def layer_nv(x, h, gs, wq, wsf, alpha):
        xq, xs = q.fusedQuantizeNv(x, h, gs)
        return q.matmul_nvf4_bf16_tn(xq.view(-1, xq.size(-1)), wq, to_blocked(xs), wsf, alpha)

    with FakeTensorMode():
        x = torch.empty(4, 64, 512, dtype=torch.bfloat16, device=DEV)
        h = torch.empty(32, 32, dtype=torch.bfloat16, device=DEV)
        wq = torch.empty(384, 256, dtype=torch.uint8, device=DEV)
        wsf = torch.empty(384 * 16, dtype=torch.float8_e8m0fnu, device=DEV)
        alpha = torch.empty(1, device=DEV)
        out = torch.compile(layer, backend="eager", fullgraph=True)(x, h, wq, wsf, alpha)
        assert out.shape == (256, 384) and out.dtype == torch.bfloat16
        gm = make_fx(layer)(x, h, wq, wsf, alpha)
        t = _targets(gm)
        assert "qutlass_amd.fusedQuantizeMx_.default" in t and "qutlass_amd.to_blocked.default" in t and "_qutlass_C.matmul_mxf4_bf16_tn.default" in t
        h16 = torch.empty(16, 16, dtype=torch.bfloat16, device=DEV)
        wsf_nv = torch.empty(384 * 32, dtype=torch.float8_e4m3fn, device=DEV)
        out = torch.compile(layer_nv, backend="eager", fullgraph=True)(x, h16, torch.empty(1, device=DEV), wq, wsf_nv, alpha)
        assert out.shape == (256, 384)


def test_remaining_ops_trace_under_fake_tensors():
    with FakeTensorMode():
        h = torch.empty(32, 32, dtype=torch.bfloat16, device=DEV)
        alpha = torch.empty(1, device=DEV)
        a8 = torch.empty(512, 256, dtype=torch.float8_e4m3fn, device=DEV)       # (K, M) for the NN op
        b8 = torch.empty(384, 512, dtype=torch.float8_e4m3fn, device=DEV)
        sf = torch.empty(4096, dtype=torch.float8_e8m0fnu, device=DEV)
        assert torch.compile(q.matmul_mxf8_bf16_nn, backend="eager", fullgraph=True)(a8, b8, sf, sf, alpha).shape == (256, 384)
        assert torch.compile(q.matmul_mxf8_bf16_tn, backend="eager", fullgraph=True)(b8, b8, sf, sf, alpha).shape == (384, 384)
        x = torch.empty(2, 128, 256, dtype=torch.bfloat16, device=DEV)
        e2m1, e8m0 = torch.compile(q.backward_t_bf16, backend="eager", fullgraph=True)(x, h)
        assert e2m1.shape == (2, 256, 64) and e8m0.shape == (2, 256, 4)
        y, rs, cs = torch.compile(q.backward_bf16_square_double_mxfp8, backend="eager", fullgraph=True)(torch.empty(200, 256, dtype=torch.bfloat16, device=DEV))
        assert y.shape == (256, 256) and rs.shape == (256, 8) and cs.shape == (256, 8)
        yt, st = torch.compile(q.mxfp4_transpose_mxfp8, backend="eager", fullgraph=True)(torch.empty(128, 128, dtype=torch.uint8, device=DEV),
                                                                                       torch.empty(128, 8, dtype=torch.float8_e8m0fnu, device=DEV))
        assert yt.shape == (256, 256) and st.shape == (256, 8)
        out = q.fusedQuantizeMx(torch.empty(64, 128, dtype=torch.bfloat16, device=DEV), h, method="quest", return_mask=True)
        assert len(out) == 3 and out[2].shape[-1] == 128 // 8


def test_inductor_compiles_callers_with_e8m0_results():
    """inductor (torch 2.10) does not lower nodes that touch `float8_e8m0fnu` tensors -- a MUTATING custom op with an e8m0 argument dies in its post-grad pass
    ("auto_functionalized_v2 was not removed"), a functional one is called as an extern kernel.  CPU tensors: compilation runs to the end, the run then finds no kernel
    (CUDA key only).  The GPU half (compiled == eager bytes) is tests/test_gpu_round6.py."""
    def layer(x, h, wq, wsf, alpha):
        xq, xs = q.fusedQuantizeMx(x, h, method="abs_max")
        y = q.matmul_mxf4_bf16_tn(xq.view(-1, xq.size(-1)), wq, to_blocked(xs), wsf, alpha)
        return y @ y.t(), xq

    def prep(g, h):
        a, b = q.backward_t_bf16(g, h)
        y, rs, cs = q.backward_bf16_square_double_mxfp8(g.view(-1, g.size(-1)))
        return a, b, y, rs, cs

    h = torch.zeros(32, 32, dtype=torch.bfloat16)
    cases = ((layer, (torch.zeros(2, 160, 512, dtype=torch.bfloat16), h, torch.zeros(384, 256, dtype=torch.uint8), torch.zeros(384 * 16, dtype=torch.float8_e8m0fnu), torch.ones(1))),
             (prep, (torch.zeros(2, 256, 384, dtype=torch.bfloat16), h)))
    for fn, args in cases:
        torch._dynamo.reset()
        with pytest.raises((NotImplementedError, RuntimeError), match="CPU"):
            torch.compile(fn, backend="inductor", fullgraph=True)(*args)
