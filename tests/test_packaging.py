"""Drop-in packaging (reference: /root/reference/setup.py:142-180): `pip install --no-build-isolation .` must yield importable
`qutlass`, `qutlass._CUDA` and `qutlass_amd` for an unmodified caller in ANY working directory -- checked by installing into a prefix inside the
repo's build/ directory and importing from /tmp with only that prefix on the path.  CPU-only: loads the libraries and checks that the 14 ops of the
reference's schema list are registered; no compute call."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OPS = ["matmul_mxf4_bf16_tn", "matmul_nvf4_bf16_tn", "matmul_ada_mxf4_bf16_tn", "matmul_mxf8_bf16_tn", "matmul_mxf8_bf16_nn", "fusedQuantizeMxQuest",
       "fusedQuantizeMxAbsMax", "fusedQuantizeNvQuest", "fusedQuantizeNvAbsMax", "fusedQuantizeMxQuestWithMask", "backward_t_bf16", "backward_qt_bf16",
       "backward_bf16_square_double_mxfp8", "mxfp4_transpose_mxfp8"]


def test_pip_install_yields_importable_drop_in_from_any_cwd(tmp_path):
    prefix = os.path.join(ROOT, "build", "pkgtest")
    shutil.rmtree(prefix, ignore_errors=True)
    os.makedirs(prefix)
    r = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--no-index", "--prefix", prefix, ROOT],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    site = [d for d in glob.glob(os.path.join(prefix, "**", "*-packages"), recursive=True) if os.path.isdir(os.path.join(d, "qutlass"))]
    assert site, "no site directory with the qutlass package under the prefix"
    code = ("import os, torch, qutlass, qutlass._CUDA, qutlass_amd, qutlass.utils\n"
            f"assert os.path.dirname(qutlass.__file__).startswith({site[0]!r}) and os.path.dirname(qutlass_amd.__file__).startswith({site[0]!r})\n"
            f"for op in {OPS!r}: getattr(torch.ops._qutlass_C, op)\n"
            "assert callable(qutlass.fusedQuantizeMx) and callable(qutlass.matmul_mxf4_bf16_tn) and callable(qutlass.utils.to_blocked)\n"
            "print(qutlass_amd._lib.load().qutlass_amd_version().decode())\n")
    env = dict(os.environ, PYTHONPATH=site[0])
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "qutlass_amd" in r.stdout and "gfx950" in r.stdout


def test_minimal_op_library_registers_the_inference_ops_only(tmp_path):
    """[r5] The reference's QUTLASS_MINIMAL_BUILD (qutlass/csrc/bindings.cpp:254, :428, :508, :537): an op library without the clip-mask quantizer, the QAT-backward data-prep
    ops and the Python module entry.  Built into a scratch directory from the same source, loaded in a fresh interpreter through QUTLASS_AMD_OP_LIBRARY: the nine
    inference ops + the four extension ops are there (and trace under FakeTensorMode), the five training-only ops are not, and the file exports no PyInit__CUDA."""
    import subprocess
    import sys

    from qutlass_amd import build

    out = build.build_extension(minimal=True, out=str(tmp_path / "_CUDA_minimal.so"))
    syms = subprocess.run(["nm", "-D", "--defined-only", out], capture_output=True, text=True, check=True).stdout
    assert "PyInit__CUDA" not in syms
    code = r"""
import torch, qutlass_amd
C = torch.ops._qutlass_C
have = [n for n in ("matmul_mxf4_bf16_tn", "matmul_nvf4_bf16_tn", "matmul_ada_mxf4_bf16_tn", "matmul_mxf8_bf16_tn", "matmul_mxf8_bf16_nn", "fusedQuantizeMxQuest",
                    "fusedQuantizeMxAbsMax", "fusedQuantizeNvQuest", "fusedQuantizeNvAbsMax") if hasattr(C, n)]
gone = [n for n in ("fusedQuantizeMxQuestWithMask", "backward_t_bf16", "backward_qt_bf16", "backward_bf16_square_double_mxfp8", "mxfp4_transpose_mxfp8") if hasattr(C, n)]
ext = [n for n in ("to_blocked", "fusedQuantizeMxBlocked", "fusedQuantizeNvBlocked", "fusedQuantizeMatmulMxf4") if hasattr(torch.ops.qutlass_amd, n)]
from torch._subclasses.fake_tensor import FakeTensorMode
with FakeTensorMode():
    a = torch.empty(64, 128, dtype=torch.uint8, device="cuda"); sf = torch.empty(128 * 8, dtype=torch.float8_e8m0fnu, device="cuda")
    y = qutlass_amd.matmul_mxf4_bf16_tn(a, a, sf, sf, torch.empty(1, device="cuda"))
print(len(have), len(gone), len(ext), tuple(y.shape), y.dtype)
"""
    env = dict(os.environ, QUTLASS_AMD_OP_LIBRARY=out, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split("\n")[-2].strip() == "9 0 4 (64, 64) torch.bfloat16", r.stdout
