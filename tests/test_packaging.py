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
