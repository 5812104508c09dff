"""Build hook of the drop-in package: runs qutlass_amd/build.py (hipcc --offload-arch=gfx950 for the kernels, g++ for the LibTorch-stable-ABI
op library) before the Python files are collected, so that `pip install --no-build-isolation .` (or `-e .`) yields importable `qutlass`,
`qutlass._CUDA` and `qutlass_amd` from any working directory.  Reference counterpart: /root/reference/setup.py:142-180 (CUDAExtension
`qutlass._CUDA`, py_limited_api)."""
import importlib.util
import os

from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

ROOT = os.path.dirname(os.path.abspath(__file__))


def _native_build():
    spec = importlib.util.spec_from_file_location("_qutlass_amd_build", os.path.join(ROOT, "qutlass_amd", "build.py"))
    qbuild = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(qbuild)
    print("qutlass_amd: native build ->", qbuild.build(force=bool(os.environ.get("QAMD_FORCE_BUILD")), verbose=True))


class BuildPy(build_py):
    def run(self):
        _native_build()
        # the C header travels inside the package (qutlass_amd/include/): qutlass_amd.build looks for it there when the source tree is gone
        import shutil
        dst = os.path.join(ROOT, "qutlass_amd", "include")
        os.makedirs(dst, exist_ok=True)
        shutil.copy2(os.path.join(ROOT, "include", "qutlass_amd.h"), os.path.join(dst, "qutlass_amd.h"))
        super().run()


class Develop(develop):
    def run(self):
        _native_build()
        super().run()


try:   # PEP 660 editable installs (`pip install -e .` with a recent pip / setuptools) do not run `develop`: hook the command they do run
    from setuptools.command.editable_wheel import editable_wheel

    class EditableWheel(editable_wheel):
        def run(self):
            _native_build()
            super().run()

    _EXTRA_CMDS = {"editable_wheel": EditableWheel}
except ImportError:   # an old setuptools: `develop` is the editable path
    _EXTRA_CMDS = {}


setup(
    name="qutlass-amd",
    version="0.2.0",
    packages=["qutlass_amd", "qutlass"],
    package_data={"qutlass_amd": ["libqutlass_amd.so", "csrc/*", "include/*.h"], "qutlass": ["_CUDA.abi3.so"]},
    include_package_data=False,
    zip_safe=False,
    cmdclass={"build_py": BuildPy, "develop": Develop, **_EXTRA_CMDS},
)
