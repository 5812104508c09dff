"""Build the two in-tree binaries (no cmake, no JIT cache):

  libqutlass_amd.so   hipcc --offload-arch=gfx950: the hand-written HIP kernels behind the C ABI (no torch headers)
  _C.so               g++: the PyTorch extension (csrc/torch_ext.cpp, LibTorch stable ABI, no device code) that
                      registers torch.ops._qutlass_C.* over that C ABI
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "capi.hip")
OUT = os.path.join(_HERE, "libqutlass_amd.so")
EXT_SRC = os.path.join(_HERE, "csrc", "torch_ext.cpp")
EXT_OUT = os.path.join(_HERE, "_C.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "qutlass_amd.h")


def _kernel_sources():
    d = os.path.join(_HERE, "csrc")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f != "torch_ext.cpp"] + [HEADER]


def _stale(out, sources) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in sources)


def needs_build() -> bool:
    return _stale(OUT, _kernel_sources()) or _stale(EXT_OUT, [EXT_SRC, HEADER, OUT])


NUM_TU = 7   # translation units of csrc/capi.hip (QAMD_TU = 1..NUM_TU; see the comment at the top of that file)


def build_kernels(force: bool = False, verbose: bool = False) -> str:
    """csrc/capi.hip is compiled once per QAMD_TU value, in parallel (each unit instantiates one kernel family), and the
    objects are linked into libqutlass_amd.so.  QAMD_SINGLE_TU=1 in the environment compiles it as one unit instead."""
    if force or _stale(OUT, _kernel_sources()):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]
        if os.environ.get("QAMD_SINGLE_TU"):
            cmd = base + ["-shared", SRC, "-o", OUT]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            return OUT
        import tempfile
        from concurrent.futures import ThreadPoolExecutor

        with tempfile.TemporaryDirectory(prefix="qamd_build_") as tmp:
            objs = [os.path.join(tmp, f"capi_tu{i}.o") for i in range(1, NUM_TU + 1)]
            cmds = [base + [f"-DQAMD_TU={i}", "-c", SRC, "-o", o] for i, o in zip(range(1, NUM_TU + 1), objs)]
            if verbose:
                print(" ".join(cmds[0]), f"   (x{NUM_TU}: QAMD_TU=1..{NUM_TU}, in parallel)")
            with ThreadPoolExecutor(max_workers=min(NUM_TU, os.cpu_count() or 1)) as ex:
                for rc, cmd in zip(ex.map(lambda c: subprocess.run(c).returncode, cmds), cmds):
                    if rc != 0:
                        raise subprocess.CalledProcessError(rc, cmd)
            link = base + ["-shared"] + objs + ["-o", OUT]
            if verbose:
                print(" ".join(link[:6]), "... -o", OUT)
            subprocess.check_call(link)
    return OUT


def build_extension(force: bool = False, verbose: bool = False) -> str:
    if force or _stale(EXT_OUT, [EXT_SRC, HEADER, OUT]):
        import torch

        tdir = os.path.dirname(torch.__file__)
        inc, lib = os.path.join(tdir, "include"), os.path.join(tdir, "lib")
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DUSE_ROCM", "-DTORCH_TARGET_VERSION=0x020a000000000000",
               EXT_SRC, "-I" + inc, "-o", EXT_OUT, "-L" + lib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip",
               "-L" + _HERE, "-lqutlass_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return EXT_OUT


def build(force: bool = False, verbose: bool = False) -> str:
    build_kernels(force, verbose)
    build_extension(force, verbose)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
