"""Build the in-tree binaries (no cmake, no JIT cache):

  libqutlass_amd.so        hipcc --offload-arch=gfx950: the hand-written HIP kernels behind the C ABI (no torch headers).
                           The PRODUCT: only the kernels the dispatch rules can reach, no kernel-selecting options.
  libqutlass_amd_bench.so  the same source with -DQAMD_BENCH=1: the LAB build with every schedule variant / ablation /
                           trace instantiation and the "gemm_variant"-style options.  Test and bench infrastructure
                           only (tests/native, tests/_benchlib.py, tools/); nothing under qutlass_amd/ loads it.
  qutlass/_CUDA.abi3.so    g++: the PyTorch extension (csrc/torch_ext.cpp, LibTorch stable ABI, no device code) that registers
                           torch.ops._qutlass_C.* over that C ABI -- same module name and entry point (PyInit__CUDA) as the
                           reference's op library (qutlass/csrc/bindings.cpp:537-540)
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "capi.hip")
OUT = os.path.join(_HERE, "libqutlass_amd.so")
BENCH_OUT = os.path.join(_HERE, "libqutlass_amd_bench.so")
EXT_SRC = os.path.join(_HERE, "csrc", "torch_ext.cpp")
EXT_OUT = os.path.join(os.path.dirname(_HERE), "qutlass", "_CUDA.abi3.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "qutlass_amd.h")


def _header() -> str:
    """include/qutlass_amd.h: next to the package in the source tree, inside it (qutlass_amd/include/) in an installed copy (setup.py ships it as package data)."""
    inpkg = os.path.join(_HERE, "include", "qutlass_amd.h")
    return HEADER if os.path.exists(HEADER) or not os.path.exists(inpkg) else inpkg


def _kernel_sources(lab: bool = False):
    """sources of libqutlass_amd.so; lab: + csrc/lab/ (read by the -DQAMD_BENCH=1 build only; an installed copy does not carry that directory)"""
    d = os.path.join(_HERE, "csrc")
    out = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f != "torch_ext.cpp" and os.path.isfile(os.path.join(d, f))] + [_header()]
    ld = os.path.join(d, "lab")
    if lab and os.path.isdir(ld):
        out += [os.path.join(ld, f) for f in sorted(os.listdir(ld))]
    return out


def _stale(out, sources) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)   # (a source that an installed copy does not carry cannot be newer than its build)


def needs_build() -> bool:
    return _stale(OUT, _kernel_sources()) or _stale(EXT_OUT, [EXT_SRC, _header(), OUT])


UNITS = [1, 2, 3, 4, 5, 8]              # translation units of csrc/capi.hip in the product build (QAMD_TU values; see the top of that file)
UNITS_BENCH = [1, 2, 3, 4, 5, 6, 7, 8]  # the lab build adds the ablation units 6 and 7
# Per-unit compiler flags.  Unit 5 holds the rotation quantizers (fused_quantize_kernel family, backward_t / backward_qt) and nothing else: their rotation MFMAs take their
# accumulators in VGPRs -- LLVM's default put them in AGPRs and copied all 16 back with v_accvgpr_read_b32 per 1024-element tile,
# a fifth of the VALU instructions of a kernel that is VALU-issue-bound at R = 32.  The GEMM units must keep
# the AGPR form (256 accumulator registers per lane), so this is not a global flag; a QAMD_SINGLE_TU build goes without it.
TU_FLAGS = {5: ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _compile_units(out: str, units: list, extra: list, verbose: bool) -> None:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"] + extra
    if os.environ.get("QAMD_SINGLE_TU"):
        cmd = base + ["-shared", "-Wl,-Bsymbolic", SRC, "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    with tempfile.TemporaryDirectory(prefix="qamd_build_") as tmp:
        objs = [os.path.join(tmp, f"capi_tu{i}.o") for i in units]
        cmds = [base + TU_FLAGS.get(i, []) + [f"-DQAMD_TU={i}", "-c", SRC, "-o", o] for i, o in zip(units, objs)]
        if verbose:
            print(" ".join(cmds[0]), f"   (x{len(units)}: QAMD_TU={units}, in parallel)")
        def run(cmd):
            rc = subprocess.run(cmd).returncode
            if rc != 0 and any(f in cmd for fl in TU_FLAGS.values() for f in fl):   # a compiler without the per-unit flag: build that unit plainly
                plain = [a for a in cmd if not any(a == f for fl in TU_FLAGS.values() for f in fl)]
                print("qutlass_amd.build: retrying without the per-unit compiler flags:", " ".join(plain[-6:]))
                rc = subprocess.run(plain).returncode
            return rc

        with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as ex:
            for rc, cmd in zip(ex.map(run, cmds), cmds):
                if rc != 0:
                    raise subprocess.CalledProcessError(rc, cmd)
        # -Bsymbolic: the product and the lab library export the same C names and may live in one process
        link = base + ["-shared", "-Wl,-Bsymbolic"] + objs + ["-o", out]
        if verbose:
            print(" ".join(link[:6]), "... -o", out)
        subprocess.check_call(link)


def build_kernels(force: bool = False, verbose: bool = False) -> str:
    """csrc/capi.hip is compiled once per QAMD_TU value, in parallel (each unit instantiates one kernel family), and the
    objects are linked into libqutlass_amd.so.  QAMD_SINGLE_TU=1 in the environment compiles it as one unit instead."""
    if force or _stale(OUT, _kernel_sources()):
        _compile_units(OUT, UNITS, [], verbose)
    return OUT


def build_bench_lib(force: bool = False, verbose: bool = False) -> str:
    """The lab library (test / bench infrastructure, see the module docstring)."""
    if force or _stale(BENCH_OUT, _kernel_sources(lab=True)):
        _compile_units(BENCH_OUT, UNITS_BENCH, ["-DQAMD_BENCH=1"], verbose)
    return BENCH_OUT


def build_extension(force: bool = False, verbose: bool = False, minimal: bool = False, out: str | None = None) -> str:
    """minimal: the reference's QUTLASS_MINIMAL_BUILD (bindings.cpp:254, :428, :508, :537) -- an op library with the inference ops only (no clip-mask quantizer, no
    QAT-backward data prep, no Python module entry: load it with torch.ops.load_library, or point QUTLASS_AMD_OP_LIBRARY at it).  Written to `out`
    (default qutlass/_CUDA_minimal.so), never over the full library."""
    if minimal:
        return _build_extension_to(out or os.path.join(os.path.dirname(EXT_OUT), "_CUDA_minimal.so"), ["-DQUTLASS_MINIMAL_BUILD"], verbose)
    if force or _stale(EXT_OUT, [EXT_SRC, _header(), OUT]):
        _build_extension_to(EXT_OUT, [], verbose)
        stale = os.path.join(_HERE, "_C.so")   # the round-1 name of the extension
        if os.path.exists(stale):
            os.remove(stale)
    return EXT_OUT


def _build_extension_to(ext_out: str, defines: list, verbose: bool) -> str:
    import sysconfig

    import torch

    tdir = os.path.dirname(torch.__file__)
    inc, lib = os.path.join(tdir, "include"), os.path.join(tdir, "lib")
    # Py_LIMITED_API: the module only needs PyModule_Create (abi3); the ops themselves use the LibTorch stable ABI
    rpaths = ["-Wl,-rpath,$ORIGIN/../qutlass_amd", "-Wl,-rpath," + lib]
    if os.path.dirname(os.path.abspath(ext_out)) != os.path.dirname(EXT_OUT):   # a copy outside the tree finds the kernel library by its absolute path
        rpaths.insert(0, "-Wl,-rpath," + _HERE)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-DUSE_ROCM", "-DTORCH_TARGET_VERSION=0x020a000000000000",
           "-DPy_LIMITED_API=0x03090000", EXT_SRC, "-I" + inc, "-I" + sysconfig.get_paths()["include"], "-o", ext_out,
           "-L" + lib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-L" + _HERE, "-lqutlass_amd"] + rpaths + defines
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return ext_out


def build(force: bool = False, verbose: bool = False) -> str:
    build_kernels(force, verbose)
    build_extension(force, verbose)
    if os.environ.get("QUTLASS_MINIMAL_BUILD"):   # in addition, never instead: the trimmed op library next to the full one
        build_extension(verbose=verbose, minimal=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
