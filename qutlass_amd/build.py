"""Build libqutlass_amd.so in-tree with hipcc for gfx950 (no torch headers, no cmake)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "capi.hip")
OUT = os.path.join(_HERE, "libqutlass_amd.so")


def _sources():
    d = os.path.join(_HERE, "csrc")
    return [os.path.join(d, f) for f in sorted(os.listdir(d))] + [
        os.path.join(os.path.dirname(_HERE), "include", "qutlass_amd.h")]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
