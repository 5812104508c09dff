"""Operator registration: ``torch.ops._qutlass_C.*`` (the ops of the reference's binding file ``qutlass/csrc/bindings.cpp``)
and ``torch.ops.qutlass_amd.to_blocked``.

The ops live in the in-tree C++ extension ``qutlass/_CUDA.abi3.so`` (``csrc/torch_ext.cpp``, LibTorch stable ABI, no device
code) -- the same module name the reference's op library has (bindings.cpp:537-540): argument validation with the
reference's order and messages, output allocation, current-stream lookup, then ONE call into the C ABI of
``libqutlass_amd.so`` (``include/qutlass_amd.h``), where the hand-written HIP kernels are.  No Python-side compute and no
fallback: if the extension is not built, loading it raises.
"""
from __future__ import annotations

import os

import torch

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(os.path.dirname(_HERE), "qutlass", "_CUDA.abi3.so")

_registered = False


def register_torch_ops() -> None:
    """Load the C++ extension, which registers ``_qutlass_C::*`` and ``qutlass_amd::to_blocked`` with the dispatcher
    (reference: bindings.cpp:498-535 + registration.h).  The file is the Python extension module ``qutlass._CUDA``; it is
    loaded by path here so that ``import qutlass_amd`` does not depend on the alias package, and ``import qutlass._CUDA``
    afterwards finds the same, already initialised library."""
    global _registered
    if _registered:
        return
    _lib.load()  # libqutlass_amd.so first (the extension links against it)
    if not os.path.exists(EXT_PATH):
        raise ImportError(
            f"{EXT_PATH} is missing: build the extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or python qutlass_amd/build.py). "
            "qutlass_amd has no CPU / eager fallback."
        )
    torch.ops.load_library(EXT_PATH)
    _registered = True


def to_blocked(input_matrix: torch.Tensor) -> torch.Tensor:
    """qutlass/utils.py:160-193 as a HIP kernel (csrc/to_blocked.hip.h) -> flat blocked byte vector, input dtype."""
    register_torch_ops()
    return torch.ops.qutlass_amd.to_blocked(input_matrix)
