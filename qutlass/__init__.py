"""Drop-in alias: ``import qutlass`` resolves to the MI355X-native implementation (qutlass_amd)."""
from qutlass_amd import *  # noqa: F401,F403
from qutlass_amd import (  # noqa: F401
    __version__, fusedQuantizeMx, fusedQuantizeNv, matmul_mxf4_bf16_tn, matmul_mxf8_bf16_nn,
    matmul_mxf8_bf16_tn, matmul_nvf4_bf16_tn, qutlass_CUDA,
)
from . import _CUDA, utils  # noqa: F401


def __getattr__(name):
    import qutlass_amd

    return getattr(qutlass_amd, name)
