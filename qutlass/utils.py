"""Drop-in alias of qutlass.utils -> qutlass_amd.utils."""
from qutlass_amd.utils import ceil_div, get_padded_shape_mx, get_padded_shape_nv, pad_to_block, to_blocked  # noqa: F401
