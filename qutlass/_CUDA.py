"""``import qutlass._CUDA`` registers ``torch.ops._qutlass_C`` in the reference (registration.h);
here the registration happens when qutlass_amd is imported -- this module only makes the import work."""
import qutlass_amd  # noqa: F401
