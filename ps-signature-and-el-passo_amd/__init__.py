"""MI355X-native (HIP / gfx950) hot path of the PS-signature / EL PASSO library.

The package holds only what the path needs: csrc/ (HIP kernels + the C-ABI of include/elpasso.h) and the host-side
binding.  Importing never touches the GPU; `elpasso.Context(...)` does, and fails loudly without one.
"""
from . import elpasso  # noqa: F401
from .elpasso import Context, ElpassoError, load_library, CURVE_BN254, CURVE_BLS12_381  # noqa: F401
