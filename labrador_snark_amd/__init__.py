"""labrador_snark_amd — MI355X-native LaBRADOR prover hot path (R_q arithmetic).

HIP kernels for gfx950 behind the C ABI in include/labrador_hip.h; this package
is the thin host-side mirror of the reference's interface for that path.
"""
from ._lib import (LabradorError, LengthMismatch, NoDeviceError, SO_PATH, load)  # noqa: F401
from .api import (Context, HatVec, D, matvec_sum, polynomial_vec_inner_product)  # noqa: F401
