"""labrador_snark_amd — MI355X-native LaBRADOR prover hot path (R_q arithmetic).

HIP kernels for gfx950 behind the C ABI in include/labrador_hip.h; this package
is the thin host-side mirror of the reference's interface for that path.
"""
from ._lib import (BppCheckFailed, JLRejected, LabradorError, LengthMismatch, NoDeviceError, SO_PATH,  # noqa: F401
                   load)
from .api import *  # noqa: F401,F403
from .api import Context, HatVec, D  # noqa: F401
from .prover import (BatchPlan, CRS, Prover, ResidentCRS, RuntimeConstants, State, Transcript, Verifier, fetch_challenge, gen_f_b, operator_norm, pack_pi, pack_pi_host, unpack_pi_host, prove_core, prove_core_batch,  # noqa: F401
                     verify_core)  # noqa: F401
