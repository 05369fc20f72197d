"""ctypes binding of liblabrador_hip.so (the C ABI in include/labrador_hip.h).

The product has no CPU fallback: if the shared object is missing this module
raises ImportError naming the build command, and if no GPU is usable
lab_ctx_create returns LAB_ERR_NO_DEVICE which `check()` raises as
`NoDeviceError`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "liblabrador_hip.so")

LAB_OK = 0
LAB_ERR_ARG = -1
LAB_ERR_LEN_MISMATCH = -2
LAB_ERR_RANGE = -3
LAB_ERR_HIP = -4
LAB_ERR_NO_DEVICE = -5
LAB_ERR_JL_REJECTED = -6
LAB_ERR_BPP_CHECK = -7
LAB_ERR_DECOMPOSE = -8

_STATUS_NAMES = {
    -1: "LAB_ERR_ARG", -2: "LAB_ERR_LEN_MISMATCH", -3: "LAB_ERR_RANGE", -4: "LAB_ERR_HIP",
    -5: "LAB_ERR_NO_DEVICE", -6: "LAB_ERR_JL_REJECTED", -7: "LAB_ERR_BPP_CHECK", -8: "LAB_ERR_DECOMPOSE",
}


class LabradorError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("%s (%d): %s" % (_STATUS_NAMES.get(code, "LAB_ERR"), code, msg))


class NoDeviceError(LabradorError):
    pass


class LengthMismatch(LabradorError, AssertionError):
    """The reference's `assert!(v1.len() == v2.len(), ...)` panic (src/util.rs:497-502)."""


class JLRejected(LabradorError):
    """Verifier::valid_projection returned false (src/verification.rs:568-579); the reference resamples Pi."""


class BppCheckFailed(LabradorError, AssertionError):
    """verify_b_prime_prime's assert! fired (src/verification.rs:550)."""


class lab_prove_input(C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("n", C.c_size_t), ("r", C.c_size_t), ("kappa", C.c_size_t), ("kappa1", C.c_size_t),
                ("kappa2", C.c_size_t), ("b1", C.c_int64), ("b2", C.c_int64), ("t1", C.c_int), ("t2", C.c_int),
                ("beta_bound", C.c_int64), ("S", C.c_void_p), ("phi", C.c_void_p), ("a", C.c_void_p),
                ("b_prime", C.c_uint64), ("A", C.c_void_p), ("a_off", C.c_size_t), ("B", C.c_void_p),
                ("b_off", C.c_size_t), ("C", C.c_void_p), ("c_off", C.c_size_t), ("Dm", C.c_void_p),
                ("d_off", C.c_size_t), ("Pi", C.c_void_p), ("psi", C.c_uint64), ("omega", C.c_void_p),
                ("alpha", C.c_void_p), ("beta", C.c_void_p), ("c", C.c_void_p), ("B8", C.c_void_p), ("A8", C.c_void_p), ("C8", C.c_void_p), ("D8", C.c_void_p), ("A8d", C.c_void_p), ("Pi2", C.c_void_p), ("crs_seed", C.c_char_p)]


class lab_transcript(C.Structure):
    _fields_ = [("u_1", C.c_void_p), ("projection_int", C.c_void_p), ("projection", C.c_void_p),
                ("b_prime_prime", C.c_void_p), ("u_2", C.c_void_p), ("z", C.c_void_p), ("t_i_all", C.c_void_p),
                ("g_mat", C.c_void_p), ("h_mat", C.c_void_p)]


class lab_info(C.Structure):
    _fields_ = [("q", C.c_uint64), ("nlimbs", C.c_int), ("primes", C.c_uint32 * 4), ("max_len", C.c_uint64),
                ("device", C.c_int)]


# name -> (restype, argtypes); also the list the "exports every symbol" test walks
SIGNATURES = {
    "lab_ctx_create": (C.c_int, [C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]),
    "lab_ctx_create_ex": (C.c_int, [C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "lab_ctx_destroy": (None, [C.c_void_p]),
    "lab_ctx_info": (C.c_int, [C.c_void_p, C.POINTER(lab_info)]),
    "lab_last_error": (C.c_char_p, [C.c_void_p]),
    "lab_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lab_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "lab_ctx_profile_begin": (C.c_int, [C.c_void_p]),
    "lab_ctx_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_double)]),
    "lab_ctx_profile_phase": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "lab_ctx_profile_kernel": (C.c_char_p, [C.c_void_p]),
    "lab_ctx_gemm_diag": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "lab_ctx_enable_digit_basis": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "lab_ctx_enable_digit_basis_mfma": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "lab_ctx_prove_flags": (C.c_int, [C.c_void_p]),
    "lab_ctx_digit_info": (C.c_int, [C.c_void_p, C.POINTER(lab_info)]),
    "lab_vec_create_digit": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "lab_vec_basis": (C.c_int, [C.c_void_p]),
    "lab_negacyclic_polymul_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "lab_ntt_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "lab_ntt_inverse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "lab_vec_create": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "lab_vec_destroy": (None, [C.c_void_p]),
    "lab_vec_count": (C.c_size_t, [C.c_void_p]),
    "lab_vec_load_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "lab_vec_store_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "lab_vec_fill_synthetic": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint64,
                                         C.c_uint64]),
    "lab_crs_expand": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_char_p, C.c_uint64, C.c_uint64]),
    "lab_crs_expand_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_uint64, C.c_uint64]),
    "lab_crs_matvec_mfma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                      C.c_void_p, C.c_size_t, C.c_int]),
    "lab_inner_product": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_size_t, C.c_size_t]),
    "lab_matvec_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                 C.c_size_t, C.c_void_p, C.c_size_t]),
    "lab_matvec_sum_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                         C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]),
    "lab_mat8_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]),
    "lab_mat8_destroy": (C.c_int, [C.c_void_p]),
    "lab_mat8_bytes": (C.c_size_t, [C.c_void_p]),
    "lab_matvec_sum_mfma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "lab_ajtai_commit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                   C.c_void_p, C.c_size_t, C.c_size_t]),
    "lab_gram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                           C.c_size_t, C.c_size_t]),
    "lab_axpy_commit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                  C.c_size_t, C.c_size_t]),
    "lab_axpy_commit_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]),
    "lab_amortize_z": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                 C.c_void_p, C.c_size_t]),
    "lab_pointwise": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                C.c_size_t, C.c_size_t]),
    "lab_poly_by_poly_vec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_size_t, C.c_size_t]),
    "lab_decompose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int]),
    "lab_norm_sq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lab_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]),
    "lab_sigma_inv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "lab_jl_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "lab_pi_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "lab_prove_core": (C.c_int, [C.c_void_p, C.POINTER(lab_prove_input), C.POINTER(lab_transcript)]),
    "lab_prove_core_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(lab_prove_input)),
                                       C.POINTER(C.POINTER(lab_transcript)), C.POINTER(C.c_int)]),
    "lab_verify_core": (C.c_int, [C.c_void_p, C.POINTER(lab_prove_input), C.POINTER(lab_transcript), C.c_void_p,
                                  C.c_int64, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Loads the shared object and binds every entry point declared in the header."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "labrador_snark_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C labrador_snark_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
    # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Import it FIRST so the dynamic loader
    # satisfies our NEEDED libamdhip64.so.7 with that already-loaded runtime: one HIP runtime per process, and
    # torch tensors' device pointers / streams are valid inside the library.  (Loaded the other way round the
    # process ends up with two runtimes and torch reports "No HIP GPUs are available".)
    import torch  # noqa: F401
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale: rebuild
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, rc):
    if rc == LAB_OK:
        return
    msg = ""
    if ctx_handle:
        m = load().lab_last_error(ctx_handle)
        msg = m.decode() if m else ""
    if rc == LAB_ERR_NO_DEVICE:
        raise NoDeviceError(rc, "no usable GPU; the product path has no CPU fallback")
    if rc == LAB_ERR_LEN_MISMATCH:
        raise LengthMismatch(rc, msg or "inner product not defined on vectors of unequal length")
    if rc == LAB_ERR_JL_REJECTED:
        raise JLRejected(rc, msg)
    if rc == LAB_ERR_BPP_CHECK:
        raise BppCheckFailed(rc, msg)
    raise LabradorError(rc, msg)
