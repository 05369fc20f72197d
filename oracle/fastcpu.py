"""ctypes binding of oracle/fastcpu.c — baseline B1 (the honest-fast OpenMP CPU prover) and the CPU arms of BASELINE
config 1.  Measurement / test infrastructure like the rest of oracle/: the product never imports it."""
import ctypes as C

import numpy as np

from . import coracle as co

D = 64


class FcProveIn(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("Dm", C.c_void_p)]


_bound = False


def lib():
    global _bound
    L = co.lib()
    if not _bound:
        u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
        u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
        L.fc_create.restype = C.c_void_p
        L.fc_create.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.fc_destroy.restype = None
        L.fc_destroy.argtypes = [C.c_void_p]
        L.fc_limbs.restype = C.c_int
        L.fc_limbs.argtypes = [C.c_void_p]
        L.fc_threads.restype = C.c_int
        L.fc_threads.argtypes = [C.c_void_p]
        L.fc_to_hat.restype = None
        L.fc_to_hat.argtypes = [C.c_void_p, u64p, C.c_size_t, u32p]
        L.fc_from_hat.restype = None
        L.fc_from_hat.argtypes = [C.c_void_p, u32p, C.c_size_t, u64p]
        L.fc_fill_synthetic_hat.restype = None
        L.fc_fill_synthetic_hat.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint64, u32p]
        L.fc_fill_random_hat.restype = None
        L.fc_fill_random_hat.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, u32p]
        L.fc_polymul.restype = None
        L.fc_polymul.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.fc_bench_polymul.restype = C.c_double
        L.fc_bench_polymul.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_uint64]
        L.fc_prove.restype = C.c_int
        L.fc_prove.argtypes = [C.c_void_p, C.POINTER(FcProveIn), C.POINTER(co.ProveOut)]
        _bound = True
    return L


class FastCpu:
    """One modulus; `threads` = OpenMP threads (0 = all the runtime offers)."""

    def __init__(self, q, max_len=0, threads=0):
        self._h = lib().fc_create(int(q), int(max_len), int(threads))
        if not self._h:
            raise ValueError("fc_create failed for q=%d" % q)
        self.q, self.L, self.threads = int(q), lib().fc_limbs(self._h), lib().fc_threads(self._h)

    def close(self):
        if self._h:
            lib().fc_destroy(self._h)
            self._h = None

    def to_hat(self, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, D)
        hat = np.empty((self.L, coeffs.shape[0], D), dtype=np.uint32)
        lib().fc_to_hat(self._h, coeffs, coeffs.shape[0], hat)
        return hat

    def from_hat(self, hat):
        out = np.empty((hat.shape[1], D), dtype=np.uint64)
        lib().fc_from_hat(self._h, np.ascontiguousarray(hat), hat.shape[1], out)
        return out

    def synthetic_hat(self, seed, count, bound=None, first_elem=0):
        hat = np.empty((self.L, count, D), dtype=np.uint32)
        lib().fc_fill_synthetic_hat(self._h, seed, first_elem, count, self.q if bound is None else bound, hat)
        return hat

    def random_hat(self, seed, count):
        """uniform residues directly in the transform domain (timing only: see fc_fill_random_hat)"""
        hat = np.empty((self.L, count, D), dtype=np.uint32)
        lib().fc_fill_random_hat(self._h, seed, count, hat)
        return hat

    def polymul(self, a, b):
        out = np.empty(D, dtype=np.uint64)
        lib().fc_polymul(self._h, np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64), out)
        return out

    def prove(self, case, A_hat, B_hat, C_hat, D_hat):
        """`case`: a coracle.ProverCase (its A, B, Cm, Dm arrays are ignored); *_hat: the CRS in the transform domain."""
        n, r = case.n, case.r
        o = {"t": np.zeros((r, case.kappa, D), np.uint64), "G": np.zeros((r, r, D), np.uint64),
             "H": np.zeros((r, r, D), np.uint64), "u1": np.zeros((case.kappa1, D), np.uint64),
             "u2": np.zeros((case.kappa2, D), np.uint64), "proj_int": np.zeros(256, np.int64),
             "proj": np.zeros(256, np.uint64), "bpp": np.zeros(D, np.uint64), "z": np.zeros((n, D), np.uint64)}
        po = co.ProveOut()
        for k, v in o.items():
            setattr(po, k, v.ctypes.data)
        po.phi_final = None
        pin = case._in()
        fin = FcProveIn()
        fin.inp = C.addressof(pin)
        keep = [np.ascontiguousarray(x) for x in (A_hat, B_hat, C_hat, D_hat)]
        fin.A, fin.B, fin.C, fin.Dm = (x.ctypes.data for x in keep)
        rc = lib().fc_prove(self._h, C.byref(fin), C.byref(po))
        if rc:
            raise ValueError("fc_prove rc=%d" % rc)
        o["jl_valid"], o["bpp_ok"] = bool(po.jl_valid), bool(po.bpp_ok)
        return o


def bench_polymul(q, arm, samples=1000, seed=1):
    """median ns of one ring product: arm 'classic' (i128 % per op, schoolbook + fold) or 'ntt'"""
    return lib().fc_bench_polymul(int(q), 0 if arm == "classic" else 1, int(samples), int(seed))
