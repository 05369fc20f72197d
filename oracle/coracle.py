"""coracle.py — ctypes/numpy loader for oracle/liblabrador_oracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Builds the shared object with gcc on first use
(no GPU, no reference tree needed).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblabrador_oracle.so")
D = 64

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")


def build(force=False):
    """(Re)build the shared object when it is missing or older than its sources.  Safe under concurrent callers
    (pytest-xdist workers, torchrun ranks): compile to a private temporary file, then rename atomically."""
    src = [os.path.join(_HERE, f) for f in ("labrador_oracle.c", "fastcpu.c", "labrador_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        tmp = "%s.%d.tmp" % (_SO, os.getpid())
        # same flags as oracle/Makefile (no -march=native: the library travels to the GPU box)
        subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-fPIC", "-Wall", "-Wextra", "-std=gnu11", "-fopenmp", "-shared",
                               "-o", tmp, src[0], src[1], "-lm"])
        os.replace(tmp, _SO)
    return _SO


class Constants(C.Structure):
    _fields_ = [("n", C.c_uint64), ("r", C.c_uint64), ("beta_bound", C.c_int64), ("std", C.c_double),
                ("b", C.c_int64), ("t1", C.c_int64), ("b1", C.c_int64), ("t2", C.c_int64), ("b2", C.c_int64),
                ("gamma", C.c_double), ("gamma1", C.c_double), ("gamma2", C.c_double), ("beta_prime", C.c_double),
                ("kappa", C.c_uint64), ("kappa1", C.c_uint64), ("kappa2", C.c_uint64)]


class ProveIn(C.Structure):
    _fields_ = [("q", C.c_uint64), ("n", C.c_size_t), ("r", C.c_size_t), ("kappa", C.c_size_t),
                ("kappa1", C.c_size_t), ("kappa2", C.c_size_t), ("b1", C.c_int64), ("b2", C.c_int64),
                ("beta_bound", C.c_int64), ("t1", C.c_int), ("t2", C.c_int),
                ("S", C.c_void_p), ("phi", C.c_void_p), ("a", C.c_void_p), ("b_prime", C.c_uint64),
                ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("Dm", C.c_void_p),
                ("Pi", C.c_void_p), ("psi", C.c_uint64), ("omega", C.c_void_p),
                ("alpha", C.c_void_p), ("beta", C.c_void_p), ("c", C.c_void_p)]


class ProveOut(C.Structure):
    _fields_ = [("t", C.c_void_p), ("G", C.c_void_p), ("H", C.c_void_p), ("u1", C.c_void_p), ("u2", C.c_void_p),
                ("proj_int", C.c_void_p), ("proj", C.c_void_p), ("bpp", C.c_void_p), ("z", C.c_void_p),
                ("phi_final", C.c_void_p), ("jl_valid", C.c_int), ("bpp_ok", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.orc_mod_positive_i128.restype = C.c_uint64
    L.orc_mod_positive_i128.argtypes = [C.c_uint64, C.c_int64, C.c_uint64]
    L.orc_trimmed_len.restype = C.c_int
    L.orc_trimmed_len.argtypes = [u64p]
    for name in ("orc_polymul", "orc_polymul_faithful", "orc_polymul_ntt_branch", "orc_poly_add", "orc_poly_sub"):
        f = getattr(L, name)
        f.restype = None
        f.argtypes = [C.c_uint64, u64p, u64p, u64p]
    L.orc_poly_scale.restype = None
    L.orc_poly_scale.argtypes = [C.c_uint64, u64p, C.c_uint64, u64p]
    L.orc_sigma_inv.restype = None
    L.orc_sigma_inv.argtypes = [C.c_uint64, u64p, u64p]
    L.orc_inner_product.restype = None
    L.orc_inner_product.argtypes = [C.c_uint64, u64p, u64p, C.c_size_t, u64p]
    L.orc_poly_by_poly_vec.restype = None
    L.orc_poly_by_poly_vec.argtypes = [C.c_uint64, u64p, u64p, C.c_size_t, u64p]
    L.orc_add_poly_vec.restype = None
    L.orc_add_poly_vec.argtypes = [C.c_uint64, u64p, u64p, C.c_size_t, u64p]
    L.orc_decompose.restype = C.c_int
    L.orc_decompose.argtypes = [C.c_uint64, u64p, C.c_int64, C.c_int, u64p]
    L.orc_decompose_vec.restype = C.c_int
    L.orc_decompose_vec.argtypes = [C.c_uint64, u64p, C.c_size_t, C.c_int64, C.c_int, u64p]
    L.orc_poly_norm.restype = C.c_double
    L.orc_poly_norm.argtypes = [u64p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_vec_poly_norm_squared.restype = C.c_double
    L.orc_vec_poly_norm_squared.argtypes = [u64p, C.c_size_t]
    L.orc_l2_norm_i64.restype = C.c_double
    L.orc_l2_norm_i64.argtypes = [i64p, C.c_size_t]
    L.orc_jl_accumulate.restype = None
    L.orc_jl_accumulate.argtypes = [i8p, u64p, C.c_size_t, i64p]
    L.orc_runtime_constants.restype = None
    L.orc_runtime_constants.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(Constants)]
    L.orc_splitmix64.restype = C.c_uint64
    L.orc_splitmix64.argtypes = [C.c_uint64]
    L.orc_synth_residue.restype = C.c_uint64
    L.orc_synth_residue.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.orc_synth_fill.restype = None
    L.orc_synth_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint64, u64p]
    L.orc_prove_core.restype = C.c_int
    L.orc_prove_core.argtypes = [C.POINTER(ProveIn), C.POINTER(ProveOut)]
    L.orc_verify_core.restype = C.c_int
    L.orc_verify_core.argtypes = [C.POINTER(ProveIn), C.POINTER(ProveOut), u64p, C.c_int64, C.c_double,
                                  C.POINTER(C.c_double)]
    L.orc_state_b.restype = None
    L.orc_state_b.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, u64p, u64p, u64p, u64p]
    L.orc_chacha20_block.restype = None
    L.orc_chacha20_block.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS"),
                                     np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")]
    L.orc_crs_coeff.restype = C.c_uint64
    L.orc_crs_coeff.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_int)]
    L.orc_crs_fill.restype = None
    L.orc_crs_fill.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint64, u64p]
    L.orc_crs_offset.restype = None
    L.orc_crs_offset.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                 C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint64)]
    L.orc_bench_inner_product_faithful.restype = C.c_uint64
    L.orc_bench_inner_product_faithful.argtypes = [C.c_uint64, u64p, u64p, C.c_size_t, C.c_int]
    _lib = L
    return L


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


# ---- thin numpy wrappers ---------------------------------------------------

def polymul(q, a, b, variant="classic"):
    a, b = _u64(a), _u64(b)
    out = np.empty(D, dtype=np.uint64)
    fn = {"classic": lib().orc_polymul, "faithful": lib().orc_polymul_faithful,
          "ntt_branch": lib().orc_polymul_ntt_branch}[variant]
    fn(q, a, b, out)
    return out


def polymul_batch(q, a, b):
    a, b = _u64(a).reshape(-1, D), _u64(b).reshape(-1, D)
    out = np.empty_like(a)
    f = lib().orc_polymul
    for i in range(a.shape[0]):
        f(q, a[i], b[i], out[i])
    return out


def poly_add(q, a, b):
    out = np.empty(D, dtype=np.uint64)
    lib().orc_poly_add(q, _u64(a), _u64(b), out)
    return out


def poly_sub(q, a, b):
    out = np.empty(D, dtype=np.uint64)
    lib().orc_poly_sub(q, _u64(a), _u64(b), out)
    return out


def poly_scale(q, p, s):
    out = np.empty(D, dtype=np.uint64)
    lib().orc_poly_scale(q, _u64(p), int(s), out)
    return out


def sigma_inv(q, a):
    out = np.empty(D, dtype=np.uint64)
    lib().orc_sigma_inv(q, _u64(a), out)
    return out


def inner_product(q, v1, v2):
    v1, v2 = _u64(v1).reshape(-1, D), _u64(v2).reshape(-1, D)
    if v1.shape != v2.shape:
        raise ValueError("inner product not defined on vectors of unequal length")
    out = np.empty(D, dtype=np.uint64)
    lib().orc_inner_product(q, v1, v2, v1.shape[0], out)
    return out


def poly_by_poly_vec(q, p, v):
    v = _u64(v).reshape(-1, D)
    out = np.empty_like(v)
    lib().orc_poly_by_poly_vec(q, _u64(p), v, v.shape[0], out)
    return out


def decompose(q, p, base, exp):
    out = np.empty((exp, D), dtype=np.uint64)
    rc = lib().orc_decompose(q, _u64(p), int(base), int(exp), out)
    if rc:
        raise ValueError("decompose failed rc=%d" % rc)
    return out


def decompose_vec(q, v, base, exp):
    v = _u64(v).reshape(-1, D)
    out = np.empty((exp, v.shape[0], D), dtype=np.uint64)
    rc = lib().orc_decompose_vec(q, v, v.shape[0], int(base), int(exp), out)
    if rc:
        raise ValueError("decompose_vec failed rc=%d" % rc)
    return out


def poly_norm(p):
    lo, hi = C.c_uint64(), C.c_uint64()
    f = lib().orc_poly_norm(_u64(p), C.byref(lo), C.byref(hi))
    return f, (hi.value << 64) | lo.value


def jl_project(pi, s_cols):
    """pi: [r][256][cols] int8, s_cols: [r][cols] uint64 -> int64[256]"""
    pi = np.ascontiguousarray(pi, dtype=np.int8)
    s_cols = _u64(s_cols)
    acc = np.zeros(256, dtype=np.int64)
    for i in range(pi.shape[0]):
        lib().orc_jl_accumulate(pi[i], s_cols[i], s_cols.shape[1], acc)
    return acc


def runtime_constants(q, n, r):
    c = Constants()
    lib().orc_runtime_constants(q, n, r, C.byref(c))
    return c


def synth_fill(seed, first_idx, count, bound):
    out = np.empty(count, dtype=np.uint64)
    lib().orc_synth_fill(seed, first_idx, count, bound, out)
    return out


class ProverCase:
    """Holds numpy inputs for orc_prove_core / orc_verify_core and runs them."""

    def __init__(self, q, n, r, kappa, kappa1, kappa2, b1, t1, b2, t2, beta_bound, S, phi, a, A, B, Cm, Dm,
                 Pi, psi, omega, alpha, beta, c):
        self.q, self.n, self.r = q, n, r
        self.kappa, self.kappa1, self.kappa2 = kappa, kappa1, kappa2
        self.b1, self.t1, self.b2, self.t2, self.beta_bound = b1, t1, b2, t2, beta_bound
        self.S, self.phi, self.a = _u64(S), _u64(phi), _u64(a)
        self.A, self.B, self.Cm, self.Dm = _u64(A), _u64(B), _u64(Cm), _u64(Dm)
        self.Pi = np.ascontiguousarray(Pi, dtype=np.int8)
        self.psi, self.omega = int(psi), _u64(omega)
        self.alpha, self.beta, self.c = _u64(alpha), _u64(beta), _u64(c)
        self.b_poly = np.empty(D, dtype=np.uint64)
        lib().orc_state_b(q, n, r, self.S, self.phi, self.a, self.b_poly)
        self.b_prime = int(self.b_poly[0])  # structs.rs:373 b.eval(0)

    def _in(self):
        p = ProveIn()
        p.q, p.n, p.r, p.kappa, p.kappa1, p.kappa2 = self.q, self.n, self.r, self.kappa, self.kappa1, self.kappa2
        p.b1, p.b2, p.beta_bound, p.t1, p.t2 = self.b1, self.b2, self.beta_bound, self.t1, self.t2
        for name, arr in (("S", self.S), ("phi", self.phi), ("a", self.a), ("A", self.A), ("B", self.B),
                          ("C", self.Cm), ("Dm", self.Dm), ("Pi", self.Pi), ("omega", self.omega),
                          ("alpha", self.alpha), ("beta", self.beta), ("c", self.c)):
            setattr(p, name, arr.ctypes.data)
        p.b_prime, p.psi = self.b_prime, self.psi
        return p

    def prove(self):
        n, r = self.n, self.r
        o = {"t": np.zeros((r, self.kappa, D), np.uint64), "G": np.zeros((r, r, D), np.uint64),
             "H": np.zeros((r, r, D), np.uint64), "u1": np.zeros((self.kappa1, D), np.uint64),
             "u2": np.zeros((self.kappa2, D), np.uint64), "proj_int": np.zeros(256, np.int64),
             "proj": np.zeros(256, np.uint64), "bpp": np.zeros(D, np.uint64), "z": np.zeros((n, D), np.uint64),
             "phi_final": np.zeros((r, n, D), np.uint64)}
        po = ProveOut()
        for k, v in o.items():
            setattr(po, k, v.ctypes.data)
        pin = self._in()
        rc = lib().orc_prove_core(C.byref(pin), C.byref(po))
        if rc:
            raise ValueError("orc_prove_core rc=%d" % rc)
        o["jl_valid"], o["bpp_ok"] = bool(po.jl_valid), bool(po.bpp_ok)
        self._last = (o, po)
        return o

    def verify(self, proof, bconst, beta_prime):
        po = ProveOut()
        keep = {}
        for k in ("t", "G", "H", "u1", "u2", "proj_int", "proj", "bpp", "z"):
            dt = np.int64 if k == "proj_int" else np.uint64
            keep[k] = np.ascontiguousarray(proof[k], dtype=dt)
            setattr(po, k, keep[k].ctypes.data)
        pin = self._in()
        s = C.c_double()
        rc = lib().orc_verify_core(C.byref(pin), C.byref(po), self.b_poly, int(bconst), float(beta_prime), C.byref(s))
        return rc, s.value


def chacha20_block(state16):
    st = np.ascontiguousarray(state16, dtype=np.uint32)
    out = np.empty(16, dtype=np.uint32)
    lib().orc_chacha20_block(st, out)
    return out


def crs_fill(base_seed, offset, count, q):
    """count coefficients for seeds base_seed + offset + k (src/structs.rs:147-171)."""
    out = np.empty(count, dtype=np.uint64)
    lib().orc_crs_fill(bytes(base_seed), offset & ((1 << 64) - 1), offset >> 64, count, q, out)
    return out


def crs_offset(which, n, r, kappa, kappa1, kappa2, t1, i=0, j=0, k=0, row=0):
    lo, hi = C.c_uint64(), C.c_uint64()
    lib().orc_crs_offset({"A": 0, "B": 1, "C": 2, "D": 3}[which], n, r, kappa, kappa1, kappa2, t1, i, j, k, row,
                         C.byref(lo), C.byref(hi))
    return (hi.value << 64) | lo.value
