"""Host-side mirror of the reference's ring/vector interface over the C ABI.

Names follow the Rust reference so parity tests read like its own tests:
`Context` stands for the (Q, PLAN) globals (src/constants.rs:193-198),
`negacyclic_polymul` for `PLAN.negacyclic_polymul` (src/algebraic.rs:396),
`polynomial_vec_inner_product` for src/util.rs:496-509, and so on.

Arrays at this level are numpy uint64 (host) or torch tensors on the GPU
(uint64/int64 storage): both are passed to the library as raw pointers.
PyTorch is only the device-memory / stream plumbing.
"""
import ctypes as C

import numpy as np

from . import _lib

D = 64


def _ptr(a):
    """Raw pointer + keep-alive for a numpy array or a torch tensor."""
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return a.data_ptr()
    raise TypeError("expected numpy array or torch tensor, got %r" % type(a))


def _as_u64(a):
    if isinstance(a, np.ndarray):
        return np.ascontiguousarray(a, dtype=np.uint64)
    if hasattr(a, "data_ptr"):
        if a.element_size() != 8:
            raise TypeError("device coefficient tensors must have 8-byte elements")
        return a.contiguous()
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _nelems(a):
    n = a.size if isinstance(a, np.ndarray) else a.numel()
    if n % D:
        raise ValueError("coefficient array size must be a multiple of 64")
    return n // D


class Context:
    """One modulus q on one GPU.  `max_len` bounds inner-product lengths (it fixes the RNS limb count)."""

    def __init__(self, q, device=0, max_len=0, rns_only=False):
        """rns_only: never run single-limb on q itself (lab_ctx_create_ex, LAB_CTX_RNS_ONLY)."""
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.lab_ctx_create_ex(int(q), int(device), int(max_len), 1 if rns_only else 0, C.byref(h))
        _lib.check(None, rc)
        self._h = h
        info = _lib.lab_info()
        _lib.check(self._h, self._lib.lab_ctx_info(self._h, C.byref(info)))
        self.q = int(info.q)
        self.nlimbs = int(info.nlimbs)
        self.primes = [int(info.primes[i]) for i in range(self.nlimbs)]
        self.max_len = int(info.max_len)
        self.device = int(info.device)
        # q is its own NTT prime in [2^30, 2^32): the strict-32 kernels (csrc/zq_device.h Zp<true>); its main basis has no
        # matrix-core form (lab_mat8_pack refuses it), the 23-bit digit basis serves that q
        self.wide = self.nlimbs == 1 and self.primes[0] == self.q and self.q >= (1 << 30)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lab_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        _lib.check(self._h, rc)

    def synchronize(self):
        self._check(self._lib.lab_ctx_synchronize(self._h))

    def use_torch_stream(self):
        """Run on torch's current HIP stream (so torch.cuda.Event timing and tensor lifetimes line up)."""
        import torch
        self._check(self._lib.lab_ctx_set_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def enable_digit_basis(self, max_terms, digit_bound, mfma=False):
        """Second, smaller RNS basis for reductions whose vector operand is a decomposition digit (< digit_bound):
        see lab_ctx_enable_digit_basis.  mfma=True picks primes below 2^23 instead (three signed byte planes per
        residue on the matrix cores, lab_ctx_enable_digit_basis_mfma).  Returns (primes, last_limb_is_u16,
        bytes_per_coefficient)."""
        fn = self._lib.lab_ctx_enable_digit_basis_mfma if mfma else self._lib.lab_ctx_enable_digit_basis
        self._check(fn(self._h, int(max_terms), int(digit_bound)))
        return self.digit_info()

    def digit_info(self):
        info = _lib.lab_info()
        self._check(self._lib.lab_ctx_digit_info(self._h, C.byref(info)))
        primes = [int(info.primes[i]) for i in range(int(info.nlimbs))]
        small = bool(info.device)
        self.digit_primes, self.digit_small = primes, small
        return primes, small, (4 * (len(primes) - 1) + 2 if small else 4 * len(primes))

    def prove_flags(self):
        """How the last prove_core[_batch] ran: 1 = step 1 in the digit basis, 2 = tried and redone in the main basis, 0 = main basis;
        + 4: sub-batches of a call of more than 64 proofs shared one pass over B."""
        return int(self._lib.lab_ctx_prove_flags(self._h))

    def profile_begin(self):
        """Start bracketing every launch of the streaming mat-vec kernel with HIP events on the context's stream."""
        self._check(self._lib.lab_ctx_profile_begin(self._h))

    def profile_end(self):
        """-> (kernel_ms_total, launches, algorithmic_bytes_total)"""
        ms, n, b = C.c_double(), C.c_uint64(), C.c_double()
        self._check(self._lib.lab_ctx_profile_end(self._h, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, int(n.value), b.value

    PHASES = ("crs_expand", "coefs_to_tiles", "gemm", "digits_to_tiles")

    def profile_phases(self):
        """The streamed outer commitment's phases as sampled by the hook since profile_begin (lab_ctx_profile_phase):
        {name: dict(sampled_ms, sampled_launches, sampled_units, launches, units)}."""
        out = {}
        for i, name in enumerate(self.PHASES):
            ms, sl, su, n, u = C.c_double(), C.c_uint64(), C.c_double(), C.c_uint64(), C.c_double()
            self._check(self._lib.lab_ctx_profile_phase(self._h, i, C.byref(ms), C.byref(sl), C.byref(su), C.byref(n), C.byref(u)))
            out[name] = dict(sampled_ms=ms.value, sampled_launches=int(sl.value), sampled_units=su.value, launches=int(n.value), units=u.value)
        return out

    def gemm_diag(self, mode, ticks=None):
        """Diagnostic variant of the timed GEMM (include/labrador_hip.h: lab_ctx_gemm_diag); ticks: uint64 device tensor."""
        ptr, words = (ticks.data_ptr(), ticks.numel()) if ticks is not None else (None, 0)
        self._check(self._lib.lab_ctx_gemm_diag(self._h, int(mode), ptr, words))

    def profile_kernel(self):
        """Name of the kernel the most recent launch timed by the profiling hook was dispatched to."""
        return self._lib.lab_ctx_profile_kernel(self._h).decode()

    # ---- PLAN.negacyclic_polymul + transform_slice_u64_to_zq (src/algebraic.rs:396-397), batched
    def negacyclic_polymul(self, lhs, rhs, out=None):
        lhs, rhs = _as_u64(lhs), _as_u64(rhs)
        n = _nelems(lhs)
        if _nelems(rhs) != n:
            raise _lib.LengthMismatch(_lib.LAB_ERR_LEN_MISMATCH, "operands of unequal length")
        if out is None:
            if isinstance(lhs, np.ndarray):
                out = np.empty(lhs.shape, dtype=np.uint64)
            else:
                import torch
                out = torch.empty_like(lhs)
        self._check(self._lib.lab_negacyclic_polymul_u64(self._h, _ptr(out), _ptr(lhs), _ptr(rhs), n))
        return out

    # ---- raw batched transforms on device residues u32 [nlimbs][batch][64]
    def ntt_forward(self, residues, batch):
        self._check(self._lib.lab_ntt_forward(self._h, _ptr(residues), batch))

    def ntt_inverse(self, residues, batch):
        self._check(self._lib.lab_ntt_inverse(self._h, _ptr(residues), batch))

    # ---- hat vectors
    def vec(self, count, digit=False, compact=False):
        return HatVec(self, count, digit=digit, compact=compact)

    def vec_from(self, coeffs):
        coeffs = _as_u64(coeffs)
        v = HatVec(self, _nelems(coeffs))
        v.load(coeffs)
        return v


class HatVec:
    """Device-resident ring elements in the transform domain (lab_vec)."""

    def __init__(self, ctx, count, digit=False, compact=False):
        self.ctx = ctx
        h = C.c_void_p()
        if digit or compact:
            ctx._check(ctx._lib.lab_vec_create_digit(ctx._h, int(count), 1 if compact else 0, C.byref(h)))
        else:
            ctx._check(ctx._lib.lab_vec_create(ctx._h, int(count), C.byref(h)))
        self._h = h
        self.count = int(count)
        self.digit, self.compact = bool(digit or compact), bool(compact)

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx._lib.lab_vec_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.count

    def load(self, coeffs, offset=0):
        coeffs = _as_u64(coeffs)
        self.ctx._check(self.ctx._lib.lab_vec_load_u64(self.ctx._h, self._h, offset, _ptr(coeffs), _nelems(coeffs)))
        return self

    def store(self, offset=0, count=None, out=None):
        count = self.count - offset if count is None else count
        if out is None:
            out = np.empty((count, D), dtype=np.uint64)
        self.ctx._check(self.ctx._lib.lab_vec_store_u64(self.ctx._h, self._h, offset, count, _ptr(out)))
        return out

    def fill_synthetic(self, seed, first_elem=0, bound=None, offset=0, count=None):
        count = self.count - offset if count is None else count
        bound = self.ctx.q if bound is None else bound
        self.ctx._check(self.ctx._lib.lab_vec_fill_synthetic(self.ctx._h, self._h, offset, count, seed, first_elem,
                                                            bound))
        return self


def polynomial_vec_inner_product(ctx, v1, v2, off1=0, off2=0, length=None, out=None, out_off=0):
    """src/util.rs:496-509.  Raises LengthMismatch (an AssertionError) like the reference's assert!."""
    if length is None:
        if v1.count - off1 != v2.count - off2:
            raise _lib.LengthMismatch(_lib.LAB_ERR_LEN_MISMATCH,
                                      "inner product not defined on vectors of unequal length. v1 length: %d, v2 length: %d"
                                      % (v1.count - off1, v2.count - off2))
        length = v1.count - off1
    if out is None:
        out = HatVec(ctx, 1)
    ctx._check(ctx._lib.lab_inner_product(ctx._h, out._h, out_off, v1._h, off1, v2._h, off2, length))
    return out


def matvec_sum(ctx, M, V, nmat, rows, cols, m_off=0, v_off=0, out=None, out_off=0):
    """out[rho] = sum_m <M[m][rho,:], V[m][:]> — src/proofgen.rs:41-49 (nmat=1) and :101-132 (nmat=r*t1)."""
    if out is None:
        out = HatVec(ctx, rows, digit=M.digit)
    ctx._check(ctx._lib.lab_matvec_sum(ctx._h, out._h, out_off, M._h, m_off, nmat, rows, cols, V._h, v_off))
    return out


def matvec_sum_batched(ctx, M, V, nmat, rows, cols, nb, m_off=0, v_off=0, out=None, out_off=0):
    """nb independent vector sets sharing the matrices M (the CRS is streamed once for all of them)."""
    if out is None:
        out = HatVec(ctx, nb * rows, digit=M.digit)
    ctx._check(ctx._lib.lab_matvec_sum_batched(ctx._h, out._h, out_off, M._h, m_off, nmat, rows, cols, V._h, v_off, nb))
    return out


class Mat8:
    """Matrices re-stored as signed base-256 digit planes in MFMA operand-tile order (lab_mat8_pack)."""

    def __init__(self, ctx, M, nmat, rows, cols, m_off=0):
        self.ctx, self.nmat, self.rows, self.cols, self.digit = ctx, nmat, rows, cols, M.digit
        h = C.c_void_p()
        ctx._check(ctx._lib.lab_mat8_pack(ctx._h, M._h, m_off, nmat, rows, cols, C.byref(h)))
        self._h = h

    @property
    def nbytes(self):
        return int(self.ctx._lib.lab_mat8_bytes(self._h))

    def close(self):
        if self._h:
            self.ctx._lib.lab_mat8_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def matvec_sum_mfma(ctx, M8, V, ns, v_off=0, out=None, out_off=0):
    """lab_matvec_sum_batched for ns in {8,16,32,64} vector sets on the matrix cores (exact int8 GEMMs)."""
    if out is None:
        out = HatVec(ctx, ns * M8.rows, digit=M8.digit)
    ctx._check(ctx._lib.lab_matvec_sum_mfma(ctx._h, out._h, out_off, M8._h, V._h, v_off, ns))
    return out


def crs_matvec_mfma(ctx, base_seed, mat_offsets, rows, cols, V, ns, v_off=0, out=None, out_off=0):
    """lab_crs_matvec_mfma: out_s[rho] = sum_m <M_m[rho,:], V_s[m]> with the matrices regenerated from the CRS seed slab by
    slab (never resident).  mat_offsets: the seed offset (coefficient units, python ints) of row 0 of every matrix."""
    nmat = len(mat_offsets)
    offs = np.empty(2 * nmat, dtype=np.uint64)
    for m, o in enumerate(mat_offsets):
        offs[2 * m], offs[2 * m + 1] = o & ((1 << 64) - 1), o >> 64
    if out is None:
        out = HatVec(ctx, ns * rows, digit=V.digit)
    ctx._check(ctx._lib.lab_crs_matvec_mfma(ctx._h, out._h, out_off, bytes(base_seed), _ptr(offs), nmat, rows, cols, V._h, v_off, ns))
    return out


def ajtai_commit(ctx, A, S, rows, inner, ncols, a_off=0, s_off=0, out=None, out_off=0):
    """t_i_all[i][rho] = <A[rho,:], s_i> for every witness column (src/proofgen.rs:41-49). out: [ncols][rows]."""
    if out is None:
        out = HatVec(ctx, rows * ncols)
    ctx._check(ctx._lib.lab_ajtai_commit(ctx._h, out._h, out_off, A._h, a_off, rows, inner, S._h, s_off, ncols))
    return out


def gram(ctx, P, S, n, r, p_off=0, s_off=0, out=None, out_off=0):
    """out[i][k] = sum_x P[x][i] * S[x][k] (src/proofgen.rs:59-70, 334-335)."""
    if out is None:
        out = HatVec(ctx, r * r)
    ctx._check(ctx._lib.lab_gram(ctx._h, out._h, out_off, P._h, p_off, S._h, s_off, n, r))
    return out


def axpy_commit_batched(ctx, G, Cm, nterms, length, nb, g_off=0, c_off=0, out=None, out_off=0):
    """nb independent digit vectors sharing C (streamed once): out[b][rho] = sum_m G[b][m] * C[m][rho]."""
    if out is None:
        out = HatVec(ctx, nb * length, digit=Cm.digit)
    ctx._check(ctx._lib.lab_axpy_commit_batched(ctx._h, out._h, out_off, G._h, g_off, Cm._h, c_off, nterms, length, nb))
    return out


def axpy_commit(ctx, G, Cm, nterms, length, g_off=0, c_off=0, out=None, out_off=0):
    """out[rho] = sum_m G[m] * C[m][rho] (src/proofgen.rs:137-148, 364-378)."""
    if out is None:
        out = HatVec(ctx, length, digit=Cm.digit)
    ctx._check(ctx._lib.lab_axpy_commit(ctx._h, out._h, out_off, G._h, g_off, Cm._h, c_off, nterms, length))
    return out


def amortize_z(ctx, S, c, n, r, s_off=0, c_off=0, out=None, out_off=0):
    """z = sum_i c_i s_i (src/proofgen.rs:380-399)."""
    if out is None:
        out = HatVec(ctx, n)
    ctx._check(ctx._lib.lab_amortize_z(ctx._h, out._h, out_off, S._h, s_off, n, r, c._h, c_off))
    return out


_OPS = {"mul": 0, "add": 1, "sub": 2}


def pointwise(ctx, op, a, b, count=None, a_off=0, b_off=0, out=None, out_off=0):
    """Element-wise Rq Mul/Add/Sub (src/algebraic.rs:441-531); `add` is add_poly_vec (src/util.rs:298-309)."""
    count = a.count - a_off if count is None else count
    if out is None:
        out = HatVec(ctx, count)
    ctx._check(ctx._lib.lab_pointwise(ctx._h, _OPS[op], out._h, out_off, a._h, a_off, b._h, b_off, count))
    return out


def add_poly_vec(ctx, a, b):
    if a.count != b.count:
        raise _lib.LengthMismatch(_lib.LAB_ERR_LEN_MISMATCH, "summation not defined on vectors of unequal length")
    return pointwise(ctx, "add", a, b)


def poly_by_poly_vec(ctx, p, v, length=None, p_off=0, v_off=0, out=None, out_off=0):
    """src/util.rs:289-296."""
    length = v.count - v_off if length is None else length
    if out is None:
        out = HatVec(ctx, length)
    ctx._check(ctx._lib.lab_poly_by_poly_vec(ctx._h, out._h, out_off, p._h, p_off, v._h, v_off, length))
    return out


def _like(a, shape):
    if isinstance(a, np.ndarray):
        return np.empty(shape, dtype=np.uint64)
    import torch
    return torch.empty(shape, dtype=a.dtype, device=a.device)


def decompose_polynomial_vec(ctx, coeffs, base, exp):
    """src/util.rs:360-442 (bug-compatible). coeffs [count][64] -> [exp][count][64]."""
    coeffs = _as_u64(coeffs)
    n = _nelems(coeffs)
    out = _like(coeffs, (max(exp, 0), n, D))
    ctx._check(ctx._lib.lab_decompose(ctx._h, _ptr(out), _ptr(coeffs), n, int(base), int(exp)))
    return out


def poly_norm_exact(ctx, coeffs):
    """Exact squared norms (python ints) of each polynomial's raw residues (src/util.rs:195-202 before the f64 cast)."""
    coeffs = _as_u64(coeffs)
    n = _nelems(coeffs)
    lo, hi = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint64)
    ctx._check(ctx._lib.lab_norm_sq(ctx._h, _ptr(coeffs), n, _ptr(lo), _ptr(hi)))
    return [(int(h) << 64) | int(l) for l, h in zip(lo, hi)]


def poly_norm(ctx, p):
    """src/util.rs:195-202: `norm as f64`."""
    return float(poly_norm_exact(ctx, p)[0])


def vec_poly_norm_squared(ctx, v):
    """src/util.rs:188-192: sequential f64 sum of poly_norm, in index order."""
    s = 0.0
    for x in poly_norm_exact(ctx, v):
        s += float(x)
    return s


def scale_poly_vec(ctx, coeffs, s):
    """src/util.rs:167-180."""
    coeffs = _as_u64(coeffs)
    out = _like(coeffs, tuple(coeffs.shape))
    ctx._check(ctx._lib.lab_scale(ctx._h, _ptr(out), _ptr(coeffs), _nelems(coeffs), int(s)))
    return out


def sigma_inv_vec(ctx, coeffs):
    """src/util.rs:107-137."""
    coeffs = _as_u64(coeffs)
    out = _like(coeffs, tuple(coeffs.shape))
    ctx._check(ctx._lib.lab_sigma_inv(ctx._h, _ptr(out), _ptr(coeffs), _nelems(coeffs)))
    return out


def jl_project(ctx, Pi, S, n, r):
    """Integer projection of Prover::jl_project (src/proofgen.rs:429-456). Pi int8 [r][256][n*64]; S [n][r][64]."""
    S = _as_u64(S)
    if isinstance(Pi, np.ndarray):
        Pi = np.ascontiguousarray(Pi, dtype=np.int8)
    out = np.zeros(256, dtype=np.int64)
    ctx._check(ctx._lib.lab_jl_project(ctx._h, _ptr(out), _ptr(Pi), _ptr(S), n, r))
    return out
