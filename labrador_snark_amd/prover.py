"""Host-side mirror of the reference's `Prover` (src/proofgen.rs:14-28) over lab_prove_core.

`Prover(witness, verifier, constants).proof_gen(st, crs)` keeps the reference's
call shape.  The verifier object supplies the interactive randomness through the
same method names the reference calls (sample_jl_projection, valid_projection is
evaluated inside the core, generate_psi, generate_omega, fetch_alpha, fetch_beta,
fetch_challenge); the CRS object holds the four matrices resident on the GPU in
the transform domain instead of regenerating rows from a ChaCha20 seed
(src/structs.rs:55-171 — on-device expansion is a later row of SURVEY §8f).
"""
import ctypes as C

import numpy as np

from . import _lib
from .api import D, HatVec, _as_u64, _ptr


class RuntimeConstants:
    """src/constants.rs:205-264.  kappa/kappa1/kappa2 are explicit here (the reference hard-wires n*64)."""

    def __init__(self, q, N, R, kappa=None, kappa1=None, kappa2=None):
        import math
        TAU = 71.0
        Q = float(q)
        self.N, self.R = N, R
        self.KAPPA = N * D if kappa is None else kappa
        self.KAPPA_1 = N * D if kappa1 is None else kappa1
        self.KAPPA_2 = N * D if kappa2 is None else kappa2

        def rround(x):  # f64::round — half away from zero
            return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)

        self.BETA_BOUND = int(math.floor(math.sqrt(30.0 / 128.0) * Q / 125.0))
        self.STD = float(self.BETA_BOUND) / math.sqrt(float(R * N * D))
        self.B = int(rround(math.sqrt(math.sqrt(12.0 * float(R) * TAU) * self.STD)))
        self.T_1 = int(rround(math.log2(Q) / math.log2(float(self.B))))
        self.B_1 = int(math.pow(Q, 1.0 / float(self.T_1)))
        self.T_2 = int(rround(math.log2(math.sqrt(24.0 * float(N * D)) * self.STD ** 2) / math.log2(float(self.B))))
        self.B_2 = int(rround(math.pow(math.sqrt(float(24 * N * D)) * self.STD ** 2, 1.0 / float(self.T_2))))
        b1, b2, t1, t2, r = float(self.B_1), float(self.B_2), float(self.T_1), float(self.T_2), float(R)
        self.GAMMA = float(self.BETA_BOUND ** 2) * TAU
        self.GAMMA_1 = ((b1 * b1 * t1) / 12.0) * r * float(self.KAPPA) * float(D) + \
            ((b2 * b2 * t2) / 12.0) * ((r * r + r) / 2.0) * float(D)
        self.GAMMA_2 = ((b1 * b1 * t1) / 12.0) * ((r * r + r) / 2.0) * float(D)
        self.BETA_PRIME = (2.0 / float(self.B) ** 2) * self.GAMMA + self.GAMMA_1 + self.GAMMA_2


class ResidentCRS:
    """A, B_ik, C_ijk, D_ijk as transform-domain device matrices (stand-in for CRS::fetch_*, src/structs.rs:55-144)."""

    def __init__(self, ctx, constants, compact=False, mfma_basis=False, with_B=True):
        """compact=True keeps B, C, D in the context's digit basis with the small limb as u16 (they are only ever
        multiplied by decomposition digits < B_1, B_2): fewer bytes per coefficient on the HBM-bound kernel.
        Falls back to the main basis when no smaller basis exists (q its own NTT prime).
        mfma_basis=True keeps them in the digit basis made for the matrix cores instead (primes below 2^23: three byte
        planes per residue once packed by pack_mfma)."""
        k = constants
        pairs = k.R * (k.R + 1) // 2
        self.ctx, self.constants = ctx, k
        self.compact = False
        # longest exact sum of digit products: step 3's r*t1*kappa terms plus the C terms added to it.  With B streamed from the
        # seed (with_B=False) the K chunks of step 3 (at most 2^17 - 2 terms each) are reduced mod q one by one when the basis
        # does not hold the whole sum, so one chunk's length is enough: three 23-bit primes instead of four at kappa = 2^14
        outer = k.R * k.T_1 * k.KAPPA
        if not with_B and k.KAPPA <= (1 << 17) - 2:
            outer = min(outer, (((1 << 17) - 2) // k.KAPPA) * k.KAPPA)
        terms = outer + pairs * max(k.T_1, k.T_2)
        if mfma_basis:
            if not getattr(ctx, "digit_primes", None):
                ctx.enable_digit_basis(terms, max(k.B_1, k.B_2), mfma=True)
            self.compact = True
        elif compact and ctx.nlimbs > 1:
            if not getattr(ctx, "digit_primes", None):
                try:
                    ctx.enable_digit_basis(terms, max(k.B_1, k.B_2))
                except _lib.LabradorError:
                    pass
            self.compact = bool(getattr(ctx, "digit_primes", None))
        dg = dict(digit=self.compact, compact=self.compact and ctx.digit_small)
        self.A = HatVec(ctx, k.KAPPA * k.N)
        self.B = HatVec(ctx, k.R * k.T_1 * k.KAPPA_1 * k.KAPPA, **dg) if with_B else None    # with_B=False: B is streamed from crs_seed
        self.C = HatVec(ctx, pairs * k.T_2 * k.KAPPA_2, **dg)
        self.Dm = HatVec(ctx, pairs * k.T_1 * k.KAPPA_2, **dg)

    def fill_synthetic(self, seed):
        """Uniform residues in [0,q) from the shared splitmix64 streams seed..seed+3 (oracle: orc_synth_fill)."""
        self.A.fill_synthetic(seed)
        if self.B is not None:
            self.B.fill_synthetic(seed + 1)
        self.C.fill_synthetic(seed + 2)
        self.Dm.fill_synthetic(seed + 3)
        return self

    def load(self, A, B, Cm, Dm):
        self.A.load(A); self.B.load(B); self.C.load(Cm); self.Dm.load(Dm)
        return self

    def pack_mfma(self, drop_B=False, main_basis_A=True, drop_CD=False):
        """Re-stores the B matrices for the matrix cores (lab_mat8_pack): step 3 of prove_core[_batch] then runs as exact
        int8 GEMMs over up to 64 proofs per pass over B.  drop_B=True frees the original B (the verifier and
        single-proof VALU path need it; a prover-only service does not).  main_basis_A=False skips A8 (12 bytes per
        coefficient of A): step 1 then has the digit-basis A8d, and the multiply-add kernel for a witness above its norm bound.
        drop_CD=True frees C and Dm after packing them (prover-only, like drop_B)."""
        from .api import Mat8
        k = self.constants
        # (sums longer than the int32 bound of one int8 GEMM stay on the multiply-add kernels when B is resident; streamed B is
        # cut into K chunks instead)
        fits = k.R * k.T_1 * k.KAPPA < (1 << 17) - 1
        self.B8 = Mat8(self.ctx, self.B, k.R * k.T_1, k.KAPPA_1, k.KAPPA) if self.B is not None and fits else None
        # (a single-limb strict-32 context — q its own NTT prime in [2^30, 2^32) — has no main-basis matrix-core form: step 1
        # then has the digit-basis A8d or the multiply-add kernel)
        self.A8 = Mat8(self.ctx, self.A, 1, k.KAPPA, k.N) if main_basis_A and not self.ctx.wide else None
        pairs = k.R * (k.R + 1) // 2
        self.C8 = Mat8(self.ctx, self.C, pairs * k.T_2, k.KAPPA_2, 1)      # element (rho, m) = C[m][rho]
        self.D8 = Mat8(self.ctx, self.Dm, pairs * k.T_1, k.KAPPA_2, 1)
        if self.compact and not self.ctx.digit_small and all(p < (1 << 23) for p in self.ctx.digit_primes):
            # matrix-core digit basis: A once more in that basis (3 byte planes); step 1 of a batch runs there whenever the
            # witness norm bound keeps |A s| inside its CRT range (lab_prove_input.A8d)
            Ad = HatVec(self.ctx, k.KAPPA * k.N, digit=True)
            step = 1 << 15
            for off in range(0, k.KAPPA * k.N, step):
                cnt = min(step, k.KAPPA * k.N - off)
                Ad.load(self.A.store(off, cnt), off)
            self.A8d = Mat8(self.ctx, Ad, 1, k.KAPPA, k.N)
            self.ctx.synchronize()
            Ad.close()
        if drop_B and self.B is not None:
            self.ctx.synchronize()
            self.B.close()
            self.B = None
        if drop_CD:      # the prover reads C8 / D8 only (the verifier needs the residues): 46 GB at kappa = 2^14
            self.ctx.synchronize()
            self.C.close(); self.Dm.close()
            self.C = self.Dm = None
        return self

    def close(self):
        for v in (self.A, self.B, self.C, self.Dm, getattr(self, "B8", None), getattr(self, "A8", None),
                  getattr(self, "C8", None), getattr(self, "D8", None), getattr(self, "A8d", None)):
            if v is not None:
                v.close()


class CRS:
    """The reference's seed-addressed CRS (src/structs.rs:27-190): `base_seed` is the 32-byte PRG seed, rows are
    ChaCha20-expanded on the GPU (lab_crs_expand) instead of one coefficient at a time on the host.  fetch_* return
    coefficient arrays like the reference's Vec<Rq>; materialize() builds the resident transform-domain matrices
    lab_prove_core consumes.  Seed offsets are the reference's own, copied literally (including B matrices that
    are spaced kappa1*kappa seeds apart although each consumes 64x that, and T_1 used for C's index)."""

    def __init__(self, ctx, constants, base_seed):
        if len(base_seed) != 32:
            raise ValueError("base_seed must be 32 bytes")
        self.ctx, self.constants, self.base_seed = ctx, constants, bytes(base_seed)

    # ---- offsets in coefficient units
    def _sum_pairs(self, i):
        R = self.constants.R
        return i * R - (i * (i - 1)) // 2 if i > 0 else 0

    def offset_A(self, row):                                  # src/structs.rs:62
        k = self.constants
        return row * (k.N * D)

    def offset_B(self, i, kk, row):                           # src/structs.rs:78-82
        k = self.constants
        return k.KAPPA * k.N * D + (i * k.T_1 + kk) * (k.KAPPA_1 * k.KAPPA) + row * (k.KAPPA * D)

    def offset_C(self, i, j, kk):                             # src/structs.rs:94-108
        k = self.constants
        return (k.KAPPA * k.N * D + (k.R * k.T_1) * (k.KAPPA_1 * k.KAPPA) * D +
                (kk + k.T_1 * (self._sum_pairs(i) + (j - i))) * (k.KAPPA_2 * D))

    def offset_D(self, i, j, kk):                             # src/structs.rs:120-138
        k = self.constants
        return (k.KAPPA * k.N * D + (k.R * k.T_1) * (k.KAPPA_1 * k.KAPPA) * D + (k.R * (k.R + 1) // 2) * (k.KAPPA_2 * D) +
                (kk + k.T_1 * (self._sum_pairs(i) + (j - i))) * (k.KAPPA_2 * D))

    # ---- fetch_* as coefficient arrays [count][64]
    def _fetch(self, offset, count):
        out = np.empty((count, D), dtype=np.uint64)
        self.ctx._check(self.ctx._lib.lab_crs_expand_u64(self.ctx._h, _ptr(out), count, self.base_seed,
                                                         offset & ((1 << 64) - 1), offset >> 64))
        return out

    def fetch_A_row(self, row):
        return self._fetch(self.offset_A(row), self.constants.N)

    def fetch_B_ik_row(self, i, kk, row):
        return self._fetch(self.offset_B(i, kk, row), self.constants.KAPPA)

    def fetch_C_ijk(self, i, j, kk):
        return self._fetch(self.offset_C(i, j, kk), self.constants.KAPPA_2)

    def fetch_D_ijk(self, i, j, kk):
        return self._fetch(self.offset_D(i, j, kk), self.constants.KAPPA_2)

    def _expand_into(self, vec, dst_off, count, offset):
        self.ctx._check(self.ctx._lib.lab_crs_expand(self.ctx._h, vec._h, dst_off, count, self.base_seed,
                                                     offset & ((1 << 64) - 1), offset >> 64))

    def materialize(self, compact=False, mfma_basis=False, stream_B=False):
        """Resident transform-domain A, B, C, D with exactly the rows the reference's fetch_* would return.
        stream_B=True: A, C, D resident, B NOT materialised — lab_prove_core[_batch] regenerates its rows from the seed
        slab by slab inside step 3 (lab_prove_input.crs_seed), like the reference does on every fetch_B_ik_row."""
        k = self.constants
        res = ResidentCRS(self.ctx, k, compact=compact, mfma_basis=mfma_basis, with_B=not stream_B)
        if stream_B:
            res.crs_seed = self.base_seed
        self._expand_into(res.A, 0, k.KAPPA * k.N, self.offset_A(0))       # rows of A are consecutive in seed space
        for i in range(k.R):
            for kk in range(k.T_1):                                           # rows of one B_ik are consecutive too
                if stream_B:
                    break
                self._expand_into(res.B, (i * k.T_1 + kk) * k.KAPPA_1 * k.KAPPA, k.KAPPA_1 * k.KAPPA, self.offset_B(i, kk, 0))
        for i in range(k.R):
            for j in range(i, k.R):
                pair = self._sum_pairs(i) + (j - i)
                for kk in range(k.T_2):
                    self._expand_into(res.C, (pair * k.T_2 + kk) * k.KAPPA_2, k.KAPPA_2, self.offset_C(i, j, kk))
                for kk in range(k.T_1):
                    self._expand_into(res.Dm, (pair * k.T_1 + kk) * k.KAPPA_2, k.KAPPA_2, self.offset_D(i, j, kk))
        return res


def gen_f_b(ctx, witness, phi, a, n, r):
    """The `b` of State::gen_f (src/structs.rs:316-341): b = sum_{i,j} a_ij <s_i, s_j> + sum_i <phi_i, s_i>, on the
    Gram / inner-product kernels.  witness, phi: [n][r][64]; a: [r][r][64] (symmetric); returns b as uint64[64].
    G goes through the coefficient domain before it is multiplied again (product depth 1, like everywhere)."""
    from .api import gram, polynomial_vec_inner_product
    hS, hphi, ha = ctx.vec_from(witness), ctx.vec_from(phi), ctx.vec_from(a)
    G = gram(ctx, hS, hS, n, r)
    Gc = ctx.vec_from(G.store())
    t1 = polynomial_vec_inner_product(ctx, ha, Gc).store()[0]           # sum_ij a_ij g_ij
    t2 = polynomial_vec_inner_product(ctx, hphi, hS).store()[0]         # sum_i <phi_i, s_i>: [n][r] x [n][r] element-wise
    q = np.uint64(ctx.q)
    b = (t1 + t2) % q                                                   # residues < 2^40: no overflow in u64
    for v in (hS, hphi, ha, G, Gc):
        v.close()
    return b


class State:
    """The parts of src/structs.rs:269-286 the prover and verifier read: phi_k[0] (= phi_prime_k[0]), a_k[0]
    (= a_prime_k[0]), b_k[0] and b_prime_k[0] = b_k[0].eval(0)."""

    def __init__(self, phi, a, b_prime, b=None):
        self.phi, self.a, self.b_prime, self.b = phi, a, int(b_prime), b

    @classmethod
    def new(cls, ctx, witness, constants, rng):
        """State::new / gen_f (src/structs.rs:289-387) with a numpy Generator standing in for thread_rng: a_constraints
        symmetric with a_ij (i <= j in generation order) uniform in [0, q)^64, phi uniform, b computed on the GPU."""
        k, q = constants, ctx.q
        a = np.zeros((k.R, k.R, D), dtype=np.uint64)
        filled = np.zeros((k.R, k.R), dtype=bool)
        for i in range(k.R):
            for j in range(k.R):
                a_ij = rng.integers(0, q, size=D, dtype=np.uint64)      # drawn for every (i, j) like the reference
                if not filled[i, j]:                                     # `== Rq::new(vec![])`: still unset (src/structs.rs:297)
                    a[i, j] = a_ij
                    a[j, i] = a_ij
                    filled[i, j] = filled[j, i] = True
        phi = np.empty((k.N, k.R, D), dtype=np.uint64)
        for i in range(k.R):
            for j in range(k.N):
                phi[j, i] = rng.integers(0, q, size=D, dtype=np.uint64)
        b = gen_f_b(ctx, witness, phi, a, k.N, k.R)
        return cls(phi, a, int(b[0]), b)


class Transcript:
    """Computed fields of src/structs.rs:192-209 (numpy uint64 / int64) plus the tape that produced them."""
    pass


class BatchTranscript(Transcript):
    """Transcript j of a batch whose outputs live in one device tensor per field ([nb, ...]): the per-proof views are
    made on first access (building 9 views for each of 64 proofs per call costs more than launching the kernels)."""

    def __init__(self, big, j, tape):
        self.__dict__["_big"], self.__dict__["_j"] = big, j
        self.tape = tape

    def __getattr__(self, name):
        big = self.__dict__["_big"]
        if name in big:
            v = big[name][self.__dict__["_j"]]
            self.__dict__[name] = v
            return v
        raise AttributeError(name)


class Prover:
    def __init__(self, witness, verifier, constants, ctx):
        self.witness, self.verifier, self.constants, self.ctx = witness, verifier, constants, ctx

    def proof_gen(self, st, crs, max_rejections=5):
        """src/proofgen.rs:30-427.  Raises RuntimeError("failed JL...") after >5 rejected projections (:175-177)
        and BppCheckFailed (an AssertionError) where the reference's verify_b_prime_prime asserts."""
        k, v = self.constants, self.verifier
        rejections = 0
        while True:
            tape = dict(Pi=v.sample_jl_projection_all(), psi=v.generate_psi(), omega=v.generate_omega(),
                        alpha=v.fetch_alpha(), beta=v.fetch_beta(), c=v.fetch_challenges())
            try:
                return prove_core(self.ctx, k, self.witness, st, crs, tape)
            except _lib.JLRejected:
                rejections += 1
                if rejections > max_rejections:
                    raise RuntimeError("failed JL...")


_OUT_FIELDS = ("u_1", "projection_int", "projection", "b_prime_prime", "u_2", "z", "t_i_all", "g_mat", "h_mat")


def _out_shapes(k):
    return {"u_1": (k.KAPPA_1, D), "projection_int": (256,), "projection": (256,), "b_prime_prime": (D,),
            "u_2": (k.KAPPA_2, D), "z": (k.N, D), "t_i_all": (k.R, k.KAPPA, D), "g_mat": (k.R, k.R, D), "h_mat": (k.R, k.R, D)}


def _io_template(k, crs):
    """The fields of lab_prove_input that are the same for every proof under one CRS."""
    pin = _lib.lab_prove_input()
    pin.struct_size = C.sizeof(_lib.lab_prove_input)
    pin.n, pin.r, pin.kappa, pin.kappa1, pin.kappa2 = k.N, k.R, k.KAPPA, k.KAPPA_1, k.KAPPA_2
    pin.b1, pin.b2, pin.t1, pin.t2, pin.beta_bound = k.B_1, k.B_2, k.T_1, k.T_2, k.BETA_BOUND
    pin.A, pin.B = crs.A._h, (crs.B._h if crs.B is not None else None)
    pin.C, pin.Dm = (crs.C._h if crs.C is not None else None), (crs.Dm._h if crs.Dm is not None else None)
    for name in ("B8", "A8", "C8", "D8", "A8d"):
        m = getattr(crs, name, None)
        setattr(pin, name, m._h if m is not None else None)
    pin.crs_seed = getattr(crs, "crs_seed", None)          # B streamed from the seed when crs.B and crs.B8 are absent
    pin.a_off = pin.b_off = pin.c_off = pin.d_off = 0
    return pin


def pack_pi(ctx, Pi):
    """sample_jl_projection's matrices, int8 {-1,0,1} of shape (..., 256, cols), to the tiled 2-bit form of
    lab_prove_input.Pi2 (lab_pi_pack): uint8 of shape (..., 256*cols/4); numpy in -> numpy out, cuda tensor in -> cuda
    tensor out."""
    cols = Pi.shape[-1]
    if isinstance(Pi, np.ndarray):
        Pi = np.ascontiguousarray(Pi, dtype=np.int8)
        out = np.empty(Pi.shape[:-2] + (256 * cols // 4,), dtype=np.uint8)
        count = Pi.size
    else:
        import torch
        Pi = Pi.contiguous()
        out = torch.empty(tuple(Pi.shape[:-2]) + (256 * cols // 4,), dtype=torch.uint8, device=Pi.device)
        count = Pi.numel()
    if Pi.shape[-2] != 256:
        raise ValueError("projection matrices have 256 rows")
    ctx._check(ctx._lib.lab_pi_pack(ctx._h, _ptr(out), _ptr(Pi), count // (256 * cols), cols))
    return out


def pack_pi_host(Pi):
    """The same packing in numpy (reference for tests; bench.py's host path): field = entry + 1 in {0, 1, 2}; dword
    ((i*4 + s)*(cols/16) + ct)*64 + 16*kq + cc = rows 64s + 16kq + 0..15 of column 16ct + cc, row 4j + k at bit 8k + 2j."""
    Pi = np.asarray(Pi, dtype=np.int8)
    lead, cols = Pi.shape[:-2], Pi.shape[-1]
    f = (Pi.astype(np.int16) + 1).astype(np.uint8).reshape((-1, 4, 4, 4, 4, cols // 16, 16))      # [mat, s, kq, j, k, ct, cc]
    b = f[:, :, :, 0] | (f[:, :, :, 1] << 2) | (f[:, :, :, 2] << 4) | (f[:, :, :, 3] << 6)          # [mat, s, kq, k, ct, cc]
    return np.ascontiguousarray(b.transpose(0, 1, 4, 2, 5, 3)).reshape(lead + (256 * cols // 4,))   # [mat, s, ct, kq, cc, k]


def unpack_pi_host(packed, cols):
    """Inverse of pack_pi_host / lab_pi_pack: uint8 (..., 256*cols/4) -> int8 (..., 256, cols)."""
    packed = np.asarray(packed, dtype=np.uint8)
    lead = packed.shape[:-1]
    b = packed.reshape((-1, 4, cols // 16, 4, 16, 4)).transpose(0, 1, 3, 5, 2, 4)                 # [mat, s, kq, k, ct, cc]
    f = np.stack([(b >> (2 * j)) & 3 for j in range(4)], axis=3)                                    # [mat, s, kq, j, k, ct, cc]
    return (f.astype(np.int8) - 1).reshape(lead + (256, cols))


def _set_pi(pin, tape, keep):
    """tape["Pi2"] (2 bits per entry, pack_pi) takes precedence over tape["Pi"] (int8)."""
    if tape.get("Pi2") is not None:
        Pi2 = tape["Pi2"]
        if isinstance(Pi2, np.ndarray):
            Pi2 = np.ascontiguousarray(Pi2, dtype=np.uint8)
        keep.append(Pi2)
        pin.Pi2, pin.Pi = _ptr(Pi2), None
        return
    Pi = tape["Pi"]
    if isinstance(Pi, np.ndarray):
        Pi = np.ascontiguousarray(Pi, dtype=np.int8)
    keep.append(Pi)
    pin.Pi, pin.Pi2 = _ptr(Pi), None


def _prove_io(ctx, k, witness, st, crs, tape, device_outputs, out_views=None, template=None):
    """ctypes input/output structs for one proof; returns (lab_prove_input, lab_transcript, Transcript, keep-alive list).
    out_views: preallocated device tensors for the nine outputs (prove_core_batch slices them out of one allocation
    per field instead of nine allocations per proof); template: _io_template(k, crs), copied instead of refilled."""
    n, r = k.N, k.R
    pin = _lib.lab_prove_input.from_buffer_copy(template) if template is not None else _io_template(k, crs)
    keep = []

    def p64(a):
        if hasattr(a, "data_ptr") and a.is_contiguous() and a.element_size() == 8:   # device/host tensor: no conversion
            keep.append(a)
            return a.data_ptr()
        a = _as_u64(a)
        keep.append(a)
        return _ptr(a)

    pin.S, pin.phi, pin.a = p64(witness), p64(st.phi), p64(st.a)
    pin.b_prime = st.b_prime
    _set_pi(pin, tape, keep)
    pin.psi = int(tape["psi"])
    pin.omega, pin.alpha, pin.beta, pin.c = p64(tape["omega"]), p64(tape["alpha"]), p64(tape["beta"]), p64(tape["c"])

    if device_outputs:
        import torch

        def mk(shape, dt=None):
            return torch.empty(shape, dtype=dt or torch.int64, device="cuda")
    else:
        def mk(shape, dt=None):
            return np.empty(shape, dtype=np.int64 if dt == "i64" else np.uint64)
    if out_views is not None:      # (big tensors, proof index, byte addresses of this proof's slices)
        big, j, addrs = out_views
        tr = _lib.lab_transcript()
        for name, addr in zip(_OUT_FIELDS, addrs):
            setattr(tr, name, addr)
        return pin, tr, BatchTranscript(big, j, tape), keep
    T = Transcript()
    T.u_1 = mk((k.KAPPA_1, D)); T.projection_int = mk((256,), None if device_outputs else "i64")
    T.projection = mk((256,)); T.b_prime_prime = mk((D,)); T.u_2 = mk((k.KAPPA_2, D)); T.z = mk((n, D))
    T.t_i_all = mk((r, k.KAPPA, D)); T.g_mat = mk((r, r, D)); T.h_mat = mk((r, r, D))
    tr = _lib.lab_transcript()
    for name in _OUT_FIELDS:
        setattr(tr, name, _ptr(getattr(T, name)))
    T.tape = tape
    return pin, tr, T, keep


def prove_core(ctx, k, witness, st, crs, tape, device_outputs=False):
    """One call of lab_prove_core.  Arrays may be numpy (host) or torch cuda tensors (device resident)."""
    pin, tr, T, keep = _prove_io(ctx, k, witness, st, crs, tape, device_outputs)
    ctx._check(ctx._lib.lab_prove_core(ctx._h, C.byref(pin), C.byref(tr)))
    return T


def prove_core_batch(ctx, k, witnesses, states, crs, tapes, device_outputs=False, raise_on_reject=True):
    """lab_prove_core_batch: independent proofs (own witness, state, tape) under ONE common reference string; the
    CRS matrices are streamed from HBM once per group of proofs (up to 64 on the matrix-core path).  Returns the list of
    Transcripts (bit-identical to separate prove_core calls); with raise_on_reject=False also the status codes."""
    nb = len(witnesses)
    views = [None] * nb
    if device_outputs:      # one device allocation per output field for the whole batch
        import torch
        big = {name: torch.empty((nb,) + shape, dtype=torch.int64, device="cuda") for name, shape in _out_shapes(k).items()}
        base = [(big[name].data_ptr(), big[name][0].numel() * 8) for name in _OUT_FIELDS]
        views = [(big, j, [p + j * step for p, step in base]) for j in range(nb)]
    template = _io_template(k, crs)
    ios = [_prove_io(ctx, k, witnesses[j], states[j], crs, tapes[j], device_outputs, views[j], template) for j in range(nb)]
    pins = (C.POINTER(_lib.lab_prove_input) * nb)(*[C.pointer(io[0]) for io in ios])
    trs = (C.POINTER(_lib.lab_transcript) * nb)(*[C.pointer(io[1]) for io in ios])
    status = (C.c_int * nb)()
    ctx._check(ctx._lib.lab_prove_core_batch(ctx._h, nb, pins, trs, status))
    codes = [int(x) for x in status]
    if raise_on_reject:
        for code in codes:
            _lib.check(ctx._h, code)
        return [io[2] for io in ios]
    return [io[2] for io in ios], codes


class BatchPlan:
    """lab_prove_core_batch with the boundary structures built ONCE: the proofs' witness / state / tape arrays and the output
    buffers are persistent (device tensors or host arrays the caller refills between runs), every run() proves what they hold
    now and returns the same Transcript objects, overwritten.  What a prover service that reuses its buffers does; it takes
    the per-call marshalling (0.4 ms for 64 proofs) out of the loop."""

    def __init__(self, ctx, k, witnesses, states, crs, tapes, device_outputs=True):
        self.ctx, self.nb = ctx, len(witnesses)
        nb = self.nb
        views = [None] * nb
        if device_outputs:
            import torch
            big = {name: torch.empty((nb,) + shape, dtype=torch.int64, device="cuda") for name, shape in _out_shapes(k).items()}
            base = [(big[name].data_ptr(), big[name][0].numel() * 8) for name in _OUT_FIELDS]
            views = [(big, j, [p + j * step for p, step in base]) for j in range(nb)]
        template = _io_template(k, crs)
        self._keep = (witnesses, states, crs, tapes)          # the arrays the structures point into
        self._ios = [_prove_io(ctx, k, witnesses[j], states[j], crs, tapes[j], device_outputs, views[j], template) for j in range(nb)]
        self._pins = (C.POINTER(_lib.lab_prove_input) * nb)(*[C.pointer(io[0]) for io in self._ios])
        self._trs = (C.POINTER(_lib.lab_transcript) * nb)(*[C.pointer(io[1]) for io in self._ios])
        self._status = (C.c_int * nb)()
        self.transcripts = [io[2] for io in self._ios]

    def run(self, raise_on_reject=True):
        ctx = self.ctx
        ctx._check(ctx._lib.lab_prove_core_batch(ctx._h, self.nb, self._pins, self._trs, self._status))
        if raise_on_reject:
            for code in self._status:
                if code:
                    _lib.check(ctx._h, int(code))
            return self.transcripts
        return self.transcripts, [int(x) for x in self._status]


def verify_core(ctx, k, st, b_poly, crs, T, tape=None):
    """Verifier::verify (src/verification.rs:25-438) on the GPU via lab_verify_core.
    Returns (failed_line, line14_sum): failed_line == 0 means the reference would return true, otherwise it is
    the number of the first failing protocol line.  `T` is a Transcript (or any object with its fields)."""
    tape = tape if tape is not None else T.tape
    n, r = k.N, k.R
    pin = _lib.lab_prove_input()
    pin.struct_size = C.sizeof(_lib.lab_prove_input)
    pin.n, pin.r, pin.kappa, pin.kappa1, pin.kappa2 = n, r, k.KAPPA, k.KAPPA_1, k.KAPPA_2
    pin.b1, pin.b2, pin.t1, pin.t2, pin.beta_bound = k.B_1, k.B_2, k.T_1, k.T_2, k.BETA_BOUND
    keep = []

    def p64(a):
        a = _as_u64(a)
        keep.append(a)
        return _ptr(a)

    pin.S = None
    pin.phi, pin.a = p64(st.phi), p64(st.a)
    pin.b_prime = st.b_prime
    pin.A, pin.B = crs.A._h, (crs.B._h if crs.B is not None else None)
    pin.C, pin.Dm = (crs.C._h if crs.C is not None else None), (crs.Dm._h if crs.Dm is not None else None)
    pin.B8 = crs.B8._h if getattr(crs, "B8", None) is not None else None
    pin.A8 = crs.A8._h if getattr(crs, "A8", None) is not None else None
    pin.C8 = crs.C8._h if getattr(crs, "C8", None) is not None else None
    pin.D8 = crs.D8._h if getattr(crs, "D8", None) is not None else None
    pin.A8d = crs.A8d._h if getattr(crs, "A8d", None) is not None else None
    pin.a_off = pin.b_off = pin.c_off = pin.d_off = 0
    _set_pi(pin, tape, keep)
    pin.psi = int(tape["psi"])
    pin.omega, pin.alpha, pin.beta, pin.c = p64(tape["omega"]), p64(tape["alpha"]), p64(tape["beta"]), p64(tape["c"])
    tr = _lib.lab_transcript()
    for name in ("u_1", "projection", "b_prime_prime", "u_2", "z", "t_i_all", "g_mat", "h_mat"):
        setattr(tr, name, p64(getattr(T, name)))
    tr.projection_int = None
    line, s14 = C.c_int(-1), C.c_double()
    ctx._check(ctx._lib.lab_verify_core(ctx._h, C.byref(pin), C.byref(tr), p64(b_poly), int(k.B), float(k.BETA_PRIME),
                                        C.byref(line), C.byref(s14)))
    return line.value, s14.value


class Verifier:
    """Mirror of src/verification.rs:12-23: `Verifier::new(b_prime_k, &constants)`; verify() runs on the GPU.
    The challenge samplers stay host-side objects supplied by the caller (they are the interactive verifier)."""

    def __init__(self, b_prime_k, constants, ctx):
        self.b_prime, self.constants, self.ctx = b_prime_k, constants, ctx

    def verify(self, st, b_poly, proof, crs, beta_prime=None):
        """-> bool, like Verifier::verify(&self, &State, &Transcript, &mut CRS)."""
        k = self.constants
        if beta_prime is not None:
            import copy
            k = copy.copy(k)
            k.BETA_PRIME = beta_prime
        line, _ = verify_core(self.ctx, k, st, b_poly, crs, proof)
        return line == 0


# ---- challenge space sampling (src/verification.rs:460-489, src/util.rs:83-104, 227-246) — SURVEY §8(f) row f4.
# The reference draws everything from thread_rng, so only the arithmetic is reproducible: given the sample
# polynomials, operator_norm is max_k sqrt(f64(|c r_k|^2)) / sqrt(f64(|r_k|^2)) with exact integer norms.

def operator_norm(ctx, c, samples):
    """util.rs:227-246 with the n_samples random polynomials `samples` ([k][64]) supplied by the caller: one batched
    ring product and two exact norm kernels on the GPU, the f64 ratio/max on the host (IEEE sqrt and division)."""
    import math
    from .api import poly_norm_exact
    samples = _as_u64(samples)
    k = samples.shape[0]
    prod = ctx.negacyclic_polymul(np.broadcast_to(_as_u64(c).reshape(1, D), (k, D)).copy(), samples)
    num, den = poly_norm_exact(ctx, prod), poly_norm_exact(ctx, samples)
    sup = 0.0
    for a, b in zip(num, den):
        ratio = math.sqrt(float(a)) / math.sqrt(float(b))       # poly_norm_strict(c*r) / poly_norm_strict(r); NaN never wins
        if ratio > sup:
            sup = ratio
    return sup


def fetch_challenge(ctx, rng, q, T=15.0, n_samples=1000):
    """Verifier::fetch_challenge (src/verification.rs:460-489) with a numpy Generator standing in for thread_rng:
    a random signed arrangement of 23 zeros, 31 ones, 10 twos, redrawn until operator_norm(candidate) <= T."""
    while True:
        vals = np.array([0] * 23 + [1] * 31 + [2] * 10, dtype=np.int64)
        rng.shuffle(vals)
        signs = rng.integers(0, 2, size=D)
        cand = np.where((vals > 0) & (signs == 1), q - vals, vals).astype(np.uint64)
        samples = rng.integers(0, q, size=(n_samples, D), dtype=np.uint64)
        if operator_norm(ctx, cand, samples) <= T:
            return cand
