"""The single-limb strict-32 path (path (b)): an NTT-friendly prime q in [2^30, 2^32) as its own NTT prime.

Policy (lab_ctx_create / lab_ctx_create_ex), the carry and exact-product corners of the strict arithmetic at primes on both
sides of 2^31 and next to 2^32 (csrc/zq_device.h Zp<true>, ntt64_*_wide; csrc/wide_kernels.hip k_wide_dot), and equality of
path (a) (LAB_CTX_RNS_ONLY: three RNS limbs) with path (b) and with the oracle on the same inputs.  Reference anchor: the
commented 32-bit Q of src/constants.rs:195-196 behind the seam src/algebraic.rs:379-404; the oracle is the same schoolbook
code with q as a parameter.  Bit-exact, through the C ABI.  (Every test parametrised over conftest.ALL_Q — ring ops, linear
algebra, prover, verifier, batch and matrix-core paths — runs at q = 4294966657 on both paths as well.)
"""
import numpy as np
import pytest

from conftest import Q_NTT32

pytestmark = pytest.mark.gpu
D = 64
# q == 1 (mod 128) primes: just above 2^30, on both sides of 2^31 (sums of two residues start to overflow 32 bits), 3 * 2^30 + 1,
# and the two largest below 2^32
WIDE_PRIMES = [1073741953, 2147479681, 2147483777, 3221225473, 4294962817, Q_NTT32]


def rand_polys(rng, q, count):
    return rng.integers(0, q, size=(count, D), dtype=np.uint64)


def test_policy_single_limb_and_rns_only():
    import labrador_snark_amd as lab
    c = lab.Context(Q_NTT32, 0)
    assert c.nlimbs == 1 and c.primes == [Q_NTT32]            # 4 bytes per coefficient, no CRT
    r = lab.Context(Q_NTT32, 0, rns_only=True)
    assert r.nlimbs == 3 and Q_NTT32 not in r.primes          # the generic exact path at the same modulus
    c.close(); r.close()
    d = lab.Context(1073738753, 0, rns_only=True)             # ... also overrides the lazy single-limb mode below 2^30
    assert d.nlimbs > 1
    d.close()
    n = lab.Context((1 << 32) + 15, 0)                        # not NTT-friendly: RNS either way
    assert n.nlimbs == 3
    n.close()
    import ctypes as C
    from labrador_snark_amd import _lib
    h = C.c_void_p()
    assert _lib.load().lab_ctx_create_ex(Q_NTT32, 0, 0, 2, C.byref(h)) == _lib.LAB_ERR_ARG and not h.value   # unknown flag


@pytest.mark.parametrize("q", WIDE_PRIMES)
def test_polymul_corners(q):
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = lab.Context(q, 0)
    assert ctx.nlimbs == 1
    rng = np.random.default_rng(q & 0xFFFF)
    a, b = rand_polys(rng, q, 200), rand_polys(rng, q, 200)
    a[0] = 0; b[1] = 0
    a[2] = q - 1; b[2] = q - 1                                 # every butterfly sum and difference at the modulus edge
    a[3, 1:] = 0; a[3, 0] = 1
    a[4, :] = 0; a[4, 63] = q - 1; b[4, :] = 0; b[4, 63] = q - 1
    a[5] = q - 1; b[5] = 1
    a[6] = (q - 1) // 2; b[6] = (q + 1) // 2                    # sums that land exactly on q and q - 1
    a[7] = 1 << 31 if q > (1 << 31) else (q - 1); b[7] = 2      # products 2^32 = carry out of the low word
    assert np.array_equal(ctx.negacyclic_polymul(a, b), co.polymul_batch(q, a, b))
    # transform round trip at ragged sizes
    for count in (1, 63, 65):
        x = rand_polys(rng, q, count)
        assert np.array_equal(ctx.vec_from(x).store(), x)
    ctx.close()


@pytest.mark.parametrize("q", WIDE_PRIMES)
@pytest.mark.parametrize("length", [1, 5, 16, 131, 4096])
def test_inner_product_extremes(q, length):
    # all-(q-1) operands: every product is (q-1)^2 ~ 2^64 - the two-accumulator sum and its final reduction at their largest
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = lab.Context(q, 0)
    rng = np.random.default_rng(length)
    a, b = rand_polys(rng, q, length), rand_polys(rng, q, length)
    if length >= 5:
        a[: length // 2] = q - 1
        b[: length // 2] = q - 1
    got = lab.polynomial_vec_inner_product(ctx, ctx.vec_from(a), ctx.vec_from(b)).store()[0]
    assert np.array_equal(got, co.inner_product(q, a, b))
    ctx.close()


def test_paths_a_and_b_agree_on_the_linear_algebra():
    # outer-commitment shaped mat-vec sums, the Ajtai product, a Gram product: path (a) == path (b) == oracle
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_NTT32
    cb, ca = lab.Context(q, 0), lab.Context(q, 0, rns_only=True)
    rng = np.random.default_rng(12)
    nmat, rows, cols = 5, 37, 29
    M = rand_polys(rng, q, nmat * rows * cols).reshape(nmat, rows, cols, D)
    V = rand_polys(rng, q, nmat * cols).reshape(nmat, cols, D)
    M[0, 0] = q - 1; V[0] = q - 1
    outs = []
    for c in (cb, ca):
        outs.append(lab.matvec_sum(c, c.vec_from(M), c.vec_from(V), nmat, rows, cols).store())
    want = np.zeros((rows, D), dtype=np.uint64)
    for r_ in range(rows):
        acc = np.zeros(D, dtype=object)
        for m in range(nmat):
            acc = (acc + co.inner_product(q, M[m, r_], V[m]).astype(object)) % q
        want[r_] = acc.astype(np.uint64)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0].reshape(rows, D), want)
    # Ajtai: t_i = A s_i
    kappa, n, r = 9, 70, 5
    A = rand_polys(rng, q, kappa * n).reshape(kappa, n, D)
    S = rand_polys(rng, q, n * r).reshape(n, r, D)
    ts = [lab.ajtai_commit(c, c.vec_from(A), c.vec_from(S), kappa, n, r).store() for c in (cb, ca)]
    assert np.array_equal(ts[0], ts[1])
    t = ts[0].reshape(r, kappa, D)
    for (i, rho) in ((0, 0), (4, 8), (2, 3)):
        assert np.array_equal(t[i, rho], co.inner_product(q, A[rho], np.ascontiguousarray(S[:, i])))
    cb.close(); ca.close()
