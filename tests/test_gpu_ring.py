"""GPU parity: the reference's operator seam and inner products vs the CPU oracle.

Mirrors tests/proptest.rs of the reference (NTT == classic, linearity) with the
oracle as the classic path.  Bit-exact: integer work, no tolerance.
"""
import numpy as np
import pytest

from conftest import ALL_Q, Q_DEFAULT, Q_REF33

pytestmark = pytest.mark.gpu

D = 64


def rand_polys(rng, q, count):
    return rng.integers(0, q, size=(count, D), dtype=np.uint64)


@pytest.mark.parametrize("q", ALL_Q)
def test_polymul_matches_classic(contexts, q):
    # tests/proptest.rs:13-24 test_NTT_preserves_result, 50 cases -> one batch of 300 incl. edge rows
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(q & 0xFFFF)
    a, b = rand_polys(rng, q, 300), rand_polys(rng, q, 300)
    a[0] = 0; b[1] = 0; a[2] = q - 1; b[2] = q - 1          # zero poly (empty Vec in the reference), max residues
    a[3, 1:] = 0; a[3, 0] = 1                                # identity
    a[4, :] = 0; a[4, 63] = q - 1; b[4, :] = 0; b[4, 63] = q - 1  # X^63 * X^63 wraps with sign
    got = ctx.negacyclic_polymul(a, b)
    want = co.polymul_batch(q, a, b)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("q", ALL_Q)
@pytest.mark.parametrize("count", [1, 63, 64, 65, 1000])
def test_load_store_roundtrip(contexts, q, count):
    ctx = contexts(q)
    rng = np.random.default_rng(count)
    a = rand_polys(rng, q, count)
    v = ctx.vec_from(a)
    assert np.array_equal(v.store(), a)


@pytest.mark.parametrize("q", ALL_Q)
@pytest.mark.parametrize("length", [1, 3, 16, 37, 130, 1024])
def test_inner_product_matches_oracle(contexts, q, length):
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(length * 7 + 1)
    a, b = rand_polys(rng, q, length), rand_polys(rng, q, length)
    got = lab.polynomial_vec_inner_product(ctx, ctx.vec_from(a), ctx.vec_from(b)).store()[0]
    assert np.array_equal(got, co.inner_product(q, a, b))


@pytest.mark.parametrize("q", [Q_DEFAULT, Q_REF33])
def test_inner_product_linearity(contexts, q):
    # tests/proptest.rs:36-64: <a, c*b> == c*<a,b>, vectors of 16 ring elements
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(5)
    for _ in range(5):
        a, b = rand_polys(rng, q, 16), rand_polys(rng, q, 16)
        c = int(rng.integers(0, q))
        cb = np.stack([co.poly_scale(q, row, c) for row in b])
        lhs = lab.polynomial_vec_inner_product(ctx, ctx.vec_from(a), ctx.vec_from(cb)).store()[0]
        ab = lab.polynomial_vec_inner_product(ctx, ctx.vec_from(a), ctx.vec_from(b)).store()[0]
        assert np.array_equal(lhs, co.poly_scale(q, ab, c))


def test_inner_product_length_mismatch(contexts):
    # src/util.rs:497-502 assert! -> LengthMismatch (an AssertionError)
    import labrador_snark_amd as lab
    ctx = contexts(Q_DEFAULT)
    a = ctx.vec_from(np.zeros((3, D), np.uint64))
    b = ctx.vec_from(np.zeros((4, D), np.uint64))
    with pytest.raises(AssertionError):
        lab.polynomial_vec_inner_product(ctx, a, b)


@pytest.mark.parametrize("q", ALL_Q)
@pytest.mark.parametrize("nmat,rows,cols", [(1, 5, 33), (3, 7, 64), (2, 4, 129), (6, 9, 20),
                                            (1, 5, 1500)])  # long rows: every wave of the block holds a partial sum
def test_matvec_sum_matches_oracle(contexts, q, nmat, rows, cols):
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(nmat * 100 + rows)
    M = rand_polys(rng, q, nmat * rows * cols).reshape(nmat, rows, cols, D)
    V = rand_polys(rng, q, nmat * cols).reshape(nmat, cols, D)
    got = lab.matvec_sum(ctx, ctx.vec_from(M), ctx.vec_from(V), nmat, rows, cols).store()
    want = np.zeros((rows, D), np.uint64)
    for m in range(nmat):
        for r in range(rows):
            want[r] = co.poly_add(q, want[r], co.inner_product(q, M[m, r], V[m]))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("q", ALL_Q)
def test_synthetic_fill_matches_oracle(contexts, q):
    from oracle import coracle as co
    ctx = contexts(q)
    v = ctx.vec(100).fill_synthetic(seed=42, first_elem=7)
    want = co.synth_fill(42, 7 * D, 100 * D, q).reshape(100, D)
    assert np.array_equal(v.store(), want)


@pytest.mark.parametrize("q", ALL_Q)
def test_raw_ntt_roundtrip_and_convolution(contexts, q):
    # BASELINE config 2 at test size: forward then inverse is the identity on every limb plane
    import torch
    ctx = contexts(q)
    batch = 4096
    g = torch.Generator(device="cpu").manual_seed(2)
    planes = []
    for p in ctx.primes:
        planes.append(torch.randint(0, p, (batch, D), generator=g, dtype=torch.int64).to(torch.int32))
    x = torch.stack(planes).cuda()
    y = x.clone()
    torch.cuda.synchronize()  # the context runs on its own stream
    ctx.ntt_forward(y, batch)
    ctx.synchronize()
    assert not torch.equal(x, y)
    ctx.ntt_inverse(y, batch)
    ctx.synchronize()
    assert torch.equal(x, y)


def test_context_limb_selection_and_argument_errors():
    # limb count follows the exactness bound; unusable moduli and oversize reductions are refused with status codes
    import labrador_snark_amd as lab
    c = lab.Context(Q_DEFAULT, 0)
    assert c.nlimbs == 2 and c.max_len > 1 << 20
    c.close()
    c = lab.Context(Q_REF33, 0)
    assert c.nlimbs == 3 and c.max_len == 524285
    a = c.vec_from(np.zeros((4, D), np.uint64))
    with pytest.raises(lab.LabradorError):                     # offsets past the end
        lab.polynomial_vec_inner_product(c, a, a, off1=2, off2=0, length=3)
    c.close()
    c = lab.Context(Q_REF33, 0, max_len=1 << 22)               # a longer reduction needs the 4th limb
    assert c.nlimbs == 4
    rng = np.random.default_rng(0)
    x, y = rand_polys(rng, Q_REF33, 50), rand_polys(rng, Q_REF33, 50)
    from oracle import coracle as co
    assert np.array_equal(c.negacyclic_polymul(x, y), co.polymul_batch(Q_REF33, x, y))
    assert np.array_equal(lab.polynomial_vec_inner_product(c, c.vec_from(x), c.vec_from(y)).store()[0],
                          co.inner_product(Q_REF33, x, y))
    c.close()
    c = lab.Context(1073738753, 0)                              # q itself is the NTT prime: single limb
    assert c.nlimbs == 1 and c.primes == [1073738753]
    c.close()
    for bad in (0, 2, 8192, 1 << 40):                           # even / out of range
        with pytest.raises(lab.LabradorError):
            lab.Context(bad, 0)
    small = lab.Context(97, 0)                                  # tiny odd q (not 1 mod 128): generic path, 1 limb suffices
    u, v = rand_polys(rng, 97, 10), rand_polys(rng, 97, 10)
    assert np.array_equal(small.negacyclic_polymul(u, v), co.polymul_batch(97, u, v))
    small.close()
    big = lab.Context((1 << 39) + 23, 0, max_len=64)            # near the 2^40 ceiling (odd, not necessarily prime)
    qb = (1 << 39) + 23
    u, v = rand_polys(rng, qb, 20), rand_polys(rng, qb, 20)
    assert np.array_equal(big.negacyclic_polymul(u, v), co.polymul_batch(qb, u, v))
    assert np.array_equal(lab.polynomial_vec_inner_product(big, big.vec_from(u), big.vec_from(v)).store()[0],
                          co.inner_product(qb, u, v))
    big.close()


def test_reduction_longer_than_the_exactness_bound_is_refused(contexts):
    import labrador_snark_amd as lab
    ctx = contexts(Q_REF33)
    n = ctx.max_len + 3
    # vectors this long are never allocated: the length check comes first
    a = ctx.vec(8)
    with pytest.raises(lab.LabradorError) as e:
        ctx._check(ctx._lib.lab_matvec_sum(ctx._h, a._h, 0, a._h, 0, n, 1, 1, a._h, 0))
    assert e.value.code in (lab._lib.LAB_ERR_RANGE, lab._lib.LAB_ERR_LEN_MISMATCH)


def test_empty_batches_are_no_ops(contexts):
    ctx = contexts(Q_DEFAULT)
    out = ctx.negacyclic_polymul(np.zeros((0, D), np.uint64), np.zeros((0, D), np.uint64))
    assert out.shape == (0, D)
