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
@pytest.mark.parametrize("nmat,rows,cols", [(1, 5, 33), (3, 7, 64), (2, 4, 129), (6, 9, 20)])
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
