"""GPU parity of the per-step operations (src/util.rs helpers and the loop nests of proof_gen) vs the oracle.
Bit-exact integer work; every call goes through the C ABI."""
import numpy as np
import pytest

from conftest import ALL_Q, Q_DEFAULT, Q_REF33, Q_DIRECT, ctx_kw

pytestmark = pytest.mark.gpu
D = 64


def rp(rng, q, *shape):
    return rng.integers(0, q, size=shape + (D,), dtype=np.uint64)


@pytest.mark.parametrize("q", ALL_Q)
# the last four shapes (rows >= 32, even inner, ncols % 8 == 0) run on the lane=row LDS-DMA kernel, the others on k_matmul_t
@pytest.mark.parametrize("rows,inner,ncols", [(5, 7, 2), (9, 33, 16), (4, 64, 32), (13, 20, 35),
                                              (40, 6, 8), (64, 34, 16), (70, 2, 24), (600, 10, 32)])
def test_ajtai_commit(contexts, q, rows, inner, ncols):
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(rows * 31 + ncols)
    A, S = rp(rng, q, rows, inner), rp(rng, q, inner, ncols)
    got = lab.ajtai_commit(ctx, ctx.vec_from(A), ctx.vec_from(S), rows, inner, ncols).store().reshape(ncols, rows, D)
    for i in range(ncols):
        col = np.ascontiguousarray(S[:, i])
        for rho in (range(rows) if rows < 100 else list(range(0, rows, 37)) + [63, 64, rows - 1]):
            assert np.array_equal(got[i, rho], co.inner_product(q, A[rho], col)), (i, rho)


@pytest.mark.parametrize("q", ALL_Q)
@pytest.mark.parametrize("n,r", [(3, 2), (37, 5), (130, 4)])
def test_gram(contexts, q, n, r):
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(n + r)
    P, S = rp(rng, q, n, r), rp(rng, q, n, r)
    got = lab.gram(ctx, ctx.vec_from(P), ctx.vec_from(S), n, r).store().reshape(r, r, D)
    for i in range(r):
        for k in range(r):
            want = co.inner_product(q, np.ascontiguousarray(P[:, i]), np.ascontiguousarray(S[:, k]))
            assert np.array_equal(got[i, k], want)


@pytest.mark.parametrize("q", ALL_Q)
@pytest.mark.parametrize("nterms,length", [(1, 5), (12, 17), (64, 16), (130, 33)])
def test_axpy_commit(contexts, q, nterms, length):
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(nterms * 7 + length)
    G, C = rp(rng, q, nterms), rp(rng, q, nterms, length)
    got = lab.axpy_commit(ctx, ctx.vec_from(G), ctx.vec_from(C), nterms, length).store()
    want = np.zeros((length, D), np.uint64)
    for m in range(nterms):
        prod = co.poly_by_poly_vec(q, G[m], C[m])
        for rho in range(length):
            want[rho] = co.poly_add(q, want[rho], prod[rho])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("q", ALL_Q)
def test_amortize_z_pointwise_and_poly_by_vec(contexts, q):
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(9)
    n, r = 21, 6
    S, c = rp(rng, q, n, r), rp(rng, q, r)
    z = lab.amortize_z(ctx, ctx.vec_from(S), ctx.vec_from(c), n, r).store()
    for x in range(n):
        assert np.array_equal(z[x], co.inner_product(q, S[x], c))
    a, b = rp(rng, q, 70), rp(rng, q, 70)
    va, vb = ctx.vec_from(a), ctx.vec_from(b)
    assert np.array_equal(lab.pointwise(ctx, "mul", va, vb).store(), co.polymul_batch(q, a, b))
    assert np.array_equal(lab.add_poly_vec(ctx, va, vb).store(), np.stack([co.poly_add(q, x, y) for x, y in zip(a, b)]))
    assert np.array_equal(lab.pointwise(ctx, "sub", va, vb).store(), np.stack([co.poly_sub(q, x, y) for x, y in zip(a, b)]))
    p = rp(rng, q, 1)
    assert np.array_equal(lab.poly_by_poly_vec(ctx, ctx.vec_from(p), va).store(), co.poly_by_poly_vec(q, p[0], a))


@pytest.mark.parametrize("q", ALL_Q)
@pytest.mark.parametrize("base,exp", [(9, 4), (1625, 3), (14, 2), (638, 4), (2, 13), (3, 1), (255, 5), (256, 5), (65535, 3),
                                      (65536, 3), (100003, 2), (8190, 3), (8191, 2), (8192, 2)])
def test_decompose_bug_compatible(contexts, q, base, exp):
    # bases below 2^16 and not above q take the 32-bit long-division path, the others the double-reciprocal one
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(base)
    v = rp(rng, q, 40)
    v[0] = 0; v[1] = q - 1; v[2, :] = base // 2; v[3, :] = base // 2 + 1; v[4, 10:] = 0
    got = lab.decompose_polynomial_vec(ctx, v, base, exp)
    assert np.array_equal(got, co.decompose_vec(q, v, base, exp))


def test_decompose_base_one_is_an_error(contexts):
    import labrador_snark_amd as lab
    ctx = contexts(Q_DEFAULT)
    with pytest.raises(lab.LabradorError):
        lab.decompose_polynomial_vec(ctx, np.ones((1, D), np.uint64), 1, 3)  # the reference loops forever (B_1 = 1 at n=1024)


@pytest.mark.parametrize("q", ALL_Q)
def test_norms_scale_sigma(contexts, q):
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(12)
    v = rp(rng, q, 33)
    v[0] = 0; v[1] = q - 1
    exact = lab.poly_norm_exact(ctx, v)
    for k in range(33):
        f, e = co.poly_norm(v[k])
        assert exact[k] == e and float(exact[k]) == f
    assert lab.vec_poly_norm_squared(ctx, v) == co.lib().orc_vec_poly_norm_squared(v, 33)
    s = int(rng.integers(0, q))
    assert np.array_equal(lab.scale_poly_vec(ctx, v, s), np.stack([co.poly_scale(q, row, s) for row in v]))
    assert np.array_equal(lab.sigma_inv_vec(ctx, v), np.stack([co.sigma_inv(q, row) for row in v]))


@pytest.mark.parametrize("q", ALL_Q)
@pytest.mark.parametrize("n,r", [(1, 1), (2, 2), (5, 3), (64, 2)])
def test_jl_project(contexts, q, n, r):
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(n * 10 + r)
    S = rp(rng, q, n, r)
    Pi = rng.choice(np.array([-1, 0, 1], dtype=np.int8), size=(r, 256, n * D), p=[0.25, 0.5, 0.25]).astype(np.int8)
    got = lab.jl_project(ctx, Pi, S, n, r)
    s_cols = np.ascontiguousarray(S.transpose(1, 0, 2).reshape(r, n * D))
    assert np.array_equal(got, co.jl_project(Pi, s_cols))


def test_device_resident_io(contexts):
    # same entry points with device pointers (torch tensors): nothing is staged through the host
    import torch
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_REF33
    ctx = contexts(q)
    rng = np.random.default_rng(77)
    a, b = rp(rng, q, 128), rp(rng, q, 128)
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    torch.cuda.synchronize()
    out = ctx.negacyclic_polymul(da, db)
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), co.polymul_batch(q, a, b))
    dec = lab.decompose_polynomial_vec(ctx, da, 1625, 3)
    ctx.synchronize()
    assert np.array_equal(dec.cpu().numpy().view(np.uint64), co.decompose_vec(q, a, 1625, 3))


@pytest.mark.parametrize("q", [Q_DEFAULT, Q_REF33])
@pytest.mark.parametrize("compact", [False, True])
def test_digit_basis_matvec_and_axpy(q, compact):
    # matrices times decomposition DIGITS in the smaller RNS basis (u16 last limb when compact): same exact result
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = lab.Context(q, 0, **ctx_kw(q))
    bound = 1625
    nmat, rows, cols = 3, 9, 40
    primes, small, nbytes = ctx.enable_digit_basis(5000, bound)
    assert nbytes < 4 * ctx.nlimbs and small
    rng = np.random.default_rng(3)
    M = rp(rng, q, nmat, rows, cols)
    V = rng.integers(0, bound, size=(nmat, cols, D), dtype=np.uint64)
    hM = ctx.vec(nmat * rows * cols, digit=True, compact=compact).load(M)
    hV = ctx.vec(nmat * cols, digit=True).load(V)
    assert np.array_equal(hM.store(), M.reshape(-1, D)) and np.array_equal(hV.store(), V.reshape(-1, D))
    got = lab.matvec_sum(ctx, hM, hV, nmat, rows, cols).store()
    want = np.zeros((rows, D), np.uint64)
    for m in range(nmat):
        for r_ in range(rows):
            want[r_] = co.poly_add(q, want[r_], co.inner_product(q, M[m, r_], V[m]))
    assert np.array_equal(got, want)
    nterms, length = 130, 21
    G = rng.integers(0, bound, size=(nterms, D), dtype=np.uint64)
    Cm = rp(rng, q, nterms, length)
    hC = ctx.vec(nterms * length, digit=True, compact=compact).load(Cm)
    got = lab.axpy_commit(ctx, ctx.vec(nterms, digit=True).load(G), hC, nterms, length).store()
    want = np.zeros((length, D), np.uint64)
    for m in range(nterms):
        prod = co.poly_by_poly_vec(q, G[m], Cm[m])
        for rho in range(length):
            want[rho] = co.poly_add(q, want[rho], prod[rho])
    assert np.array_equal(got, want)
    # synthetic fill lands on the same coefficients in either storage
    f = ctx.vec(70, digit=True, compact=compact).fill_synthetic(9)
    assert np.array_equal(f.store(), co.synth_fill(9, 0, 70 * D, q).reshape(70, D))
    # misuse is refused: compact vectors are not general operands, bases do not mix, lengths are bounded
    with pytest.raises(lab.LabradorError):
        lab.pointwise(ctx, "add", hM, hM) if compact else lab.matvec_sum(ctx, hM, ctx.vec_from(V), nmat, rows, cols)
    with pytest.raises(lab.LabradorError):
        lab.matvec_sum(ctx, ctx.vec(2 * 6000, digit=True), ctx.vec(6000, digit=True), 1, 2, 6000)
    ctx.close()


def test_digit_basis_not_offered_for_direct_modulus(contexts):
    import labrador_snark_amd as lab
    ctx = lab.Context(Q_DIRECT, 0)
    with pytest.raises(lab.LabradorError):
        ctx.enable_digit_basis(1000, 1625)
    ctx.close()


@pytest.mark.parametrize("q", [Q_DEFAULT, Q_REF33])
def test_operator_norm_and_fetch_challenge(contexts, q):
    # util.rs:227-246 / verification.rs:460-489 with the random draws supplied: same f64 value as the oracle route
    import math
    import labrador_snark_amd as lab
    from oracle import coracle as co
    ctx = contexts(q)
    rng = np.random.default_rng(15)
    c = rp(rng, 3, 1)[0]
    samples = rp(rng, q, 200)
    want = 0.0
    for rpoly in samples:
        num = co.poly_norm(co.polymul(q, c, rpoly))[0]
        den = co.poly_norm(rpoly)[0]
        want = max(want, math.sqrt(num) / math.sqrt(den))
    assert lab.operator_norm(ctx, c, samples) == want
    ch = lab.fetch_challenge(ctx, np.random.default_rng(1), q, T=float("inf"), n_samples=50)
    small = np.minimum(ch, q - ch)
    assert sorted(small.tolist()) == [0] * 23 + [1] * 31 + [2] * 10


def test_sharded_outer_commitment_single_rank(contexts):
    # parallel.sharded_outer_commitment with world size 1 equals lab_matvec_sum over all (i,k) (src/proofgen.rs:101-132)
    import torch
    import labrador_snark_amd as lab
    from labrador_snark_amd import parallel
    q = Q_REF33
    ctx = contexts(q)
    r, t1, kappa1, kappa = 3, 3, 6, 20
    B = ctx.vec(r * t1 * kappa1 * kappa).fill_synthetic(5)
    V = ctx.vec(r * t1 * kappa).fill_synthetic(6, bound=1625)
    full = lab.matvec_sum(ctx, B, V, r * t1, kappa1, kappa).store()
    lo, hi = parallel.ik_shard(r, t1, 0, 1)
    assert (lo, hi) == (0, r * t1)
    got = parallel.sharded_outer_commitment(ctx, B, V, hi - lo, kappa1, kappa, q)
    assert np.array_equal(got.cpu().numpy().view(np.uint64)[0], full)


@pytest.mark.parametrize("nb", [2, 4, 8])
@pytest.mark.parametrize("rows,cols,nmat", [(9, 40, 3), (20, 37, 2), (33, 1, 1), (64, 37, 3), (100, 8, 2), (130, 515, 1),
                                            (32, 2, 1), (64, 4, 2), (65, 6, 1), (200, 26, 2), (1024, 16, 1)])
@pytest.mark.parametrize("compact", [False, True])
def test_matvec_sum_batched_all_kernels(nb, rows, cols, nmat, compact):
    # nb vector sets sharing one matrix stream (the batched outer commitment, src/proofgen.rs:101-132 for nb proofs):
    # every dispatch target (chunk-split, LDS "sets", lane=row "bcast" kernel register-staged (odd cols) and with the
    # LDS-DMA ring (even cols: 1, 2, 3, 4, 8, 13 tiles); u32 and u16 matrix limbs) gives
    # the single-set result set by set, and the oracle's result on a sample of rows
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_REF33
    ctx = lab.Context(q, 0, **ctx_kw(q))
    bound = 1625
    ctx.enable_digit_basis(5000, bound)
    rng = np.random.default_rng(nb * 1000 + rows)
    M = rp(rng, q, nmat, rows, cols)
    V = rng.integers(0, bound, size=(nb, nmat, cols, D), dtype=np.uint64)
    hM = ctx.vec(nmat * rows * cols, digit=True, compact=compact).load(M)
    hV = ctx.vec(nb * nmat * cols, digit=True).load(V)
    got = lab.matvec_sum_batched(ctx, hM, hV, nmat, rows, cols, nb).store().reshape(nb, rows, D)
    for b in range(nb):
        one = lab.matvec_sum(ctx, hM, hV, nmat, rows, cols, v_off=b * nmat * cols).store()
        assert np.array_equal(got[b], one), (b,)
    for b, r_ in ((0, 0), (nb - 1, rows - 1), (nb // 2, rows // 2)):
        want = np.zeros(D, np.uint64)
        for m in range(nmat):
            want = co.poly_add(q, want, co.inner_product(q, M[m, r_], V[b, m]))
        assert np.array_equal(got[b, r_], want)
    ctx.close()


@pytest.mark.parametrize("q", [Q_REF33, Q_DIRECT])
def test_matvec_sum_batched_main_basis_full_residues(contexts, q):
    # same, operands in the context's main basis with full-range residues (worst case for the 64-bit accumulators:
    # the lazy fold keeps the high word below 2p across 8-term groups)
    import labrador_snark_amd as lab
    ctx = contexts(q)
    rng = np.random.default_rng(77)
    nb, nmat, rows, cols = 8, 2, 70, 300
    M = np.full((nmat, rows, cols, D), q - 1, np.uint64)
    M[:, ::3] = rp(rng, q, nmat, len(range(0, rows, 3)), cols)
    V = np.full((nb, nmat, cols, D), q - 1, np.uint64)
    V[1::2] = rp(rng, q, nb // 2, nmat, cols)
    hM, hV = ctx.vec_from(M), ctx.vec_from(V)
    got = lab.matvec_sum_batched(ctx, hM, hV, nmat, rows, cols, nb).store().reshape(nb, rows, D)
    for b in range(nb):
        assert np.array_equal(got[b], lab.matvec_sum(ctx, hM, hV, nmat, rows, cols, v_off=b * nmat * cols).store())


@pytest.mark.parametrize("nb", [1, 2, 4, 8])
@pytest.mark.parametrize("nterms,length", [(130, 21), (72, 40), (96, 64), (66, 100), (1056, 33), (17, 70), (2, 32)])
@pytest.mark.parametrize("compact", [False, True])
def test_axpy_commit_batched_all_kernels(nb, nterms, length, compact):
    # nb digit vectors sharing C (the C / D commitments of nb proofs, src/proofgen.rs:137-148, 364-378): the
    # register kernel (short commitments, odd term counts) and the lane=rho LDS-DMA kernel (length >= 32 and an even
    # chunk length: 9x8, 8x12, 11x6 (7 < 8 chunks: 6x11 odd -> 3x22), 8x132, 1x2) agree with the single-vector op
    # and with the oracle
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_REF33
    ctx = lab.Context(q, 0, **ctx_kw(q))
    bound = 1625
    ctx.enable_digit_basis(5000, bound)
    rng = np.random.default_rng(nb * 7 + nterms)
    G = rng.integers(0, bound, size=(nb, nterms, D), dtype=np.uint64)
    Cm = rp(rng, q, nterms, length)
    hC = ctx.vec(nterms * length, digit=True, compact=compact).load(Cm)
    hG = ctx.vec(nb * nterms, digit=True).load(G)
    got = lab.axpy_commit_batched(ctx, hG, hC, nterms, length, nb).store().reshape(nb, length, D)
    for b in range(nb):
        assert np.array_equal(got[b], lab.axpy_commit(ctx, hG, hC, nterms, length, g_off=b * nterms).store()), (b,)
    b = nb - 1
    for rho in (0, length // 2, length - 1):
        want = np.zeros(D, np.uint64)
        for m in range(nterms):
            want = co.poly_add(q, want, co.polymul(q, G[b, m], Cm[m, rho]))
        assert np.array_equal(got[b, rho], want)
    ctx.close()


@pytest.mark.parametrize("ns", [8, 16, 32, 64])
@pytest.mark.parametrize("nmat,rows,cols", [(3, 40, 37), (2, 64, 64), (1, 100, 200), (5, 31, 7)])
@pytest.mark.parametrize("compact", [False, True])
def test_matvec_sum_mfma_digit_basis(ns, nmat, rows, cols, compact):
    # the batched outer commitment as exact int8 GEMMs on the matrix cores (signed base-256 digit planes, int32
    # accumulation, recombination mod each limb prime): bit-identical to lab_matvec_sum set by set, and to the oracle
    # on sample rows; ragged rows / k (padding to 32-row and 128-k tiles), u32 and u16 source limbs (4 and 2 planes)
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_REF33
    ctx = lab.Context(q, 0, **ctx_kw(q))
    bound = 1625
    ctx.enable_digit_basis(5000, bound)
    rng = np.random.default_rng(ns + rows)
    M = rp(rng, q, nmat, rows, cols)
    V = rng.integers(0, bound, size=(ns, nmat, cols, D), dtype=np.uint64)
    hM = ctx.vec(nmat * rows * cols, digit=True, compact=compact).load(M)
    hV = ctx.vec(ns * nmat * cols, digit=True).load(V)
    M8 = lab.Mat8(ctx, hM, nmat, rows, cols)
    assert M8.nbytes > 0
    got = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
    for b in range(ns):
        one = lab.matvec_sum(ctx, hM, hV, nmat, rows, cols, v_off=b * nmat * cols).store()
        assert np.array_equal(got[b], one), (b,)
    for b, r_ in ((0, 0), (ns - 1, rows - 1)):
        want = np.zeros(D, np.uint64)
        for m in range(nmat):
            want = co.poly_add(q, want, co.inner_product(q, M[m, r_], V[b, m]))
        assert np.array_equal(got[b, r_], want)
    M8.close()
    ctx.close()


@pytest.mark.parametrize("ns", [8, 16, 32, 64])
@pytest.mark.parametrize("nmat,rows,cols", [(3, 40, 37), (2, 64, 64), (1, 100, 200), (5, 31, 7), (2, 130, 70), (1, 256, 33), (3, 97, 129)])
def test_matvec_sum_mfma_three_plane_basis(ns, nmat, rows, cols):
    # the digit basis made for the matrix cores: primes below 2^23, residues as three signed byte planes, GEMM blocks of
    # 3 (one row tile) or 6 waves, 1/2/3/6 column tiles, and from 4 row tiles on (32 / 64 sets) k_gemm_i8_w3 — one row tile
    # with its three planes per wave, assembly K loop; ragged last tiles, waves past the last tile (5 tiles = 2 blocks), 6,
    # 12 and 18 stages; bit-identical to lab_matvec_sum and the oracle
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_REF33
    ctx = lab.Context(q, 0, **ctx_kw(q))
    bound = 1625
    primes, small, nbytes = ctx.enable_digit_basis(5000, bound, mfma=True)
    assert not small and all(pr < (1 << 23) - 0x8080 and pr % 128 == 1 for pr in primes) and nbytes == 4 * len(primes)
    rng = np.random.default_rng(ns + rows)
    M = rp(rng, q, nmat, rows, cols)
    V = rng.integers(0, bound, size=(ns, nmat, cols, D), dtype=np.uint64)
    M[0, :2] = q - 1
    V[ns - 1] = bound - 1
    hM = ctx.vec(nmat * rows * cols, digit=True).load(M)
    hV = ctx.vec(ns * nmat * cols, digit=True).load(V)
    M8 = lab.Mat8(ctx, hM, nmat, rows, cols)
    got = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
    for b in range(ns):
        one = lab.matvec_sum(ctx, hM, hV, nmat, rows, cols, v_off=b * nmat * cols).store()
        assert np.array_equal(got[b], one), (b,)
    for b, r_ in ((0, 0), (ns - 1, rows - 1)):
        want = np.zeros(D, np.uint64)
        for m in range(nmat):
            want = co.poly_add(q, want, co.inner_product(q, M[m, r_], V[b, m]))
        assert np.array_equal(got[b, r_], want)
    M8.close()
    ctx.close()


@pytest.mark.parametrize("q", ALL_Q)
def test_matvec_sum_mfma_main_basis_full_residues(contexts, q):
    # operands in the context's main basis with full-range residues (every byte value, -128 included, occurs in the
    # digit planes of both operands)
    import labrador_snark_amd as lab
    ctx = contexts(q)
    rng = np.random.default_rng(5)
    ns, nmat, rows, cols = 8, 2, 70, 150
    M = rp(rng, q, nmat, rows, cols)
    V = rp(rng, q, ns, nmat, cols)
    M[0, :3] = q - 1
    V[1] = q - 1
    hM, hV = ctx.vec_from(M), ctx.vec_from(V)
    if ctx.wide:      # single-limb strict-32 context (q = 4294966657 does not even fit four signed byte planes; 1073741953 would): refused
        with pytest.raises(lab.LabradorError):
            lab.Mat8(ctx, hM, nmat, rows, cols)
        return
    M8 = lab.Mat8(ctx, hM, nmat, rows, cols)
    got = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
    for b in range(ns):
        assert np.array_equal(got[b], lab.matvec_sum(ctx, hM, hV, nmat, rows, cols, v_off=b * nmat * cols).store())
    M8.close()


def test_mat8_misuse_is_refused(contexts):
    # exactness bound of the int8 formulation (int32 accumulators: fewer than 2^17 - 1 products per sum), set counts
    # the GEMM is instantiated for, basis mismatches
    import labrador_snark_amd as lab
    ctx = contexts(Q_REF33)
    big = ctx.vec(131072)
    with pytest.raises(lab.LabradorError):
        lab.Mat8(ctx, big, 1, 1, 131072)
    M = ctx.vec(2 * 3 * 5).fill_synthetic(1)
    M8 = lab.Mat8(ctx, M, 2, 3, 5)
    V = ctx.vec(16 * 2 * 5).fill_synthetic(2)
    with pytest.raises(lab.LabradorError):
        lab.matvec_sum_mfma(ctx, M8, V, 12)
    with pytest.raises(lab.LabradorError):
        lab.matvec_sum_mfma(ctx, M8, ctx.vec(8), 8)          # V too short for 8 sets
    M8.close()
    big.close()


def _signed_digits(x, planes):
    """signed base-256 digits of x as the packing kernels produce them (mfma_kernels.hip: signed_digits)"""
    d = []
    for i in range(planes):
        if i < planes - 1:
            lo = ((x & 0xFF) ^ 0x80) - 0x80
            d.append(lo)
            x = (x - lo) >> 8
        else:
            d.append(x)
    return d


def test_matvec_sum_mfma_prime_between_8355711_and_2p23():
    # a directly NTT-friendly prime just below 2^23 (the Dilithium prime 8380417 = 2^23 - 2^13 + 1): residues above
    # 127 * 65793 = 8355711 do not fit three signed bytes, so the packed form must use four planes — full-range residues
    # (q - 1 in both operands) against the oracle
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = 8380417
    assert _signed_digits(q - 1, 3)[2] > 127          # the three-plane form would wrap its top digit
    ctx = lab.Context(q, 0, **ctx_kw(q))
    assert ctx.nlimbs == 1 and ctx.primes == [q]
    rng = np.random.default_rng(23)
    ns, nmat, rows, cols = 8, 2, 40, 150
    M = rp(rng, q, nmat, rows, cols)
    V = rp(rng, q, ns, nmat, cols)
    M[0, :3] = q - 1
    M[1, 5] = 8355712
    V[1] = q - 1
    V[2, :, :, ::2] = 8355712
    hM, hV = ctx.vec_from(M), ctx.vec_from(V)
    M8 = lab.Mat8(ctx, hM, nmat, rows, cols)
    assert M8.nbytes == 64 * 4 * 2 * 12 * 1024        # 4 planes x 2 row tiles x 12 k-tiles
    got = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
    for b in range(ns):
        assert np.array_equal(got[b], lab.matvec_sum(ctx, hM, hV, nmat, rows, cols, v_off=b * nmat * cols).store())
    for b, r_ in ((1, 0), (2, 5), (7, rows - 1)):
        want = np.zeros(D, np.uint64)
        for m in range(nmat):
            want = co.poly_add(q, want, co.inner_product(q, M[m, r_], V[b, m]))
        assert np.array_equal(got[b, r_], want)
    M8.close()
    ctx.close()


@pytest.mark.parametrize("K", [98304, 131070])
def test_matvec_sum_mfma_long_sums_three_plane_basis(K):
    # the reduction lengths the bench runs at (K = r*t1*kappa = 98304: 768 trips of the GEMM's K loop) and the longest the
    # int32 accumulators allow (2^17 - 2), 64 sets, basis of primes below 2^23.  (a) random data from the shared synthetic
    # streams: the matrix cores against the oracle on sample (set, row) pairs and against the multiply-add kernel on whole
    # sets; (b) extreme digit patterns: every matrix and vector residue of every NTT slot has the digits (-128, -128, 127)
    # resp. (127, 127, 126), so the plane products are K * 16384, K * -16256, K * 16129: the int32 bound |D| <= K * 2^14.
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_REF33
    ctx = lab.Context(q, 0, 1 << 17, **ctx_kw(q))
    bound = 1625
    primes, _, _ = ctx.enable_digit_basis(1 << 17, bound, mfma=True)
    assert all(pr < (1 << 23) - 0x8080 for pr in primes)
    ns, rows = 64, 128                   # 4 row tiles: k_gemm_i8_w3<6>, the kernel of the bench's step 3
    nmat, cols = (96, 1024) if K == 98304 else (2, 65535)
    assert nmat * cols == K
    hM = ctx.vec(nmat * rows * cols, digit=True).fill_synthetic(71)
    hV = ctx.vec(ns * K, digit=True).fill_synthetic(72, bound=bound)
    M8 = lab.Mat8(ctx, hM, nmat, rows, cols)
    got = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
    for b in (0, 37, 63):
        assert np.array_equal(got[b], lab.matvec_sum(ctx, hM, hV, nmat, rows, cols, v_off=b * K).store()), (b,)
    for b, r_ in ((0, 0), (63, 127), (21, 40)):
        v = co.synth_fill(72, b * K * D, K * D, bound).reshape(K, D)
        want = np.zeros(D, np.uint64)
        for m in range(nmat):
            row = co.synth_fill(71, ((m * rows + r_) * cols) * D, cols * D, q).reshape(cols, D)
            want = co.poly_add(q, want, co.inner_product(q, row, v[m * cols:(m + 1) * cols]))
        assert np.array_equal(got[b, r_], want), (b, r_)
    M8.close()
    hM.close()
    hV.close()
    # (b) a constant polynomial x has the residue x in every NTT slot of every limb (x below all three primes)
    lo, hi = 127 * 65536 - 128 * 256 - 128, 126 * 65536 + 127 * 256 + 127
    assert _signed_digits(lo, 3) == [-128, -128, 127] and _signed_digits(hi, 3) == [127, 127, 126] and hi < min(primes)
    for rows in (128, 32):               # k_gemm_i8_w3<6> and k_gemm_i8<6, 3>
      for xm, xv in ((lo, lo), (hi, hi), (lo, hi)):
        Mc = np.zeros((K, D), np.uint64)
        Mc[:, 0] = xm
        Vc = np.zeros((K, D), np.uint64)
        Vc[:, 0] = xv
        hM = ctx.vec(rows * K, digit=True)
        for r_ in range(rows):
            hM.load(Mc, r_ * K)
        hV = ctx.vec(ns * K, digit=True)
        for s in range(ns):
            hV.load(Vc, s * K)
        M8 = lab.Mat8(ctx, hM, 1, rows, K)
        got = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
        want = np.zeros(D, np.uint64)
        want[0] = (K * xm * xv) % q
        assert np.array_equal(got, np.broadcast_to(want, got.shape)), (rows, xm, xv)
        assert np.array_equal(got[5, 7], co.inner_product(q, Mc, Vc))
        M8.close()
        hM.close()
        hV.close()
    ctx.close()


@pytest.mark.parametrize("K", [98304, 131070])
def test_matvec_sum_mfma_long_sums_main_basis(K):
    # the same for the 30-bit limb primes (four planes, k_gemm_i8<NT, 8>): extreme patterns (-128, -128, -128, 63) and
    # (127, 127, 127, 62) in both operands, full-range residues, 64 sets
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_REF33
    ctx = lab.Context(q, 0, 1 << 17, **ctx_kw(q))
    ns, rows = 64, 32
    lo = 63 * (1 << 24) - 128 * 65536 - 128 * 256 - 128
    hi = 62 * (1 << 24) + 127 * 65536 + 127 * 256 + 127
    assert _signed_digits(lo, 4) == [-128, -128, -128, 63] and _signed_digits(hi, 4) == [127, 127, 127, 62] and hi < min(ctx.primes)
    for xm, xv in ((lo, lo), (hi, hi), (lo, hi)):
        Mc = np.zeros((rows * K, D), np.uint64)
        Mc[:, 0] = xm
        Vc = np.zeros((K, D), np.uint64)
        Vc[:, 0] = xv
        hM = ctx.vec_from(Mc)
        hV = ctx.vec(ns * K)
        for s in range(ns):
            hV.load(Vc, s * K)
        M8 = lab.Mat8(ctx, hM, 1, rows, K)
        got = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
        want = np.zeros(D, np.uint64)
        want[0] = (K * xm * xv) % q
        assert np.array_equal(got, np.broadcast_to(want, got.shape)), (xm, xv)
        M8.close()
        hM.close()
        hV.close()
    # random full-range residues at the same length: 8 sets against the multiply-add kernel and the oracle
    nmat, cols = (96, 1024) if K == 98304 else (2, 65535)
    hM = ctx.vec(nmat * rows * cols).fill_synthetic(81)
    hV = ctx.vec(8 * K).fill_synthetic(82)
    M8 = lab.Mat8(ctx, hM, nmat, rows, cols)
    got = lab.matvec_sum_mfma(ctx, M8, hV, 8).store().reshape(8, rows, D)
    for b in (0, 7):
        assert np.array_equal(got[b], lab.matvec_sum(ctx, hM, hV, nmat, rows, cols, v_off=b * K).store()), (b,)
    b, r_ = 3, 17
    v = co.synth_fill(82, b * K * D, K * D, q).reshape(K, D)
    want = np.zeros(D, np.uint64)
    for m in range(nmat):
        row = co.synth_fill(81, ((m * rows + r_) * cols) * D, cols * D, q).reshape(cols, D)
        want = co.poly_add(q, want, co.inner_product(q, row, v[m * cols:(m + 1) * cols]))
    assert np.array_equal(got[b, r_], want)
    M8.close()
    ctx.close()


def test_matvec_sum_mfma_full_range_operands_in_the_23_bit_basis():
    # lab_ctx_enable_digit_basis_mfma with digit_bound = q: the "digit" operand may be any residue — FOUR primes below 2^23 then hold
    # 1024-term sums of products of two full-range residues.  bench.py's config-4 leg with 32 right-hand sides runs there
    # (k_gemm_i8_w3<3>: 32 sets x 3 planes); results equal the multiply-add kernel in the main basis and the oracle.
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_REF33
    ctx = lab.Context(q, 0)
    ns, nmat, rows, cols = 32, 1, 260, 1024
    primes, small, _ = ctx.enable_digit_basis(cols, q, mfma=True)
    assert len(primes) == 4 and not small and all(pr < 8355712 for pr in primes)
    rng = np.random.default_rng(77)
    M = rp(rng, q, nmat, rows, cols)
    V = rp(rng, q, ns, nmat, cols)
    M[0, :2] = q - 1
    V[ns - 1] = q - 1
    hM = ctx.vec(nmat * rows * cols, digit=True).load(M)
    hV = ctx.vec(ns * nmat * cols, digit=True).load(V)
    M8 = lab.Mat8(ctx, hM, nmat, rows, cols)
    got = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
    mM, mV = ctx.vec_from(M), ctx.vec_from(V)               # main basis, multiply-add kernel
    for b in (0, 5, ns - 1):
        assert np.array_equal(got[b], lab.matvec_sum(ctx, mM, mV, nmat, rows, cols, v_off=b * nmat * cols).store()), (b,)
    for b, r_ in ((ns - 1, 0), (ns - 1, 1), (3, rows - 1)):
        assert np.array_equal(got[b, r_], co.inner_product(q, M[0, r_], V[b, 0])), (b, r_)
    M8.close()
    ctx.close()
