"""CPU tests of the oracle itself (no GPU): the reference's own property tests re-asserted on the C
restatement, agreement of the two independent restatements, and the reference's CI invariant
verify(proof_gen(..)) == true.  This is what pins the oracle — the reference holds no golden vectors."""
import random
import re

import numpy as np
import pytest

from conftest import ALL_Q, Q_DEFAULT, Q_REF33

from oracle import coracle as co
from oracle import pyref
import cases

D = 64


def rnd_poly(rng, q):
    return rng.integers(0, q, size=D, dtype=np.uint64)


@pytest.mark.parametrize("q", ALL_Q)
def test_c_and_python_twins_agree_on_ring_ops(q):
    R = pyref.Ring(q)
    rnd = random.Random(q)
    for it in range(8):
        a = [rnd.randrange(q) for _ in range(D)]
        b = [rnd.randrange(q) for _ in range(D)]
        if it == 0:
            a[30:] = [0] * 34  # short polynomial: the reference strips trailing zeros
        if it == 1:
            a = [0] * D
        assert list(map(int, co.polymul(q, a, b))) == R.pad(R.mul(R.trim(a), R.trim(b)))
        assert list(map(int, co.polymul(q, a, b, "faithful"))) == R.pad(R.mul(R.trim(a), R.trim(b)))
        assert list(map(int, co.polymul(q, a, b, "ntt_branch"))) == R.pad(R.mul_ntt_branch(R.trim(a), R.trim(b)))
        assert list(map(int, co.poly_add(q, a, b))) == R.pad(R.add(R.trim(a), R.trim(b)))
        assert list(map(int, co.poly_sub(q, a, b))) == R.pad(R.sub(R.trim(a), R.trim(b)))
        assert list(map(int, co.sigma_inv(q, a))) == R.pad(R.sigma_inv(R.trim(a)))
        s = rnd.randrange(q)
        assert list(map(int, co.poly_scale(q, a, s))) == R.pad(R.scale(R.trim(a), s))
        for base, exp in ((9, 4), (1625, 3), (14, 2), (638, 4), (2, 13), (3, 1)):
            want = [R.pad(x) for x in R.decompose(R.trim(a), base, exp)]
            assert [list(map(int, row)) for row in co.decompose(q, a, base, exp)] == want
        f, exact = co.poly_norm(a)
        assert exact == R.poly_norm_int(a) and f == R.poly_norm(a)


def test_ntt_preserves_result_default_q():
    # tests/proptest.rs:13-24 at the default Q = 8191: the reference's NTT branch equals its classic branch
    rng = np.random.default_rng(1)
    for _ in range(50):
        a, b = rnd_poly(rng, Q_DEFAULT), rnd_poly(rng, Q_DEFAULT)
        assert np.array_equal(co.polymul(Q_DEFAULT, a, b, "ntt_branch"), co.polymul(Q_DEFAULT, a, b))


def test_ntt_branch_wraps_at_33_bit_q():
    # at Q = 2^32+15 the reference's own NTT branch wraps mod 2^64 and disagrees with classic;
    # the classic branch is the semantic definition the GPU path must match.
    rng = np.random.default_rng(2)
    a, b = rnd_poly(rng, Q_REF33), rnd_poly(rng, Q_REF33)
    assert not np.array_equal(co.polymul(Q_REF33, a, b, "ntt_branch"), co.polymul(Q_REF33, a, b))


@pytest.mark.parametrize("q", ALL_Q)
def test_linearity_of_poly_vec_inner_product(q):
    # tests/proptest.rs:36-48 / 51-64, N = 16, 50 cases
    rng = np.random.default_rng(3)
    for _ in range(50):
        a = rng.integers(0, q, size=(16, D), dtype=np.uint64)
        b = rng.integers(0, q, size=(16, D), dtype=np.uint64)
        c = int(rng.integers(0, q))
        cb = np.stack([co.poly_scale(q, row, c) for row in b])
        assert np.array_equal(co.inner_product(q, a, cb), co.poly_scale(q, co.inner_product(q, a, b), c))


@pytest.mark.parametrize("q", ALL_Q)
def test_sigma_inv_invariant(q):
    # tests/proptest.rs:67-81: <a,b> over Z_q equals the constant term of <sigma_inv(a), b> over R_q
    rng = np.random.default_rng(4)
    for _ in range(50):
        a = rng.integers(0, q, size=(16, D), dtype=np.uint64)
        b = rng.integers(0, q, size=(16, D), dtype=np.uint64)
        inner = sum(int(x) * int(y) for x, y in zip(a.ravel(), b.ravel())) % q
        inv_a = np.stack([co.sigma_inv(q, row) for row in a])
        assert int(co.inner_product(q, inv_a, b)[0]) == inner


def _parse_regressions():
    with open(__file__.replace("test_oracle.py", "golden/proptest_regressions_9_11.txt")) as f:
        lines = [l for l in f.read().split("\n") if l.startswith("a=")]
    out = []
    for l in lines:
        m = re.match(r"a=(.*);b=(.*);c=(\d+)$", l)
        polys = lambda s: [[int(x) for x in p.split(",") if x] for p in s.split("|")]
        out.append((polys(m.group(1)), polys(m.group(2)), int(m.group(3))))
    return out


@pytest.mark.parametrize("q", [Q_DEFAULT, Q_REF33])
def test_reference_regression_inputs_linearity(q):
    # tests/proptest.proptest-regressions:9-11 (saved failing inputs of the linearity tests, as data)
    for a, b, c in _parse_regressions():
        pad = lambda v: np.array([p + [0] * (D - len(p)) for p in v], dtype=np.uint64) % np.uint64(q)
        A, B = pad(a), pad(b)
        cB = np.stack([co.poly_scale(q, row, c) for row in B])
        assert np.array_equal(co.inner_product(q, A, cB), co.poly_scale(q, co.inner_product(q, A, B), c))


def test_inner_product_length_mismatch_oracle():
    with pytest.raises(ValueError):
        co.inner_product(Q_DEFAULT, np.zeros((3, D), np.uint64), np.zeros((4, D), np.uint64))


def test_runtime_constants_match_survey_table():
    # SURVEY.md Appendix B (restatement of src/constants.rs:234-247)
    rows = {(8191, 2, 2): (31, 9, 4, 9, 2, 14), (8191, 16, 16): (31, 5, 6, 4, 1, 9),
            (Q_REF33, 2, 2): (16634336, 6551, 3, 1625, 4, 2782), (Q_REF33, 16, 16): (16634336, 3895, 3, 1625, 3, 13834),
            (Q_REF33, 1024, 32): (16634336, 1377, 3, 1625, 4, 638)}
    for (q, n, r), (beta, b, t1, b1, t2, b2) in rows.items():
        c = co.runtime_constants(q, n, r)
        assert (c.beta_bound, c.b, c.t1, c.b1, c.t2, c.b2) == (beta, b, t1, b1, t2, b2)
        p = pyref.runtime_constants(q, n, r)
        assert (p["beta_bound"], p["b"], p["t1"], p["b1"], p["t2"], p["b2"]) == (beta, b, t1, b1, t2, b2)
    # the default Q cannot run n=1024, r=32: B_1 = 1, T_2 = -1
    c = co.runtime_constants(8191, 1024, 32)
    assert c.b1 == 1 and c.t2 == -1


def test_host_runtime_constants_match_oracle():
    import labrador_snark_amd.prover as pr  # host logic only; does not load the HIP library
    for q, n, r in ((8191, 2, 2), (Q_REF33, 16, 16), (Q_REF33, 1024, 32), (1073738753, 64, 8)):
        k, c = pr.RuntimeConstants(q, n, r), co.runtime_constants(q, n, r)
        assert (k.BETA_BOUND, k.B, k.T_1, k.B_1, k.T_2, k.B_2) == (c.beta_bound, c.b, c.t1, c.b1, c.t2, c.b2)
        assert k.STD == c.std and k.GAMMA == c.gamma and k.GAMMA_1 == c.gamma1 and k.GAMMA_2 == c.gamma2
        assert k.BETA_PRIME == c.beta_prime


@pytest.mark.parametrize("q,n,r,kappa,kappa1", [(Q_DEFAULT, 2, 2, 4, 3), (Q_REF33, 2, 2, 5, 4), (Q_REF33, 4, 3, 6, 5)])
def test_prove_then_verify_passes_and_twins_agree(q, n, r, kappa, kappa1):
    # the reference's CI invariant (src/main.rs:106-107): verify(proof_gen(..)) is true — on the C oracle,
    # and the pure-Python twin produces the identical transcript.
    p = cases.Params(q, n, r, kappa, kappa1)
    case, (S, phi, a, tape) = cases.oracle_case(p, seed=11)
    proof = case.prove()
    assert proof["bpp_ok"]
    # Line 14 (the f64 norm sum against BETA_PRIME) is evaluated with the real bound at the reference's CI size
    # n = r = 2; it is not a robust check in the reference itself (its own tests/res log shows it failing), so for
    # other shapes only checks 8-13 and 15-20 are asserted.
    rc, s14 = case.verify(proof, p.b, p.beta_prime if (n, r) == (2, 2) else float("inf"))
    assert rc == 0, "verify failed at protocol line %d (norm sum %g vs %g)" % (rc, s14, p.beta_prime)
    # tamper: any changed field must be caught
    bad = dict(proof); bad["z"] = proof["z"].copy(); bad["z"][0, 0] ^= np.uint64(1)
    assert case.verify(bad, p.b, float("inf"))[0] in (15, 16, 17)
    bad = dict(proof); bad["u1"] = proof["u1"].copy(); bad["u1"][0, 0] ^= np.uint64(1)
    assert case.verify(bad, p.b, float("inf"))[0] == 19
    # python twin
    R = pyref.Ring(q)
    A, B, C, Dm = cases.make_crs_arrays(p, 1000)
    tl = lambda arr: [R.trim([int(x) for x in row]) for row in arr.reshape(-1, D)]
    d = dict(n=n, r=r, kappa=kappa, kappa1=kappa1, kappa2=p.kappa2, b1=p.b1, t1=p.t1, b2=p.b2, t2=p.t2,
             beta_bound=p.beta_bound, b_prime=case.b_prime, psi=tape["psi"], omega=[int(x) for x in tape["omega"]],
             alpha=R.trim([int(x) for x in tape["alpha"]]), beta=R.trim([int(x) for x in tape["beta"]]),
             c=tl(tape["c"]), Pi=[[[int(v) for v in row] for row in m] for m in tape["Pi"]])
    d["S"] = [[R.trim([int(x) for x in S[x_, i]]) for i in range(r)] for x_ in range(n)]
    d["phi"] = [[R.trim([int(x) for x in phi[x_, i]]) for i in range(r)] for x_ in range(n)]
    d["a"] = [[R.trim([int(x) for x in a[i, j]]) for j in range(r)] for i in range(r)]
    d["A"] = [tl(A[rho]) for rho in range(kappa)]
    d["B"] = [[tl(B[m, rho]) for rho in range(kappa1)] for m in range(r * p.t1)]
    d["C"] = [tl(C[m]) for m in range(p.pairs * p.t2)]
    d["Dm"] = [tl(Dm[m]) for m in range(p.pairs * p.t1)]
    py = pyref.prove_core(R, d)
    flat = lambda polys: np.array([R.pad(x) for x in polys], dtype=np.uint64)
    assert np.array_equal(flat([x for row in py["t"] for x in row]).reshape(r, kappa, D), proof["t"])
    assert np.array_equal(flat([x for row in py["G"] for x in row]).reshape(r, r, D), proof["G"])
    assert np.array_equal(flat([x for row in py["H"] for x in row]).reshape(r, r, D), proof["H"])
    assert np.array_equal(flat(py["u1"]), proof["u1"]) and np.array_equal(flat(py["u2"]), proof["u2"])
    assert np.array_equal(flat(py["z"]), proof["z"]) and np.array_equal(flat([py["bpp"]])[0], proof["bpp"])
    assert py["proj_int"] == [int(x) for x in proof["proj_int"]]
    assert py["jl_valid"] == proof["jl_valid"] and py["bpp_ok"] == proof["bpp_ok"]


def test_synth_stream_is_stable():
    # the shared synthetic stream (oracle side); the GPU fill kernel is compared against it in the gpu tests
    v = co.synth_fill(42, 7 * D, 4, 8191)
    assert [int(x) for x in v] == [int(co.lib().orc_synth_residue(42, 7 * D + i, 8191)) for i in range(4)]
    assert all(int(x) < 8191 for x in co.synth_fill(1, 0, 1000, 8191))
