"""Seeded synthetic prover instances shared by the oracle tests, the GPU parity tests and bench.py.

Shapes follow Prover::proof_gen (src/proofgen.rs:30-427); the random choices restate
generate_witness (src/proofgen.rs:460-518, as small uniform coefficients so the norm bound holds without the
halving loop), State::gen_f (src/structs.rs:289-350), the verifier samplers (src/verification.rs:441-513,
553-566) and CRS rows as plain uniform matrices (src/structs.rs:55-171).
"""
import math

import numpy as np

D = 64


def pair_index(i, j, r):  # src/structs.rs:101-106
    return i * r - (i * (i - 1)) // 2 + (j - i)


class Params:
    """RuntimeConstants (src/constants.rs:232-264) with explicit commitment ranks."""

    def __init__(self, q, n, r, kappa, kappa1, kappa2=None):
        from oracle import coracle as co
        c = co.runtime_constants(q, n, r)
        self.q, self.n, self.r = q, n, r
        self.kappa, self.kappa1, self.kappa2 = kappa, kappa1, kappa1 if kappa2 is None else kappa2
        self.beta_bound, self.std, self.b = c.beta_bound, c.std, c.b
        self.b1, self.t1, self.b2, self.t2 = c.b1, c.t1, c.b2, c.t2
        b1, b2, t1, t2, rr = float(c.b1), float(c.b2), float(c.t1), float(c.t2), float(r)
        # src/constants.rs:250-254 with the explicit kappa
        self.gamma = float(c.beta_bound ** 2) * 71.0
        self.gamma1 = ((b1 * b1 * t1) / 12.0) * rr * float(kappa) * float(D) + ((b2 * b2 * t2) / 12.0) * ((rr * rr + rr) / 2.0) * float(D)
        self.gamma2 = ((b1 * b1 * t1) / 12.0) * ((rr * rr + rr) / 2.0) * float(D)
        self.beta_prime = (2.0 / float(c.b) ** 2) * self.gamma + self.gamma1 + self.gamma2
        self.pairs = r * (r + 1) // 2


def challenge_poly(rng, q):
    """generate_polynomial_picky with fetch_challenge's multiset (src/verification.rs:460-479, src/util.rs:83-104):
    23 zeros, 31 ones, 10 twos, random order, random sign (the operator-norm rejection loop is skipped)."""
    vals = np.array([0] * 23 + [1] * 31 + [2] * 10, dtype=np.int64)
    rng.shuffle(vals)
    signs = rng.integers(0, 2, size=D)
    out = np.where((vals > 0) & (signs == 1), q - vals, vals)
    return out.astype(np.uint64)


def make_inputs(p, seed, witness_bound=None):
    """Witness, state and challenge tape as numpy arrays (no CRS)."""
    rng = np.random.default_rng(seed)
    q, n, r = p.q, p.n, p.r
    if witness_bound is None:
        witness_bound = int(p.beta_bound / math.sqrt(n * r * D))
    if witness_bound >= 1:
        S = rng.integers(0, witness_bound + 1, size=(n, r, D), dtype=np.uint64)
    else:
        # default-Q regime (beta = 31): generate_witness's halving loop leaves an almost-all-zero witness
        # (SURVEY Appendix B); restate that as beta^2/4 unit coefficients at random positions.
        S = np.zeros(n * r * D, dtype=np.uint64)
        S[rng.choice(n * r * D, size=max(1, p.beta_bound ** 2 // 4), replace=False)] = 1
        S = S.reshape(n, r, D)
    phi = rng.integers(0, q, size=(n, r, D), dtype=np.uint64)
    a = rng.integers(0, q, size=(r, r, D), dtype=np.uint64)
    for i in range(r):
        for j in range(i):
            a[i, j] = a[j, i]  # src/structs.rs:297-300 symmetric
    Pi = rng.choice(np.array([-1, 0, 1], dtype=np.int8), size=(r, 256, n * D), p=[0.25, 0.5, 0.25]).astype(np.int8)
    tape = dict(Pi=Pi, psi=int(rng.integers(0, q)), omega=rng.integers(0, q, size=256, dtype=np.uint64),
                alpha=rng.integers(0, q, size=D, dtype=np.uint64), beta=rng.integers(0, q, size=D, dtype=np.uint64),
                c=np.stack([challenge_poly(rng, q) for _ in range(r)]))
    return S, phi, a, tape


def make_crs_arrays(p, seed):
    """Explicit uniform CRS matrices from the shared splitmix64 streams seed..seed+3 (same data as
    ResidentCRS.fill_synthetic / lab_vec_fill_synthetic)."""
    from oracle import coracle as co
    q = p.q
    A = co.synth_fill(seed, 0, p.kappa * p.n * D, q).reshape(p.kappa, p.n, D)
    B = co.synth_fill(seed + 1, 0, p.r * p.t1 * p.kappa1 * p.kappa * D, q).reshape(p.r * p.t1, p.kappa1, p.kappa, D)
    C = co.synth_fill(seed + 2, 0, p.pairs * p.t2 * p.kappa2 * D, q).reshape(p.pairs * p.t2, p.kappa2, D)
    Dm = co.synth_fill(seed + 3, 0, p.pairs * p.t1 * p.kappa2 * D, q).reshape(p.pairs * p.t1, p.kappa2, D)
    return A, B, C, Dm


def oracle_case(p, seed, crs_seed=1000):
    from oracle import coracle as co
    S, phi, a, tape = make_inputs(p, seed)
    A, B, C, Dm = make_crs_arrays(p, crs_seed)
    case = co.ProverCase(p.q, p.n, p.r, p.kappa, p.kappa1, p.kappa2, p.b1, p.t1, p.b2, p.t2, p.beta_bound, S, phi, a,
                         A, B, C, Dm, tape["Pi"], tape["psi"], tape["omega"], tape["alpha"], tape["beta"], tape["c"])
    return case, (S, phi, a, tape)
