"""The CPU baselines bench.py reports are baselines for the SAME job: B1 (oracle/fastcpu.c, the honest-fast OpenMP
prover: RNS/NTT residues, transform-domain accumulation, phi never formed) produces the transcript orc_prove_core —
the restatement of Prover::proof_gen (src/proofgen.rs:30-427) — produces, bit for bit; its single ring product equals
the classic product (the reference's test_NTT_preserves_result, tests/proptest.rs:13-24), also at the NTT-friendly
32-bit modulus 4294966657 (path (b) of SURVEY §7) where the reference's own NTT branch would wrap."""
import numpy as np
import pytest

from conftest import ALL_Q, Q_NTT32

import cases

D = 64


@pytest.mark.parametrize("q", ALL_Q)
def test_fast_product_equals_classic(q):
    from oracle import coracle as co
    from oracle import fastcpu as fc
    f = fc.FastCpu(q, 1, 1)
    assert f.L == (1 if q in (1073738753, 1073741953, Q_NTT32) else 2 if q == 8191 else 3)
    rng = np.random.default_rng(q % 1000)
    for _ in range(50):
        a = rng.integers(0, q, size=D, dtype=np.uint64)
        b = rng.integers(0, q, size=D, dtype=np.uint64)
        assert np.array_equal(f.polymul(a, b), co.polymul(q, a, b))
    a = np.full(D, q - 1, np.uint64)
    assert np.array_equal(f.polymul(a, a), co.polymul(q, a, a))
    x = rng.integers(0, q, size=(7, D), dtype=np.uint64)
    assert np.array_equal(f.from_hat(f.to_hat(x)), x)
    f.close()


@pytest.mark.parametrize("q", ALL_Q)
@pytest.mark.parametrize("n,r,kappa,kappa1", [(2, 2, 8, 6), (5, 3, 7, 4)])
def test_fast_prover_equals_the_oracle_prover(q, n, r, kappa, kappa1):
    from oracle import fastcpu as fc
    p = cases.Params(q, n, r, kappa, kappa1)
    case, _ = cases.oracle_case(p, seed=77 + n)
    want = case.prove()
    f = fc.FastCpu(q, 65536, 2)
    crs = [f.to_hat(M) for M in (case.A, case.B, case.Cm, case.Dm)]
    got = f.prove(case, *crs)
    for k in ("t", "G", "u1", "proj_int", "proj", "bpp", "H", "u2", "z"):
        assert np.array_equal(got[k], want[k]), k
    assert got["jl_valid"] == want["jl_valid"] and got["bpp_ok"] == want["bpp_ok"] and got["bpp_ok"]
    # the synthetic CRS stream straight into the transform domain equals transforming the oracle's stream
    A2 = f.synthetic_hat(1000, kappa * n)
    assert np.array_equal(A2, crs[0])
    f.close()


def test_config1_microbench_runs():
    # benches/ntt_speed.rs:51-81 restated: both arms return a positive median; the transform arm is not slower than the
    # reference-faithful classic arm by construction of either (no assertion on the ratio: host dependent)
    from oracle import fastcpu as fc
    for q in (8191, (1 << 32) + 15, Q_NTT32):
        assert fc.bench_polymul(q, "classic", samples=50) > 0
        assert fc.bench_polymul(q, "ntt", samples=50) > 0
