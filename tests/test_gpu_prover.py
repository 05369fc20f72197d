"""GPU parity of the fused prover core (lab_prove_core) vs the oracle's restatement of
Prover::proof_gen (src/proofgen.rs:30-427): every transcript field bit-exact, then the oracle's restated
Verifier::verify accepts the GPU transcript (the reference's CI invariant, src/main.rs:106-107)."""
import copy

import numpy as np
import pytest

from conftest import ALL_Q, Q_DEFAULT, Q_REF33, Q_DIRECT, Q_NTT32, ctx_kw

import cases

pytestmark = pytest.mark.gpu
D = 64


def run_gpu(ctx, p, S, phi, a, tape, b_prime, crs_seed=1000):
    import labrador_snark_amd as lab
    k = lab.RuntimeConstants(p.q, p.n, p.r, p.kappa, p.kappa1, p.kappa2)
    assert (k.B_1, k.T_1, k.B_2, k.T_2, k.BETA_BOUND) == (p.b1, p.t1, p.b2, p.t2, p.beta_bound)
    crs = lab.ResidentCRS(ctx, k).fill_synthetic(crs_seed)
    st = lab.State(phi, a, b_prime)
    try:
        return lab.prove_core(ctx, k, S, st, crs, tape)
    finally:
        crs.close()


@pytest.mark.parametrize("q,n,r,kappa,kappa1", [
    (Q_DEFAULT, 2, 2, 4, 3),       # the reference's CI shape (src/main.rs:62-63) with small ranks
    (Q_REF33, 2, 2, 5, 4),
    (Q_REF33, 4, 3, 6, 5),
    (Q_DIRECT, 3, 2, 7, 4),
    (Q_REF33, 16, 8, 24, 16),
    (Q_DEFAULT, 16, 16, 16, 8),
    (Q_NTT32, 2, 2, 5, 4),         # the NTT-friendly 32-bit modulus (SURVEY §8(d) "q: report both")
    (Q_NTT32, 4, 3, 6, 5),
    (Q_NTT32, 16, 8, 24, 16),
])
def test_prove_core_matches_oracle(contexts, q, n, r, kappa, kappa1):
    ctx = contexts(q)
    p = cases.Params(q, n, r, kappa, kappa1)
    case, (S, phi, a, tape) = cases.oracle_case(p, seed=21)
    want = case.prove()
    assert want["bpp_ok"]
    if not want["jl_valid"]:
        pytest.skip("synthetic tape rejected by valid_projection")
    T = run_gpu(ctx, p, S, phi, a, tape, case.b_prime)
    assert np.array_equal(T.t_i_all, want["t"])
    assert np.array_equal(T.g_mat, want["G"])
    assert np.array_equal(T.u_1, want["u1"])
    assert np.array_equal(T.projection_int, want["proj_int"])
    assert np.array_equal(T.projection, want["proj"])
    assert np.array_equal(T.b_prime_prime, want["bpp"])
    assert np.array_equal(T.h_mat, want["H"])
    assert np.array_equal(T.u_2, want["u2"])
    assert np.array_equal(T.z, want["z"])
    proof = dict(t=T.t_i_all, G=T.g_mat, H=T.h_mat, u1=T.u_1, u2=T.u_2, proj_int=T.projection_int, proj=T.projection,
                 bpp=T.b_prime_prime, z=T.z)
    rc, _ = case.verify(proof, p.b, float("inf"))
    assert rc == 0


def test_prove_core_rejections(contexts):
    import labrador_snark_amd as lab
    q, n, r = Q_REF33, 2, 2
    ctx = contexts(q)
    p = cases.Params(q, n, r, 4, 3)
    case, (S, phi, a, tape) = cases.oracle_case(p, seed=5)
    # a witness far above the norm bound makes Verifier::valid_projection reject (src/verification.rs:568-579)
    S_big = np.full_like(S, q - 1)
    with pytest.raises(lab.JLRejected):
        run_gpu(ctx, p, S_big, phi, a, tape, case.b_prime)
    # a wrong b' trips verify_b_prime_prime's assertion (src/verification.rs:550)
    with pytest.raises(AssertionError):
        run_gpu(ctx, p, S, phi, a, tape, (case.b_prime + 1) % q)
    # kappa1 != kappa2: u_1 = lhs + rhs is undefined (add_poly_vec asserts, src/util.rs:299-302)
    k = lab.RuntimeConstants(q, n, r, 4, 3, 2)
    crs = lab.ResidentCRS(ctx, k).fill_synthetic(1)
    with pytest.raises(AssertionError):
        lab.prove_core(ctx, k, S, lab.State(phi, a, case.b_prime), crs, tape)
    crs.close()


def test_prover_class_mirrors_reference_call_shape(contexts):
    # Prover::new(&witness, &verifier, &constants) / proof_gen(&st, &mut crs) with a scripted verifier
    import labrador_snark_amd as lab
    q, n, r = Q_REF33, 2, 2
    ctx = contexts(q)
    p = cases.Params(q, n, r, 4, 3)
    case, (S, phi, a, tape) = cases.oracle_case(p, seed=8)

    class ScriptedVerifier:
        def sample_jl_projection_all(self): return tape["Pi"]
        def generate_psi(self): return tape["psi"]
        def generate_omega(self): return tape["omega"]
        def fetch_alpha(self): return tape["alpha"]
        def fetch_beta(self): return tape["beta"]
        def fetch_challenges(self): return tape["c"]

    k = lab.RuntimeConstants(q, n, r, 4, 3, 3)
    crs = lab.ResidentCRS(ctx, k).fill_synthetic(1000)
    T = lab.Prover(S, ScriptedVerifier(), k, ctx).proof_gen(lab.State(phi, a, case.b_prime), crs)
    want = case.prove()
    assert np.array_equal(T.u_1, want["u1"]) and np.array_equal(T.u_2, want["u2"]) and np.array_equal(T.z, want["z"])
    crs.close()


@pytest.mark.parametrize("q,n,r,kappa,kappa1", [(Q_DEFAULT, 2, 2, 4, 3), (Q_REF33, 2, 2, 5, 4), (Q_REF33, 4, 3, 6, 5),
                                               (Q_DIRECT, 3, 2, 7, 4), (Q_REF33, 16, 8, 24, 16)])
def test_verify_core_matches_oracle_verifier(contexts, q, n, r, kappa, kappa1):
    # Verifier::verify (src/verification.rs:25-438) on the GPU: same verdict, same failing line and the same
    # f64 line-14 norm sum (bit-for-bit: the exact integer norms are folded in the reference's order) as the oracle
    import copy
    import labrador_snark_amd as lab
    ctx = contexts(q)
    p = cases.Params(q, n, r, kappa, kappa1)
    case, (S, phi, a, tape) = cases.oracle_case(p, seed=21)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k).fill_synthetic(1000)
    st = lab.State(phi, a, case.b_prime)
    T = lab.prove_core(ctx, k, S, st, crs, tape)
    proof = dict(t=T.t_i_all, G=T.g_mat, H=T.h_mat, u1=T.u_1, u2=T.u_2, proj_int=T.projection_int, proj=T.projection,
                 bpp=T.b_prime_prime, z=T.z)
    kinf = copy.copy(k); kinf.BETA_PRIME = float("inf")
    line, s14 = lab.verify_core(ctx, kinf, st, case.b_poly, crs, T)
    orc_line, orc_s14 = case.verify(proof, k.B, float("inf"))
    assert (line, s14) == (orc_line, orc_s14) == (0, orc_s14)
    # line 14 with a bound just below / at the sum
    ktight = copy.copy(k); ktight.BETA_PRIME = s14
    assert lab.verify_core(ctx, ktight, st, case.b_poly, crs, T)[0] == 0
    ktight.BETA_PRIME = float(np.nextafter(s14, 0.0))
    assert lab.verify_core(ctx, ktight, st, case.b_poly, crs, T)[0] == 14
    # tampering: the GPU verifier reports the same first failing line as the oracle
    for field, okey, idx in (("z", "z", (0, 0)), ("u_1", "u1", (0, 3)), ("u_2", "u2", (1, 0)), ("t_i_all", "t", (0, 0, 0)),
                             ("g_mat", "G", (0, 1, 2)), ("h_mat", "H", (1, 0, 5)), ("b_prime_prime", "bpp", (7,))):
        T2 = copy.copy(T)
        arr = getattr(T, field).copy()
        arr[idx] = (int(arr[idx]) + 1) % q
        setattr(T2, field, arr)
        bad = dict(proof); bad[okey] = arr
        got = lab.verify_core(ctx, kinf, st, case.b_poly, crs, T2)[0]
        want = case.verify(bad, k.B, float("inf"))[0]
        assert got == want != 0, (field, got, want)
    assert lab.Verifier([case.b_prime], k, ctx).verify(st, case.b_poly, T, crs, beta_prime=float("inf")) is True
    crs.close()


@pytest.mark.parametrize("q,n,r,kappa,kappa1", [(Q_DEFAULT, 2, 2, 4, 3), (Q_REF33, 4, 3, 6, 5), (Q_REF33, 16, 8, 24, 16),
                                               (Q_REF33, 4, 8, 64, 4)])   # the last one needs the u16 limb
def test_prove_and_verify_with_compact_crs(q, n, r, kappa, kappa1):
    # B, C, D in the digit basis with a u16 limb (10 instead of 12 bytes per coefficient at the 33-bit Q): the
    # transcript is bit-identical to the oracle's and to the main-basis run, and the GPU verifier accepts it
    import copy
    import labrador_snark_amd as lab
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    case, (S, phi, a, tape) = cases.oracle_case(p, seed=21)
    want = case.prove()
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, compact=True).fill_synthetic(1000)
    assert crs.compact and crs.B.digit        # digit basis; the u16 limb appears once the bounds need it
    assert crs.B.compact == ctx.digit_small
    if kappa == 64:
        assert ctx.digit_small and len(ctx.digit_primes) == 3
    st = lab.State(phi, a, case.b_prime)
    T = lab.prove_core(ctx, k, S, st, crs, tape)
    for f, o in (("t_i_all", "t"), ("g_mat", "G"), ("h_mat", "H"), ("u_1", "u1"), ("u_2", "u2"), ("z", "z"), ("b_prime_prime", "bpp")):
        assert np.array_equal(getattr(T, f), want[o]), f
    kinf = copy.copy(k); kinf.BETA_PRIME = float("inf")
    assert lab.verify_core(ctx, kinf, st, case.b_poly, crs, T)[0] == 0
    T2 = copy.copy(T); T2.u_1 = T.u_1.copy(); T2.u_1[0, 0] = (int(T2.u_1[0, 0]) + 1) % q
    assert lab.verify_core(ctx, kinf, st, case.b_poly, crs, T2)[0] == 19
    # seed-addressed CRS expands into the compact form as well
    base = bytes(range(32))
    res = lab.CRS(ctx, k, base).materialize(compact=True)
    ref = lab.CRS(ctx, k, base).materialize(compact=False)
    assert np.array_equal(res.B.store(), ref.B.store()) and np.array_equal(res.Dm.store(), ref.Dm.store())
    for v in (crs, res, ref):
        v.close()
    ctx.close()


@pytest.mark.parametrize("nb,kappa1", [(2, 20), (3, 20), (5, 20), (8, 20), (7, 64), (8, 96)])
@pytest.mark.parametrize("compact", [False, True])
def test_prove_core_batch_equals_separate_proofs(nb, kappa1, compact):
    # independent proofs under one CRS: batched call (B streamed once per group) == separate calls == oracle
    # (kappa1 = 20 runs the LDS "sets" kernel in step 3, kappa1 >= 32 the lane=row kernel with the LDS-DMA ring)
    import labrador_snark_amd as lab
    q, n, r, kappa = Q_REF33, 4, 8, 64
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, compact=compact).fill_synthetic(1000)
    ws, sts, tapes, wants = [], [], [], []
    for j in range(nb):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=100 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
        wants.append(case.prove() if j in (0, nb - 1) else None)
    Ts = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    for j in range(nb):
        single = lab.prove_core(ctx, k, ws[j], sts[j], crs, tapes[j])
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int", "projection"):
            assert np.array_equal(getattr(Ts[j], f), getattr(single, f)), (j, f)
        if wants[j] is not None:
            assert np.array_equal(Ts[j].u_1, wants[j]["u1"]) and np.array_equal(Ts[j].u_2, wants[j]["u2"])
            assert np.array_equal(Ts[j].t_i_all, wants[j]["t"]) and np.array_equal(Ts[j].z, wants[j]["z"])
    # per-proof status: a wrong b' in one proof does not affect the others
    bad = list(sts)
    bad[1] = lab.State(sts[1].phi, sts[1].a, (sts[1].b_prime + 1) % q)
    Ts2, codes = lab.prove_core_batch(ctx, k, ws, bad, crs, tapes, raise_on_reject=False)
    assert codes[1] == lab._lib.LAB_ERR_BPP_CHECK and all(c == 0 for i, c in enumerate(codes) if i != 1)
    assert np.array_equal(Ts2[0].u_1, Ts[0].u_1)
    crs.close()
    ctx.close()


@pytest.mark.parametrize("nb,kappa1", [(1, 20), (5, 20), (9, 64), (20, 40), (33, 20), (64, 20), (66, 20)])
@pytest.mark.parametrize("compact", [False, True, "mfma23"])
def test_prove_core_batch_on_matrix_cores(nb, kappa1, compact):
    # step 3 as exact int8 GEMMs (B packed with lab_mat8_pack, up to 32 proofs per pass, groups padded to 8/16/32
    # sets): transcripts bit-identical to the VALU path and the oracle; B itself can be dropped afterwards
    import labrador_snark_amd as lab
    q, n, r, kappa = Q_REF33, 4, 8, 64
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    # "mfma23": B, C, D in the digit basis of primes below 2^23 (three byte planes, 3- and 6-wave GEMM blocks)
    crs = lab.ResidentCRS(ctx, k, compact=compact is True, mfma_basis=compact == "mfma23").fill_synthetic(1000)
    if compact == "mfma23":
        assert all(pr < (1 << 23) for pr in ctx.digit_primes) and not ctx.digit_small
    ws, sts, tapes = [], [], []
    want0 = None
    for j in range(nb):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=300 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
        if j == 0:
            want0 = case.prove()
    ref = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    crs.pack_mfma(drop_B=True)
    assert crs.B is None and crs.B8.nbytes > 0
    got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    for j in range(nb):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int", "projection"):
            assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (j, f)
    assert np.array_equal(got[0].u_1, want0["u1"]) and np.array_equal(got[0].u_2, want0["u2"])
    crs.close()
    ctx.close()


def test_digit_basis_ajtai_and_its_fallback():
    # with the matrix-core digit basis step 1 (t = A s) runs in that basis as long as the witness keeps its norm bound
    # (checked on the device by the pass that transforms S); a witness above beta_bound makes the call redo the batch in
    # the main basis — either way every transcript equals the multiply-add path's
    import labrador_snark_amd as lab
    q, n, r, kappa, kappa1 = Q_REF33, 4, 8, 64, 20
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, mfma_basis=True).fill_synthetic(1000)
    ws, sts, tapes = [], [], []
    for j in range(3):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=900 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
    ref, _ = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
    assert ctx.prove_flags() == 0
    crs.pack_mfma()
    assert crs.A8d.nbytes > 0
    got, _ = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
    assert ctx.prove_flags() == 1
    for j in range(3):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (j, f)
    # one coefficient of proof 1's witness far above beta_bound: |A s| no longer fits the digit basis
    big = [w.copy() for w in ws]
    big[1][2, 3, 5] = q - 1
    bad = [sts[0], lab.State(sts[1].phi, sts[1].a, 0), sts[2]]
    crs_plain = lab.ResidentCRS(ctx, k, mfma_basis=True).fill_synthetic(1000)
    ref2, codes_ref = lab.prove_core_batch(ctx, k, big, bad, crs_plain, tapes, raise_on_reject=False)
    got2, codes = lab.prove_core_batch(ctx, k, big, bad, crs, tapes, raise_on_reject=False)
    assert ctx.prove_flags() == 2 and list(codes) == list(codes_ref)
    for j in range(3):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z"):
            assert np.array_equal(getattr(got2[j], f), getattr(ref2[j], f)), (j, f)
    crs.close(); crs_plain.close()
    ctx.close()


@pytest.mark.parametrize("q", [Q_DIRECT, Q_DEFAULT])
def test_matrix_core_basis_with_other_moduli(q):
    # the digit basis of primes below 2^23 next to a single-prime main basis (q its own NTT prime: 3 digit limbs, 1 main
    # limb) and next to a small q: same transcripts as the multiply-add path, whichever basis step 1 ends up in
    import labrador_snark_amd as lab
    n, r, kappa, kappa1 = 4, 8, 64, 20
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    if k.T_1 < 1 or k.T_2 < 1 or k.B_1 < 2 or k.B_2 < 2:
        pytest.skip("RuntimeConstants give no usable decomposition at this modulus and size")
    plain = lab.ResidentCRS(ctx, k).fill_synthetic(1000)
    ws, sts, tapes = [], [], []
    for j in range(3):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=40 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
    ref, codes_ref = lab.prove_core_batch(ctx, k, ws, sts, plain, tapes, raise_on_reject=False)
    crs = lab.ResidentCRS(ctx, k, mfma_basis=True).fill_synthetic(1000)
    crs.pack_mfma(drop_B=True)
    got, codes = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
    assert list(codes) == list(codes_ref) and ctx.prove_flags() in (0, 1, 2)
    for j in range(3):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (j, f)
    crs.close(); plain.close()
    ctx.close()


def test_verifier_needs_the_unpacked_B():
    # a prover-only process may drop B after packing it for the matrix cores; lab_verify_core then refuses cleanly
    import labrador_snark_amd as lab
    q, n, r, kappa, kappa1 = Q_REF33, 4, 8, 64, 20
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, compact=True).fill_synthetic(1000)
    case, (S, phi, a, tape) = cases.oracle_case(p, seed=7)
    st = lab.State(phi, a, case.b_prime)
    crs.pack_mfma(drop_B=True)
    T = lab.prove_core(ctx, k, S, st, crs, tape)                     # single proof on the matrix-core path
    want = case.prove()
    assert np.array_equal(T.u_1, want["u1"]) and np.array_equal(T.u_2, want["u2"])
    with pytest.raises(lab.LabradorError):
        lab.verify_core(ctx, k, st, case.b_poly, crs, T)
    crs.close()
    ctx.close()


@pytest.mark.parametrize("nb,kappa", [(8, 130), (12, 256), (16, 128), (34, 130)])
def test_ajtai_product_on_persistent_blocks(nb, kappa):
    # r = 32, at least 8 proofs, at least 128 rows of A: the Ajtai product t = A S of the batch runs on k_gemm_i8_w3p — blocks that
    # stay on their (slot, 128 rows) for several two-proof jobs, the next job's operands under way while the last one's sums are
    # recombined (130 rows: a second row group whose waves 2, 3 have no tile; 34 proofs: 17 jobs in chunks of 16 + 1).  Every field
    # equals the multiply-add path, t and u_1 of the first and last proof equal the oracle.
    import labrador_snark_amd as lab
    q, n, r, kappa1 = Q_REF33, 6, 32, 36
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, mfma_basis=True).fill_synthetic(1000)
    ws, sts, tapes, wants = [], [], [], {}
    for j in range(nb):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=2100 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
        if j in (0, nb - 1):
            wants[j] = case.prove()
    ref = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    crs.pack_mfma()
    got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    assert ctx.prove_flags() == 1
    for j in range(nb):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (j, f)
    for j, w in wants.items():
        assert np.array_equal(got[j].t_i_all, w["t"]) and np.array_equal(got[j].u_1, w["u1"]) and np.array_equal(got[j].h_mat, w["H"])
    crs.close()
    ctx.close()


def test_t_chain_in_slices_of_the_batch():
    # kappa = 8320, r = 32, 32 proofs: the dense inverse-transform output of t for all proofs at once would be 6.5 GB, so the chain
    # Ajtai GEMM -> inverse transform -> CRT (+ digits) runs 16 proofs at a time (what keeps 160 proofs per streamed pass in memory at
    # kappa = 2^14); the witness tiles of the slices land in one array for steps 6-8.  Every field of every proof == the
    # multiply-add path on unpacked matrices; entries of t and G == the oracle.
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q, n, r, kappa, kappa1, nb = Q_REF33, 2, 32, 8320, 2, 32
    ctx = lab.Context(q, 0)
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, mfma_basis=True).fill_synthetic(1000)
    ws, sts, tapes = [], [], []
    for j in range(nb):
        S, phi, a, tape = cases.make_inputs(p, seed=3100 + j)
        ws.append(S); tapes.append(tape)
        sts.append(lab.State(phi, a, int(lab.gen_f_b(ctx, S, phi, a, n, r)[0])))
    ref, c_ref = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
    crs.pack_mfma(main_basis_A=False)
    got, c_got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
    assert ctx.prove_flags() == 1 and c_ref == c_got
    for j in range(nb):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (j, f)
    A = co.synth_fill(1000, 0, kappa * n * D, q).reshape(kappa, n, D)
    for j, i, row in ((0, 0, 0), (15, 31, kappa - 1), (16, 7, 4097), (31, 31, 8319)):
        assert np.array_equal(got[j].t_i_all[i, row], co.inner_product(q, A[row], ws[j][:, i])), (j, i, row)
    for j, i, kk in ((0, 0, 0), (17, 5, 31), (31, 31, 31)):
        assert np.array_equal(got[j].g_mat[i, kk], co.inner_product(q, ws[j][:, i], ws[j][:, kk])), (j, i, kk)
    crs.close()
    ctx.close()


@pytest.mark.parametrize("nproofs,mfma_basis", [(70, True), (130, True), (200, True), (270, True), (130, False)])
def test_more_than_64_proofs_share_one_pass_over_resident_B(nproofs, mfma_basis):
    # B resident as byte planes (B8): a call of more than 64 proofs runs its sub-batches of 64 up to the digit tiles of t, multiplies
    # ALL of them in one launch per limb (the full sub-batches as jobs that share the matrix through L2, a shorter last one by
    # itself) and lets each sub-batch finish.  Every transcript == the same proofs proved 64 at a time == the multiply-add path on
    # unpacked matrices (sample) == the oracle (last proof).
    import labrador_snark_amd as lab
    q, n, r, kappa, kappa1 = Q_REF33, 2, 3, 6, 130      # 130 rows: 5 row tiles, the last with 2 rows
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, compact=True, mfma_basis=mfma_basis).fill_synthetic(1000)
    ws, sts, tapes = [], [], []
    want_last = None
    for j in range(nproofs):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=1500 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
        if j == nproofs - 1:
            want_last = case.prove()
    ref = lab.prove_core_batch(ctx, k, ws[:5] + ws[-3:], sts[:5] + sts[-3:], crs, tapes[:5] + tapes[-3:])
    crs.pack_mfma()
    got, codes = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
    flags = ctx.prove_flags()
    by64 = []
    for j0 in range(0, nproofs, 64):
        by64 += lab.prove_core_batch(ctx, k, ws[j0:j0 + 64], sts[j0:j0 + 64], crs, tapes[j0:j0 + 64])
    assert codes == [0] * nproofs
    assert bool(flags & 4) == (nproofs >= 128)      # two or more full sub-batches went through one launch per limb
    fields = ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int")
    for j in range(nproofs):
        for f in fields:
            assert np.array_equal(getattr(got[j], f), getattr(by64[j], f)), (j, f)
    for a_, b_ in zip(list(range(5)) + list(range(nproofs - 3, nproofs)), range(8)):
        for f in fields:
            assert np.array_equal(getattr(got[a_], f), getattr(ref[b_], f)), (a_, f)
    assert np.array_equal(got[-1].t_i_all, want_last["t"]) and np.array_equal(got[-1].u_1, want_last["u1"])
    crs.close()
    ctx.close()


@pytest.mark.parametrize("nb,r,kappa", [(3, 17, 37), (2, 5, 33), (4, 20, 70), (6, 32, 96), (2, 9, 64)])
def test_t_from_the_staging_arrays_of_every_limb(nb, r, kappa):
    # t = A S of a batch leaves the matrix cores in one slot-major staging array per limb; k_stage_to_coeffs turns them into t mod q
    # and its digits in one pass (inverse transform of each limb by a wave of its own, CRT from LDS).  Ragged cases: rows not a
    # multiple of 32 (padded staging rows), fewer valid sets than the padded 8 / 16 / 32, unpaired (odd, or fewer than 32 columns) and
    # paired jobs.  Every field == the multiply-add path; t and u_1 of the last proof == the oracle.
    import labrador_snark_amd as lab
    q, n, kappa1 = Q_REF33, 3, 7
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, compact=True, mfma_basis=True).fill_synthetic(1000)
    ws, sts, tapes = [], [], []
    want_last = None
    for j in range(nb):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=900 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
        if j == nb - 1:
            want_last = case.prove()
    ref = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    crs.pack_mfma()
    got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    assert ctx.prove_flags() & 1
    for j in range(nb):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (j, f)
    assert np.array_equal(got[-1].t_i_all, want_last["t"]) and np.array_equal(got[-1].u_1, want_last["u1"])
    crs.close()
    ctx.close()


@pytest.mark.parametrize("mfma_basis", [False, True])
@pytest.mark.parametrize("nb", [2, 4, 5])
def test_prove_core_batch_matrix_cores_r32(nb, mfma_basis):
    # r = 32 witness columns: the Ajtai product of an even number of proofs multiplies two proofs as one 64-set job
    # (paired jobs), an odd number keeps one 32-set job per proof; both equal the multiply-add path and the oracle
    import labrador_snark_amd as lab
    q, n, r, kappa, kappa1 = Q_REF33, 6, 32, 40, 36
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, compact=True, mfma_basis=mfma_basis).fill_synthetic(1000)
    ws, sts, tapes = [], [], []
    want_last = None
    for j in range(nb):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=500 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
        if j == nb - 1:
            want_last = case.prove()
    ref = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    crs.pack_mfma()
    got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    for j in range(nb):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (j, f)
    assert np.array_equal(got[-1].t_i_all, want_last["t"]) and np.array_equal(got[-1].u_1, want_last["u1"])
    crs.close()
    ctx.close()


@pytest.mark.parametrize("r", [16, 17, 24, 31])
@pytest.mark.parametrize("nb", [2, 3])
def test_prove_core_batch_digit_basis_gram_products_ragged_r(nb, r):
    # steps 6-8 and 10 from the digit-basis witness (phi'^T S and W^T S on the matrix cores from packed tiles, z from the
    # digit-basis transform) for 17 <= r <= 32 columns (zero columns in the 32-wide tiles, ragged k tiles: n = 70), the
    # multiply-add Gram kernels for r = 16: every field equals the multiply-add path and the oracle
    import labrador_snark_amd as lab
    q, n, kappa, kappa1 = Q_REF33, 70, 33, 36
    ctx = lab.Context(q, 0)
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, compact=True, mfma_basis=True).fill_synthetic(1000)
    ws, sts, tapes, wants = [], [], [], []
    for j in range(nb):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=900 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
        wants.append(case.prove())
    ref = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    crs.pack_mfma()
    got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    assert ctx.prove_flags() == 1                       # the digit-basis path ran (no fallback)
    for j in range(nb):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (j, f)
        w = wants[j]
        assert np.array_equal(got[j].h_mat, w["H"]) and np.array_equal(got[j].z, w["z"]) and np.array_equal(got[j].b_prime_prime, w["bpp"])
    crs.close()
    ctx.close()


def test_batch_plan_reuses_its_buffers():
    # lab.BatchPlan: the boundary structures built once; run() proves what the persistent buffers hold NOW
    import labrador_snark_amd as lab
    q, n, r, kappa, kappa1 = Q_REF33, 6, 3, 8, 5
    ctx = lab.Context(q, 0)
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k).fill_synthetic(1000)
    ws, sts, tapes = [], [], []
    for j in range(3):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=700 + j)
        ws.append(np.ascontiguousarray(S)); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
    plan = lab.BatchPlan(ctx, k, ws, sts, crs, tapes, device_outputs=False)
    fields = ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int")
    for rounds in range(2):
        got, codes = plan.run(raise_on_reject=False)
        ref, rcodes = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
        assert codes == rcodes
        for j in range(3):
            for f in fields:
                assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (rounds, j, f)
        ws[1][0, 0, :] = (ws[1][0, 0, :] + 1) % 3          # refill a witness buffer in place; b' no longer matches: status only
    crs.close()
    ctx.close()


def _pack_pi_numpy(Pi):
    """independent restatement of the tiled 2-bit layout (include/labrador_hip.h, lab_prove_input.Pi2), entry by entry"""
    mats, rows, cols = Pi.shape
    out = np.zeros((mats, 4 * (cols // 16) * 64), dtype=np.uint32)
    for i in range(mats):
        for m in range(256):
            s, kq, t = m // 64, (m % 64) // 16, m % 16
            shift = 8 * (t % 4) + 2 * (t // 4)
            for col in range(cols):
                ct, cc = col // 16, col % 16
                out[i, (s * (cols // 16) + ct) * 64 + 16 * kq + cc] |= np.uint32((int(Pi[i, m, col]) + 1) << shift)
    return out.view(np.uint8).reshape(mats, 256 * cols // 4)


@pytest.mark.parametrize("q", ALL_Q)
@pytest.mark.parametrize("n,r", [(4, 3), (65, 2), (128, 1)])
def test_packed_pi_gives_the_same_transcript(q, n, r):
    # Pi at 2 bits per entry (lab_prove_input.Pi2, lab_pi_pack) against the int8 form: every transcript field, for a
    # norm-bounded witness and for a full-range one (raw residues up to q - 1: the projection is then rejected, its
    # integer value is still produced); 256, 4160 (one full + one ragged 4096-column chunk) and 8192 columns
    import labrador_snark_amd as lab
    kappa, kappa1 = 8, 5
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k).fill_synthetic(1000)
    S, phi, a, tape = cases.make_inputs(p, seed=4242 + n)
    packed = lab.pack_pi(ctx, tape["Pi"])
    assert packed.dtype == np.uint8 and np.array_equal(packed, lab.pack_pi_host(tape["Pi"]))
    assert np.array_equal(lab.unpack_pi_host(packed, n * D), tape["Pi"])
    if n == 4:
        assert np.array_equal(packed, _pack_pi_numpy(tape["Pi"]))
    tape2 = dict(tape)
    del tape2["Pi"]
    tape2["Pi2"] = packed
    rng = np.random.default_rng(7)
    S_full = rng.integers(0, q, size=S.shape, dtype=np.uint64)
    S_full[0, 0, :4] = q - 1
    for W in (S, S_full):
        st = lab.State(phi, a, int(lab.gen_f_b(ctx, W, phi, a, n, r)[0]))
        (T1,), c1 = lab.prove_core_batch(ctx, k, [W], [st], crs, [tape], raise_on_reject=False)
        (T2,), c2 = lab.prove_core_batch(ctx, k, [W], [st], crs, [tape2], raise_on_reject=False)
        assert c1 == c2
        for f in ("projection_int", "projection", "b_prime_prime", "h_mat", "u_2", "u_1", "z", "t_i_all", "g_mat"):
            assert np.array_equal(getattr(T1, f), getattr(T2, f)), f
    # the verifier takes the packed form too
    st = lab.State(phi, a, int(lab.gen_f_b(ctx, S, phi, a, n, r)[0]))
    b_poly = lab.gen_f_b(ctx, S, phi, a, n, r)
    T = lab.prove_core(ctx, k, S, st, crs, tape2)
    kinf = copy.copy(k)
    kinf.BETA_PRIME = float("inf")
    assert lab.verify_core(ctx, kinf, st, b_poly, crs, T, tape=tape2)[0] == lab.verify_core(ctx, kinf, st, b_poly, crs, T, tape=tape)[0] == 0
    crs.close()
    ctx.close()


def test_packed_pi_on_the_norm_bounded_fast_path_and_its_fallback():
    # with step 1 in the digit basis (A8d) the witness is inside its norm bound or the batch is redone, so the packed-Pi pass
    # uses one multiply-add per entry for the projection (coefficients < 2^24); a witness above the bound takes the general
    # pass in the redone batch — both equal the int8 form
    import labrador_snark_amd as lab
    q, n, r, kappa, kappa1 = Q_REF33, 4, 8, 64, 20
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    crs = lab.ResidentCRS(ctx, k, mfma_basis=True).fill_synthetic(1000)
    crs.pack_mfma()
    ws, sts, tapes, tapes2 = [], [], [], []
    # per proof the pass picks the 16-bit form (two entries per v_dot2_u32_u16) or the 24-bit form from a flag the witness transform
    # sets: proof 0 has a coefficient that needs all 24 bits, proof 1 one at 2^16 - 1 (the largest the 16-bit form takes, in the
    # last column tile of a pair), proof 2 one at 2^16 (the smallest that does not), proof 3 is small throughout
    special = {0: ((0, 0, 0), min(k.BETA_BOUND // 2, (1 << 24) - 1)), 1: ((n - 1, r - 1, 63), 65535), 2: ((1, 2, 17), 65536)}
    for j in range(4):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=900 + j)
        S = S.copy()
        if j in special:
            S[special[j][0]] = special[j][1]
        st = lab.State(phi, a, int(lab.gen_f_b(ctx, S, phi, a, n, r)[0]))
        ws.append(S); sts.append(st); tapes.append(tape)
        t2 = {kk: v for kk, v in tape.items() if kk != "Pi"}
        t2["Pi2"] = lab.pack_pi(ctx, tape["Pi"])
        tapes2.append(t2)
    ref, c_ref = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
    assert ctx.prove_flags() == 1
    got, c_got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes2, raise_on_reject=False)
    assert ctx.prove_flags() == 1 and c_ref == c_got
    big = [w.copy() for w in ws]
    big[1][2, 3, 5] = q - 1
    ref2, c_ref2 = lab.prove_core_batch(ctx, k, big, sts, crs, tapes, raise_on_reject=False)
    got2, c_got2 = lab.prove_core_batch(ctx, k, big, sts, crs, tapes2, raise_on_reject=False)
    assert ctx.prove_flags() == 2 and c_ref2 == c_got2
    for A, B in ((ref, got), (ref2, got2)):
        for j in range(4):
            for f in ("projection_int", "projection", "b_prime_prime", "h_mat", "u_2", "u_1", "z", "t_i_all", "g_mat"):
                assert np.array_equal(getattr(A[j], f), getattr(B[j], f)), (j, f)
    # the integer projection itself against the oracle's jl_project (src/proofgen.rs:429-456), every proof
    from oracle import coracle as co
    for j in range(4):
        s_cols = np.ascontiguousarray(ws[j].transpose(1, 0, 2)).reshape(r, n * 64)        # column i of the witness, flattened
        assert np.array_equal(np.asarray(got[j].projection_int, dtype=np.int64), co.jl_project(tapes[j]["Pi"], s_cols)), j
    crs.close()
    ctx.close()


@pytest.mark.parametrize("q", [Q_DEFAULT, Q_REF33])
def test_state_gen_f_and_transcript_bytes_from_a_gpu_proof(q):
    # State::new / gen_f (src/structs.rs:289-387) on the Gram / inner-product kernels: b equals the oracle's orc_state_b for
    # the same (phi, a); then the proof made from that state serialises (bincode layout of src/structs.rs:192-220) to bytes
    # an independent reader decodes back to the device results, trimmed polynomials included
    import struct
    import labrador_snark_amd as lab
    from labrador_snark_amd import transcript as tr
    from oracle import coracle as co
    from test_transcript import Reader
    n, r, kappa, kappa1 = 4, 3, 6, 5
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    S, _, _, tape = cases.make_inputs(p, seed=5)
    st = lab.State.new(ctx, S, k, np.random.default_rng(9))
    assert np.array_equal(st.a, st.a.transpose(1, 0, 2))                        # a_constraints symmetric (src/structs.rs:297-300)
    want_b = np.empty(D, np.uint64)
    co.lib().orc_state_b(q, n, r, np.ascontiguousarray(S), np.ascontiguousarray(st.phi), np.ascontiguousarray(st.a), want_b)
    assert np.array_equal(st.b, want_b) and st.b_prime == int(want_b[0])
    crs = lab.ResidentCRS(ctx, k).fill_synthetic(1000)
    (T,), (code,) = lab.prove_core_batch(ctx, k, [S], [st], crs, [tape], raise_on_reject=False)
    assert code in (0, -6)                                                       # b'' check passes with gen_f's b'
    data = tr.serialize_transcript(T, q, tape)
    rd = Reader(data)
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64), T.u_1)
    assert rd.u64() == r
    for i in range(r):
        assert rd.u8() == 1 and (rd.u64(), rd.u64()) == (256, n * D)
        rd.o += 8 + 16 * 256 * n * D                                             # the lifted Pi_i block
    assert rd.vec_zq() == [int(x) for x in T.projection]
    assert rd.u64() == 1 and rd.vec_zq() == [int(tape["psi"]) % q]
    assert rd.u64() == 1 and rd.vec_zq() == [int(x) for x in tape["omega"]]
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64)[0], T.b_prime_prime)
    rd.vec_rq(); rd.vec_rq()                                                     # alpha, beta
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64), T.u_2)
    rd.vec_rq()                                                                  # c
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64), T.z)
    assert rd.u64() == r
    for i in range(r):
        assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64), T.t_i_all[i])
    for M in (T.g_mat, T.h_mat):
        assert rd.u8() == 1 and (rd.u64(), rd.u64()) == (r, r)
        assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64).reshape(r, r, D), M)
    assert rd.o == len(data)
    crs.close()
    ctx.close()


# ------------------------------------------------------------------ four 23-bit digit primes (what kappa = 2^14 selects)
def test_four_prime_digit_basis_matvec_on_matrix_cores():
    # lab_ctx_enable_digit_basis_mfma(2^21 terms): r*t1*kappa at kappa = 2^14 is 1.5 * 2^20 — FOUR primes below 8355712.  The int8 GEMM,
    # its 4-limb CRT (k_crt_to_u64<4>) and the fused digits -> tiles pass in that basis against the multiply-add kernel and the oracle
    import labrador_snark_amd as lab
    from oracle import coracle as co
    q = Q_REF33
    ctx = lab.Context(q, 0)
    primes, small, _ = ctx.enable_digit_basis(1 << 21, 1625, mfma=True)
    assert len(primes) == 4 and not small and all(pr < 8355712 and pr % 128 == 1 for pr in primes) and len(set(primes)) == 4
    rng = np.random.default_rng(44)
    for ns, nmat, rows, cols in ((8, 3, 40, 70), (64, 2, 130, 256), (32, 5, 128, 100)):
        M = rng.integers(0, q, size=(nmat, rows, cols, D), dtype=np.uint64)
        V = rng.integers(0, 1625, size=(ns, nmat, cols, D), dtype=np.uint64)
        M[0, :2] = q - 1
        V[ns - 1] = 1624
        hM = ctx.vec(nmat * rows * cols, digit=True).load(M)
        hV = ctx.vec(ns * nmat * cols, digit=True).load(V)
        M8 = lab.Mat8(ctx, hM, nmat, rows, cols)
        got = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
        for b in sorted({0, 3, ns - 1}):
            assert np.array_equal(got[b], lab.matvec_sum(ctx, hM, hV, nmat, rows, cols, v_off=b * nmat * cols).store()), (ns, b)
        for b, r_ in ((0, 0), (ns - 1, rows - 1), (ns - 1, 1)):
            want = np.zeros(D, np.uint64)
            for m in range(nmat):
                want = co.poly_add(q, want, co.inner_product(q, M[m, r_], V[b, m]))
            assert np.array_equal(got[b, r_], want), (ns, b, r_)
        M8.close(); hM.close(); hV.close()
    ctx.close()


@pytest.mark.parametrize("nb", [3, 4, 9])
def test_prove_core_batch_in_the_four_prime_digit_basis(nb):
    # lab_prove_core_batch with B8, C8, D8 and A8d in the four-prime basis, r = 32 (the witness products of steps 1, 2, 6-8, 10 all in
    # that basis; nb = 4: two proofs per 64-set Ajtai job, 3 and 9: one per job): every field == the multiply-add path in the main
    # basis == the oracle
    import labrador_snark_amd as lab
    q, n, r, kappa, kappa1 = Q_REF33, 6, 32, 40, 36
    ctx = lab.Context(q, 0)
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
    plain = lab.ResidentCRS(ctx, k).fill_synthetic(1000)            # main basis, multiply-add kernels
    primes, _, _ = ctx.enable_digit_basis(1 << 21, max(k.B_1, k.B_2), mfma=True)
    assert len(primes) == 4
    crs = lab.ResidentCRS(ctx, k, mfma_basis=True).fill_synthetic(1000)
    crs.pack_mfma(drop_B=True)
    assert crs.A8d.nbytes > 0 and crs.B is None
    ws, sts, tapes, wants = [], [], [], []
    for j in range(nb):
        case, (S, phi, a, tape) = cases.oracle_case(p, seed=1300 + j)
        ws.append(S); sts.append(lab.State(phi, a, case.b_prime)); tapes.append(tape)
        wants.append(case.prove() if j in (0, nb - 1) else None)
    ref = lab.prove_core_batch(ctx, k, ws, sts, plain, tapes)
    assert ctx.prove_flags() == 0
    got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes)
    assert ctx.prove_flags() == 1                                   # digit-basis step 1, no fallback
    for j in range(nb):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int", "projection"):
            assert np.array_equal(getattr(got[j], f), getattr(ref[j], f)), (j, f)
        w = wants[j]
        if w is not None:
            for f, o in (("t_i_all", "t"), ("g_mat", "G"), ("h_mat", "H"), ("u_1", "u1"), ("u_2", "u2"), ("z", "z"), ("b_prime_prime", "bpp")):
                assert np.array_equal(getattr(got[j], f), w[o]), (j, f)
    # Pi at 2 bits per entry on the same path (the bench's form)
    tapes2 = [dict(t, Pi2=lab.pack_pi_host(t["Pi"])) for t in tapes]
    got2 = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes2)
    for j in range(nb):
        for f in ("h_mat", "u_2", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got2[j], f), getattr(ref[j], f)), (j, f)
    crs.close(); plain.close()
    ctx.close()
