"""The configuration bench.py TIMES, under the oracle at the size it is timed at.

bench.py's default step = lab_prove_core_batch over 64 proofs, q = 2^32+15, n = 1024, r = 32,
kappa = kappa1 = kappa2 = 1024, ResidentCRS(mfma_basis=True).pack_mfma(drop_B=True): step 3 on k_gemm_i8 with
K = r*t1*kappa = 98304, step 1 in the digit basis (A8d), k_digits_to_v8, k_intt_from_stage, k_gram_v8.  The workload is
built by bench.build_workload itself; for proofs 0, 31 and 63 the oracle recomputes spot rows of every computed
transcript field (src/proofgen.rs:41-49, 59-70, 101-153, 161-186, 205-289, 301-358, 364-399), and a second test redoes
three proofs on the multiply-add kernels (single-proof path, B as u32 residues) and compares EVERY output word.
Everything is bit-exact and goes through the C ABI.
"""
import copy

import numpy as np
import pytest

from conftest import Q_REF33, ctx_kw

pytestmark = pytest.mark.gpu
D = 64
NB = 64
CHECKED = (0, 31, 63)


def _host(t):
    return t.cpu().numpy().view(np.uint64) if hasattr(t, "cpu") else np.asarray(t)


@pytest.fixture(scope="module")
def bench_run():
    import torch
    import bench
    import labrador_snark_amd as lab
    q, n, r, kappa = Q_REF33, 1024, 32, 1024
    k = lab.RuntimeConstants(q, n, r, kappa, kappa, kappa)
    ctx = lab.Context(q, 0, max(65536, n * r + r * r, **ctx_kw(q)))
    ctx.use_torch_stream()
    dev = torch.device("cuda", 0)
    dSs, sts, dtapes, crs = bench.build_workload(lab, ctx, k, q, dev, NB, 50, 1000, use_mfma=True, basis23=True, compact=True)
    assert crs.B is None and crs.B8 is not None and crs.A8d is not None      # exactly bench.py's default objects
    plan = lab.BatchPlan(ctx, k, dSs, sts, crs, dtapes, device_outputs=True)     # bench.py's step() is plan.run()
    plan.run()
    Ts = plan.run()                                                              # second run: the reused buffers
    torch.cuda.synchronize()
    flags = ctx.prove_flags()
    run = dict(ctx=ctx, k=k, q=q, n=n, r=r, kappa=kappa, dSs=dSs, sts=sts, dtapes=dtapes, crs=crs, Ts=Ts, flags=flags)
    yield run
    crs.close()
    ctx.close()


def _fields(T):
    return {name: _host(getattr(T, name)) for name in ("u_1", "projection", "b_prime_prime", "u_2", "z", "t_i_all", "g_mat", "h_mat")}


def test_bench_step_ran_in_the_digit_basis(bench_run):
    assert bench_run["flags"] == 1          # step 1 on A8d, no fallback to the main basis
    assert bench_run["ctx"].digit_primes == [8355329, 8355073, 8353153]


@pytest.mark.parametrize("j", CHECKED)
def test_bench_transcript_spot_rows_against_the_oracle(bench_run, j):
    from oracle import coracle as co
    R = bench_run
    q, n, r, kappa, k = R["q"], R["n"], R["r"], R["kappa"], R["k"]
    T = R["Ts"][j]
    S = _host(R["dSs"][j]).reshape(n, r, D)
    phi = _host(R["sts"][j].phi).reshape(n, r, D)
    a = _host(R["sts"][j].a).reshape(r, r, D)
    tape = R["dtapes"][j]
    import labrador_snark_amd as lab
    Pi = lab.unpack_pi_host(tape["Pi2"].cpu().numpy(), n * D)   # bench.py's default: Pi at 2 bits per entry, tiled (lab_prove_input.Pi2)
    omega, alpha, beta, c = (_host(tape[x]) for x in ("omega", "alpha", "beta", "c"))
    psi = int(tape["psi"])
    F = _fields(T)
    col = lambda M, i: np.ascontiguousarray(M[:, i])
    # step 1: t_i[rho] = <A[rho,:], s_i>
    for (i, rho) in ((0, 0), (5, 7), (31, 1023)):
        a_row = co.synth_fill(1000, rho * n * D, n * D, q).reshape(n, D)
        assert np.array_equal(F["t_i_all"][i, rho], co.inner_product(q, a_row, col(S, i))), (j, i, rho)
    # step 2: G
    for (i, jj) in ((0, 5), (31, 31), (17, 2)):
        assert np.array_equal(F["g_mat"][i, jj], co.inner_product(q, col(S, i), col(S, jj)))
    # step 10: z
    assert np.array_equal(F["z"][3], co.inner_product(q, S[3], c))
    assert np.array_equal(F["z"][1023], co.inner_product(q, S[1023], c))
    # step 5: projection over the integers, and its lift
    s_cols = np.ascontiguousarray(S.transpose(1, 0, 2).reshape(r, n * D))
    proj = co.jl_project(Pi, s_cols)
    assert np.array_equal(_host(T.projection_int).view(np.int64), proj)
    assert np.array_equal(F["projection"], np.array([int(x) % q for x in proj], dtype=np.uint64))
    # steps 3 + 4: one full row of u_1 = sum_ik <B_ik[rho,:], t_i^(k)> + sum_{i<=j,k} g_ij^(k) C_ijk[rho]
    rho = (2, 517, 1023)[CHECKED.index(j)]
    acc = np.zeros(D, np.uint64)
    for i in range(r):
        dec = co.decompose_vec(q, F["t_i_all"][i], k.B_1, k.T_1)
        for kk in range(k.T_1):
            m = i * k.T_1 + kk
            b_row = co.synth_fill(1001, ((m * kappa + rho) * kappa) * D, kappa * D, q).reshape(kappa, D)
            acc = co.poly_add(q, acc, co.inner_product(q, b_row, dec[kk]))
    pair = 0
    for i in range(r):
        for jj in range(i, r):
            gdec = co.decompose(q, F["g_mat"][i, jj], k.B_2, k.T_2)
            for kk in range(k.T_2):
                c_elem = co.synth_fill(1002, (((pair * k.T_2 + kk) * kappa) + rho) * D, D, q)
                acc = co.poly_add(q, acc, co.polymul(q, gdec[kk], c_elem))
            pair += 1
    assert np.array_equal(F["u_1"][rho], acc)
    # steps 6-8: phi''_i = psi phi'_i + sum_m omega_m sigma_inv(pi_i^(m)), phi_i = alpha phi'_i + beta phi''_i,
    # h_ij = 2^-1 (<phi_i, s_j> + <phi_j, s_i>) — columns hi, hj only
    om = omega.astype(np.int64)

    def phi_pp_col(i):
        P = Pi[i].reshape(256, n, D).astype(np.int64)
        Wraw = np.einsum("m,mxk->xk", om, P)                    # sum_m omega_m pi^(m)[x][k] over the integers (|.| < 2^42)
        W = np.empty_like(Wraw)
        W[:, 0] = Wraw[:, 0]
        W[:, 1:] = -Wraw[:, :0:-1]                              # sigma_inv: out[64-k] = -c_k
        W = np.mod(W, q).astype(np.uint64)
        return np.stack([co.poly_add(q, co.poly_scale(q, phi[x, i], psi), W[x]) for x in range(n)])

    def phi_col(i, pp):
        return np.stack([co.poly_add(q, co.polymul(q, alpha, phi[x, i]), co.polymul(q, beta, pp[x])) for x in range(n)])

    hi, hj = (3, 29)
    ppi, ppj = phi_pp_col(hi), phi_pp_col(hj)
    fi, fj = phi_col(hi, ppi), phi_col(hj, ppj)
    inv2 = (q + 1) // 2
    h = co.poly_scale(q, co.poly_add(q, co.inner_product(q, fi, col(S, hj)), co.inner_product(q, fj, col(S, hi))), inv2)
    assert np.array_equal(F["h_mat"][hi, hj], h) and np.array_equal(F["h_mat"][hj, hi], h)
    # step 9: one full row of u_2 = sum_{i<=j,k} h_ij^(k) D_ijk[rho]
    acc = np.zeros(D, np.uint64)
    pair = 0
    for i in range(r):
        for jj in range(i, r):
            hdec = co.decompose(q, F["h_mat"][i, jj], k.B_1, k.T_1)
            for kk in range(k.T_1):
                d_elem = co.synth_fill(1003, (((pair * k.T_1 + kk) * kappa) + rho) * D, D, q)
                acc = co.poly_add(q, acc, co.polymul(q, hdec[kk], d_elem))
            pair += 1
    assert np.array_equal(F["u_2"][rho], acc)
    if j == 0:
        # step 6: b'' = sum_ij (psi a'_ij) g_ij + sum_i <phi''_i, s_i>, every column
        bpp = np.zeros(D, np.uint64)
        for i in range(r):
            for jj in range(r):
                bpp = co.poly_add(q, bpp, co.polymul(q, co.poly_scale(q, a[i, jj], psi), F["g_mat"][i, jj]))
        for i in range(r):
            bpp = co.poly_add(q, bpp, co.inner_product(q, phi_pp_col(i), col(S, i)))
        assert np.array_equal(F["b_prime_prime"], bpp)


def test_bench_batch_equals_the_multiply_add_path_word_for_word(bench_run):
    """Proofs 0, 17 and 63 once more, one at a time, on the VALU kernels (k_matvec_partial over B as u32 residues in the
    same digit basis, k_matmul / k_gram for steps 1-2 in the main basis): every output word equals the batch's."""
    import labrador_snark_amd as lab
    R = bench_run
    ctx, k, crs = R["ctx"], R["k"], R["crs"]
    B = lab.HatVec(ctx, k.R * k.T_1 * k.KAPPA_1 * k.KAPPA, digit=True)
    B.fill_synthetic(1001)
    plain = copy.copy(crs)
    plain.B = B
    plain.B8 = plain.A8 = plain.C8 = plain.D8 = plain.A8d = None
    try:
        for j in (0, 17, 63):
            T1 = lab.prove_core(ctx, k, R["dSs"][j], R["sts"][j], plain, R["dtapes"][j], device_outputs=True)
            assert ctx.prove_flags() == 0
            want, got = _fields(T1), _fields(R["Ts"][j])
            for name in want:
                assert np.array_equal(want[name], got[name]), (j, name)
            assert np.array_equal(_host(T1.projection_int), _host(R["Ts"][j].projection_int))
    finally:
        B.close()


def test_gemm_diag_clock_stamps_leave_the_commitments_unchanged(bench_run):
    """lab_ctx_gemm_diag mode 1 (tools/gemm_ablation.py): the product loop with s_memtime / s_memrealtime read around it —
    the transcripts equal the product kernel's, and every block of the last limb launch left its two tick counts."""
    import torch
    import labrador_snark_amd as lab
    R = bench_run
    ctx = R["ctx"]
    ticks = torch.zeros(2 * 512, dtype=torch.int64, device="cuda:0")      # 64 slots x 8 row groups of 4 tiles
    ctx.gemm_diag(1, ticks)
    try:
        Ts = lab.prove_core_batch(ctx, R["k"], R["dSs"], R["sts"], R["crs"], R["dtapes"], device_outputs=True)
        torch.cuda.synchronize()
    finally:
        ctx.gemm_diag(0, None)
    for j in CHECKED:
        for name in ("u_1", "u_2", "z"):
            assert np.array_equal(_host(getattr(Ts[j], name)), _host(getattr(R["Ts"][j], name))), (j, name)
    t = ticks.cpu().numpy().reshape(-1, 2)
    assert (t > 0).all()
    assert 0.5 < (t[:, 0] / t[:, 1]).mean() * 0.1 < 3.0          # shader clock in GHz against the 100 MHz constant clock
    with pytest.raises(lab.LabradorError):
        ctx.gemm_diag(7, None)
