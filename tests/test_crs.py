"""CRS random oracle (src/structs.rs:27-190).  CPU part: the ChaCha20 block function against
the RFC 7539 / djb known-answer vectors and (when available) an independent OpenSSL keystream; the C and Python
restatements of rand's seed handling + gen_range against each other; the literal seed-offset formulas.
The rand_chacha/rand internals are restated from memory of their published algorithms: PARITY UNPINNED for
those (documented in oracle/labrador_oracle.h)."""
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import Q_DEFAULT, Q_REF33, Q_DIRECT, ctx_kw

from oracle import coracle as co
from oracle import pyref

SIGMA = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
D = 64


def test_chacha20_block_known_answers():
    # all-zero key/nonce, block counter 0 (djb's original vector; also rand_chacha's test_chacha_true_values_a)
    want0 = bytes.fromhex("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                          "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    st = SIGMA + [0] * 12
    assert struct.pack("<16I", *[int(x) for x in co.chacha20_block(st)]) == want0
    assert struct.pack("<16I", *pyref.chacha20_block(st)) == want0
    # RFC 7539 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00
    want1 = bytes.fromhex("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                          "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    st = SIGMA + list(struct.unpack("<8I", bytes(range(32)))) + [1, 0x09000000, 0x4A000000, 0]
    assert struct.pack("<16I", *[int(x) for x in co.chacha20_block(st)]) == want1
    assert struct.pack("<16I", *pyref.chacha20_block(st)) == want1


@pytest.mark.skipif(shutil.which("openssl") is None, reason="openssl CLI not available")
def test_chacha20_block_matches_openssl_keystream():
    rng = np.random.default_rng(1)
    for _ in range(3):
        key = bytes(int(x) for x in rng.integers(0, 256, 32))
        # OpenSSL's 16-byte IV = 32-bit LE counter || 96-bit nonce; all zero = rand_chacha's counter 0, stream 0
        ks = subprocess.run(["openssl", "enc", "-chacha20", "-K", key.hex(), "-iv", "00" * 16],
                            input=bytes(128), capture_output=True, check=True).stdout
        for counter in (0, 1):
            st = SIGMA + list(struct.unpack("<8I", key)) + [counter, 0, 0, 0]
            assert struct.pack("<16I", *[int(x) for x in co.chacha20_block(st)]) == ks[64 * counter:64 * counter + 64]


@pytest.mark.parametrize("q", [Q_DEFAULT, Q_REF33, Q_DIRECT])
def test_crs_fill_twins_agree_and_are_uniform_enough(q):
    base = bytes(range(100, 132))
    got = co.crs_fill(base, 12345, 300, q)
    want = pyref.crs_fill(base, 12345, 300, q)
    assert [int(x) for x in got] == want
    assert all(0 <= x < q for x in want)
    # consecutive offsets continue the same stream (fetch_next_n, src/structs.rs:147-153)
    assert np.array_equal(co.crs_fill(base, 12345 + 100, 50, q), got[100:150])


def test_seed_addition_carries_across_the_256_bit_seed():
    q = Q_DEFAULT
    base = bytes([0] * 15 + [0xFF] * 17)             # low 136 bits all ones: +1 carries into byte 14
    a = co.crs_fill(base, 0, 3, q)
    b = pyref.crs_fill(base, 0, 3, q)
    assert [int(x) for x in a] == b
    big = (1 << 100) + 77                             # offsets beyond 64 bits (reference kappa puts B offsets near 2^45..)
    assert [int(x) for x in co.crs_fill(base, big, 2, q)] == pyref.crs_fill(base, big, 2, q)


def test_rejection_sampling_is_exercised_at_33_bit_q():
    # at Q = 2^32+15 the zone is ~2^127: about half of the u128 candidates are rejected
    import ctypes as C
    draws = []
    for k in range(200):
        d = C.c_int()
        seed = (k * 7919 + 3).to_bytes(32, "big")
        co.lib().orc_crs_coeff(seed, Q_REF33, C.byref(d))
        draws.append(d.value)
    assert max(draws) >= 3 and 1.6 < sum(draws) / len(draws) < 2.6


def test_offsets_copied_literally():
    n, r, kappa, kappa1, kappa2, t1 = 2, 2, 128, 128, 128, 4      # RuntimeConstants::new(2,2) at Q = 8191
    off = pyref.crs_offsets(n, r, kappa, kappa1, kappa2, t1)
    assert co.crs_offset("A", n, r, kappa, kappa1, kappa2, t1, row=5) == off["A"](5) == 5 * n * D
    for (i, k, row) in ((0, 0, 0), (1, 3, 7), (1, 0, 127)):
        assert co.crs_offset("B", n, r, kappa, kappa1, kappa2, t1, i=i, k=k, row=row) == off["B"](i, k, row)
    for (i, j, k) in ((0, 0, 0), (0, 1, 1), (1, 1, 1)):
        assert co.crs_offset("C", n, r, kappa, kappa1, kappa2, t1, i=i, j=j, k=k) == off["C"](i, j, k)
        assert co.crs_offset("D", n, r, kappa, kappa1, kappa2, t1, i=i, j=j, k=k) == off["D"](i, j, k)
    # the B matrices are spaced kappa1*kappa apart although each consumes kappa1*kappa*64 seeds (sic, :81)
    assert off["B"](0, 1, 0) - off["B"](0, 0, 0) == kappa1 * kappa


# ------------------------------------------------------------------ GPU side
@pytest.mark.gpu
@pytest.mark.parametrize("q", [Q_DEFAULT, Q_REF33, Q_DIRECT])
def test_gpu_crs_expand_matches_oracle(contexts, q):
    import labrador_snark_amd as lab
    ctx = contexts(q)
    base = bytes(range(7, 39))
    k = lab.RuntimeConstants(q, 2, 2, 4, 3, 3)
    crs = lab.CRS(ctx, k, base)
    # coefficient-form fetches vs the oracle at the same literal offsets
    row = crs.fetch_A_row(3)
    assert np.array_equal(row.ravel(), co.crs_fill(base, co.crs_offset("A", 2, 2, 4, 3, 3, k.T_1, row=3), 2 * D, q))
    b = crs.fetch_B_ik_row(1, 2, 1)
    assert np.array_equal(b.ravel(), co.crs_fill(base, co.crs_offset("B", 2, 2, 4, 3, 3, k.T_1, i=1, k=2, row=1), 4 * D, q))
    c = crs.fetch_C_ijk(0, 1, 1)
    assert np.array_equal(c.ravel(), co.crs_fill(base, co.crs_offset("C", 2, 2, 4, 3, 3, k.T_1, i=0, j=1, k=1), 3 * D, q))
    d = crs.fetch_D_ijk(1, 1, 0)
    assert np.array_equal(d.ravel(), co.crs_fill(base, co.crs_offset("D", 2, 2, 4, 3, 3, k.T_1, i=1, j=1, k=0), 3 * D, q))
    # a longer stream with a carry-heavy seed and an offset beyond 64 bits
    base2 = bytes([0] * 15 + [0xFF] * 17)
    crs2 = lab.CRS(ctx, k, base2)
    big = (1 << 70) + 12345
    got = crs2._fetch(big, 40)
    assert np.array_equal(got.ravel(), co.crs_fill(base2, big, 40 * D, q))


@pytest.mark.gpu
@pytest.mark.parametrize("q", [Q_REF33, Q_DEFAULT])
def test_gpu_crs_expand_large_count_is_position_independent(contexts, q):
    # k_crs_expand walks a long range in groups of 8 passes of 1024 coefficients per block and queues the coefficients whose first
    # ChaCha20 block is rejected (one in 16 at the 33-bit q): a 6.3-million-polynomial range equals the same range fetched in
    # pieces of odd sizes (other grids, other pass groups), and polynomials at group / pass / range boundaries equal the oracle
    import labrador_snark_amd as lab
    ctx = contexts(q)
    base = bytes((5 * i + 1) % 256 for i in range(32))
    k = lab.RuntimeConstants(q, 2, 2, 4, 3, 3)
    crs = lab.CRS(ctx, k, base)
    off0, count = (1 << 40) + 977, 6 * (1 << 20) + 37
    whole = crs._fetch(off0, count)
    pos = 0
    for piece in (1, 15, 16, 17, 1023, 131072 + 5, 1 << 20, 3 * (1 << 20) + 11):
        part = crs._fetch(off0 + pos * D, piece)
        assert np.array_equal(part, whole[pos:pos + piece]), (pos, piece)
        pos += piece
    for poly in (0, 15, 16, 131071, 131072, 8 * 131072 - 1, 8 * 131072, count - 1):   # 8192 blocks x 1024 / 64 = 131072 polynomials per pass
        assert np.array_equal(whole[poly], co.crs_fill(base, off0 + poly * D, D, q)), poly


@pytest.mark.gpu
def test_gpu_prove_on_seed_addressed_crs_matches_oracle(contexts):
    # the reference's own flow: CRS::new(seed) -> proof_gen fetches rows from the seed (src/main.rs:82-97)
    import labrador_snark_amd as lab
    import cases
    q, n, r, kappa, kappa1 = Q_REF33, 2, 2, 5, 4
    ctx = contexts(q)
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, kappa1)
    base = bytes((i * 37 + 11) % 256 for i in range(32))
    crs = lab.CRS(ctx, k, base)
    S, phi, a, tape = cases.make_inputs(p, seed=41)
    # oracle side: expand the same matrices with the oracle's CRS restatement at the reference's offsets
    off = lambda which, **kw: co.crs_offset(which, n, r, kappa, kappa1, kappa1, p.t1, **kw)
    A = np.stack([co.crs_fill(base, off("A", row=rho), n * D, q).reshape(n, D) for rho in range(kappa)])
    B = np.stack([np.stack([co.crs_fill(base, off("B", i=m // p.t1, k=m % p.t1, row=rho), kappa * D, q).reshape(kappa, D)
                            for rho in range(kappa1)]) for m in range(r * p.t1)])
    pairs = [(i, j) for i in range(r) for j in range(i, r)]
    C = np.stack([co.crs_fill(base, off("C", i=i, j=j, k=kk), kappa1 * D, q).reshape(kappa1, D)
                  for (i, j) in pairs for kk in range(p.t2)])
    Dm = np.stack([co.crs_fill(base, off("D", i=i, j=j, k=kk), kappa1 * D, q).reshape(kappa1, D)
                   for (i, j) in pairs for kk in range(p.t1)])
    case = co.ProverCase(q, n, r, kappa, kappa1, kappa1, p.b1, p.t1, p.b2, p.t2, p.beta_bound, S, phi, a, A, B, C, Dm,
                         tape["Pi"], tape["psi"], tape["omega"], tape["alpha"], tape["beta"], tape["c"])
    want = case.prove()
    assert want["jl_valid"] and want["bpp_ok"]
    res = crs.materialize()
    assert np.array_equal(res.A.store().reshape(kappa, n, D), A)
    assert np.array_equal(res.B.store().reshape(B.shape), B)
    T = lab.prove_core(ctx, k, S, lab.State(phi, a, case.b_prime), res, tape)
    res.close()
    assert np.array_equal(T.u_1, want["u1"]) and np.array_equal(T.u_2, want["u2"])
    assert np.array_equal(T.t_i_all, want["t"]) and np.array_equal(T.z, want["z"])
    assert np.array_equal(T.h_mat, want["H"]) and np.array_equal(T.g_mat, want["G"])


# ------------------------------------------------------------------ streamed mode (row f1 as written)
@pytest.mark.gpu
@pytest.mark.parametrize("mfma_basis", [True, False])
# cols % 64 == 0 on the 3-plane basis: the slab goes from coefficients to tiles in one pass (k_coefs_to_a8) — 131 rows: a 3-row
# ragged slab; 4096 columns: expansion chunks of 64 rows inside a slab
# 300 rows: slabs of 256 rows (one full, one ragged of 44); 600 and 1100 rows: slabs of 512 (ragged 88 resp. 76 rows), on the
# one-pass and on the transform + pack path
@pytest.mark.parametrize("nmat,rows,cols,ns", [(3, 40, 37, 8), (2, 300, 64, 32), (5, 130, 70, 64), (3, 131, 128, 16), (1, 70, 4096, 8),
                                               (2, 600, 70, 16), (3, 1100, 128, 8)])
def test_streamed_crs_matvec_equals_materialise_then_multiply(mfma_basis, nmat, rows, cols, ns):
    # lab_crs_matvec_mfma: slabs of 128 rows expanded from the seed, transformed, packed, multiplied on the matrix cores
    # and discarded == lab_crs_expand of whole matrices + lab_mat8_pack + lab_matvec_sum_mfma, bit for bit; and the oracle's
    # CRS restatement + schoolbook products on sample rows.  300 and 130 rows: three resp. two slabs, the last one ragged.
    import labrador_snark_amd as lab
    q = Q_REF33
    ctx = lab.Context(q, 0, **ctx_kw(q))
    ctx.enable_digit_basis(nmat * cols, 1625, mfma=mfma_basis)
    base = bytes((i * 29 + 5) % 256 for i in range(32))
    # matrix m starts at an arbitrary offset (one beyond 64 bits); its rows are consecutive in seed space
    offs = [1000 + m * (rows * cols * D + 77) + ((1 << 66) if m == 1 else 0) for m in range(nmat)]
    rng = np.random.default_rng(ns + rows)
    V = rng.integers(0, 1625, size=(ns, nmat, cols, D), dtype=np.uint64)
    hV = ctx.vec(ns * nmat * cols, digit=True).load(V)
    got = lab.crs_matvec_mfma(ctx, base, offs, rows, cols, hV, ns).store().reshape(ns, rows, D)
    M = ctx.vec(nmat * rows * cols, digit=True)
    for m in range(nmat):
        ctx._check(ctx._lib.lab_crs_expand(ctx._h, M._h, m * rows * cols, rows * cols, base, offs[m] & ((1 << 64) - 1), offs[m] >> 64))
    M8 = lab.Mat8(ctx, M, nmat, rows, cols)
    want = lab.matvec_sum_mfma(ctx, M8, hV, ns).store().reshape(ns, rows, D)
    assert np.array_equal(got, want)
    for s, rho in ((0, 0), (ns - 1, rows - 1), (3, 129 % rows)):
        acc = np.zeros(D, np.uint64)
        for m in range(nmat):
            row = co.crs_fill(base, offs[m] + rho * cols * D, cols * D, q).reshape(cols, D)
            acc = co.poly_add(q, acc, co.inner_product(q, row, V[s, m]))
        assert np.array_equal(got[s, rho], acc), (s, rho)
    M8.close(); M.close()
    ctx.close()


@pytest.mark.gpu
def test_batch_refuses_sub_batches_with_different_constants():
    # more than 64 proofs are proved in sub-batches that share the commitment pass: a later sub-batch that names
    # another seed or another norm bound than proof 0 is refused as a whole (LAB_ERR_ARG), nothing is proved
    import labrador_snark_amd as lab
    import cases
    from labrador_snark_amd import prover
    q, n, r, kappa, kappa1 = Q_REF33, 2, 2, 5, 4
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, kappa1)
    base = bytes((i * 37 + 11) % 256 for i in range(32))
    streamed = lab.CRS(ctx, k, base).materialize(mfma_basis=True, stream_B=True)
    ws, sts, tapes = [], [], []
    for j in range(65):
        S, phi, a, tape = cases.make_inputs(p, seed=41 + j)
        ws.append(S); tapes.append(tape)
        sts.append(lab.State(phi, a, int(lab.gen_f_b(ctx, S, phi, a, n, r)[0])))
    plan = prover.BatchPlan(ctx, k, ws, sts, streamed, tapes, device_outputs=False)
    plan.run(raise_on_reject=False)
    pin = plan._ios[64][0]
    kept = pin.beta_bound
    pin.beta_bound = kept + 1
    with pytest.raises(lab.LabradorError, match="share dimensions"):
        plan.run(raise_on_reject=False)
    pin.beta_bound = kept
    other = bytes(range(1, 33))
    pin.crs_seed = other
    with pytest.raises(lab.LabradorError, match="share dimensions"):
        plan.run(raise_on_reject=False)
    pin.crs_seed = base
    plan.run(raise_on_reject=False)
    streamed.close()
    ctx.close()


@pytest.mark.gpu
# 70 and 130 proofs: more than one call's 64 — the sub-batches run steps 1-2, suspend, share ONE pass over the regenerated rows
# (one group of vector sets each: 64 + 6, 64 + 64 + 2) and resume for steps 4-10
# 260 proofs: more sub-batches (5) than one shared pass takes (4): every sub-batch then streams B for itself
@pytest.mark.parametrize("nproofs", [1, 3, 70, 130, 260])
def test_prover_with_B_streamed_from_the_seed(nproofs):
    # lab_prove_input.crs_seed with B == B8 == NULL: step 3 regenerates the B_ik rows from the seed inside the call
    # (fetch_B_ik_row's literal offsets) — same transcripts as with the materialised CRS, and as the oracle on its own
    # CRS restatement
    import labrador_snark_amd as lab
    import cases
    q, n, r, kappa, kappa1 = Q_REF33, 2, 2, 5, 4
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, kappa1)
    base = bytes((i * 37 + 11) % 256 for i in range(32))
    crs = lab.CRS(ctx, k, base)
    full = crs.materialize(mfma_basis=True)
    streamed = crs.materialize(mfma_basis=True, stream_B=True)
    assert streamed.B is None and streamed.crs_seed == base
    ws, sts, tapes = [], [], []
    for j in range(nproofs):
        S, phi, a, tape = cases.make_inputs(p, seed=41 + j)
        ws.append(S); tapes.append(tape)
        sts.append(lab.State(phi, a, int(lab.gen_f_b(ctx, S, phi, a, n, r)[0])))
    want, c1 = lab.prove_core_batch(ctx, k, ws, sts, full, tapes, raise_on_reject=False)
    got, c2 = lab.prove_core_batch(ctx, k, ws, sts, streamed, tapes, raise_on_reject=False)
    assert c1 == c2
    for j in range(nproofs):
        for f in ("u_1", "u_2", "t_i_all", "g_mat", "h_mat", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(want[j], f)), (j, f)
    # oracle: u_1 of proof 0 from the oracle's own expansion of B, C at the reference's offsets
    off = lambda which, **kw: co.crs_offset(which, n, r, kappa, kappa1, kappa1, p.t1, **kw)
    T = got[0]
    acc = [np.zeros(D, np.uint64) for _ in range(kappa1)]
    for i in range(r):
        dec = co.decompose_vec(q, T.t_i_all[i], k.B_1, k.T_1)
        for kk in range(k.T_1):
            for rho in range(kappa1):
                b_row = co.crs_fill(base, off("B", i=i, k=kk, row=rho), kappa * D, q).reshape(kappa, D)
                acc[rho] = co.poly_add(q, acc[rho], co.inner_product(q, b_row, dec[kk]))
    pairs = [(i, j) for i in range(r) for j in range(i, r)]
    for (i, j) in pairs:
        gdec = co.decompose(q, T.g_mat[i, j], k.B_2, k.T_2)
        for kk in range(k.T_2):
            c_vec = co.crs_fill(base, off("C", i=i, j=j, k=kk), kappa1 * D, q).reshape(kappa1, D)
            for rho in range(kappa1):
                acc[rho] = co.poly_add(q, acc[rho], co.polymul(q, gdec[kk], c_vec[rho]))
    assert np.array_equal(T.u_1, np.stack(acc))
    full.close(); streamed.close()
    ctx.close()


# ------------------------------------------------------------------ the regime kappa = 2^14 runs in
def _digit_basis(ctx, which, terms):
    """'mfma23': primes below 2^23 sized for `terms`; 'mfma23x4': the same with 2^21 terms (what r*t1*kappa at kappa = 2^14 asks
    for: FOUR primes); 'u16': the compact basis (30-bit prefix + a prime below 2^16: 4- and 2-plane limbs, slab path)"""
    if which == "u16":
        primes, small, _ = ctx.enable_digit_basis(terms, 1625)
        assert small
        return primes
    if which in ("chunk17", "chunk70000"):
        # a basis that holds ONE K chunk but not the whole sum: the chunks are reduced mod q one by one (crs_stream.hip `reduce`) —
        # 2^17 terms: the chunking of the int8 GEMM stands (MC = 31 / 7 / 3 matrices); 70000 terms: the basis cuts shorter chunks
        primes, small, _ = ctx.enable_digit_basis((1 << 17) if which == "chunk17" else 70000, 1625, mfma=True)
        assert not small and len(primes) == 3 and terms > (1 << 17)
        return primes
    primes, small, _ = ctx.enable_digit_basis((1 << 21) if which == "mfma23x4" else terms, 1625, mfma=True)
    assert not small and all(p < 8355712 for p in primes)
    assert len(primes) == (4 if which == "mfma23x4" else 3), primes
    return primes


@pytest.mark.gpu
@pytest.mark.parametrize("basis", ["mfma23", "mfma23x4", "u16", "chunk17", "chunk70000"])
# more matrices than fit one K chunk of the int8 GEMM (MC = floor((2^17 - 2) / cols) matrices per chunk): chunks after the first
# land in the side buffer and are added mod every limb prime — 4096 columns: MC = 31 -> 40 = 31 + 9 and 70 = 31 + 31 + 8;
# 16384 columns (kappa = 2^14's own shape): MC = 7 -> 9 = 7 + 2, 130 rows = a full slab + a ragged one of 2 rows;
# 32768 columns: MC = 3 -> 5 = 3 + 2.  The last chunk is always shorter than MC (the tiles past it must read as zero).
@pytest.mark.parametrize("nmat,rows,cols,ns", [(40, 8, 4096, 8), (70, 33, 4096, 64), (9, 130, 16384, 32), (5, 40, 32768, 16)])
def test_streamed_crs_matvec_with_k_chunks(basis, nmat, rows, cols, ns):
    # lab_crs_matvec_mfma where nmat * cols exceeds the int32 bound of one GEMM: == lab_crs_expand of the whole matrices + the
    # multiply-add kernel (lab_matvec_sum, one set at a time; the resident int8 GEMM refuses these lengths), bit for bit, and
    # == the oracle's CRS restatement + schoolbook products on sample (set, row) pairs
    import labrador_snark_amd as lab
    q = Q_REF33
    assert nmat * cols > (1 << 17) - 2 and nmat % (((1 << 17) - 2) // cols) != 0
    ctx = lab.Context(q, 0, **ctx_kw(q))
    _digit_basis(ctx, basis, nmat * cols)
    base = bytes((i * 31 + 9) % 256 for i in range(32))
    offs = [555 + m * (rows * cols * D + 13) + ((1 << 65) if m == 2 else 0) for m in range(nmat)]
    hV = ctx.vec(ns * nmat * cols, digit=True).fill_synthetic(91, bound=1625)
    got = lab.crs_matvec_mfma(ctx, base, offs, rows, cols, hV, ns).store().reshape(ns, rows, D)
    ref = ctx
    if basis.startswith("chunk"):
        # the multiply-add reference needs a basis that holds the whole sum: a second context (same q, four digit primes)
        ref = lab.Context(q, 0, **ctx_kw(q))
        ref.enable_digit_basis(1 << 21, 1625, mfma=True)
        hV.close()
        hV = ref.vec(ns * nmat * cols, digit=True).fill_synthetic(91, bound=1625)
    M = ref.vec(nmat * rows * cols, digit=True)
    for m in range(nmat):
        ref._check(ref._lib.lab_crs_expand(ref._h, M._h, m * rows * cols, rows * cols, base, offs[m] & ((1 << 64) - 1), offs[m] >> 64))
    for s in sorted({0, 1, ns // 2, ns - 1}):
        want = lab.matvec_sum(ref, M, hV, nmat, rows, cols, v_off=s * nmat * cols).store()
        assert np.array_equal(got[s], want), (s,)
    K = nmat * cols
    for s, rho in ((ns - 1, rows - 1), (2, 0)):
        v = co.synth_fill(91, s * K * D, K * D, 1625).reshape(nmat, cols, D)
        acc = np.zeros(D, np.uint64)
        for m in range(nmat):
            row = co.crs_fill(base, offs[m] + rho * cols * D, cols * D, q).reshape(cols, D)
            acc = co.poly_add(q, acc, co.inner_product(q, row, v[m]))
        assert np.array_equal(got[s, rho], acc), (s, rho)
    M.close(); hV.close()
    if ref is not ctx:
        ref.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nproofs,chunk_basis", [(2, False), (9, False), (2, True), (9, True), (67, True)])
def test_prover_streamed_with_k_chunks(nproofs, chunk_basis):
    # the streamed prover in the configuration of the kappa = 2^14 bench run, scaled down in kappa1 only: kappa = 16384 columns
    # per B_ik (MC = 7 matrices per K chunk, r*t1 = 9 matrices -> two chunks), A8d in the basis of the CRS.
    #   chunk_basis = False: digit basis sized for 2^21 terms (FOUR primes below 2^23; chunks added mod the limb primes)
    #   chunk_basis = True:  the basis ResidentCRS(with_B=False) asks for by itself — one chunk's length, THREE primes; the chunks
    #                        are reduced mod q one by one (what bench.py's kappa = 2^14 record runs)
    # == the resident CRS on the multiply-add kernels (B as residues, four-prime basis), field by field; u_1 row 0 of proof 0 ==
    # the oracle on its own CRS expansion at the reference's literal offsets (src/structs.rs:78-82); entries of t == the oracle
    import labrador_snark_amd as lab
    import cases
    q, n, r, kappa, kappa1 = Q_REF33, 2, 3, 16384, 3
    ref = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, kappa1)
    assert k.T_1 == 3 and r * k.T_1 * kappa > (1 << 17)
    primes, _, _ = ref.enable_digit_basis(1 << 21, max(k.B_1, k.B_2), mfma=True)
    assert len(primes) == 4
    base = bytes((i * 41 + 3) % 256 for i in range(32))
    full = lab.CRS(ref, k, base).materialize(mfma_basis=True)            # B resident as residues: the multiply-add kernels
    if chunk_basis:
        ctx = lab.Context(q, 0, **ctx_kw(q))
    else:
        ctx = ref
    streamed = lab.CRS(ctx, k, base).materialize(mfma_basis=True, stream_B=True)
    assert len(ctx.digit_primes) == (3 if chunk_basis else 4)
    # A8d, C8, D8 in the CRS basis; B stays a seed; in the bench's configuration no main-basis A8 and no C / D residues either
    streamed.pack_mfma(main_basis_A=not chunk_basis, drop_CD=chunk_basis)
    assert (streamed.C is None) == chunk_basis
    assert streamed.B is None and getattr(streamed, "B8", None) is None and streamed.A8d.nbytes > 0
    ws, sts, tapes = [], [], []
    for j in range(nproofs):
        S, phi, a, tape = cases.make_inputs(p, seed=61 + j)
        ws.append(S); tapes.append(tape)
        sts.append(lab.State(phi, a, int(lab.gen_f_b(ref, S, phi, a, n, r)[0])))
    want, c1 = lab.prove_core_batch(ref, k, ws, sts, full, tapes, raise_on_reject=False)
    got, c2 = lab.prove_core_batch(ctx, k, ws, sts, streamed, tapes, raise_on_reject=False)
    assert c1 == c2 and ctx.prove_flags() & 3 == 1              # step 1 ran in the digit basis (A8d), no fallback
    assert bool(ctx.prove_flags() & 4) == (nproofs > 64)        # the sub-batches shared one pass over the regenerated rows
    for j in range(nproofs):
        for f in ("u_1", "u_2", "t_i_all", "g_mat", "h_mat", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(want[j], f)), (j, f)
    off = lambda which, **kw: co.crs_offset(which, n, r, kappa, kappa1, kappa1, p.t1, **kw)
    T, rho = got[0], 0
    acc = np.zeros(D, np.uint64)
    for i in range(r):
        dec = co.decompose_vec(q, T.t_i_all[i], k.B_1, k.T_1)
        for kk in range(k.T_1):
            b_row = co.crs_fill(base, off("B", i=i, k=kk, row=rho), kappa * D, q).reshape(kappa, D)
            acc = co.poly_add(q, acc, co.inner_product(q, b_row, dec[kk]))
    for (i, j) in [(i, j) for i in range(r) for j in range(i, r)]:
        gdec = co.decompose(q, T.g_mat[i, j], k.B_2, k.T_2)
        for kk in range(k.T_2):
            c_vec = co.crs_fill(base, off("C", i=i, j=j, k=kk), kappa1 * D, q).reshape(kappa1, D)
            acc = co.poly_add(q, acc, co.polymul(q, gdec[kk], c_vec[rho]))
    assert np.array_equal(T.u_1[rho], acc)
    for (i, row) in ((0, 0), (r - 1, kappa - 1), (1, 7777)):
        a_row = co.crs_fill(base, off("A", row=row), n * D, q).reshape(n, D)
        assert np.array_equal(T.t_i_all[i, row], co.inner_product(q, a_row, ws[0][:, i]))
    # a witness above its norm bound: the call falls back to the main basis for step 1 (the multiply-add kernel when A8 is
    # absent) and still equals the resident run
    big = [w.copy() for w in ws]
    big[1][1, 2, 5] = q - 1
    bad = [sts[0], lab.State(sts[1].phi, sts[1].a, 0)] + sts[2:]
    want2, codes_ref = lab.prove_core_batch(ref, k, big, bad, full, tapes, raise_on_reject=False)
    got2, codes = lab.prove_core_batch(ctx, k, big, bad, streamed, tapes, raise_on_reject=False)
    # (more than 64 proofs: the sub-batch with the big witness is redone in place — flag 2 — while the other one keeps its
    # digit-basis run — flag 1: the call reports both)
    assert ctx.prove_flags() & 3 == (2 if nproofs <= 64 else 3) and list(codes) == list(codes_ref)
    for j in range(nproofs):
        for f in ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z"):
            assert np.array_equal(getattr(got2[j], f), getattr(want2[j], f)), (j, f)
    full.close(); streamed.close()
    if ctx is not ref:
        ctx.close()
    ref.close()


@pytest.mark.gpu
def test_streamed_groups_share_one_gemm_launch():
    # 128 proofs = two full sub-batches: their digit tiles sit in the two sub-batches' workspaces (equally spaced) and go through ONE
    # GEMM launch per limb and (chunk, slab) as its two jobs — the slab is read from HBM once for both.  kappa1 = 130 rows (five row
    # tiles: the assembly K loop, ragged), kappa = 64, a digit basis that holds four of the nine B_ik per K chunk (chunks reduced mod q).
    # Every field of every proof == the resident CRS on the multiply-add kernels; u_1 rows of the first and the last proof == the oracle.
    import labrador_snark_amd as lab
    import cases
    q, n, r, kappa, kappa1, nproofs = Q_REF33, 2, 3, 64, 130, 128
    ref = lab.Context(q, 0, **ctx_kw(q))
    ctx = lab.Context(q, 0, **ctx_kw(q))
    p = cases.Params(q, n, r, kappa, kappa1)
    k = lab.RuntimeConstants(q, n, r, kappa, kappa1, kappa1)
    primes, _, _ = ctx.enable_digit_basis(300, max(k.B_1, k.B_2), mfma=True)
    assert len(primes) == 3 and r * k.T_1 * kappa > 300
    base = bytes((i * 53 + 7) % 256 for i in range(32))
    full = lab.CRS(ref, k, base).materialize(mfma_basis=True)
    streamed = lab.CRS(ctx, k, base).materialize(mfma_basis=True, stream_B=True)
    streamed.pack_mfma(main_basis_A=False, drop_CD=True)
    ws, sts, tapes = [], [], []
    for j in range(nproofs):
        S, phi, a, tape = cases.make_inputs(p, seed=4000 + j)
        ws.append(S); tapes.append(tape)
        sts.append(lab.State(phi, a, int(lab.gen_f_b(ref, S, phi, a, n, r)[0])))
    want, c1 = lab.prove_core_batch(ref, k, ws, sts, full, tapes, raise_on_reject=False)
    got, c2 = lab.prove_core_batch(ctx, k, ws, sts, streamed, tapes, raise_on_reject=False)
    assert c1 == c2
    for j in range(nproofs):
        for f in ("u_1", "u_2", "t_i_all", "g_mat", "h_mat", "z", "b_prime_prime", "projection_int"):
            assert np.array_equal(getattr(got[j], f), getattr(want[j], f)), (j, f)
    off = lambda which, **kw: co.crs_offset(which, n, r, kappa, kappa1, kappa1, p.t1, **kw)
    pairs = [(i, j) for i in range(r) for j in range(i, r)]
    for jj, rho in ((0, 0), (nproofs - 1, kappa1 - 1), (64, 128)):
        T = got[jj]
        acc = np.zeros(D, np.uint64)
        for i in range(r):
            dec = co.decompose_vec(q, T.t_i_all[i], k.B_1, k.T_1)
            for kk in range(k.T_1):
                b_row = co.crs_fill(base, off("B", i=i, k=kk, row=rho), kappa * D, q).reshape(kappa, D)
                acc = co.poly_add(q, acc, co.inner_product(q, b_row, dec[kk]))
        for (i, j) in pairs:
            gdec = co.decompose(q, T.g_mat[i, j], k.B_2, k.T_2)
            for kk in range(k.T_2):
                c_vec = co.crs_fill(base, off("C", i=i, j=j, k=kk), kappa1 * D, q).reshape(kappa1, D)
                acc = co.poly_add(q, acc, co.polymul(q, gdec[kk], c_vec[rho]))
        assert np.array_equal(T.u_1[rho], acc), (jj, rho)
    full.close(); streamed.close()
    ctx.close(); ref.close()
