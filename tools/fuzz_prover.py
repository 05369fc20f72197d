"""Developer fuzz (GPU box): random shapes of lab_prove_core_batch against the oracle.

Every iteration draws (q, n, r, kappa, kappa1, number of proofs) and one of the ways the prover can be fed —
multiply-add kernels on unpacked matrices, packed matrices in the main or in the digit basis, B regenerated from the
seed inside the call — and compares every transcript field of every proof with the multiply-add run and the whole
transcript of one proof with the oracle's Prover::proof_gen restatement (src/proofgen.rs:30-427).

    python tools/fuzz_prover.py [seed] [iterations] [large]

`large`: shapes that span several row tiles / k-tiles / row groups of the matrix-core kernels (n up to 70, kappa up to 300,
kappa1 up to 150), bounded so that the oracle's proof of one instance stays within a second or two.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import cases  # noqa: E402
import labrador_snark_amd as lab  # noqa: E402

FIELDS = ("t_i_all", "g_mat", "h_mat", "u_1", "u_2", "z", "b_prime_prime", "projection_int")
ORACLE = {"t_i_all": "t", "g_mat": "G", "h_mat": "H", "u_1": "u1", "u_2": "u2", "z": "z"}


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    large = len(sys.argv) > 3
    rng = np.random.default_rng(seed)
    qs = [(1 << 32) + 15, 1073738753, 4294966657, 1073741953, 8191]   # three limbs; one 30-bit limb; single-limb strict-32 (two); the default Q
    bad = 0
    for it in range(iters):
        q = qs[int(rng.integers(0, len(qs)))]
        n, r = int(rng.integers(1, 7)), int(rng.choice([1, 2, 5, 8, 9, 16, 17, 24, 32, 33, 40]))
        kappa, kappa1 = int(rng.integers(1, 100)), int(rng.integers(1, 40))
        nb = int(rng.choice([1, 2, 3, 4, 5, 8, 66, 70, 130, 260]))
        if large:
            while True:
                n, r = int(rng.integers(1, 71)), int(rng.choice([2, 8, 17, 32]))
                kappa, kappa1 = int(rng.integers(1, 301)), int(rng.integers(1, 151))
                if r * 3 * kappa1 * kappa + kappa * n * r + r * r * n < 1200000:
                    break
            nb = int(rng.choice([1, 2, 3, 5]))
        if nb > 8:                       # several sub-batches: keep the instances small
            n, r, kappa, kappa1 = min(n, 2), min(r, 5), min(kappa, 9), min(kappa1, 6)
        mode = int(rng.integers(0, 4))   # 0: packed, main basis; 1: packed, digit basis; 2: digit basis, no main-basis A; 3: B from the seed
        desc = dict(q=q, n=n, r=r, kappa=kappa, kappa1=kappa1, nb=nb, mode=mode)
        try:
            ctx = lab.Context(q, 0)
            if ctx.wide and mode != 3:
                mode = 1                 # single-limb strict-32 contexts take no main-basis packing
            p = cases.Params(q, n, r, kappa, kappa1)
            k = lab.RuntimeConstants(q, n, r, kappa, kappa1, p.kappa2)
            ws, sts, tapes = [], [], []
            for j in range(nb):
                S, phi, a, tape = cases.make_inputs(p, seed=7000 + 100 * it + j)
                ws.append(S); tapes.append(tape)
                sts.append(lab.State(phi, a, int(lab.gen_f_b(ctx, S, phi, a, n, r)[0])))
            if mode == 3:
                base = bytes((i * 29 + it) % 256 for i in range(32))
                crs0 = lab.CRS(ctx, k, base)
                full = crs0.materialize(mfma_basis=True)
                ref, c_ref = lab.prove_core_batch(ctx, k, ws, sts, full, tapes, raise_on_reject=False)
                crs = crs0.materialize(mfma_basis=True, stream_B=True)
                got, c_got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
                full.close()
                want = None
            else:
                crs = lab.ResidentCRS(ctx, k, compact=True, mfma_basis=mode >= 1).fill_synthetic(1000)
                ref, c_ref = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
                crs.pack_mfma(main_basis_A=mode != 2) if mode >= 1 else crs.pack_mfma()
                got, c_got = lab.prove_core_batch(ctx, k, ws, sts, crs, tapes, raise_on_reject=False)
                case, _ = cases.oracle_case(p, seed=7000 + 100 * it + nb - 1)
                want = case.prove() if c_got[-1] == 0 else None
            ok = c_ref == c_got
            for j in range(nb):
                if c_got[j]:
                    continue
                for f in FIELDS:
                    if not np.array_equal(getattr(got[j], f), getattr(ref[j], f)):
                        ok = False
                        print("  field", f, "of proof", j, "differs from the multiply-add run")
            if want is not None:
                for f, key in ORACLE.items():
                    if key in want and not np.array_equal(getattr(got[-1], f), want[key]):
                        ok = False
                        print("  field", f, "of the last proof differs from the oracle")
            crs.close()
            ctx.close()
        except lab.LabradorError as e:   # a shape the boundary refuses (stated in its message) is not a mismatch
            print("refused", desc, str(e)[:120])
            continue
        print("ok      " if ok else "MISMATCH", desc, flush=True)
        bad += 0 if ok else 1
    print("fuzz done, mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
