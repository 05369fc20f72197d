"""Developer fuzz: random shapes through the C ABI vs the oracle (tail handling of every tiled kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import labrador_snark_amd as lab
from oracle import coracle as co
D = 64
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
qs = [8191, (1 << 32) + 15, 1073738753, 4294966657, 2147483777]    # the last two: single-limb strict-32 path (k_wide_dot)
ctxs = {q: lab.Context(q, 0) for q in qs}
def rp(q, *shape): return rng.integers(0, q, size=shape + (D,), dtype=np.uint64)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    q = qs[it % len(qs)]; ctx = ctxs[q]
    kind = (it // len(qs)) % 5
    if kind == 0:
        nmat, rows, cols = int(rng.integers(1, 5)), int(rng.integers(1, 19)), int(rng.integers(1, 300))
        M, V = rp(q, nmat, rows, cols), rp(q, nmat, cols)
        got = lab.matvec_sum(ctx, ctx.vec_from(M), ctx.vec_from(V), nmat, rows, cols).store()
        want = np.zeros((rows, D), np.uint64)
        for m in range(nmat):
            for r in range(rows): want[r] = co.poly_add(q, want[r], co.inner_product(q, M[m, r], V[m]))
        ok = np.array_equal(got, want); desc = ("matvec", nmat, rows, cols)
    elif kind == 1:
        rows, inner, ncols = int(rng.integers(1, 14)), int(rng.integers(1, 90)), int(rng.integers(1, 40))
        A, S = rp(q, rows, inner), rp(q, inner, ncols)
        got = lab.ajtai_commit(ctx, ctx.vec_from(A), ctx.vec_from(S), rows, inner, ncols).store().reshape(ncols, rows, D)
        ok = all(np.array_equal(got[i, rho], co.inner_product(q, A[rho], np.ascontiguousarray(S[:, i])))
                 for i in range(ncols) for rho in range(rows)); desc = ("ajtai", rows, inner, ncols)
    elif kind == 2:
        n, r = int(rng.integers(1, 200)), int(rng.integers(1, 12))
        P, S = rp(q, n, r), rp(q, n, r)
        got = lab.gram(ctx, ctx.vec_from(P), ctx.vec_from(S), n, r).store().reshape(r, r, D)
        ok = all(np.array_equal(got[i, k], co.inner_product(q, np.ascontiguousarray(P[:, i]), np.ascontiguousarray(S[:, k])))
                 for i in range(r) for k in range(r)); desc = ("gram", n, r)
    elif kind == 3:
        nterms, length = int(rng.integers(1, 200)), int(rng.integers(1, 50))
        G, C = rp(q, nterms), rp(q, nterms, length)
        got = lab.axpy_commit(ctx, ctx.vec_from(G), ctx.vec_from(C), nterms, length).store()
        want = np.zeros((length, D), np.uint64)
        for m in range(nterms):
            pr = co.poly_by_poly_vec(q, G[m], C[m])
            for rho in range(length): want[rho] = co.poly_add(q, want[rho], pr[rho])
        ok = np.array_equal(got, want); desc = ("axpy", nterms, length)
    else:
        n, r = int(rng.integers(1, 9)), int(rng.integers(1, 6))
        S = rp(q, n, r)
        Pi = rng.choice(np.array([-1, 0, 1], dtype=np.int8), size=(r, 256, n * D)).astype(np.int8)
        got = lab.jl_project(ctx, Pi, S, n, r)
        ok = np.array_equal(got, co.jl_project(Pi, np.ascontiguousarray(S.transpose(1, 0, 2).reshape(r, n * D)))); desc = ("jl", n, r)
    if not ok:
        bad += 1
        print("MISMATCH", q, desc)
print("fuzz done, mismatches:", bad)
sys.exit(1 if bad else 0)
