import sys, os; sys.path.insert(0, '.')
import torch, numpy as np, labrador_snark_amd as lab
from oracle import coracle as co
q = (1 << 32) + 15
ctx = lab.Context(q, 0, 1 << 17); ctx.use_torch_stream()
ctx.enable_digit_basis(96 * 1024 + 3000, 1625)
# correctness at a small shape, all nb
rng = np.random.default_rng(0)
for nb in (1, 2, 4, 8):
    nmat, rows, cols = (3, 7, 133) if nb == 1 else (2, 37, 133)
    M = rng.integers(0, q, size=(nmat, rows, cols, 64), dtype=np.uint64)
    V = rng.integers(0, 1625, size=(nb, nmat, cols, 64), dtype=np.uint64)
    for compact in (False, True):
        hM = ctx.vec(nmat * rows * cols, digit=True, compact=compact).load(M)
        hV = ctx.vec(nb * nmat * cols, digit=True).load(V)
        got = lab.matvec_sum_batched(ctx, hM, hV, nmat, rows, cols, nb).store().reshape(nb, rows, 64)
        for b in range(nb):
            want = np.zeros((rows, 64), np.uint64)
            for m in range(nmat):
                for r in range(rows): want[r] = co.poly_add(q, want[r], co.inner_product(q, M[m, r], V[b, m]))
            assert np.array_equal(got[b], want), (nb, compact, b)
print("batched parity ok")
nmat, rows, cols = 48, 1024, 1024
M = ctx.vec(nmat * rows * cols, digit=True, compact=True).fill_synthetic(1)
for nb in (1, 2, 4, 8):
    V = ctx.vec(nb * nmat * cols, digit=True).fill_synthetic(2, bound=1625); out = ctx.vec(nb * rows, digit=True)
    def run(it):
        for _ in range(2): lab.matvec_sum_batched(ctx, M, V, nmat, rows, cols, nb, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): lab.matvec_sum_batched(ctx, M, V, nmat, rows, cols, nb, out=out)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it
    ms = min(run(8) for _ in range(3))
    gb = nmat * rows * cols * 64 * 10 / 1e9
    print("nb=%d  %.3f ms total  %.3f ms per set  matrix stream %.0f GB/s  %.2e mul-adds/s" % (nb, ms, ms / nb, gb / ms * 1e3, nb * nmat * rows * cols * 64 * 3 / ms * 1e3))
    V.close(); out.close()
