"""Developer perf probe (not the judged bench): times the NTT and mat-vec kernels with HIP events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import labrador_snark_amd as lab

D = 64

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def main():
    which = sys.argv[1:] or ["ntt", "mv"]
    for q in (1073738753, (1 << 32) + 15):
        ctx = lab.Context(q, 0, 1 << 17)
        ctx.use_torch_stream()
        L = ctx.nlimbs
        print("== q=%d nlimbs=%d" % (q, L))
        if "ntt" in which:
            for batch in (4096, 1 << 16, 1 << 20):
                x = torch.randint(0, 1 << 29, (L, batch, D), dtype=torch.int32, device="cuda")
                tf = timeit(lambda: ctx.ntt_forward(x, batch))
                ti = timeit(lambda: ctx.ntt_inverse(x, batch))
                byt = L * batch * 512
                print("ntt batch=%8d fwd %8.1f us %7.1f GB/s | inv %8.1f us %7.1f GB/s | %.3e NTT/s(fwd, per limb-plane)" % (
                    batch, tf * 1e6, byt / tf / 1e9, ti * 1e6, byt / ti / 1e9, L * batch / tf))
                del x
        if "mv" in which:
            for (nmat, rows, cols) in ((1, 1 << 14, 1 << 10), (96, 1024, 1024) if L == 1 else (32, 1024, 1024), (8, 4096, 64)):
                M = ctx.vec(nmat * rows * cols); V = ctx.vec(nmat * cols); out = ctx.vec(rows)
                M.fill_synthetic(1); V.fill_synthetic(2)
                t = timeit(lambda: lab.matvec_sum(ctx, M, V, nmat, rows, cols, out=out), iters=10)
                byt = L * nmat * rows * cols * 256
                print("matvec nmat=%3d rows=%6d cols=%6d  %9.1f us  %7.1f GB/s (matrix bytes %.2f GiB)" % (
                    nmat, rows, cols, t * 1e6, byt / t / 1e9, byt / 2**30))
                M.close(); V.close(); out.close()
        ctx.close()

main()
