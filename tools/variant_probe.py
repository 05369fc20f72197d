"""Developer A/B probe: times the step-3 mat-vec shape with alternative builds of the library (tools/variants/*.so),
interleaved rounds in one process per variant is not possible (one .so per process), so each variant runs in its
own subprocess on the same device, three rounds each, alternating."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import torch
import labrador_snark_amd._lib as L
L.SO_PATH = sys.argv[1]
import labrador_snark_amd as lab
q = (1 << 32) + 15
ctx = lab.Context(q, 0, 1 << 17); ctx.use_torch_stream()
nmat, rows, cols = 48, 1024, 1024
M = ctx.vec(nmat * rows * cols).fill_synthetic(1); V = ctx.vec(nmat * cols).fill_synthetic(2, bound=1625); out = ctx.vec(rows)
def run(it):
    for _ in range(3): lab.matvec_sum(ctx, M, V, nmat, rows, cols, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): lab.matvec_sum(ctx, M, V, nmat, rows, cols, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
ms = min(run(10) for _ in range(3))
print("%%s %%.4f ms %%.1f GB/s" %% (os.path.basename(sys.argv[1]), ms, ctx.nlimbs * nmat * rows * cols * 256 / ms / 1e6))
''' % ROOT
variants = sorted(f for f in os.listdir(os.path.join(HERE, "variants")) if f.endswith(".so"))
for rnd in range(2):
    for v in variants:
        r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(HERE, "variants", v)], capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-300:])
