#!/usr/bin/env python3
"""Times the int8-MFMA outer commitment (lab_matvec_sum_mfma) at the bench's step-3 size: nmat = 96, 1024 x 1024,
q = 2^32+15, B in the compact digit basis.  usage: mfma_probe.py [ns ...]"""
import sys, time
import numpy as np
import torch
import labrador_snark_amd as lab

q = (1 << 32) + 15
nmat, rows, cols = 96, 1024, 1024
ctx = lab.Context(q, 0, 1 << 17)
ctx.enable_digit_basis(nmat * cols + 2000, 1625)
B = ctx.vec(nmat * rows * cols, digit=True, compact=True).fill_synthetic(1001)
t0 = time.time()
M8 = lab.Mat8(ctx, B, nmat, rows, cols)
ctx.synchronize()
print("pack: %.2f s, %.1f GiB" % (time.time() - t0, M8.nbytes / 2**30), flush=True)
for ns in [int(x) for x in sys.argv[1:]] or [8, 16, 32]:
    V = ctx.vec(ns * nmat * cols, digit=True).fill_synthetic(7, bound=1625)
    out = ctx.vec(ns * rows, digit=True)
    for _ in range(2):
        lab.matvec_sum_mfma(ctx, M8, V, ns, out=out)
    ctx.synchronize()
    t0 = time.time()
    it = 5
    for _ in range(it):
        lab.matvec_sum_mfma(ctx, M8, V, ns, out=out)
    ctx.synchronize()
    ms = (time.time() - t0) / it * 1e3
    print("ns %2d: %.2f ms per call, %.3f ms per set, matrix stream %.2f TB/s" % (ns, ms, ms / ns, M8.nbytes / ms / 1e9), flush=True)
    if ns == 8:   # spot check against the VALU path
        ref = lab.matvec_sum_batched(ctx, B, V, nmat, rows, cols, 8)
        assert np.array_equal(ref.store(), out.store()), "MFMA result differs from the VALU kernel"
        print("ns 8 result == lab_matvec_sum_batched", flush=True)
