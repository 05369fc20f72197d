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
L_ = ctx.nlimbs; batch = 1 << 20
x = torch.randint(0, 1 << 29, (L_, batch, 64), dtype=torch.int32, device="cuda")
def run(fn, it):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
f = min(run(lambda: ctx.ntt_forward(x, batch), 20) for _ in range(3))
i = min(run(lambda: ctx.ntt_inverse(x, batch), 20) for _ in range(3))
print("%%s fwd %%.1f us %%.0f GB/s | inv %%.1f us %%.0f GB/s" %% (os.path.basename(sys.argv[1]), f * 1e3, L_ * batch * 512 / f / 1e6, i * 1e3, L_ * batch * 512 / i / 1e6))
''' % ROOT
variants = sorted(f for f in os.listdir(os.path.join(HERE, "variants")) if f.endswith(".so"))
for rnd in range(2):
    for v in variants:
        r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(HERE, "variants", v)], capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-300:])
