#!/usr/bin/env python3
"""Stand-alone rate of the CRS expansion kernel (k_crs_expand: lab_crs_expand_u64 of 2^22 polynomials = 2^28 coefficients per
launch at q = 2^32 + 15), HIP-event timed.  Run directly for the rate; run under
    rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d <dir> -- python3 tools/crs_expand_rate.py
and feed <dir> to tools/crs_expand_valu.py for the vector-instruction count per coefficient (profiles/*_crs_expand_valu.json)."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import labrador_snark_amd as lab

q = (1 << 32) + 15
ctx = lab.Context(q, 0)
ctx.use_torch_stream()
count = 1 << 22
out = torch.empty(count * 64, dtype=torch.int64, device="cuda")
seed = bytes(range(32))
f = lambda: ctx._check(ctx._lib.lab_crs_expand_u64(ctx._h, out.data_ptr(), count, seed, 12345, 0))
f()
torch.cuda.synchronize()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    f()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps / 1e3
print(json.dumps({"kernel": "lab::k_crs_expand", "q": q, "coefficients_per_launch": count * 64, "launch_ms": t * 1e3,
                  "coefficients_per_s": count * 64 / t}))
