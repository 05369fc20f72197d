#!/usr/bin/env python3
"""Times lab_jl_project-style fused Pi pass through lab_prove_core? No: times launch of the fused Pi kernel via
lab_jl_project is not the fused kernel, so this probe runs a batch-1 prove step under the profiling hook is overkill;
instead it times whole lab_prove_core_batch steps and prints ms/proof (cheap A/B for kernel variants)."""
import json, subprocess, sys
for b in sys.argv[1:] or ["8"]:
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extras", "--batch", b, "--steps", "5"],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("batch", b, "proofs/s %.1f ms/proof %.3f" % (d["value"], d["ms_per_proof"]))
