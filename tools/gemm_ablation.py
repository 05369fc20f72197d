#!/usr/bin/env python3
"""Load side / multiply side / in-kernel clock of the dominant kernel (k_gemm_i8_w3<6>, the outer commitment of the
bench workload), through lab_ctx_gemm_diag (include/labrador_hip.h).  Runs on the GPU box:

    python tools/gemm_ablation.py [--steps 3] > gpurun_out/gemm_ablation.json

For each mode the bench step (64 proofs, n=1024, r=32, kappa=1024) is run `steps` times with the measurement hook on, so
the launch time is the one bench.py's roofline object reports; in modes 1-3 every block also records the ticks its K
loop took on the shader clock (s_memtime) and on the constant 100 MHz clock (s_memrealtime): their ratio is the shader
clock the matrix instructions actually ran at.
  mode 0  the product kernel
  mode 1  the same loop with the two clock reads around it
  mode 2  the loop WITHOUT global loads / LDS-DMA   -> the multiply side alone (matrix instructions, LDS reads, barriers)
  mode 3  the loop WITHOUT matrix instructions / LDS reads -> the load side alone (HBM stream, LDS-DMA, barriers)
If the two sides overlap, t(mode 0) ~ max(t(2), t(3)); if they serialise (round 1), t(0) ~ t(2) + t(3).
The proofs computed in modes 2 and 3 are meaningless and are discarded.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--kappa", type=int, default=1024)
    args = ap.parse_args()
    import torch
    import bench
    import labrador_snark_amd as lab

    q, n, r = bench.Q_REF33, 1024, 32
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    k = lab.RuntimeConstants(q, n, r, args.kappa, args.kappa, args.kappa)
    ctx = lab.Context(q, 0, max(65536, n * r + r * r))
    ctx.use_torch_stream()
    dSs, sts, dtapes, crs = bench.build_workload(lab, ctx, k, q, dev, 64, 50, 1000, use_mfma=True, basis23=True, compact=True,
                                                 packed_pi=True, stream_b=False)

    def step():
        return lab.prove_core_batch(ctx, k, dSs, sts, crs, dtapes, device_outputs=True)[-1]

    step()
    torch.cuda.synchronize()
    ticks = torch.zeros(2 * 64 * 8 * 4, dtype=torch.int64, device=dev)      # 64 slots x 8 row groups, 2 words each (+ slack)
    names = {0: "product kernel", 1: "product loop + clock reads", 2: "multiply side alone (no global loads / LDS-DMA)",
             3: "load side alone (no matrix instructions / LDS reads)"}
    out = {"workload": "bench.py default step (64 proofs, n=1024 r=32 kappa=%d): the 3 limb launches of the outer commitment" % args.kappa,
           "steps": args.steps, "modes": {}}
    for mode in (0, 1, 2, 3, 0):
        ticks.zero_()
        ctx.gemm_diag(mode, ticks if mode else None)
        step()
        torch.cuda.synchronize()
        ctx.profile_begin()
        for _ in range(args.steps):
            step()
        ms, launches, _ = ctx.profile_end()
        rec = {"what": names[mode], "avg_launch_ms": ms / max(launches, 1), "launches": launches, "kernel": ctx.profile_kernel()}
        if mode:
            t = ticks.cpu().numpy().reshape(-1, 2)
            t = t[t[:, 1] > 0]
            if len(t):
                shader, real = t[:, 0].astype(float), t[:, 1].astype(float)
                rec.update({"blocks_recorded": int(len(t)), "loop_shader_ticks_median": float(sorted(shader)[len(t) // 2]),
                            "loop_const_ticks_median": float(sorted(real)[len(t) // 2]),
                            "shader_clock_GHz_at_100MHz_const": float((shader / real).mean() * 0.1),
                            "loop_us_median_at_100MHz_const": float(sorted(real)[len(t) // 2] / 100.0)})
        key = "mode%d" % mode if ("mode%d" % mode) not in out["modes"] else "mode%d_again" % mode
        out["modes"][key] = rec
    ctx.gemm_diag(0, None)
    m = out["modes"]
    full, mul, ld = m["mode0"]["avg_launch_ms"], m["mode2"]["avg_launch_ms"], m["mode3"]["avg_launch_ms"]
    out["summary"] = {"product_ms": full, "multiply_side_ms": mul, "load_side_ms": ld, "max_of_sides_ms": max(mul, ld),
                      "sum_of_sides_ms": mul + ld, "overlap": (mul + ld - full) / min(mul, ld) if min(mul, ld) > 0 else None,
                      "note": "overlap = (t_mul + t_load - t_product) / min(t_mul, t_load): 1 = the shorter side is hidden completely, 0 = the sides serialise"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
