#!/usr/bin/env python3
"""bench.py — prover throughput on MI355X (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[2]): one full Prover::proof_gen
(src/proofgen.rs:30-427) per step on a synthetic instance n=1024, r=32, d=64 with commitment ranks
kappa = kappa1 = kappa2 = 1024 (stated explicitly: the reference hard-wires n*64 = 65536, which makes each
B_ik a 1 TiB matrix).  Witness, state, challenge tape and the CRS matrices (transform domain) are resident
in HBM before the timed region; for the headline the CRS expansion from a seed (src/structs.rs:55-171) is NOT in the timed
region.  The same prover at the contract's kappa = 2^14 — where B cannot be resident and is regenerated from the ChaCha20 seed
INSIDE the step — is measured right after as extra.config3_kappa16384_streamed (one step; --crs streamed runs that mode as
the headline).  The modulus is the reference's "32-bit" choice
q = 2^32+15 (src/constants.rs:196; 33 bits, not NTT-friendly -> exact 3-limb RNS path).

metric  = prover constraints/sec  (K = 1 constraint per proof, src/constants.rs:26)  => proofs/sec.
N > 1   = N independent instances, one per GPU (weak scaling, BASELINE configs[4]); the only collective is one
          RCCL all-reduce of the folded (u_1, u_2) commitment vector (plus a count of the ranks) after the timed steps.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

D = 64
Q_REF33 = (1 << 32) + 15
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def ring_products_per_proof(n, r, kappa, kappa1, kappa2, t1, t2):
    """Counts of `&v1[i] * &v2[i]` in proof_gen, per step."""
    pairs = r * (r + 1) // 2
    return {"1_inner_ajtai": kappa * r * n, "2_gram": r * r * n, "3_outer_B": r * t1 * kappa1 * kappa,
            "4_outer_C": pairs * t2 * kappa2, "6_aggregate": r * r * n + r * r + r * n, "7_phi": 2 * r * n,
            "8_H": 2 * r * r * n, "9_u2": pairs * t1 * kappa2, "10_z": r * n}


def cpu_share():
    """Threads this process may keep busy: the affinity mask, cut to the cgroup CPU quota when there is one and to 16 — the
    CPU share of a one-GPU box on this pool (its 256 logical CPUs are shared with the jobs of the other 7 GPUs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("LAB_CPU_THREADS", "16"))))


def cpu_baseline_b0(q, products_per_proof, seconds=6.0):
    """B0, reference-faithful CPU port (oracle orc_polymul_faithful: i128 `%` after every coefficient op, like
    src/algebraic.rs:30-38,95-134) timed on this host on a bounded sample of ring products, all threads
    busy; proofs/s is extrapolated from the measured products/s and the per-proof product count."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import coracle as co
    lib = co.lib()
    cores = cpu_share()
    rng = np.random.default_rng(123)
    length = 64
    data = [(rng.integers(0, q, size=(length, D), dtype=np.uint64), rng.integers(0, q, size=(length, D), dtype=np.uint64))
            for _ in range(cores)]
    with ThreadPoolExecutor(cores) as ex:      # calibrate with every thread busy (one call each)
        t0 = time.perf_counter()
        list(ex.map(lambda i: lib.orc_bench_inner_product_faithful(q, data[i][0], data[i][1], length, 1), range(cores)))
        per_call = max(time.perf_counter() - t0, 1e-6)
    reps = max(1, int(seconds / per_call))

    def work(i):
        return lib.orc_bench_inner_product_faithful(q, data[i][0], data[i][1], length, reps)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    products = cores * reps * length
    rate = products / dt
    return {"value": rate / products_per_proof, "unit": "constraints/s", "cores": cores, "kind": "port", "products_per_s": rate,
            "extrapolated": True,
            "sample": "%d reference-faithful ring products (schoolbook 64x64, i128 %% per op + X^64+1 fold + accumulate) "
                      "on %d threads in %.1f s; proofs/s = products/s / %d products per proof" % (products, cores, dt, products_per_proof)}


def cpu_baseline_b1(k, q, proofs=3):
    """B1, the honest-fast CPU prover (oracle/fastcpu.c: RNS/NTT residues, transform-domain accumulation, phi never
    formed, OpenMP over output rows, CRS resident in the transform domain like on the GPU), measured DIRECTLY at the
    bench's sizes: `proofs` full proof_gen calls after one warm-up, all host threads.  The 6.4e9-coefficient B is filled
    with uniform residues directly in the transform domain (a uniform matrix is uniform in either domain; transforming
    1e8 polynomials on the host would take longer than the measurement); A, C, D come from the shared synthetic streams."""
    from oracle import coracle as co
    from oracle import fastcpu as fc
    n, r, kappa = k.N, k.R, k.KAPPA
    pairs = r * (r + 1) // 2
    f = fc.FastCpu(q, max(65536, n * r + r * r), cpu_share())
    t0 = time.perf_counter()
    A = f.synthetic_hat(1000, kappa * n)
    B = f.random_hat(1001, r * k.T_1 * k.KAPPA_1 * kappa)
    Cm = f.synthetic_hat(1002, pairs * k.T_2 * k.KAPPA_2)
    Dm = f.synthetic_hat(1003, pairs * k.T_1 * k.KAPPA_2)
    t_fill = time.perf_counter() - t0
    S, phi, a, tape = make_inputs(k, q, seed=50)
    dummy = np.zeros((1, D), np.uint64)
    case = co.ProverCase(q, n, r, kappa, k.KAPPA_1, k.KAPPA_2, k.B_1, k.T_1, k.B_2, k.T_2, k.BETA_BOUND, S, phi, a, dummy, dummy, dummy,
                         dummy, tape["Pi"], tape["psi"], tape["omega"], tape["alpha"], tape["beta"], tape["c"])
    f.prove(case, A, B, Cm, Dm)          # warm-up (first touch of the scratch allocations)
    t0 = time.perf_counter()
    for _ in range(proofs):
        out = f.prove(case, A, B, Cm, Dm)
    dt = time.perf_counter() - t0
    ok = bool(out["bpp_ok"])
    threads, limbs = f.threads, f.L
    f.close()
    del A, B, Cm, Dm
    return {"value": proofs / dt, "unit": "constraints/s", "cores": threads, "kind": "port", "extrapolated": False,
            "s_per_proof": dt / proofs, "rns_limbs": limbs, "crs_fill_s": t_fill, "bpp_check_ok": ok,
            "sample": "%d full proof_gen calls (n=%d r=%d kappa=kappa1=kappa2=%d, q=%d) of the honest-fast CPU prover "
                      "(oracle/fastcpu.c: %d-limb RNS/NTT, transform-domain accumulation, OpenMP) on %d threads in %.1f s, "
                      "CRS resident in the transform domain" % (proofs, n, r, kappa, q, limbs, threads, dt)}


def cpu_config1(samples=1000):
    """BASELINE configs[0] (benches/ntt_speed.rs:51-81 restated): one pair of uniform polynomials, a single ring product
    timed `samples` times per arm, median ns — classic (i128 `%` per op, schoolbook + fold) vs through the NTT — at the
    reference's default Q, its commented-out "32-bit" Q and the largest NTT-friendly prime below 2^32."""
    from oracle import fastcpu as fc
    out = {}
    for name, qq in (("Q8191", 8191), ("Q2p32p15", (1 << 32) + 15), ("Q4294966657_ntt_friendly", 4294966657)):
        out[name] = {"q": qq, "classic_median_ns": fc.bench_polymul(qq, "classic", samples), "ntt_median_ns": fc.bench_polymul(qq, "ntt", samples),
                     "samples": samples}
    return out


def cpu_baseline(k, q, products_per_proof):
    """The `cpu_baseline` object of the bench line: `value` is B1 (measured directly, the honest comparison); B0 (the
    reference-faithful port, extrapolated) and config 1's single-product medians ride along."""
    b1 = cpu_baseline_b1(k, q)
    b0 = cpu_baseline_b0(q, products_per_proof)
    res = dict(b1)
    res["host_cpus"] = os.cpu_count()
    res["b1_honest_fast"] = {kk: b1[kk] for kk in ("value", "s_per_proof", "cores", "rns_limbs", "crs_fill_s", "extrapolated")}
    res["b0_reference_faithful"] = b0
    res["config1_single_product"] = cpu_config1()
    # SURVEY section 8(d) "q: report both": the same CPU prover at the NTT-friendly 32-bit q (one limb, like the GPU's path (b))
    try:
        import labrador_snark_amd as lab
        q32 = 4294966657
        b32 = cpu_baseline_b1(lab.RuntimeConstants(q32, k.N, k.R, k.KAPPA, k.KAPPA_1, k.KAPPA_2), q32)
        res["b1_q4294966657_single_limb"] = {kk: b32[kk] for kk in ("value", "s_per_proof", "cores", "rns_limbs", "crs_fill_s", "bpp_check_ok")}
    except Exception as e:
        res["b1_q4294966657_single_limb"] = {"error": repr(e)}
    return res


def challenge_poly(rng, q):
    """fetch_challenge's multiset (src/verification.rs:460-479): 23 zeros, 31 ones, 10 twos, random order and sign."""
    vals = np.array([0] * 23 + [1] * 31 + [2] * 10, dtype=np.int64)
    rng.shuffle(vals)
    signs = rng.integers(0, 2, size=D)
    return np.where((vals > 0) & (signs == 1), q - vals, vals).astype(np.uint64)


def make_inputs(k, q, seed, dev=None, packed_pi=False):
    """Seeded synthetic witness / state / challenge tape (same recipe as tests/cases.py, without the oracle):
    witness coefficients uniform in [0, beta/sqrt(n r d)] so the norm bound of generate_witness holds
    (src/proofgen.rs:460-518), phi and a uniform (src/structs.rs:289-314), Pi in {-1,0,1} with weights
    1/4,1/2,1/4 (src/verification.rs:553-566), psi, omega, alpha, beta uniform, c_i challenge-shaped."""
    rng = np.random.default_rng(seed)
    n, r = k.N, k.R
    wb = max(1, int(k.BETA_BOUND / math.sqrt(n * r * D)))
    S = rng.integers(0, wb + 1, size=(n, r, D), dtype=np.uint64)
    phi = rng.integers(0, q, size=(n, r, D), dtype=np.uint64)
    a = rng.integers(0, q, size=(r, r, D), dtype=np.uint64)
    for i in range(r):
        for j in range(i):
            a[i, j] = a[j, i]
    if dev is not None:
        import torch
        # the 512 MiB projection tape is drawn on the device (numpy needs seconds per proof for it): two fair bits
        # per entry, {00 -> -1, 01/10 -> 0, 11 -> 1}
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        bits = torch.randint(0, 4, (r, 256, n * D), dtype=torch.int8, device=dev, generator=g)
        Pi = (bits == 3).to(torch.int8) - (bits == 0).to(torch.int8)
        del bits
        if packed_pi:
            # 2 bits per entry in the tiled order lab_prove_input.Pi2 wants (the same arithmetic as lab.pack_pi_host, on the device)
            f = (Pi + 1).to(torch.uint8).reshape(r, 4, 4, 4, 4, n * D // 16, 16)                 # [i, s, kq, j, k, ct, cc]
            b = f[:, :, :, 0] | (f[:, :, :, 1] << 2) | (f[:, :, :, 2] << 4) | (f[:, :, :, 3] << 6)
            Pi = b.permute(0, 1, 4, 2, 5, 3).contiguous().reshape(r, 256 * n * D // 4)
            del f, b
    else:
        bits = rng.integers(0, 4, size=(r, 256, n * D), dtype=np.int8)       # two fair bits per entry, as on the device
        Pi = (bits == 3).astype(np.int8) - (bits == 0).astype(np.int8)
        del bits
    if packed_pi and dev is None:
        from labrador_snark_amd import pack_pi_host
        Pi = pack_pi_host(Pi)
    tape = dict(psi=int(rng.integers(0, q)), omega=rng.integers(0, q, size=256, dtype=np.uint64),
                alpha=rng.integers(0, q, size=D, dtype=np.uint64), beta=rng.integers(0, q, size=D, dtype=np.uint64),
                c=np.stack([challenge_poly(rng, q) for _ in range(r)]))
    tape["Pi2" if packed_pi else "Pi"] = Pi
    return S, phi, a, tape


def folded_commitments(T):
    """(u_1, u_2) of one transcript folded over their rows: int64[128] = 64 coefficient-wise sums of u_1 | 64 of u_2, each mod 2^62
    (the aggregate the ranks all-reduce is a VECTOR the host reads as one buffer)."""
    import torch
    m = (1 << 62) - 1
    return torch.cat([T.u_1.to(torch.int64).sum(dim=0) & m, T.u_2.to(torch.int64).sum(dim=0) & m])


def aggregate_over_ranks(dist, vec, elapsed, dev):
    """After the timed steps: one all-reduce(SUM) of the folded commitment vector, one of a tensor of ones (how many ranks the
    collective really saw) and one MAX of the step time.  dist is None for a single process."""
    import torch
    seen = torch.ones(1, dtype=torch.int64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    backend = None
    if dist is not None:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        backend = dist.get_backend()
    host = vec.cpu()                      # ONE buffer read by the host
    return {"ranks_seen": int(seen.item()), "backend": backend, "vector": [int(x) for x in host.tolist()],
            "scalar": int(sum(int(x) for x in host.tolist()) & ((1 << 62) - 1)), "tmax": float(tmax.item())}


def checksum_u64(t):
    """Order-independent 64-bit checksum of a device tensor of residues (sum mod 2^63)."""
    import torch
    return (t.to(torch.int64).sum() & ((1 << 62) - 1)).reshape(1)


def build_inputs(lab, ctx, k, q, dev, nbatch, seed0, packed_pi=True):
    """`nbatch` instances (seeds seed0 .. seed0+nbatch-1): witness, state, tape as device tensors.  They do not depend on the
    commitment ranks kappa, so the kappa = 2^14 record reuses the headline's."""
    import torch
    n, r = k.N, k.R

    def dev64(x):
        return torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev)

    dSs, sts, dtapes = [], [], []
    for j in range(nbatch):
        S, phi, a, tape = make_inputs(k, q, seed=seed0 + j, dev=dev, packed_pi=packed_pi)
        # State::gen_f's b (src/structs.rs:320-341, 373) on the product's own kernels: b' = its constant term
        b_poly = lab.gen_f_b(ctx, S, phi, a, n, r)
        dSs.append(dev64(S))
        sts.append(lab.State(dev64(phi), dev64(a), int(b_poly[0])))
        pikey = "Pi2" if packed_pi else "Pi"
        dtapes.append({pikey: tape[pikey] if torch.is_tensor(tape[pikey]) else torch.from_numpy(tape[pikey]).to(dev),
                       "psi": tape["psi"], "omega": dev64(tape["omega"]),
                       "alpha": dev64(tape["alpha"]), "beta": dev64(tape["beta"]), "c": dev64(tape["c"])})
    return dSs, sts, dtapes


def build_crs(lab, ctx, k, crs_seed, use_mfma=True, basis23=True, compact=True, stream_b=False):
    import torch
    if stream_b:
        # the reference's own CRS: A, C, D expanded from the 32-byte seed once (resident), the B_ik rows regenerated from it
        # inside every step (lab_prove_input.crs_seed) — B is never materialised, so its expansion IS in the timed region
        seed = bytes((crs_seed * 7 + 13 * i) % 256 for i in range(32))
        crs = lab.CRS(ctx, k, seed).materialize(mfma_basis=True, stream_B=True)
        # (no main-basis A8: step 1 runs from A8d; a witness above its norm bound would fall back to the multiply-add kernel)
        # (... and no C / D residues: the prover reads their packed forms only)
        crs.pack_mfma(main_basis_A=False, drop_CD=True)
        torch.cuda.synchronize()
        return crs
    # B, C, D are only ever multiplied by decomposition digits: keep them in the digit basis
    crs = lab.ResidentCRS(ctx, k, compact=compact, mfma_basis=basis23).fill_synthetic(crs_seed)
    if use_mfma:
        crs.pack_mfma(drop_B=True)     # one-time, like the CRS expansion: outside every timed region
    torch.cuda.synchronize()
    return crs


def build_workload(lab, ctx, k, q, dev, nbatch, seed0, crs_seed, use_mfma=True, basis23=True, compact=True, packed_pi=True,
                   stream_b=False):
    """The resident inputs of one bench step: `nbatch` instances (seeds seed0 .. seed0+nbatch-1; witness, state, tape as
    device tensors) and the CRS they share.  tests/test_gpu_bench_config.py builds the same objects and checks the
    transcripts lab_prove_core_batch returns for them against the oracle."""
    dSs, sts, dtapes = build_inputs(lab, ctx, k, q, dev, nbatch, seed0, packed_pi=packed_pi)
    crs = build_crs(lab, ctx, k, crs_seed, use_mfma=use_mfma, basis23=basis23, compact=compact, stream_b=stream_b)
    return dSs, sts, dtapes, crs


VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12   # MI355X_MICROARCH.md: 256 CUs x 4 SIMD16 at 2.4 GHz = 39.3e12 32-bit lane operations per second


def expand_valu_per_coefficient():
    """Vector instructions k_crs_expand issues per 64 coefficients (one wave instruction = 64 lane operations), from the committed
    rocprofv3 --pmc SQ_INSTS_VALU pass (profiles/*_crs_expand_valu.json); None when no such record is tracked."""
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
        if name.endswith("_crs_expand_valu.json"):
            try:
                rec = json.load(open(os.path.join(pdir, name)))
                return float(rec["valu_instructions_per_coefficient_lane"]), "profiles/" + name
            except Exception:
                pass
    return None, None


def streamed_roofline(phases, elapsed, steps):
    """The roofline object of a --crs streamed run: the dominant kernel is the CRS expansion (k_crs_expand, ChaCha20 on the
    vector ALUs), timed live with HIP events around a uniform sample of its launches (lab_ctx_profile_phase); its bound is the
    32-bit integer issue rate, not HBM or the matrix cores.  achieved = coefficients/s of the sampled launches x the vector
    instructions per coefficient of the committed SQ_INSTS_VALU pass."""
    ex = phases["crs_expand"]
    per, src = expand_valu_per_coefficient()
    rate = ex["sampled_units"] / (ex["sampled_ms"] / 1e3) if ex["sampled_ms"] > 0 else 0.0
    out = {"bound": "valu", "kernel": "lab::k_crs_expand — CRS::fetch_B_ik_row's coefficients (src/structs.rs:147-171) regenerated inside step 3: one ChaCha20 "
                                        "block + rand's gen_range per coefficient",
           "unit": "TOP/s (32-bit integer lane operations)", "peak": VALU_PEAK_TOPS,
           "coefficients_per_s": rate, "valu_instructions_per_coefficient": per, "valu_count_source": src,
           "achieved": rate * per / 1e12 if per else None, "frac": rate * per / 1e12 / VALU_PEAK_TOPS if per else None,
           "traffic": None, "launches": ex["launches"], "sampled_launches": ex["sampled_launches"],
           "avg_launch_ms": ex["sampled_ms"] / max(ex["sampled_launches"], 1)}
    # where the step goes: phase time = sampled time scaled to all launches of the phase (launches are of equal size)
    shares = {}
    for name, ph in phases.items():
        if ph["sampled_launches"]:
            total_ms = ph["sampled_ms"] * (ph["units"] / ph["sampled_units"] if ph["sampled_units"] else ph["launches"] / ph["sampled_launches"])
            shares[name] = {"ms_per_step": total_ms / steps, "share_of_step": total_ms / 1e3 / elapsed, "launches_per_step": ph["launches"] / steps,
                            "units_per_s_inside_phase": ph["sampled_units"] / (ph["sampled_ms"] / 1e3) if ph["sampled_ms"] > 0 else None}
    out["phases"] = shares
    out["end_to_end_over_expand_alone"] = (phases["crs_expand"]["units"] / elapsed) / rate if rate else None
    return out


def host_chacha_rate(q, seconds=4.0):
    """CRS coefficients per second the host regenerates (oracle orc_crs_fill: one ChaCha20 key set-up + block + gen_range per
    coefficient, scalar C) with every thread of this process's CPU share busy."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import coracle as co
    cores = cpu_share()
    base = bytes(range(32))
    n0 = 1 << 18
    t0 = time.perf_counter()
    co.crs_fill(base, 0, n0, q)
    per = max(time.perf_counter() - t0, 1e-4)
    reps = max(1, int(seconds / per))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda i: [co.crs_fill(base, (i * reps + j) * n0, n0, q) for j in range(reps)], range(cores)))
    dt = time.perf_counter() - t0
    return cores * reps * n0 / dt, cores, dt


def config3_streamed(lab, dev, q, n, r, kappa, inputs, b1_s_per_proof_k1024=None, with_cpu=True, proofs=(256, 224, 192, 160, 128, 96, 64)):
    """SURVEY section 8(d) config 3 at the contract's commitment rank: kappa = kappa1 = kappa2 = 2^14 (the floor BASELINE.md names;
    the reference hard-wires n*d = 2^16, src/constants.rs:237-239).  One B_ik is 2^28 ring elements, the 96 of them 6 TiB as byte
    planes: B is STREAMED — regenerated from the ChaCha20 seed inside the step (CRS::fetch_B_ik_row, src/structs.rs:74-88,
    147-171) — so the expansion is inside the timed region.  ONE step, no warm-up (a step is ~80 s), all proofs of the step share
    one pass over the regenerated rows."""
    import torch
    k = lab.RuntimeConstants(q, n, r, kappa, kappa, kappa)
    rec = None
    # as many proofs per pass over the regenerated rows as the 288 GB hold (a call of more than 64 proofs shares ONE pass among its
    # sub-batches of 64; 0.23 GB of workspace, 0.27 GB of t and 0.16 GB of inputs per proof at kappa = 2^14): the first size that fits
    failed = []
    for nbatch in proofs:
        rec = _config3_streamed_once(lab, dev, q, n, r, kappa, k, inputs, nbatch)
        if "error" not in rec:
            break
        failed.append({"proofs_per_step": nbatch, "error": rec["error"]})
        torch.cuda.empty_cache()
    if failed:
        rec["sizes_that_did_not_fit"] = failed
    if with_cpu:
        rec["cpu_baseline"] = _config3_cpu_baseline(q, r, k, kappa, b1_s_per_proof_k1024)
    return rec


def _config3_streamed_once(lab, dev, q, n, r, kappa, k, inputs, nbatch):
    import torch
    dSs, sts, dtapes = (list(x) for x in inputs)
    ctx = lab.Context(q, dev.index or 0, max(65536, n * r + r * r))
    ctx.use_torch_stream()
    rec = {"workload": "full Prover::proof_gen n=%d r=%d d=64, kappa=kappa1=kappa2=%d, q=%d; B regenerated from the ChaCha20 seed inside "
                       "the timed region (A resident, C and D as byte planes); %d proofs per step share one pass over the regenerated rows; "
                       "1 step, no warm-up" % (n, r, kappa, q, nbatch), "kappa": kappa, "proofs_per_step": nbatch}
    crs = plan = T = None
    try:
        if nbatch > len(dSs):      # more instances than the headline had: seeds continue where it stopped
            more = build_inputs(lab, ctx, k, q, dev, nbatch - len(dSs), 50 + len(dSs))
            dSs += more[0]; sts += more[1]; dtapes += more[2]
        dSs, sts, dtapes = dSs[:nbatch], sts[:nbatch], dtapes[:nbatch]
        t0 = time.perf_counter()
        crs = build_crs(lab, ctx, k, 1000, stream_b=True)
        rec["crs_setup_s"] = time.perf_counter() - t0
        rec["digit_primes"] = ctx.digit_primes
        plan = lab.BatchPlan(ctx, k, dSs, sts, crs, dtapes, device_outputs=True)
        torch.cuda.synchronize()
        ctx.profile_begin()
        t0 = time.perf_counter()
        T = plan.run()[-1]
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        ctx.profile_end()
        phases = ctx.profile_phases()
        coefs = r * k.T_1 * kappa * kappa * D
        rec.update({"proofs_per_s": nbatch / elapsed, "ms_per_step": elapsed * 1e3, "s_per_proof": elapsed / nbatch,
                    "prove_flags": ctx.prove_flags(), "checksum_u1_u2": int((checksum_u64(T.u_1) + checksum_u64(T.u_2)).item()),
                    "B_coefficients_regenerated_per_step": coefs, "coefficients_per_s_end_to_end": coefs / elapsed,
                    "roofline": streamed_roofline(phases, elapsed, 1)})
    except Exception as e:
        rec["error"] = repr(e)
    del plan, T
    if crs is not None:
        crs.close()
    ctx.close()
    del dSs, sts, dtapes
    torch.cuda.empty_cache()
    return rec


def _config3_cpu_baseline(q, r, k, kappa, b1_s_per_proof_k1024):
    if True:
        try:
            rate, cores, dt = host_chacha_rate(q)
            coefs = r * k.T_1 * kappa * kappa * D
            regen_s = coefs / rate
            cb = {"unit": "constraints/s", "cores": cores, "kind": "port", "extrapolated": True,
                  "host_chacha_coefficients_per_s": rate,
                  "sample": "CRS regeneration: %.2e coefficients of the oracle's scalar ChaCha20 + gen_range on %d threads in %.1f s -> %.0f s for "
                            "the %.2e coefficients of B per proof (one proof per pass over B, as the reference regenerates rows per fetch)"
                            % (rate * dt, cores, dt, regen_s, coefs)}
            arith_s = None
            if b1_s_per_proof_k1024:
                # the arithmetic of step 3 grows with kappa1 * kappa: (2^14 / 2^10)^2 = 256 x the kappa = 1024 measurement (upper
                # bound: the whole kappa = 1024 proof is scaled, not only its step 3)
                arith_s = b1_s_per_proof_k1024 * (kappa / 1024.0) ** 2
                cb["sample"] += "; arithmetic: B1's measured %.2f s per proof at kappa = 1024 x %d" % (b1_s_per_proof_k1024, int((kappa / 1024) ** 2))
            cb["value"] = 1.0 / (regen_s + (arith_s or 0.0))
            cb["s_per_proof"] = regen_s + (arith_s or 0.0)
            return cb
        except Exception as e:
            return {"value": None, "unit": "constraints/s", "cores": cpu_share(), "kind": "port", "error": repr(e)}


def launch_ranks(n_gpus, argv):
    """`bench.py --gpus N` without a launcher around it: start N ranks (one process per GPU) with torch.distributed.run
    BEFORE this process imports torch or touches the GPU, relay their output (rank 0 prints the JSON line) and exit
    with the children's status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def launcher_selftest(args):
    """LAB_BENCH_SELFTEST=1: the rank side of the launcher without a GPU (tests/test_distributed_cpu.py): rendezvous over
    gloo, count the ranks with an all-reduce, rank 0 prints a stub line.  LAB_BENCH_SELFTEST=fail: rank 1 exits non-zero."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ["LAB_BENCH_SELFTEST"] == "fail" and rank == world - 1:
        return 3
    # the same aggregation main() runs after its timed steps, on a stand-in vector (rank k contributes k+1 in every entry)
    vec = torch.full((128,), rank + 1, dtype=torch.int64)
    if world > 1:
        dist.init_process_group("gloo")
    agg = aggregate_over_ranks(dist if world > 1 else None, vec, float(rank + 1), torch.device("cpu"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ranks_seen": agg["ranks_seen"],
                          "collective_backend": agg["backend"], "aggregate_commitment_vector": agg["vector"], "tmax": agg["tmax"]}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--q", type=int, default=Q_REF33)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--r", type=int, default=32)
    ap.add_argument("--kappa", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=64,
                    help="independent proofs per step and GPU; they share the common reference string, which is then "
                         "streamed from HBM once per batch of up to 64 (1 = one proof per call)")
    ap.add_argument("--step3", choices=["auto", "valu", "mfma"], default="auto",
                    help="outer commitment sum_ik B_ik t_i^(k): 64-bit multiply-add kernels (B as RNS residues) or exact "
                         "int8 GEMMs on the matrix cores (B packed as signed byte planes, up to 64 proofs per pass over B); "
                         "auto = mfma when --batch > 8")
    ap.add_argument("--crs", choices=["resident", "streamed"], default="resident",
                    help="resident: A, B, C, D in HBM in the transform domain (expansion outside the timed region); streamed: B "
                         "regenerated from the ChaCha20 seed inside every step (SURVEY §8(f) row f1: the only way to the reference's "
                         "ranks, e.g. --kappa 16384), expansion INSIDE the timed region")
    ap.add_argument("--pi", choices=["packed", "int8"], default="packed",
                    help="projection matrices Pi on the device: 2 bits per entry (lab_prove_input.Pi2) or one int8 per entry")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-kappa16384", action="store_true",
                    help="skip extra.config3_kappa16384_streamed (one ~80 s step of the streamed prover at kappa = 2^14 after the headline)")
    ap.add_argument("--no-compact-crs", action="store_true", help="keep B, C, D in the main RNS basis (12 B/coefficient at the 33-bit q)")
    ap.add_argument("--digit-basis", choices=["auto", "u16", "mfma23"], default="auto",
                    help="digit basis of B, C, D: u16 = a prefix of the main primes plus one prime below 2^16 (10 B per coefficient, "
                         "the single-proof path streams it); mfma23 = primes below 2^23 (three signed byte planes per residue on "
                         "the matrix cores); auto = mfma23 on the matrix-core path, else u16")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("LAB_BENCH_SELFTEST"):
        sys.exit(launcher_selftest(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # one process per GPU; LAB_DIST_BACKEND=gloo lets several ranks share one GPU for a functional rehearsal
    backend = os.environ.get("LAB_DIST_BACKEND", "nccl")
    gpu_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import labrador_snark_amd as lab

    q, n, r, kappa = args.q, args.n, args.r, args.kappa
    k = lab.RuntimeConstants(q, n, r, kappa, kappa, kappa)
    # longest reduction in the prover: r*t1*kappa digit terms (small operands) — ask for enough CRT range
    ctx = lab.Context(q, gpu_index, max(65536, n * r + r * r))
    ctx.use_torch_stream()

    # instances 50 + rank*batch + j: witness/state/tape resident on the device
    nbatch = max(1, args.batch)
    use_mfma = args.step3 == "mfma" or (args.step3 == "auto" and nbatch > 8)
    basis23 = not args.no_compact_crs and (args.digit_basis == "mfma23" or (args.digit_basis == "auto" and use_mfma))
    dSs, sts, dtapes = build_inputs(lab, ctx, k, q, dev, nbatch, 50 + rank * nbatch, packed_pi=args.pi == "packed")
    crs = build_crs(lab, ctx, k, 1000 + 10 * rank, use_mfma=use_mfma, basis23=basis23, compact=not args.no_compact_crs,
                    stream_b=args.crs == "streamed")

    # the boundary structures of the 64 proofs are built once (persistent input and output buffers, as a prover service
    # keeps them): a step is one lab_prove_core_batch call
    plan = lab.BatchPlan(ctx, k, dSs, sts, crs, dtapes, device_outputs=True) if nbatch > 1 else None

    def step():
        if nbatch == 1:
            return lab.prove_core(ctx, k, dSs[0], sts[0], crs, dtapes[0], device_outputs=True)
        return plan.run()[-1]

    for _ in range(args.warmup):
        T = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches, alg_bytes = ctx.profile_end()
    step3_kernel = ctx.profile_kernel()
    phases = ctx.profile_phases() if args.crs == "streamed" else None

    # aggregate of the commitments across instances: one RCCL all-reduce of the folded (u_1, u_2) vector (outside the timed steps)
    agg = aggregate_over_ranks(dist, folded_commitments(T), elapsed, dev)
    elapsed = agg["tmax"]

    result = None
    if rank == 0:
        proofs = world * args.steps * nbatch
        value = proofs / elapsed
        per_launch_s = (kernel_ms / 1e3) / max(launches, 1)
        achieved = (alg_bytes / max(launches, 1)) / per_launch_s / 1e9 if launches else 0.0
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, gfx950 corrections applied) — only when they were taken on this very workload
        # (PMC counters cannot be read from inside the run: `traffic` is the value of the committed rocprofv3 --pmc passes of
        # this command, and `traffic_source` says so; null when no committed pass matches this workload and kernel)
        traffic, traffic_source = None, None
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
            if not name.endswith("_pmc_traffic.json"):
                continue
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", name)))
                if rec.get("workload_key") == [q, n, r, kappa, ctx.nlimbs, bool(crs.compact), nbatch] and \
                        rec.get("digit_primes") == getattr(ctx, "digit_primes", None) and \
                        rec.get("step3", "valu") == ("mfma" if use_mfma else "valu") and step3_kernel in rec.get("kernel", ""):
                    traffic, traffic_source = rec.get("hbm_bytes_per_launch"), "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run)" % name
                    break
            except Exception:
                pass
        counts = ring_products_per_proof(n, r, kappa, kappa, kappa, k.T_1, k.T_2)
        result = {
            "metric": "prover constraints/sec", "value": value, "unit": "constraints/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "full Prover::proof_gen n=%d r=%d d=64, kappa=kappa1=kappa2=%d, q=%d (%d RNS limbs), "
                                   "%s; %d independent proofs per step share the CRS; Pi resident at %s"
                                   % (n, r, kappa, q, ctx.nlimbs,
                                      "B STREAMED from the ChaCha20 seed inside the timed region (A, C, D resident)" if args.crs == "streamed"
                                      else "CRS resident in transform domain, CRS expansion excluded",
                                      nbatch, "2 bits per entry" if args.pi == "packed" else "one int8 per entry"),
                       "q": q, "n": n, "r": r, "kappa": kappa, "t1": k.T_1, "t2": k.T_2, "b1": k.B_1, "b2": k.B_2,
                       "nlimbs": ctx.nlimbs, "crs_basis": ("digit basis %s%s" % (getattr(ctx, "digit_primes", None), ", u16 last limb" if ctx.digit_small else
                                                          " (three signed byte planes per residue)"))
                       if crs.compact else "main basis", "digit_primes": getattr(ctx, "digit_primes", None) if crs.compact else None, "proofs_per_step_per_gpu": nbatch,
                       "step3": "int8 MFMA GEMMs (signed base-256 digit planes, int32 accumulation, exact)" if use_mfma
                       else "64-bit multiply-add kernels", "parallelism": "%d independent proofs per gpu x %d gpus" % (nbatch, world)},
            "constraints_per_proof": 1,        # K = 1 dot-product constraint per proof (src/constants.rs:26): constraints/s == proofs/s
            "ms_per_proof": elapsed / (args.steps * nbatch) * 1e3,
            "proofs_per_s": value, "witness_Rq_elems_per_s": value * n * r,
            "ring_products_per_proof": sum(counts.values()),
            "checksum_u1_u2": agg["scalar"], "aggregate_commitment_vector": agg["vector"],
            "ranks_seen": agg["ranks_seen"], "collective_backend": agg["backend"],
            "roofline": {"bound": "hbm", "kernel": (("lab::%s — one 23-bit limb (3 digit planes) of the outer commitment "
                                                     "sum_ik B_ik t_i^(k) (src/proofgen.rs:101-132) for the %d proofs of a step, "
                                                     "one launch per limb" if basis23 else
                                                     "lab::%s — one 30-bit limb (4 digit planes) of the outer commitment "
                                                     "sum_ik B_ik t_i^(k) (src/proofgen.rs:101-132) for the %d proofs of a step, "
                                                     "one launch per limb; the 257 limb (2 planes) is a smaller launch of the "
                                                     "same template") if use_mfma else
                                                    "lab::%s — the u32 limbs of the outer commitment sum_ik B_ik t_i^(k) "
                                                    "(src/proofgen.rs:101-132) for the %d proofs of a step; the u16 limb runs as a "
                                                    "second launch of the same kernel template") % (step3_kernel, nbatch),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "launches": launches, "avg_launch_ms": per_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes / max(launches, 1),
                         "kernel_share_of_step": (kernel_ms / 1e3) / elapsed if elapsed > 0 else None},
        }
        if use_mfma and launches:
            # the same launch against the matrix-core roofline: int8 multiply-adds of one limb's GEMMs (64 slots x pl matrix
            # planes x kappa1 rows x r*t1*kappa terms x (padded sets x pl vector planes) columns), dense int8 peak 5 PFLOP/s
            # (more than 64 proofs per step: the full sub-batches of 64 are the jobs of ONE launch per limb)
            ns_pad = 64 * (nbatch // 64) if nbatch >= 64 else 32 if nbatch > 16 else 16 if nbatch > 8 else 8
            pl = 3 if basis23 else 4
            ops = 2.0 * D * pl * kappa * (r * k.T_1 * kappa) * (ns_pad * pl)
            mfma_frac = ops / per_launch_s / 1e12 / 5000.0
            result["roofline"].update({"mfma_int8_ops_per_launch": ops, "mfma_achieved_TOPs": ops / per_launch_s / 1e12,
                                       "mfma_peak_TOPs": 5000.0, "mfma_frac": mfma_frac})
            # the roofline this launch sits closer to (both fractions are printed either way)
            if mfma_frac > result["roofline"]["frac"]:
                result["roofline"].update({"bound": "mfma", "achieved": ops / per_launch_s / 1e12, "peak": 5000.0, "unit": "TOP/s",
                                           "frac": mfma_frac, "hbm_achieved_GBps": achieved, "hbm_frac": achieved / HBM_PEAK_GBS})
            else:
                result["roofline"]["bound_note"] = ("HBM: %.2f of the 8 TB/s spec (%.2f of the ~6.3 TB/s a streaming kernel reaches); "
                                                    "matrix cores at %.2f of the dense int8 peak" % (achieved / HBM_PEAK_GBS, achieved / 6300.0, mfma_frac))
            # the committed load-side / multiply-side ablation of this kernel (tools/gemm_ablation.py), when there is one
            for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
                if name.endswith("_gemm_ablation.json") and nbatch == 64 and kappa == 1024 and basis23:
                    try:
                        ab = json.load(open(os.path.join(ROOT, "profiles", name)))
                        sm, m1 = ab["summary"], ab["modes"]["mode1"]
                        result["roofline"]["ablation"] = {
                            "source": "profiles/%s (tools/gemm_ablation.py, not this run)" % name, "product_ms": sm["product_ms"],
                            "load_side_alone_ms": sm["load_side_ms"], "multiply_side_alone_ms": sm["multiply_side_ms"],
                            "overlap": sm["overlap"], "in_kernel_shader_clock_GHz": m1["shader_clock_GHz_at_100MHz_const"],
                            "note": "the product launch runs within ~10% of its load side alone; under the combined load the part "
                                    "sustains the in-kernel clock above, at which the matrix pipe is ~96% busy: the launch sits on "
                                    "the HBM limit and the power-limited matrix-core limit at once"}
                    except Exception:
                        pass
                    break

    if rank == 0 and args.crs == "streamed":
        coefs = r * k.T_1 * kappa * kappa * D
        result["crs_expansion"] = {"B_coefficients_regenerated_per_step": coefs, "coefficients_per_s": coefs / (elapsed / args.steps),
                                   "note": "every coefficient = one ChaCha20 key set-up + block + gen_range on i128 (src/structs.rs:167-171): "
                                           "the step is bound by that VALU work (k_crs_expand), not by HBM or the matrix cores"}
        result["roofline"] = streamed_roofline(phases, elapsed, args.steps)
    extras = {}
    shared_inputs = (dSs, sts, dtapes)
    if rank == 0 and world == 1 and not args.no_extras and args.crs == "resident" and use_mfma and basis23 and nbatch == 64 and kappa == 1024:
        # more proofs per pass over the resident B: the next 192 instances next to the headline's 64 (the kappa = 2^14 record below reuses them)
        try:
            more = build_inputs(lab, ctx, k, q, dev, 192, 50 + 64, packed_pi=args.pi == "packed")
            shared_inputs = (dSs + more[0], sts + more[1], dtapes + more[2])
            extras["resident_shared_pass_256"] = resident_shared_pass(lab, ctx, k, crs, shared_inputs, 256, result["value"] if result else None)
        except Exception as e:
            extras["resident_shared_pass_256"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_extras:
        # release the big CRS before the side measurements
        crs.close()
        torch.cuda.synchronize()
        extras.update(side_measurements(lab, ctx, q, dev))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            counts = ring_products_per_proof(n, r, kappa, kappa, kappa, k.T_1, k.T_2)
            try:
                result["cpu_baseline"] = cpu_baseline(k, q, sum(counts.values()))
            except Exception as e:      # e.g. a host without the 77 GB B1 keeps resident: never lose the GPU line over it
                result["cpu_baseline"] = {"value": None, "unit": "constraints/s", "cores": cpu_share(), "kind": "port", "error": repr(e)}
        else:
            result["cpu_baseline"] = None
    if rank == 0 and world == 1 and not args.no_extras and not args.no_kappa16384 and args.crs == "resident" and kappa < 16384:
        # config 3 at the contract's commitment rank (kappa = 2^14, B streamed): the same instances, one step
        b1 = (result.get("cpu_baseline") or {}).get("s_per_proof") if kappa == 1024 else None
        plan = None
        ctx.close()
        torch.cuda.empty_cache()
        extras["config3_kappa16384_streamed"] = config3_streamed(lab, dev, q, n, r, 16384, shared_inputs, b1_s_per_proof_k1024=b1,
                                                                 with_cpu=not args.no_cpu_baseline)
    if rank == 0:
        if extras:
            result["extra"] = extras
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def resident_shared_pass(lab, ctx, k, crs, inputs, nbatch, headline_value, steps=4):
    """The headline's workload with MORE than 64 proofs per step (same resident CRS, the headline's instances and the next ones): the
    call runs its sub-batches of 64 up to the digit tiles of t and multiplies all of them with B in ONE matrix-core launch per limb —
    B comes from HBM once per `nbatch` proofs, the other sub-batches' blocks find its tiles in L2 (lab_ctx_prove_flags() & 4).  The
    step-3 launch is then bound by the matrix cores, not by HBM: the record carries that roofline."""
    import torch
    dSs, sts, dtapes = (list(x)[:nbatch] for x in inputs)
    plan = lab.BatchPlan(ctx, k, dSs, sts, crs, dtapes, device_outputs=True)
    plan.run()
    torch.cuda.synchronize()
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        T = plan.run()[-1]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches, alg_bytes = ctx.profile_end()
    kname = ctx.profile_kernel()
    value = steps * nbatch / elapsed
    per_launch_s = (kernel_ms / 1e3) / max(launches, 1)
    jobs = nbatch // 64
    kappa, r = k.KAPPA, k.R
    ops = 2.0 * D * 3 * kappa * (r * k.T_1 * kappa) * (64 * jobs * 3)
    hbm = (alg_bytes / max(launches, 1)) / per_launch_s / 1e9
    tops = ops / per_launch_s / 1e12
    rec = {"workload": "the headline's Prover::proof_gen workload (n=%d r=%d kappa=kappa1=kappa2=%d, same resident CRS) with %d proofs per step: "
                       "sub-batches of 64 whose digit tiles are the %d jobs of ONE launch per limb of the outer commitment (B read from HBM once "
                       "per %d proofs)" % (k.N, r, kappa, nbatch, jobs, nbatch),
           "proofs_per_step": nbatch, "steps": steps, "proofs_per_s": value, "ms_per_step": elapsed / steps * 1e3,
           "ms_per_64_proofs": elapsed / steps * 1e3 * 64 / nbatch, "vs_headline_64_per_step": value / headline_value if headline_value else None,
           "prove_flags": ctx.prove_flags(), "checksum_u1_u2": int(folded_commitments(T).sum().item() & ((1 << 62) - 1)),
           "roofline": {"bound": "mfma" if tops / 5000.0 > hbm / HBM_PEAK_GBS else "hbm",
                        "kernel": "lab::%s — one 23-bit limb of the outer commitment for the %d proofs of a step (%d jobs of 64 sets)" % (kname, nbatch, jobs),
                        "achieved": tops, "peak": 5000.0, "unit": "TOP/s", "frac": tops / 5000.0, "traffic": None,
                        "hbm_achieved_GBps": hbm, "hbm_frac": hbm / HBM_PEAK_GBS, "launches": launches, "avg_launch_ms": per_launch_s * 1e3,
                        "mfma_int8_ops_per_launch": ops, "algorithmic_bytes_per_launch": alg_bytes / max(launches, 1),
                        "ceiling_note": "tools/ubench_mfma_i8.hip (profiles/r03_mfma_ceiling.json): v_mfma_i32_32x32x32_i8 with random register-resident "
                                        "operands sustains 3.55 POP/s on this part (power-limited clock), 2.2-2.4 with the operand traffic of one or two "
                                        "jobs per pass over B"}}
    del plan
    return rec


def _time(fn, iters, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def side_measurements(lab, ctx, q, dev):
    """BASELINE configs[1] (batched NTT) and configs[3] (Ajtai mat-vec 2^14 x 2^10), reported next to the headline."""
    import torch
    out = {}
    L = ctx.nlimbs
    for batch in (4096, 1 << 20):
        x = torch.randint(0, 1 << 29, (L, batch, D), dtype=torch.int32, device=dev)
        tf = _time(lambda: ctx.ntt_forward(x, batch), 20)
        ti = _time(lambda: ctx.ntt_inverse(x, batch), 20)
        out["ntt_batch_%d" % batch] = {"limb_planes": L, "fwd_us": tf * 1e6, "inv_us": ti * 1e6,
                                       "ntts_per_s": 2 * L * batch / (tf + ti),
                                       "algorithmic_GBps": 2 * L * batch * 512 / (tf + ti) / 1e9}
        del x
    rows, cols = 1 << 14, 1 << 10
    A = ctx.vec(rows * cols).fill_synthetic(4)
    for ncols in (1, 32):
        S = ctx.vec(cols * ncols).fill_synthetic(5)
        o = ctx.vec(rows * ncols)
        if ncols == 1:
            t = _time(lambda: lab.matvec_sum(ctx, A, S, 1, rows, cols, out=o), 10)
        else:
            t = _time(lambda: lab.ajtai_commit(ctx, A, S, rows, cols, ncols, out=o), 5)
        byt = L * (rows * cols + cols * ncols + rows * ncols) * 256
        out["ajtai_2p14x2p10_rhs%d" % ncols] = {"ms": t * 1e3, "algorithmic_GBps": byt / t / 1e9,
                                                "modmuladds_per_s": L * rows * cols * ncols * 64 / t}
        S.close(); o.close()
    # the same 32 right-hand sides on the matrix cores (the path the prover's step 1 takes for a batch): A packed once as
    # signed byte planes, the 32 columns as 32 vector sets, exact int8 GEMMs
    try:
        A8 = lab.Mat8(ctx, A, 1, rows, cols)
        Vs = ctx.vec(32 * cols).fill_synthetic(5)
        o = ctx.vec(32 * rows)
        t = _time(lambda: lab.matvec_sum_mfma(ctx, A8, Vs, 32, out=o), 5)
        out["ajtai_2p14x2p10_rhs32_matrix_cores"] = {"ms": t * 1e3, "packed_matrix_GBps": A8.nbytes / t / 1e9,
                                                     "modmuladds_per_s": L * rows * cols * 32 * 64 / t,
                                                     "int8_TOPs": 2.0 * 16 * L * rows * cols * 32 * 64 / t / 1e12}
        A8.close(); Vs.close(); o.close()
    except Exception as e:
        out["ajtai_2p14x2p10_rhs32_matrix_cores"] = {"error": str(e)}
    A.close()
    # ... and once more with A re-stored in FOUR primes below 2^23 (lab_ctx_enable_digit_basis_mfma with digit_bound = q: the basis
    # holds 1024-term sums of products of two full-range residues, 2^80 < M/2): 4 limbs x 9 byte products = 36 per coefficient
    # product instead of 3 x 16 = 48, the same 12 bytes of A per coefficient, and every limb runs on the assembly K loop
    # (k_gemm_i8_w3) instead of round 1's 4-plane kernel.  Same residues mod q in and out (tests/test_gpu_ops.py).
    try:
        c4 = lab.Context(q, ctx.device, 65536)
        c4.use_torch_stream()
        primes4, _, _ = c4.enable_digit_basis(cols, q, mfma=True)
        A4 = c4.vec(rows * cols, digit=True).fill_synthetic(4)
        A84 = lab.Mat8(c4, A4, 1, rows, cols)
        c4.synchronize()
        A4.close()
        Vs = c4.vec(32 * cols, digit=True).fill_synthetic(5)
        o = c4.vec(32 * rows, digit=True)
        t = _time(lambda: lab.matvec_sum_mfma(c4, A84, Vs, 32, out=o), 5)
        out["ajtai_2p14x2p10_rhs32_matrix_cores_23bit_basis"] = {"ms": t * 1e3, "primes": primes4, "packed_matrix_GBps": A84.nbytes / t / 1e9,
                                                                "packed_matrix_bytes": A84.nbytes,
                                                                "int8_TOPs": 2.0 * 9 * len(primes4) * rows * cols * 32 * 64 / t / 1e12}
        A84.close(); Vs.close(); o.close()
        c4.close()
    except Exception as e:
        out["ajtai_2p14x2p10_rhs32_matrix_cores_23bit_basis"] = {"error": str(e)}
    # the same mat-vec shape as the prover's step 3 when q is its own NTT prime (single limb, 4 B per coefficient):
    # the data point for an NTT-friendly modulus (q = 1073738753, 30 bits); 24 GiB streamed per call
    try:
        c1 = lab.Context(1073738753, ctx.device)
        c1.use_torch_stream()
        M = c1.vec(96 * 1024 * 1024).fill_synthetic(7)
        V = c1.vec(96 * 1024).fill_synthetic(8, bound=1023)
        o = c1.vec(1024)
        t = _time(lambda: lab.matvec_sum(c1, M, V, 96, 1024, 1024, out=o), 5)
        out["outer_commitment_single_limb_q1073738753"] = {"ms": t * 1e3, "algorithmic_GBps": 96 * 1024 * 1024 * 256 / t / 1e9}
        for v in (M, V, o):
            v.close()
        c1.close()
    except Exception as e:  # never let a side measurement break the headline line
        out["outer_commitment_single_limb_q1073738753"] = {"error": str(e)}
    # one proof per call at the headline modulus (latency, next to the batch-64 throughput of the headline line): B, C, D in the
    # compact digit basis (10 B per coefficient), streamed by the multiply-add kernels
    try:
        out["single_proof_headline_q"] = single_proof_latency(lab, dev, q)
    except Exception as e:
        out["single_proof_headline_q"] = {"error": repr(e)}
    # SURVEY section 8(d) "q: report both": the NTT-friendly 32-bit q = 4294966657 on path (b) (q its own NTT prime: one limb,
    # 4 B per coefficient, strict-32 kernels) and on path (a) (LAB_CTX_RNS_ONLY: three 30-bit RNS limbs)
    try:
        out["q4294966657_both_paths"] = both_paths_q32(lab, dev)
    except Exception as e:
        out["q4294966657_both_paths"] = {"error": repr(e)}
    return out


def single_proof_latency(lab, dev, q, kappa=1024, rns_only=False):
    """One proof per lab_prove_core call (n=1024, r=32, kappa=kappa1=kappa2=1024) in the best single-proof configuration: B, C, D in
    the compact digit basis when one exists (else the main basis), B streamed from HBM by the multiply-add kernel."""
    import torch
    n, r = 1024, 32
    ctx = lab.Context(q, dev.index or 0, max(65536, n * r + r * r), rns_only=rns_only)
    ctx.use_torch_stream()
    k = lab.RuntimeConstants(q, n, r, kappa, kappa, kappa)
    L = ctx.nlimbs
    dSs, sts, dtapes, crs = build_workload(lab, ctx, k, q, dev, 1, 50, 1000, use_mfma=False, basis23=False, compact=True)
    step = lambda: lab.prove_core(ctx, k, dSs[0], sts[0], crs, dtapes[0], device_outputs=True)
    step()
    torch.cuda.synchronize()
    ctx.profile_begin()
    t = _time(step, 5, warm=1)
    ms, launches, alg = ctx.profile_end()
    rec = {"q": q, "limbs": L, "proofs_per_s": 1.0 / t, "ms_per_proof": t * 1e3,
           "crs_basis": "digit basis %s" % getattr(ctx, "digit_primes", None) if crs.compact else "main basis (q itself)" if L == 1 else "main basis",
           "step3_kernel": ctx.profile_kernel(), "step3_launch_ms": ms / max(launches, 1),
           "step3_algorithmic_GBps": alg / max(launches, 1) / (ms / max(launches, 1) / 1e3) / 1e9 if launches else None}
    crs.close()
    ctx.close()
    torch.cuda.synchronize()
    return rec


def both_paths_q32(lab, dev, q=4294966657, kappa=1024):
    import torch
    res = {"q": q, "workload": "n=1024 r=32 kappa=kappa1=kappa2=%d; one proof per call (B streamed on the multiply-add kernels)" % kappa}
    n, r = 1024, 32
    for name, rns in (("path_b_single_limb", False), ("path_a_rns", True)):
        ctx = lab.Context(q, dev.index or 0, max(65536, n * r + r * r), rns_only=rns)
        ctx.use_torch_stream()
        k = lab.RuntimeConstants(q, n, r, kappa, kappa, kappa)
        L = ctx.nlimbs
        rec = {"limbs": L}
        x = torch.randint(0, 1 << 29, (L, 1 << 20, D), dtype=torch.int32, device=dev)
        tf = _time(lambda: ctx.ntt_forward(x, 1 << 20), 10)
        ti = _time(lambda: ctx.ntt_inverse(x, 1 << 20), 10)
        rec["ntt_batch_2p20"] = {"fwd_us": tf * 1e6, "inv_us": ti * 1e6, "polynomials_per_s": 2 * (1 << 20) / (tf + ti),
                                 "algorithmic_GBps": 2 * L * (1 << 20) * 512 / (tf + ti) / 1e9}
        del x
        # the best single-proof configuration of each path: (b) B, C, D in the main basis = 4 B per coefficient; (a) B, C, D in
        # the compact digit basis (10 B per coefficient)
        dSs, sts, dtapes, crs = build_workload(lab, ctx, k, q, dev, 1, 50, 1000, use_mfma=False, basis23=False, compact=True)
        step = lambda: lab.prove_core(ctx, k, dSs[0], sts[0], crs, dtapes[0], device_outputs=True)
        step()
        torch.cuda.synchronize()
        ctx.profile_begin()
        t = _time(step, 5, warm=1)
        ms, launches, alg = ctx.profile_end()
        rec["single_proof"] = {"proofs_per_s": 1.0 / t, "ms_per_proof": t * 1e3, "crs_bytes_per_coefficient": 4 * L if not crs.compact else None,
                               "crs_basis": "digit basis %s" % getattr(ctx, "digit_primes", None) if crs.compact else "main basis (q itself)" if L == 1 else "main basis",
                               "step3_kernel": ctx.profile_kernel(), "step3_launch_ms": ms / max(launches, 1),
                               "step3_algorithmic_GBps": alg / max(launches, 1) / (ms / max(launches, 1) / 1e3) / 1e9 if launches else None}
        crs.close()
        if not rns:
            # 64 proofs per call at this q: the main basis is the single limb, B, C, D, A go to the matrix cores in the 23-bit
            # digit basis exactly as at the headline q
            dSs, sts, dtapes, crs = build_workload(lab, ctx, k, q, dev, 64, 50, 1000, use_mfma=True, basis23=True, compact=True)
            plan = lab.BatchPlan(ctx, k, dSs, sts, crs, dtapes, device_outputs=True)
            t = _time(plan.run, 3, warm=1)
            rec["batch64"] = {"proofs_per_s": 64.0 / t, "ms_per_step": t * 1e3}
            del plan
            crs.close()
        ctx.close()
        torch.cuda.synchronize()
        res[name] = rec
    return res


if __name__ == "__main__":
    main()
