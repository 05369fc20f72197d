"""The N>1 path on real kernels: two ranks (processes) share the one GPU of the test box and talk over gloo
(LAB_DIST_BACKEND=gloo is the same rehearsal for bench.py); on the 8-GPU node the identical code runs over RCCL.
Each rank owns a slice of the (i,k) pairs of the outer commitment and ONLY that slice of B (src/proofgen.rs:101-132),
computes its partial on the matrix cores (lab_matvec_sum_mfma) resp. the multiply-add kernel, and
parallel.sharded_outer_commitment all-reduces the residues and reduces mod q; rank 0 compares with the unsharded
product and with the oracle."""
import os
import sys

import numpy as np
import pytest

from conftest import ctx_kw

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = 64


def _worker(rank, world, port, q, r, t1, kappa1, kappa, ns, use_mfma, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import labrador_snark_amd as lab
    from labrador_snark_amd import parallel
    torch.cuda.set_device(0)
    ctx = lab.Context(q, 0, 1 << 16, **ctx_kw(q))
    ctx.enable_digit_basis(r * t1 * kappa, 1625, mfma=use_mfma)
    nmat = r * t1
    lo, hi = parallel.ik_shard(r, t1, rank, world)
    nloc = hi - lo
    # this rank's B_ik matrices and the matching digit vectors of every set: slices of the shared synthetic streams
    Bl = ctx.vec(max(nloc, 1) * kappa1 * kappa, digit=True)
    Vl = ctx.vec(ns * max(nloc, 1) * kappa, digit=True)
    if nloc:
        Bl.fill_synthetic(7, first_elem=lo * kappa1 * kappa)
        for b in range(ns):
            Vl.fill_synthetic(8, first_elem=(b * nmat + lo) * kappa, bound=1625, offset=b * nloc * kappa, count=nloc * kappa)
    Bop = lab.Mat8(ctx, Bl, nloc, kappa1, kappa) if (use_mfma and nloc) else Bl
    got = parallel.sharded_outer_commitment(ctx, Bop, Vl, nloc, kappa1, kappa, q, ns=ns)
    if rank == 0:
        from oracle import coracle as co
        B = ctx.vec(nmat * kappa1 * kappa, digit=True).fill_synthetic(7)
        V = ctx.vec(ns * nmat * kappa, digit=True).fill_synthetic(8, bound=1625)
        full = np.stack([lab.matvec_sum(ctx, B, V, nmat, kappa1, kappa, v_off=b * nmat * kappa).store() for b in range(ns)])
        ok = bool(np.array_equal(got.cpu().numpy().view(np.uint64), full))
        b, rho = ns - 1, kappa1 - 1
        v = co.synth_fill(8, b * nmat * kappa * D, nmat * kappa * D, 1625).reshape(nmat, kappa, D)
        want = np.zeros(D, np.uint64)
        for m in range(nmat):
            row = co.synth_fill(7, ((m * kappa1 + rho) * kappa) * D, kappa * D, q).reshape(kappa, D)
            want = co.poly_add(q, want, co.inner_product(q, row, v[m]))
        ret["ok"] = ok and bool(np.array_equal(full[b, rho], want))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("use_mfma,ns", [(True, 8), (False, 2)])
def test_ik_sharded_outer_commitment_two_ranks_one_gpu(use_mfma, ns):
    import torch.multiprocessing as mp
    world = 2
    port = 29700 + (os.getpid() % 2000) + (1 if use_mfma else 0)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, (1 << 32) + 15, 3, 3, 40, 96, ns, use_mfma, ret), nprocs=world, join=True)
    assert ret["ok"] is True


def test_bench_line_of_two_ranks_and_of_one_without_extras():
    # bench.py itself with N = 2 (its own launcher, two ranks sharing this box's GPU over gloo — on the 8-GPU node the same code
    # runs over RCCL): rank 0 prints exactly ONE JSON line whose collective really saw both ranks and summed the folded commitment
    # vectors; and the single-rank line without extras (a run that once printed nothing).  Small shapes: kappa = 256, 8 proofs.
    import json
    import subprocess
    common = ["--batch", "8", "--kappa", "256", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, LAB_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    recs = {}
    for n in (1, 2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + common, env=env, capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        recs[n] = json.loads(lines[0])
    one, two = recs[1], recs[2]
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1 and one["collective_backend"] is None
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["collective_backend"] == "gloo" and two["scaling"] == "weak"
    assert len(two["aggregate_commitment_vector"]) == 128 and two["value"] > 0 and two["steps"] == 2
    # rank 0 proves the same instances in both runs (seeds 50..57): its share of the summed vector is the single-rank vector
    assert all(a >= b for a, b in zip(two["aggregate_commitment_vector"], one["aggregate_commitment_vector"]))
    assert two["aggregate_commitment_vector"] != one["aggregate_commitment_vector"]
