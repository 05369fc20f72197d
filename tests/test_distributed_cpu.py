"""world_size-2 gloo tests of the N>1 path on CPU (no GPU in this container).

(1) parallel.sharded_outer_commitment — the (i,k)-sharded outer commitment, all-reduce(SUM) and `mod q` of
src/proofgen.rs:101-132 — runs on every rank exactly as on the GPU box, with the ORACLE standing in for the device
product of the rank's own slice (`local_partial`): ranks with an empty slice, uneven slices and the 33-bit modulus.
(2) `bench.py --gpus 2`'s launcher: the parent starts the ranks itself before it imports torch, relays rank 0's JSON
line and returns the children's status (exercised with a stub workload, LAB_BENCH_SELFTEST=1, since the real one
needs a GPU).
The same code runs over RCCL on the GPU box (tests/test_gpu_distributed.py rehearses it with two ranks on one GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = 64


def _worker(rank, world, port, q, r, t1, kappa1, kappa, ns, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from labrador_snark_amd import parallel
    from oracle import coracle as co
    nmat = r * t1
    B = co.synth_fill(7, 0, nmat * kappa1 * kappa * D, q).reshape(nmat, kappa1, kappa, D)
    V = co.synth_fill(8, 0, ns * nmat * kappa * D, 1625).reshape(ns, nmat, kappa, D)  # decomposed digits are < b1
    lo, hi = parallel.ik_shard(r, t1, rank, world)

    def partial(m_lo, m_hi):
        part = np.zeros((ns, kappa1, D), np.uint64)
        for b in range(ns):
            for m in range(m_lo, m_hi):
                for rho in range(kappa1):
                    part[b, rho] = co.poly_add(q, part[b, rho], co.inner_product(q, B[m, rho], V[b, m]))
        return part

    t = parallel.sharded_outer_commitment(None, None, None, hi - lo, kappa1, kappa, q, ns=ns, device="cpu",
                                          local_partial=lambda nloc: partial(lo, lo + nloc))
    tmax = parallel.allreduce_max(torch.tensor([float(rank + 1)], dtype=torch.float64))
    chk = parallel.allreduce_sum(torch.tensor([rank + 10], dtype=torch.int64))
    if rank == 0:
        ret["ok"] = bool(np.array_equal(t.numpy().view(np.uint64), partial(0, nmat)))
        ret["tmax"] = float(tmax.item())
        ret["chk"] = int(chk.item())
        ret["slices"] = [parallel.ik_shard(r, t1, k, world) for k in range(world)]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("q,r,t1,ns", [(8191, 3, 3, 1), ((1 << 32) + 15, 3, 3, 2), ((1 << 32) + 15, 1, 1, 1)])
def test_sharded_outer_commitment_gloo_world2(q, r, t1, ns):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, q, r, t1, 4, 5, ns, ret), nprocs=world, join=True)
    assert ret["ok"] is True
    assert ret["tmax"] == 2.0 and ret["chk"] == 21
    lo0, hi0 = ret["slices"][0]
    lo1, hi1 = ret["slices"][1]
    assert lo0 == 0 and hi0 == lo1 and hi1 == r * t1          # (1, 1): rank 1 owns nothing and still takes part


def test_bench_launcher_starts_the_ranks_itself():
    env = dict(os.environ, LAB_BENCH_SELFTEST="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["selftest"] is True and rec["ranks_seen"] == 2
    # bench.aggregate_over_ranks, the code main() runs after its timed steps: the collective saw both ranks, says which backend
    # carried it, summed the 128-entry commitment vector (rank k contributes k + 1 everywhere) and took the MAX of the times
    assert rec["collective_backend"] == "gloo" and rec["aggregate_commitment_vector"] == [3] * 128 and rec["tmax"] == 2.0
    # a failing rank makes the launcher fail
    env["LAB_BENCH_SELFTEST"] = "fail"
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert bad.returncode != 0
