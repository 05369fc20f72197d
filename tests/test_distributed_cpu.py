"""world_size-2 gloo test of the N>1 path on CPU: (i,k)-sharded outer commitment + all-reduce + mod q equals the
single-process result (src/proofgen.rs:101-132), and the bench-style aggregation (max time, checksum sum).
Per-rank partial sums are computed with the ORACLE here (no GPU in this container); the code under test is the
sharding/aggregation glue in labrador_snark_amd/parallel.py, which the GPU path uses unchanged over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = 64


def _worker(rank, world, port, q, r, t1, kappa1, kappa, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from labrador_snark_amd import parallel
    from oracle import coracle as co
    nmat = r * t1
    B = co.synth_fill(7, 0, nmat * kappa1 * kappa * D, q).reshape(nmat, kappa1, kappa, D)
    V = co.synth_fill(8, 0, nmat * kappa * D, 1625).reshape(nmat, kappa, D)  # decomposed digits are < b1
    lo, hi = parallel.ik_shard(r, t1, rank, world)
    part = np.zeros((kappa1, D), np.uint64)
    for m in range(lo, hi):
        for rho in range(kappa1):
            part[rho] = co.poly_add(q, part[rho], co.inner_product(q, B[m, rho], V[m]))
    t = torch.from_numpy(part.view(np.int64).copy())
    parallel.allreduce_residues_mod_q(t, q)
    tmax = parallel.allreduce_max(torch.tensor([float(rank + 1)], dtype=torch.float64))
    chk = parallel.allreduce_sum(torch.tensor([rank + 10], dtype=torch.int64))
    if rank == 0:
        full = np.zeros((kappa1, D), np.uint64)
        for m in range(nmat):
            for rho in range(kappa1):
                full[rho] = co.poly_add(q, full[rho], co.inner_product(q, B[m, rho], V[m]))
        ret["ok"] = bool(np.array_equal(t.numpy().view(np.uint64), full))
        ret["tmax"] = float(tmax.item())
        ret["chk"] = int(chk.item())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("q", [8191, (1 << 32) + 15])
def test_sharded_outer_commitment_gloo_world2(q):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, q, 3, 3, 4, 5, ret), nprocs=world, join=True)
    assert ret["ok"] is True
    assert ret["tmax"] == 2.0 and ret["chk"] == 21
