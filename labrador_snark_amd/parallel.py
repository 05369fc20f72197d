"""Multi-GPU glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).

The prover shards at two levels:
  1. independent instances — one proof per GPU, no data-path collective; bench.py uses this (weak scaling) and
     all-reduces only a checksum of the aggregated commitments and the max step time;
  2. inside one proof, the outer commitment  lhs = sum_{i<r,k<t1} B_ik t_i^(k)  is a sum of r*t1 independent
     mat-vecs — the reference's rayon map followed by a sequential add_poly_vec fold
     (src/proofgen.rs:101-132).  Each rank owns a contiguous slice of the (i,k) pairs (and only that slice of
     the B matrices), computes its partial sum, and one all-reduce(SUM) of kappa1*64 residues + `mod q`
     finishes it.  Residues are < q < 2^40 so up to 2^23 ranks sum without overflow in int64; RCCL has no
     modular-sum op, hence sum-then-reduce.
"""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(total, rank, world):
    """Contiguous balanced slice [lo, hi) of `total` work items for `rank` (first total%world ranks get one more)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def ik_shard(r, t1, rank, world):
    """(i,k) pairs of the outer commitment owned by `rank`, as matrix indices m = i*t1 + k (src/proofgen.rs:101-124)."""
    lo, hi = shard_range(r * t1, rank, world)
    return lo, hi


def _all_reduce(t, op, group=None):
    """dist.all_reduce that also takes a GPU tensor under the gloo backend (functional rehearsals of several ranks on
    one GPU): gloo reduces a host copy.  Under nccl (= RCCL) the tensor is reduced in place over xGMI."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def allreduce_residues_mod_q(t, q, group=None):
    """In-place: element-wise sum over ranks of canonical residues (int64 tensor), then reduce mod q.
    This is the sequential `lhs = add_poly_vec(&prod, &lhs)` fold of src/proofgen.rs:126-132 done as a collective."""
    import torch.distributed as dist
    _all_reduce(t, dist.ReduceOp.SUM, group)
    t.remainder_(q)
    return t


def allreduce_max(x, group=None):
    import torch.distributed as dist
    return _all_reduce(x, dist.ReduceOp.MAX, group)


def allreduce_sum(x, group=None):
    import torch.distributed as dist
    return _all_reduce(x, dist.ReduceOp.SUM, group)


def sharded_outer_commitment(ctx, B_local, tdec_local, nmat_local, kappa1, kappa, q, group=None, ns=1, local_partial=None,
                             device="cuda"):
    """lhs of u_1 across ranks (src/proofgen.rs:101-132: the rayon map over (i,k) + the sequential fold).
    `B_local` holds only this rank's B_ik matrices ([nmat_local][kappa1][kappa]): a hat vector (multiply-add kernels) or
    a Mat8 packed from it (matrix cores, ns in {8,16,32,64} proofs at once); `tdec_local` the matching decomposed t
    vectors ([ns][nmat_local][kappa]).  Returns an int64 tensor [ns][kappa1][64] of residues mod q, identical on every
    rank.  `local_partial(nmat_local) -> uint64 array [ns][kappa1][64]` replaces the GPU product in the CPU (gloo)
    tests, where the oracle stands in for the device."""
    import numpy as np
    import torch
    from . import api
    out = torch.zeros((ns, kappa1, 64), dtype=torch.int64, device=device)
    if nmat_local > 0:
        if local_partial is not None:
            out.copy_(torch.from_numpy(np.ascontiguousarray(local_partial(nmat_local)).view(np.int64).reshape(ns, kappa1, 64)))
        elif isinstance(B_local, api.Mat8):
            part = api.matvec_sum_mfma(ctx, B_local, tdec_local, ns)
            part.store(out=out)
            ctx.synchronize()
            part.close()
        else:
            for b in range(ns):
                part = api.matvec_sum(ctx, B_local, tdec_local, nmat_local, kappa1, kappa, v_off=b * nmat_local * kappa)
                part.store(out=out[b])
                ctx.synchronize()
                part.close()
    return allreduce_residues_mod_q(out, q, group)
