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


def allreduce_residues_mod_q(t, q, group=None):
    """In-place: element-wise sum over ranks of canonical residues (int64 tensor), then reduce mod q.
    This is the sequential `lhs = add_poly_vec(&prod, &lhs)` fold of src/proofgen.rs:126-132 done as a collective."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t.remainder_(q)
    return t


def allreduce_max(x, group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(x, op=dist.ReduceOp.MAX, group=group)
    return x


def allreduce_sum(x, group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x


def sharded_outer_commitment(ctx, B_local, tdec_local, nmat_local, kappa1, kappa, q, group=None):
    """lhs of u_1 across ranks.  `B_local` holds only this rank's B_ik matrices ([nmat_local][kappa1][kappa], hat
    domain), `tdec_local` the matching decomposed t vectors ([nmat_local][kappa]).  Returns an int64 CUDA tensor
    [kappa1][64] of residues mod q, identical on every rank."""
    import torch
    from . import api
    out = torch.zeros((kappa1, 64), dtype=torch.int64, device="cuda")
    if nmat_local > 0:
        part = api.matvec_sum(ctx, B_local, tdec_local, nmat_local, kappa1, kappa)
        part.store(out=out)
        ctx.synchronize()
        part.close()
    return allreduce_residues_mod_q(out, q, group)
