"""Multi-GPU plumbing: how reads are partitioned over ranks and how results are stitched.

The hot path shards naturally: trimming is per read, counts are integer sums over reads
(associative and commutative, so any partition gives bit-identical tables) and insertion
events are a multiset union.  One process per GPU; the only data-path collective is ONE
reduce (sum, uint32) of the device table -- 7 x ref_len words: counts [ref_len][6] followed
by the insertion-event tally [ref_len] -- over RCCL/xGMI (torch.distributed backend "nccl";
"gloo" on CPU in the tests).  Insertion strings are only needed for the positions the calling
kernel flags, so they travel as a small object gather to rank 0.
"""
from __future__ import annotations

import numpy as np


def init_from_env():
    """(dist, rank, world) for the calling process: torch.distributed over RCCL (backend "nccl"; "gloo" without a GPU)
    when launched by torchrun with WORLD_SIZE > 1 or when AMPLIPY_FORCE_DIST=1, else (None, 0, 1)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("AMPLIPY_FORCE_DIST", "0") != "1":
        return None, 0, 1
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        if torch.cuda.is_available():
            dev = torch.device("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist, rank, world


def finish(dist):
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def amplicon_range(n_amplicons, rank, world):
    """Contiguous run of amplicons owned by ``rank`` (coordinate-range partition)."""
    return (n_amplicons * rank) // world, (n_amplicons * (rank + 1)) // world


def shard_bounds(batch, world):
    """Split a coordinate-sorted batch into ``world`` contiguous slices with (nearly) equal BASE
    counts (not read counts: reads differ in length).  Returns world+1 row indices."""
    cum = np.concatenate([[0], np.cumsum(batch.lseq.astype(np.int64))])
    total = int(cum[-1])
    cuts = [int(np.searchsorted(cum, total * k // world, side="left")) for k in range(world + 1)]
    cuts[0], cuts[-1] = 0, batch.n
    return cuts


def shard_bounds_from_lseq(lseq, world):
    """shard_bounds for a batch known by its read lengths only (device-resident batches)."""
    cum = np.concatenate([[0], np.cumsum(np.asarray(lseq).astype(np.int64))])
    total = int(cum[-1])
    cuts = [int(np.searchsorted(cum, total * k // world, side="left")) for k in range(world + 1)]
    cuts[0], cuts[-1] = 0, int(len(lseq))
    return cuts


def exchange_notes(dist, world, seam, error):
    """Before the run's one collective: did every rank get through its share (a rank that raised must not leave the others
    waiting in the all-reduce), and do the shares of neighbouring ranks meet?  ``seam`` = [first, end) offsets of the rank's
    share in the input's inflated stream (ampbam_open_range; [None, None] for an empty share or text input), ``error`` = the
    exception the rank caught (or None).  Returns None when all is well, else the message every rank should fail with."""
    notes = [None] * world
    dist.all_gather_object(notes, (list(seam), None if error is None else "%s: %s" % (type(error).__name__, error)))
    errs = ["rank %d: %s" % (r, e_) for r, (_, e_) in enumerate(notes) if e_]
    if errs:
        return "; ".join(errs)
    ends = [sm for sm, _ in notes if sm[0] is not None]
    for x, y in zip(ends[:-1], ends[1:]):
        if x[1] != y[0]:
            return "the shares of two ranks do not meet (inflated offsets %d / %d)" % (x[1], y[0])
    return None


def reduce_table(dist, table, dst=0):
    """Sum the device table (torch int32/uint32-as-int32 tensor) onto ``dst``."""
    dist.reduce(table, dst=dst, op=dist.ReduceOp.SUM)


def allreduce_table(dist, table):
    """Sum the device table on every rank (each can then make the calls itself: no second collective)."""
    dist.all_reduce(table, op=dist.ReduceOp.SUM)


def allgather_relevant_events(dist, world, pairs):
    """Union of [(ref_pos, string)] lists on every rank."""
    gathered = [None] * world
    dist.all_gather_object(gathered, pairs)
    return [p for part in gathered for p in part]


def gather_relevant_events(dist, rank, world, pairs, dst=0):
    """Union of [(ref_pos, string)] lists on ``dst`` (other ranks get [])."""
    gathered = [None] * world if rank == dst else None
    dist.gather_object(pairs, gathered, dst=dst)
    return [p for part in gathered for p in part] if rank == dst else []


def gather_objects(dist, rank, world, obj, dst=0):
    """[obj of rank 0, obj of rank 1, ...] on ``dst`` (None on the other ranks)."""
    gathered = [None] * world if rank == dst else None
    dist.gather_object(obj, gathered, dst=dst)
    return gathered


def agree_on_positions(dist, rank, positions, src=0):
    """Broadcast the list of insertion-relevant positions decided on ``src``."""
    box = [positions if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]
