"""Multi-GPU: one process per GPU, the corpus sharded by contiguous blocks of graphs.

Every graph is an independent unit: ranks tokenize their block with no data-path
collective, the SENT RNG is keyed by the GLOBAL graph index so results do not depend on the number
of ranks, and one all-gather (RCCL over xGMI on the GPU box, gloo in the CPU tests) reassembles the
padded token slab in dataset order when a caller wants every rank to hold it.
"""
from typing import List, Tuple

import torch
import torch.distributed as dist


def block_bounds(num_graphs: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous blocks of ceil(G/P) graphs; keeps output order == dataset order (the val/test loaders
    of trainer/train_agtt.py:602-607 run with shuffle=False and rely on it)."""
    per = -(-num_graphs // world_size) if world_size > 0 else num_graphs
    return [(min(r * per, num_graphs), min((r + 1) * per, num_graphs)) for r in range(world_size)]


def all_reduce_max_int(value: int, device) -> int:
    """max_num_nodes / slab width agreed across ranks (trainer/train_agtt.py:534 is a full pass)."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def shard_of(batch, rank: int = None, world: int = None):
    """(this rank's block of `batch` as its own GraphBatch, lo, hi): tokenize it with graph_base=lo."""
    if world is None:
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if rank is None:
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    lo, hi = block_bounds(batch.num_graphs, world)[rank]
    return batch.shard(lo, hi), lo, hi


def gather_tokens(ids: torch.Tensor, ln: torch.Tensor, num_graphs: int, pad_id: int, force: bool = False):
    """All-gather the per-rank [G_local, ld] slabs + lengths into the full [G, ld] slab on every rank.

    Ranks hold blocks from block_bounds(); the last blocks may be short, so every rank pads its
    block to ceil(G/P) rows first (one fixed-size all_gather_into_tensor, no size exchange).
    force: issue the collective even in a one-rank group (rehearsal of the N>1 path on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return ids, ln
    world = dist.get_world_size()
    per = -(-num_graphs // world)
    ld = ids.shape[1]
    if ids.shape[0] < per:
        fill = per - ids.shape[0]
        ids = torch.cat([ids, torch.full((fill, ld), pad_id, dtype=ids.dtype, device=ids.device)])
        ln = torch.cat([ln, torch.zeros(fill, dtype=ln.dtype, device=ln.device)])
    all_ids = torch.empty((world * per, ld), dtype=ids.dtype, device=ids.device)
    all_ln = torch.empty((world * per,), dtype=ln.dtype, device=ln.device)
    dist.all_gather_into_tensor(all_ids, ids.contiguous())
    dist.all_gather_into_tensor(all_ln, ln.contiguous())
    return all_ids[:num_graphs], all_ln[:num_graphs]


def reduce_vocab_stats(count: torch.Tensor, first: torch.Tensor):
    """Combine the ranks' node-id token statistics (ops.vocab_stats_synth on each rank's block, graph_base = the
    block's first global index): counts add, first positions take the minimum.  In place; every rank ends with
    the corpus-wide tables and builds the same vocab (data_loader.vocab_from_stats)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(count, op=dist.ReduceOp.SUM)
        dist.all_reduce(first, op=dist.ReduceOp.MIN)
    return count, first
