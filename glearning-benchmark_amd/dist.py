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


def gather_tokens(ids: torch.Tensor, ln: torch.Tensor, num_graphs: int, pad_id: int, force: bool = False,
                  compact: bool = False, elem_bytes: int = 2, capacity: int = None, rows_impl=None, stats: dict = None,
                  packed=None, ld: int = None, as_packed: bool = False):
    """All-gather the per-rank [G_local, ld] slabs + lengths into the full [G, ld] slab on every rank.

    Ranks hold blocks from block_bounds(); the last blocks may be short, so every rank pads its
    block to ceil(G/P) rows first (one fixed-size all_gather_into_tensor, no size exchange).
    force: issue the collective even in a one-rank group (rehearsal of the N>1 path on a one-GPU box).
    ids: the int32 slab, or the 16-bit slab of ops.sent(..., u16=True) (int16 storage; both exchanges then return the 16-bit
    slab: the padded one moves half the bytes, the compact one packs the rows as they are and re-pads at 16 bits).

    compact=True moves the PACKED form over the links instead of the padded slab (include/gtok.h, "packed rows"): every
    rank packs its rows back to back at `elem_bytes` (2: every SENT / IBTT id fits 16 bits; 4 otherwise) per id
    (gtok_pack_rows), the packed buffers - sized to the largest rank's total, one all_reduce(MAX) of an int, or to
    `capacity` elements when the caller knows a bound (then nothing waits for the host) - and the lengths are gathered, and
    every rank re-pads locally at HBM speed (gtok_unpack_rows).  More than half of a ZINC slab is padding and ids are 32
    bits wide there: 208 MB per corpus become ~46 MB.  Same result as the padded path, bit for bit.
    Every rank's pack status travels WITH its lengths (one more int per rank in the same collective), so all ranks see the
    same verdict: bit 0 = an id did not fit 16 bits, bit 1 = a rank's rows did not fit `capacity` (those rows come out as
    all pad: gtok_unpack_rows never reads beyond a rank's segment).  Without a caller-given capacity a nonzero status
    raises - on every rank, after the collectives, so nobody is left waiting in one; with it, stats["status"] holds the
    verdict as a device tensor for the caller to read when it likes.
    packed (with compact=True): the ops.PackedRows that ops.sent(..., packed=) filled together with `ids` - this rank's rows are
    packed already (by the walk itself: gtok_sent_packed), so nothing is packed here; every rank must pass a buffer of the SAME
    capacity (that is the agreed size: no size exchange), each row's start travels with its length, and gtok_unpack_rows_at
    re-pads.  Verdict as with a caller-given capacity: stats["status"].  `ids` may then be None (ops.sent(..., slab=False) wrote
    no slab): pass the slab width as `ld`.  as_packed=True (with packed=): the gathered rows are NOT re-padded - the call returns
    ((all_packed, row_start), len) with row_start[r] the absolute position of global row r in all_packed (what gtok_collate_packed /
    gtok_collate_batch / gtok_collate_epoch read in place through row_ptr = row_start, and gtok_unpack_rows_at re-pads if a slab is
    wanted after all); rows a rank had to skip come back with length 0.
    rows_impl: the module providing row_offsets / pack_rows / unpack_rows (default: ops, i.e. the HIP kernels; the CPU
    tests of the collective pass the oracle's).  stats: a dict that receives the bytes each rank contributed."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return ids, ln
    world = dist.get_world_size()
    per = -(-num_graphs // world)
    if ids is None:
        if packed is None or not compact or ld is None:
            raise ValueError("gather_tokens without a slab needs compact=True, packed= and ld=")
        if ln.numel() < per:
            ln = torch.cat([ln.reshape(-1), torch.zeros(per - ln.numel(), dtype=ln.dtype, device=ln.device)])
        if rows_impl is None:
            from . import ops as rows_impl
        return _gather_prepacked(None, ln.reshape(-1).contiguous(), num_graphs, pad_id, packed, rows_impl, stats, world, per, int(ld), as_packed)
    ld = ids.shape[1]
    if ids.shape[0] < per:
        fill = per - ids.shape[0]
        ids = torch.cat([ids, torch.full((fill, ld), pad_id, dtype=ids.dtype, device=ids.device)])
        ln = torch.cat([ln, torch.zeros(fill, dtype=ln.dtype, device=ln.device)])
    ids, ln = ids.contiguous(), ln.contiguous()
    if not compact:
        all_ln = torch.empty((world * per,), dtype=ln.dtype, device=ln.device)
        all_ids = torch.empty((world * per, ld), dtype=ids.dtype, device=ids.device)
        dist.all_gather_into_tensor(all_ids.view(torch.uint8), ids.view(torch.uint8))     # bytes: every backend moves uint8
        dist.all_gather_into_tensor(all_ln, ln)
        if stats is not None:
            stats.update(bytes_sent_per_rank=ids.numel() * ids.element_size() + ln.numel() * 4, compact=False)
        return all_ids[:num_graphs], all_ln[:num_graphs]
    if rows_impl is None:
        from . import ops as rows_impl
    if packed is not None:
        return _gather_prepacked(ids, ln, num_graphs, pad_id, packed, rows_impl, stats, world, per, ld, as_packed)
    pack = rows_impl.pack_rows_u16 if ids.dtype == torch.int16 else rows_impl.pack_rows
    caller_bound = capacity is not None
    row_ptr = None
    if not caller_bound:
        row_ptr = rows_impl.row_offsets(ln, ld)
        capacity = all_reduce_max_int(int(row_ptr[-1]), ids.device)
    capacity = max(8, -(-int(capacity) // 8) * 8)           # keeps every rank's segment 16-byte aligned
    # (a caller-given bound - e.g. last epoch's size plus a margin - means no size exchange, no host round trip at all, and
    # offsets + packing in ONE pass: gtok_pack_rows_scan)
    packed, _, status = pack(ids, ln, row_ptr, elem_bytes, capacity=capacity, check_status=False)
    all_packed = torch.empty(world * capacity, dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(all_packed.view(torch.uint8), packed[:capacity].contiguous().view(torch.uint8))
    ext = torch.cat([ln, status.to(ln.dtype).reshape(1)])   # lengths + this rank's pack status, one collective
    all_ext = torch.empty((world * (per + 1),), dtype=ln.dtype, device=ln.device)
    dist.all_gather_into_tensor(all_ext, ext)
    all_ext = all_ext.view(world, per + 1)
    all_ln = all_ext[:, :per].reshape(-1).contiguous()
    all_ptr = rows_impl.row_offsets(all_ln, ld)
    ustatus = torch.zeros(1, dtype=torch.int32, device=ln.device)
    # 16-bit rows in, 16-bit slab out (as the padded exchange does): the re-padding pass writes half the bytes
    u16 = ids.dtype == torch.int16
    all_ids = rows_impl.unpack_rows(all_packed, all_ptr, all_ln, ld, pad_id, segment_rows=per, segment_stride=capacity, status=ustatus,
                                    **({"u16": True} if u16 else {}))
    verdict = torch.maximum(all_ext[:, per].max().to(torch.int32).reshape(1), ustatus)
    if not caller_bound:
        st = int(verdict.item())
        if st:
            from ._lib import GtokError
            raise GtokError("gather_tokens: " + ("an id does not fit 16 bits (pass elem_bytes=4)" if st & 1 else f"a rank's rows did not fit the agreed size (status {st})"))
    if stats is not None:
        stats.update(bytes_sent_per_rank=capacity * packed.element_size() + (ln.numel() + 1) * 4, compact=True,
                     elem_bytes=elem_bytes, capacity=capacity, status=verdict)
    return all_ids[:num_graphs], all_ln[:num_graphs]


def _gather_prepacked(ids, ln, num_graphs, pad_id, packed, rows_impl, stats, world, per, ld, as_packed=False):
    """the compact exchange of rows the walk has packed already (gather_tokens(packed=)): packed buffer, row starts, lengths + status of
    every rank - three collectives, one re-padding pass"""
    dev = ln.device
    capacity = int(packed.capacity)
    start = packed.row_start[:per]
    if start.numel() < per:                                   # a short last block: rows of length 0
        start = torch.cat([start, torch.zeros(per - start.numel(), dtype=torch.int64, device=dev)])
    all_packed = torch.empty(world * capacity, dtype=packed.buf.dtype, device=dev)
    dist.all_gather_into_tensor(all_packed.view(torch.uint8), packed.buf[:capacity].view(torch.uint8))
    all_start = torch.empty(world * per, dtype=torch.int64, device=dev)                  # the row starts as they are (no widening,
    dist.all_gather_into_tensor(all_start, start.contiguous())                          # no slicing on the way back) ...
    ext = torch.cat([ln, packed.status().to(ln.dtype).reshape(1)])                       # ... lengths + this rank's status, one collective
    all_ext = torch.empty((world, per + 1), dtype=ln.dtype, device=dev)
    dist.all_gather_into_tensor(all_ext.view(-1), ext)
    all_ln = all_ext[:, :per].reshape(-1).contiguous()
    if as_packed:
        # the rows stay where the collective put them: a row's absolute start = its rank's segment + its start there
        seg = torch.arange(world, dtype=torch.int64, device=dev).mul_(capacity).repeat_interleave(per)
        skipped = all_start < 0
        all_start = torch.where(skipped, all_start, all_start + seg)
        all_ln = torch.where(skipped, torch.zeros_like(all_ln), all_ln)
        verdict = all_ext[:, per].max().to(torch.int32).reshape(1)
        if stats is not None:
            stats.update(bytes_sent_per_rank=capacity * packed.buf.element_size() + per * 8 + (per + 1) * 4, compact=True,
                         elem_bytes=packed.buf.element_size(), capacity=capacity, status=verdict, prepacked=True, as_packed=True)
        return (all_packed, all_start[:num_graphs]), all_ln[:num_graphs]
    ustatus = torch.zeros(1, dtype=torch.int32, device=dev)
    u16 = packed.buf.dtype == torch.int16
    all_ids = rows_impl.unpack_rows_at(all_packed, all_start, all_ln, ld, pad_id, segment_rows=per, segment_stride=capacity, status=ustatus,
                                       **({"u16": True} if u16 else {}))
    verdict = torch.maximum(all_ext[:, per].max().to(torch.int32).reshape(1), ustatus)
    if stats is not None:
        stats.update(bytes_sent_per_rank=capacity * packed.buf.element_size() + per * 8 + (per + 1) * 4, compact=True,
                     elem_bytes=packed.buf.element_size(), capacity=capacity, status=verdict, prepacked=True)
    return all_ids[:num_graphs], all_ln[:num_graphs]


def reduce_vocab_stats(count: torch.Tensor, first: torch.Tensor):
    """Combine the ranks' node-id token statistics (ops.vocab_stats_synth on each rank's block, graph_base = the
    block's first global index): counts add, first positions take the minimum.  In place; every rank ends with
    the corpus-wide tables and builds the same vocab (data_loader.vocab_from_stats)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(count, op=dist.ReduceOp.SUM)
        dist.all_reduce(first, op=dist.ReduceOp.MIN)
    return count, first
