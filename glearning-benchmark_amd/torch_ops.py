"""`torch.ops.gtok.*` — the tokenizer kernels registered as PyTorch custom ops (device_types="cuda" only:
there is deliberately no CPU implementation, a CPU tensor raises NotImplementedError).

The ops take the batched-CSR arrays as plain tensors (layout: include/gtok.h); `ops.py` holds the friendlier
GraphBatch-level wrappers, both end in the same C-ABI calls.  Fake (meta) implementations give output shapes,
so the ops trace under torch.compile / FakeTensor without touching the GPU.
"""
import sys
import weakref
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from . import ops as _ops
from .csr import GraphBatch


# A raw-tensor call carries no GraphBatch: the batch object - and with it everything ops.sent prepares once per resident
# batch (the verified GTOK_CSR_SIMPLE_SYMMETRIC flag, the reordered copy, the byte / bit-matrix mirrors) - is kept here per
# SET OF INPUT TENSORS: the same tensor objects, unchanged since (their version counters), find their batch again; tensors
# that have died or were written to in place do not.  (Round 4 built a fresh flags=0 batch per call: the op then ran
# sent_reg_kernel on ZINC - 8 x the time of the benchmarked sent_lane_kernel - and rebuilt the bit-matrix mirror every call.)
_BATCHES: "dict[tuple, tuple]" = {}
_BATCHES_MAX = 8
_PREPARED: "dict[tuple, tuple]" = {}      # the same for batches the caller prepared (csr_prepare outputs passed back in)


def _batch(node_ptr, edge_ptr, rowptr, col, eorder, nattr, eattr, max_nodes, max_edges, verify: bool = False) -> GraphBatch:
    ts = (node_ptr, edge_ptr, rowptr, col, eorder, nattr, eattr)
    key = tuple(id(t) for t in ts) + (int(max_nodes), int(max_edges))
    hit = _BATCHES.get(key)
    if hit is not None:
        refs, vers, b = hit
        if all((r is None and t is None) or (r is not None and r() is t) for r, t in zip(refs, ts)) \
                and vers == tuple(None if t is None else t._version for t in ts):
            if verify and not b.checked:
                _verify(b)
            return b
        del _BATCHES[key]
    b = GraphBatch(int(node_ptr.numel()) - 1, max_nodes, max_edges, node_ptr, edge_ptr, rowptr, col, eorder, nattr, eattr)
    if verify:
        _verify(b)
    _purge(_BATCHES)
    if len(_BATCHES) >= _BATCHES_MAX:
        _BATCHES.pop(next(iter(_BATCHES)))
    _BATCHES[key] = (tuple(None if t is None else weakref.ref(t) for t in ts), tuple(None if t is None else t._version for t in ts), b)
    return b


def _purge(cache: dict) -> None:
    """Drop the entries nobody else can use any more.  A cached batch holds its input tensors (its C struct points into them), so a
    weak reference alone never dies; what tells a dead entry is that the batch object is the ONLY holder left of every one of its
    tensors (reference count: the cached batches' fields + this loop's variable + getrefcount's argument).  Without this a caller that
    tokenizes corpus after corpus through the raw-tensor ops would keep up to _BATCHES_MAX dead corpora resident."""
    names = ("node_ptr", "edge_ptr", "rowptr", "col", "eorder", "nattr", "eattr", "graph_ids", "unit_ptr")
    in_cache: dict = {}                       # tensor -> how many cached batches hold it (two entries may share their tensors)
    for entry in cache.values():
        for n in names:
            t = getattr(entry[-1], n)
            if t is not None:
                in_cache[id(t)] = in_cache.get(id(t), 0) + 1
            del t
    for key in list(cache):
        b = cache[key][-1]
        fields = [getattr(b, n) for n in names]
        held = False
        while fields:
            t = fields.pop()
            if t is not None and sys.getrefcount(t) > 2 + in_cache[id(t)]:
                held = True
                break
            del t
        if not held:
            del cache[key]


def _verify(b: GraphBatch) -> None:
    """gtok_csr_check once per cached batch: what the flag claims is established on the device (one 32-byte read-back)."""
    b.checked = True
    if b.num_graphs == 0 or b.col.device.type != "cuda" or b.num_edges_total == 0:
        return
    r = _ops.csr_check(b)
    if r["violations"] == 0 and r["max_nodes"] <= b.max_nodes and r["max_edges"] <= b.max_edges:
        b.flags |= _lib.CSR_SIMPLE_SYMMETRIC
        b.max_degree = r["max_degree"]


def _prepared_batch(node_ptr, edge_ptr, rowptr, col, nattr, eattr, max_nodes, max_edges, graph_ids, unit_ptr, unit_info, rowptr8, col8,
                    adj_rows, adj_planes, lane_order, layout) -> GraphBatch:
    """The batch a caller has prepared itself (csr_prepare / csr_adjbits outputs passed back in): nothing is built or cached."""
    if layout is None or len(layout) != len(LAYOUT_FIELDS):
        raise ValueError(f"layout must hold {LAYOUT_FIELDS}")
    # (the batch object - and the C struct it caches - is kept per set of tensor objects, like _batch's: a call costs ~30 us of
    # dispatcher before this function is reached, and a ZINC-full launch lasts 75)
    ts = (node_ptr, edge_ptr, rowptr, col, nattr, eattr, graph_ids, unit_ptr, unit_info, rowptr8, col8, adj_rows, adj_planes, lane_order)
    key = tuple(id(t) for t in ts) + (int(max_nodes), int(max_edges)) + tuple(int(v) for v in layout)
    hit = _PREPARED.get(key)
    if hit is not None:
        refs, b = hit
        if all((r is None and t is None) or (r is not None and r() is t) for r, t in zip(refs, ts)):
            return b
        del _PREPARED[key]
    L = dict(zip(LAYOUT_FIELDS, (int(v) for v in layout)))
    b = GraphBatch(int(node_ptr.numel()) - 1, max_nodes, max_edges, node_ptr, edge_ptr, rowptr, col, None, nattr, eattr,
                   L["flags"], L["chunk_nodes"], L["chunk_edges"], L["max_degree"])
    b.checked = b.prepared = True
    b.rowptr8, b.col8 = rowptr8, col8
    if graph_ids is not None:
        if unit_ptr is None:
            raise ValueError("graph_ids comes with unit_ptr (both are csr_prepare outputs)")
        b.graph_ids, b.unit_ptr, b.unit_info, b.num_units = graph_ids, unit_ptr, unit_info, L["num_units"]
    if adj_rows is not None:
        b.adj_rows, b.adj_planes, b.lane_order = adj_rows, adj_planes, lane_order
        b.adj_words, b.adj_max_degree = L["adj_words"], L["adj_max_degree"]
    _purge(_PREPARED)
    if len(_PREPARED) >= _BATCHES_MAX:
        _PREPARED.pop(next(iter(_PREPARED)))
    _PREPARED[key] = (tuple(None if t is None else weakref.ref(t) for t in ts), b)
    return b


LAYOUT_FIELDS = ("flags", "max_degree", "num_units", "chunk_nodes", "chunk_edges", "adj_words", "adj_max_degree")


@torch.library.custom_op("gtok::csr_prepare", mutates_args=(), device_types="cuda")
def csr_prepare(node_ptr: Tensor, edge_ptr: Tensor, rowptr: Tensor, col: Tensor, nattr: Optional[Tensor], eattr: Optional[Tensor],
                max_nodes: int, max_edges: int) -> List[Tensor]:
    """Everything gtok_sent's fastest kernel for a batch of small graphs needs, made ONCE per resident batch on the device
    (gtok_csr_lane_sort with its check): -> [layout (int64 [7] on the HOST: LAYOUT_FIELDS), node_ptr, edge_ptr, rowptr, col,
    nattr, eattr, graph_ids, unit_ptr, unit_info, rowptr8, col8] - the batch reordered by expected walk length with its unit
    table and byte mirror.  Pass the arrays back to gtok::sent / gtok::sent_epochs in place of the originals, together with
    graph_ids ... col8 and layout=layout.tolist() (prepared_args() does that): rows, lengths and RNG identities are those of
    the batch in dataset order.  A batch the lane-per-graph kernel does not take (more than 64 nodes / 255 entries per graph,
    or not simple and symmetric) comes back as empty arrays with layout[2] (num_units) == 0: keep calling with the
    originals and flags = layout[0]."""
    b = GraphBatch(int(node_ptr.numel()) - 1, max_nodes, max_edges, node_ptr, edge_ptr, rowptr, col, None, nattr, eattr)
    dev = col.device
    e = lambda dt: torch.empty(0, dtype=dt, device=dev)
    sb = _ops.lane_sorted(b, verify=True)
    if sb is None:
        flags = maxdeg = 0
        if b.num_graphs and b.num_edges_total:
            r = _ops.csr_check(b)
            ok = r["violations"] == 0 and r["max_nodes"] <= max_nodes and r["max_edges"] <= max_edges
            flags, maxdeg = (_lib.CSR_SIMPLE_SYMMETRIC if ok else 0), r["max_degree"]
        lay = torch.tensor([flags, maxdeg, 0, 0, 0, 0, 0], dtype=torch.int64)
        return [lay, e(torch.int32), e(torch.int64), e(torch.int32), e(torch.int32), e(torch.uint8), e(torch.uint8), e(torch.int32),
                e(torch.int32), e(torch.int32), e(torch.uint8), e(torch.uint8)]
    lay = torch.tensor([sb.flags, sb.max_degree, sb.num_units, sb.chunk_nodes, sb.chunk_edges, 0, 0], dtype=torch.int64)
    u8 = lambda t: e(torch.uint8) if t is None else t
    return [lay, sb.node_ptr, sb.edge_ptr, sb.rowptr, sb.col, u8(sb.nattr), u8(sb.eattr), sb.graph_ids, sb.unit_ptr, sb.unit_info.reshape(-1),
            u8(sb.rowptr8), u8(sb.col8)]


@csr_prepare.register_fake
def _(node_ptr, edge_ptr, rowptr, col, nattr, eattr, max_nodes, max_edges):
    ctx = torch.library.get_ctx()
    G, U = node_ptr.shape[0] - 1, ctx.new_dynamic_size()
    e = lambda n, dt: col.new_empty((n,), dtype=dt)
    return [torch.empty(7, dtype=torch.int64, device="cpu"), e(G + 1, torch.int32), e(G + 1, torch.int64), e(rowptr.shape[0], torch.int32),
            e(col.shape[0], torch.int32), e(0 if nattr is None else nattr.shape[0], torch.uint8), e(0 if eattr is None else eattr.shape[0], torch.uint8),
            e(G, torch.int32), e(U, torch.int32), e(ctx.new_dynamic_size(), torch.int32), e(rowptr.shape[0] + 16, torch.uint8), e(col.shape[0] + 16, torch.uint8)]


def prepared_args(node_ptr, edge_ptr, rowptr, col, nattr, eattr, max_nodes: int, max_edges: int) -> dict:
    """torch.ops.gtok.csr_prepare(...) as the keyword arguments of torch.ops.gtok.sent / sent_epochs: the batch arrays (the
    reordered ones when the batch qualifies, else the originals), the unit table and mirrors, and `layout`."""
    lay, np2, ep2, rp2, c2, na2, ea2, gid, up, ui, r8, c8 = torch.ops.gtok.csr_prepare(node_ptr, edge_ptr, rowptr, col, nattr, eattr, max_nodes, max_edges)
    layout = [int(v) for v in lay.tolist()]
    if layout[2] == 0:
        return dict(node_ptr=node_ptr, edge_ptr=edge_ptr, rowptr=rowptr, col=col, nattr=nattr, eattr=eattr, max_nodes=max_nodes,
                    max_edges=max_edges, layout=layout)
    opt = lambda t: t if t.numel() else None
    return dict(node_ptr=np2, edge_ptr=ep2, rowptr=rp2, col=c2, nattr=opt(na2) if nattr is not None else None,
                eattr=opt(ea2) if eattr is not None else None, max_nodes=max_nodes, max_edges=max_edges, graph_ids=gid, unit_ptr=up,
                unit_info=ui, rowptr8=opt(r8), col8=opt(c8), layout=layout)


def _sent_batch(node_ptr, edge_ptr, rowptr, col, nattr, eattr, max_nodes, max_edges, graph_ids, unit_ptr, unit_info, rowptr8, col8,
                adj_rows, adj_planes, lane_order, layout) -> GraphBatch:
    if layout is not None:
        return _prepared_batch(node_ptr, edge_ptr, rowptr, col, nattr, eattr, max_nodes, max_edges, graph_ids, unit_ptr, unit_info,
                               rowptr8, col8, adj_rows, adj_planes, lane_order, layout)
    if any(t is not None for t in (graph_ids, unit_ptr, unit_info, rowptr8, col8, adj_rows, adj_planes, lane_order)):
        raise ValueError("prepared arrays come with `layout` (torch.ops.gtok.csr_prepare / prepared_args)")
    return _batch(node_ptr, edge_ptr, rowptr, col, None, nattr, eattr, max_nodes, max_edges, verify=True)


# gtok::sent / gtok::sent_epochs are registered through torch.library.Library (schema + a CUDA kernel + a fake kernel) rather than
# through torch.library.custom_op: an epoch of ZINC-full is a 75 us launch, and custom_op's wrapper costs ~30-38 us per call on
# the host where this route costs ~15 - with it the op was host-bound (0.082 ms per step against the kernel's 0.075).
_FRAG = torch.library.Library("gtok", "FRAGMENT")
_PREPARED_SCHEMA = ("Tensor? graph_ids=None, Tensor? unit_ptr=None, Tensor? unit_info=None, Tensor? rowptr8=None, Tensor? col8=None, "
                    "Tensor? adj_rows=None, Tensor? adj_planes=None, Tensor? lane_order=None, int[]? layout=None")


def sent(node_ptr: Tensor, edge_ptr: Tensor, rowptr: Tensor, col: Tensor, nattr: Optional[Tensor],
         eattr: Optional[Tensor], query: Optional[Tensor], max_nodes: int, max_edges: int, max_num_nodes: int,
         max_len: int, ld: int, seed: int, epoch: int, labeled: bool, num_node_types: int, num_edge_types: int,
         remap_zinc: bool, pad_id: int, graph_base: int, graph_ids: Optional[Tensor] = None, unit_ptr: Optional[Tensor] = None,
         unit_info: Optional[Tensor] = None, rowptr8: Optional[Tensor] = None, col8: Optional[Tensor] = None,
         adj_rows: Optional[Tensor] = None, adj_planes: Optional[Tensor] = None, lane_order: Optional[Tensor] = None,
         layout: Optional[List[int]] = None) -> Tuple[Tensor, Tensor]:
    """gtok_sent.  Without `layout` the batch behind these tensors is prepared on the first call (verified on the device,
    reordered / mirrored as the kernel choice needs) and found again on later calls with the SAME tensor objects; with
    `layout` (csr_prepare / prepared_args) the call uses exactly what it is given."""
    b = _sent_batch(node_ptr, edge_ptr, rowptr, col, nattr, eattr, max_nodes, max_edges, graph_ids, unit_ptr, unit_info, rowptr8, col8,
                    adj_rows, adj_planes, lane_order, layout)
    return _ops.sent(b, max_num_nodes, max_len, seed, epoch, labeled=labeled, num_node_types=num_node_types,
                     num_edge_types=num_edge_types, remap_zinc=remap_zinc, pad_id=pad_id, graph_base=graph_base,
                     query=query, ld=ld)


def _sent_fake(node_ptr, edge_ptr, rowptr, col, nattr, eattr, query, max_nodes, max_edges, max_num_nodes, max_len, ld, seed,
               epoch, labeled, num_node_types, num_edge_types, remap_zinc, pad_id, graph_base, graph_ids=None, unit_ptr=None, unit_info=None,
               rowptr8=None, col8=None, adj_rows=None, adj_planes=None, lane_order=None, layout=None):
    G = node_ptr.shape[0] - 1
    return node_ptr.new_empty((G, ld), dtype=torch.int32), node_ptr.new_empty((G,), dtype=torch.int32)


_FRAG.define("sent(Tensor node_ptr, Tensor edge_ptr, Tensor rowptr, Tensor col, Tensor? nattr, Tensor? eattr, Tensor? query, int max_nodes, "
             "int max_edges, int max_num_nodes, int max_len, int ld, int seed, int epoch, bool labeled, int num_node_types, int num_edge_types, "
             "bool remap_zinc, int pad_id, int graph_base, " + _PREPARED_SCHEMA + ") -> (Tensor, Tensor)")
_FRAG.impl("sent", sent, "CUDA")
torch.library.register_fake("gtok::sent", _sent_fake, lib=_FRAG)


def sent_epochs(node_ptr: Tensor, edge_ptr: Tensor, rowptr: Tensor, col: Tensor, nattr: Optional[Tensor],
                eattr: Optional[Tensor], query: Optional[Tensor], max_nodes: int, max_edges: int, max_num_nodes: int,
                max_len: int, ld: int, seed: int, epoch: int, epochs: int, labeled: bool, num_node_types: int, num_edge_types: int,
                remap_zinc: bool, pad_id: int, graph_base: int, pad: bool, u16: bool, graph_ids: Optional[Tensor] = None,
                unit_ptr: Optional[Tensor] = None, unit_info: Optional[Tensor] = None, rowptr8: Optional[Tensor] = None,
                col8: Optional[Tensor] = None, adj_rows: Optional[Tensor] = None, adj_planes: Optional[Tensor] = None,
                lane_order: Optional[Tensor] = None, layout: Optional[List[int]] = None) -> Tuple[Tensor, Tensor]:
    """gtok_sent with ABI v4's epoch_count and row flags: epochs epoch .. epoch + epochs - 1 in ONE launch ->
    (ids [epochs * G, ld] int32 - or int16 storage holding 16-bit ids when u16 -, len int32 [epochs * G]), epoch-major;
    pad=False leaves the pad tails unwritten (GTOK_SENT_NO_PAD).  Prepared arrays / `layout`: see gtok::sent."""
    b = _sent_batch(node_ptr, edge_ptr, rowptr, col, nattr, eattr, max_nodes, max_edges, graph_ids, unit_ptr, unit_info, rowptr8, col8,
                    adj_rows, adj_planes, lane_order, layout)
    ids, ln = _ops.sent(b, max_num_nodes, max_len, seed, epoch, labeled=labeled, num_node_types=num_node_types,
                        num_edge_types=num_edge_types, remap_zinc=remap_zinc, pad_id=pad_id, graph_base=graph_base,
                        query=query, ld=ld, pad=pad, epochs=epochs, u16=u16)
    return ids.reshape(-1, ld), ln.reshape(-1)


def _sent_epochs_fake(node_ptr, edge_ptr, rowptr, col, nattr, eattr, query, max_nodes, max_edges, max_num_nodes, max_len, ld, seed,
                      epoch, epochs, labeled, num_node_types, num_edge_types, remap_zinc, pad_id, graph_base, pad, u16, graph_ids=None,
                      unit_ptr=None, unit_info=None, rowptr8=None, col8=None, adj_rows=None, adj_planes=None, lane_order=None, layout=None):
    rows = (node_ptr.shape[0] - 1) * max(1, epochs)
    return node_ptr.new_empty((rows, ld), dtype=torch.int16 if u16 else torch.int32), node_ptr.new_empty((rows,), dtype=torch.int32)


_FRAG.define("sent_epochs(Tensor node_ptr, Tensor edge_ptr, Tensor rowptr, Tensor col, Tensor? nattr, Tensor? eattr, Tensor? query, int max_nodes, "
             "int max_edges, int max_num_nodes, int max_len, int ld, int seed, int epoch, int epochs, bool labeled, int num_node_types, "
             "int num_edge_types, bool remap_zinc, int pad_id, int graph_base, bool pad, bool u16, " + _PREPARED_SCHEMA + ") -> (Tensor, Tensor)")
_FRAG.impl("sent_epochs", sent_epochs, "CUDA")
torch.library.register_fake("gtok::sent_epochs", _sent_epochs_fake, lib=_FRAG)


def sent_packed(node_ptr: Tensor, edge_ptr: Tensor, rowptr: Tensor, col: Tensor, nattr: Optional[Tensor],
                eattr: Optional[Tensor], query: Optional[Tensor], max_nodes: int, max_edges: int, max_num_nodes: int,
                max_len: int, ld: int, seed: int, epoch: int, epochs: int, labeled: bool, num_node_types: int, num_edge_types: int,
                remap_zinc: bool, pad_id: int, graph_base: int, u16: bool, capacity: int, graph_ids: Optional[Tensor] = None,
                unit_ptr: Optional[Tensor] = None, unit_info: Optional[Tensor] = None, rowptr8: Optional[Tensor] = None,
                col8: Optional[Tensor] = None, adj_rows: Optional[Tensor] = None, adj_planes: Optional[Tensor] = None,
                lane_order: Optional[Tensor] = None, layout: Optional[List[int]] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """gtok_sent_packed with GTOK_SENT_PACK_ONLY (ABI v6): the walk of `epochs` epochs packs its rows itself and writes no slab ->
    (len int32 [epochs * G], packed [capacity rounded up to 8] ids - int16 storage when u16 -, row_start int64 [epochs * G], state
    int64 [2] = (ids used, status bits: 2 = rows did not fit)).  Readers: gtok::unpack_rows_at, gtok::collate_packed with row_ptr =
    row_start.  Batches another kernel walks go through a temporary slab and gtok_pack_rows_scan (same outputs, dataset order)."""
    b = _sent_batch(node_ptr, edge_ptr, rowptr, col, nattr, eattr, max_nodes, max_edges, graph_ids, unit_ptr, unit_info, rowptr8, col8,
                    adj_rows, adj_planes, lane_order, layout)
    rows = b.num_graphs * max(1, epochs)
    pk = _ops.PackedRows(rows, capacity, u16, b.device)
    _, ln = _ops.sent(b, max_num_nodes, max_len, seed, epoch, labeled=labeled, num_node_types=num_node_types,
                      num_edge_types=num_edge_types, remap_zinc=remap_zinc, pad_id=pad_id, graph_base=graph_base,
                      query=query, ld=ld, epochs=epochs, u16=u16, packed=pk, slab=False)
    return ln.reshape(-1), pk.buf, pk.row_start[:rows], torch.stack([pk.used(), pk.state[0]])


def _sent_packed_fake(node_ptr, edge_ptr, rowptr, col, nattr, eattr, query, max_nodes, max_edges, max_num_nodes, max_len, ld, seed,
                      epoch, epochs, labeled, num_node_types, num_edge_types, remap_zinc, pad_id, graph_base, u16, capacity, graph_ids=None,
                      unit_ptr=None, unit_info=None, rowptr8=None, col8=None, adj_rows=None, adj_planes=None, lane_order=None, layout=None):
    rows = (node_ptr.shape[0] - 1) * max(1, epochs)
    cap = max(8, -(-capacity // 8) * 8)
    return (node_ptr.new_empty((rows,), dtype=torch.int32), node_ptr.new_empty((cap,), dtype=torch.int16 if u16 else torch.int32),
            node_ptr.new_empty((rows,), dtype=torch.int64), node_ptr.new_empty((2,), dtype=torch.int64))


_FRAG.define("sent_packed(Tensor node_ptr, Tensor edge_ptr, Tensor rowptr, Tensor col, Tensor? nattr, Tensor? eattr, Tensor? query, int max_nodes, "
             "int max_edges, int max_num_nodes, int max_len, int ld, int seed, int epoch, int epochs, bool labeled, int num_node_types, "
             "int num_edge_types, bool remap_zinc, int pad_id, int graph_base, bool u16, int capacity, " + _PREPARED_SCHEMA
             + ") -> (Tensor, Tensor, Tensor, Tensor)")
_FRAG.impl("sent_packed", sent_packed, "CUDA")
torch.library.register_fake("gtok::sent_packed", _sent_packed_fake, lib=_FRAG)


@torch.library.custom_op("gtok::unpack_rows_at", mutates_args=(), device_types="cuda")
def unpack_rows_at(packed: Tensor, row_start: Tensor, lens: Tensor, ld: int, pad_id: int, segment_rows: int, segment_stride: int,
                   u16: bool) -> Tensor:
    """gtok_unpack_rows_at: rows with explicit starts (gtok::sent_packed's row_start) -> [rows, ld] slab, 16-bit when u16."""
    return _ops.unpack_rows_at(packed, row_start, lens, ld, pad_id, segment_rows, segment_stride, u16=u16)


@unpack_rows_at.register_fake
def _(packed, row_start, lens, ld, pad_id, segment_rows, segment_stride, u16):
    return packed.new_empty((lens.shape[0], ld), dtype=torch.int16 if u16 else torch.int32)


@torch.library.custom_op("gtok::ibtt_zinc", mutates_args=(), device_types="cuda")
def ibtt_zinc(node_ptr: Tensor, edge_ptr: Tensor, rowptr: Tensor, col: Tensor, eorder: Optional[Tensor],
              nattr: Optional[Tensor], eattr: Optional[Tensor], lut: Tensor, max_nodes: int, max_edges: int,
              max_len: int, pad_id: int, ld: int) -> Tuple[Tensor, Tensor]:
    b = _batch(node_ptr, edge_ptr, rowptr, col, eorder, nattr, eattr, max_nodes, max_edges)
    return _ops.ibtt_zinc(b, lut, max_len, pad_id, ld=ld)


@ibtt_zinc.register_fake
def _(node_ptr, edge_ptr, rowptr, col, eorder, nattr, eattr, lut, max_nodes, max_edges, max_len, pad_id, ld):
    G = node_ptr.shape[0] - 1
    return node_ptr.new_empty((G, ld), dtype=torch.int32), node_ptr.new_empty((G,), dtype=torch.int32)


@torch.library.custom_op("gtok::ibtt_synth", mutates_args=(), device_types="cuda")
def ibtt_synth(node_ptr: Tensor, edge_ptr: Tensor, rowptr: Tensor, col: Tensor, eorder: Optional[Tensor], lut: Tensor,
               query: Optional[Tensor], max_nodes: int, max_edges: int, max_len: int, pad_id: int,
               ld: int) -> Tuple[Tensor, Tensor]:
    b = _batch(node_ptr, edge_ptr, rowptr, col, eorder, None, None, max_nodes, max_edges)
    return _ops.ibtt_synth(b, lut, query, max_len, pad_id, ld=ld)


@ibtt_synth.register_fake
def _(node_ptr, edge_ptr, rowptr, col, eorder, lut, query, max_nodes, max_edges, max_len, pad_id, ld):
    G = node_ptr.shape[0] - 1
    return node_ptr.new_empty((G, ld), dtype=torch.int32), node_ptr.new_empty((G,), dtype=torch.int32)


@torch.library.custom_op("gtok::remap_zinc", mutates_args=(), device_types="cuda")
def remap_zinc(ids: Tensor, lens: Tensor, idx_offset: int, node_idx_offset: int, edge_idx_offset: int) -> Tensor:
    return _ops.remap_zinc(ids, lens, idx_offset, node_idx_offset, edge_idx_offset)


@remap_zinc.register_fake
def _(ids, lens, idx_offset, node_idx_offset, edge_idx_offset):
    return torch.empty_like(ids)


@torch.library.custom_op("gtok::collate", mutates_args=(), device_types="cuda")
def collate(ids: Tensor, lens: Tensor, index: Tensor, pad_id: int, out_ld: int) -> Tuple[Tensor, Tensor]:
    return _ops.collate(ids, lens, index, pad_id, out_ld)


@collate.register_fake
def _(ids, lens, index, pad_id, out_ld):
    B = index.shape[0]
    return ids.new_empty((B, out_ld), dtype=torch.int64), ids.new_empty((B, out_ld), dtype=torch.bool)


class _Table:
    """ops.VocabTable rebuilt from its tensors (the custom op takes tensors only)."""

    def __init__(self, key_off, key_len, ids, key_bytes, pad_id):
        self.key_off, self.key_len, self.ids, self.key_bytes = key_off, key_len, ids, key_bytes
        self.capacity, self.pad_id = int(key_off.numel()), int(pad_id)

    c_struct = _ops.VocabTable.c_struct


@torch.library.custom_op("gtok::text_to_ids", mutates_args=(), device_types="cuda")
def text_to_ids(text_bytes: Tensor, text_ptr: Tensor, key_off: Tensor, key_len: Tensor, key_id: Tensor, key_bytes: Tensor,
                pad_id: int, max_len: int, strip_label: bool, ld: int) -> Tuple[Tensor, Tensor]:
    """TokenDataset's text -> ids (data_loader.py:465-484); the vocab travels as the open-addressing table of
    ops.VocabTable (key_off / key_len / key_id int32 [capacity], key_bytes uint8)."""
    return _ops.text_to_ids(text_bytes, text_ptr, _Table(key_off, key_len, key_id, key_bytes, pad_id), max_len, strip_label, ld=ld)


@text_to_ids.register_fake
def _(text_bytes, text_ptr, key_off, key_len, key_id, key_bytes, pad_id, max_len, strip_label, ld):
    G = text_ptr.shape[0] - 1
    return text_bytes.new_empty((G, ld), dtype=torch.int32), text_bytes.new_empty((G,), dtype=torch.int32)


@torch.library.custom_op("gtok::find_token", mutates_args=(), device_types="cuda")
def find_token(x: Tensor, token: int) -> Tensor:
    return _ops.find_token(x, token)


@find_token.register_fake
def _(x, token):
    return x.new_empty((x.shape[0],), dtype=torch.int32)


@torch.library.custom_op("gtok::vocab_stats_synth", mutates_args=(), device_types="cuda")
def vocab_stats_synth(node_ptr: Tensor, edge_ptr: Tensor, rowptr: Tensor, col: Tensor, eorder: Optional[Tensor],
                      query_nodes: Optional[Tensor], max_nodes: int, max_edges: int, num_ids: int,
                      graph_base: int) -> Tuple[Tensor, Tensor]:
    b = _batch(node_ptr, edge_ptr, rowptr, col, eorder, None, None, max_nodes, max_edges)
    return _ops.vocab_stats_synth(b, num_ids, query_nodes, graph_base)


@vocab_stats_synth.register_fake
def _(node_ptr, edge_ptr, rowptr, col, eorder, query_nodes, max_nodes, max_edges, num_ids, graph_base):
    return node_ptr.new_empty((num_ids,), dtype=torch.int64), node_ptr.new_empty((num_ids,), dtype=torch.int64)


@torch.library.custom_op("gtok::parse_graph_text", mutates_args=(), device_types="cuda")
def parse_graph_text(text_bytes: Tensor, text_ptr: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(num_edges, num_nodes, query [G,2], label, status, edge_ptr [G+1], src, dst): ops.parse_graph_texts as a tuple;
    src / dst have a data-dependent length (sum of the canonical texts' edge counts)."""
    r = _ops.parse_graph_texts(text_bytes, text_ptr)
    return (r["num_edges"], r["num_nodes"], r["query"], r["label"], r["status"], r["edge_ptr"], r["src"].clone(), r["dst"].clone())


@parse_graph_text.register_fake
def _(text_bytes, text_ptr):
    G = text_ptr.shape[0] - 1
    E = torch.library.get_ctx().new_dynamic_size()
    i32 = lambda *s: text_bytes.new_empty(s, dtype=torch.int32)
    return i32(G), i32(G), i32(G, 2), i32(G), i32(G), text_bytes.new_empty((G + 1,), dtype=torch.int64), i32(E), i32(E)


@torch.library.custom_op("gtok::sent_decode", mutates_args=(), device_types="cuda")
def sent_decode(ids: Tensor, lens: Tensor, max_num_nodes: int, labeled: bool, num_node_types: int, edge_cap: int,
                node_cap: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(num_nodes, num_edges, status, edge_a, edge_b, edge_type, node_type): ops.sent_decode as a tuple."""
    r = _ops.sent_decode(ids, lens, max_num_nodes, labeled, num_node_types, edge_cap, node_cap)
    return (r["num_nodes"], r["num_edges"], r["status"], r["edge_a"], r["edge_b"], r["edge_type"], r["node_type"])


@sent_decode.register_fake
def _(ids, lens, max_num_nodes, labeled, num_node_types, edge_cap, node_cap):
    G = ids.shape[0]
    i32 = lambda *s: ids.new_empty(s, dtype=torch.int32)
    return i32(G), i32(G), i32(G), i32(G, edge_cap), i32(G, edge_cap), i32(G, edge_cap), i32(G, node_cap)


# ---- packed (ragged) rows: include/gtok.h, "packed rows" -----------------------------------------------------------------
@torch.library.custom_op("gtok::row_offsets", mutates_args=(), device_types="cuda")
def row_offsets(lens: Tensor, ld: int, align: int) -> Tensor:
    return _ops.row_offsets(lens, ld, align)


@row_offsets.register_fake
def _(lens, ld, align):
    return lens.new_empty((lens.shape[0] + 1,), dtype=torch.int64)


@torch.library.custom_op("gtok::pack_rows", mutates_args=(), device_types="cuda")
def pack_rows(ids: Tensor, lens: Tensor, row_ptr: Tensor, elem_bytes: int, capacity: int) -> Tuple[Tensor, Tensor]:
    """(packed int16 | int32 [capacity], status int32 [1]): gtok_pack_rows into a caller-sized buffer; nothing waits
    for the host (status bit 0: an id needs more than 16 bits, bit 1: capacity too small)."""
    packed, _, status = _ops.pack_rows(ids, lens, row_ptr, elem_bytes, capacity=capacity, check_status=False)
    return packed, status


@pack_rows.register_fake
def _(ids, lens, row_ptr, elem_bytes, capacity):
    return (ids.new_empty((max(capacity, 1),), dtype=torch.int16 if elem_bytes == 2 else torch.int32),
            ids.new_empty((1,), dtype=torch.int32))


@torch.library.custom_op("gtok::pack_rows_u16", mutates_args=(), device_types="cuda")
def pack_rows_u16(ids16: Tensor, lens: Tensor, row_ptr: Tensor, elem_bytes: int, capacity: int) -> Tuple[Tensor, Tensor]:
    """gtok_pack_rows_u16: a slab of 16-bit ids (sent_epochs(..., u16=True)) packed at 2 / 4 / 8 bytes per id."""
    packed, _, status = _ops.pack_rows_u16(ids16, lens, row_ptr, elem_bytes, capacity=capacity, check_status=False)
    return packed, status


@pack_rows_u16.register_fake
def _(ids16, lens, row_ptr, elem_bytes, capacity):
    return (ids16.new_empty((max(capacity, 1),), dtype={2: torch.int16, 4: torch.int32, 8: torch.int64}[elem_bytes]),
            ids16.new_empty((1,), dtype=torch.int32))


@torch.library.custom_op("gtok::pack_rows_scan", mutates_args=(), device_types="cuda")
def pack_rows_scan(ids: Tensor, lens: Tensor, elem_bytes: int, capacity: int, align: int) -> Tuple[Tensor, Tensor, Tensor]:
    """gtok_pack_rows_scan: (packed [capacity], row_ptr int64 [rows + 1], status int32 [1]) from an int32 or 16-bit slab in one pass."""
    return _ops.pack_rows_scan(ids, lens, elem_bytes, capacity, align)


@pack_rows_scan.register_fake
def _(ids, lens, elem_bytes, capacity, align):
    return (ids.new_empty((max(capacity, 1),), dtype={2: torch.int16, 4: torch.int32, 8: torch.int64}[elem_bytes]),
            ids.new_empty((lens.shape[0] + 1,), dtype=torch.int64), ids.new_empty((1,), dtype=torch.int32))


@torch.library.custom_op("gtok::unpack_rows", mutates_args=(), device_types="cuda")
def unpack_rows(packed: Tensor, row_ptr: Optional[Tensor], lens: Tensor, ld: int, pad_id: int, segment_rows: int,
                segment_stride: int) -> Tensor:
    """row_ptr None: `packed` is a [rows, ld] slab of 16- / 32-bit ids read in place (the strided form)."""
    return _ops.unpack_rows(packed, row_ptr, lens, ld, pad_id, segment_rows, segment_stride)


@unpack_rows.register_fake
def _(packed, row_ptr, lens, ld, pad_id, segment_rows, segment_stride):
    return packed.new_empty((lens.shape[0], ld), dtype=torch.int32)


@torch.library.custom_op("gtok::collate_batch", mutates_args=(), device_types="cuda")
def collate_batch(packed: Tensor, row_ptr: Optional[Tensor], lens: Tensor, ld: int, index: List[int], pad_id: int, out_ld: int,
                  y: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """gtok_collate_batch: collate_packed for a row list held on the HOST (what a sampler yields) - the indices ride in the launch's
    arguments; y (labels, 4- or 8-byte elements) gathered by the same launch (an empty tensor comes back when y is None)."""
    import numpy as _np
    X, A, Y = _ops.collate_batch(packed, row_ptr, lens, ld, _np.asarray(index, dtype=_np.int64), pad_id, out_ld, y)
    return X, A, Y if Y is not None else packed.new_empty((0,), dtype=torch.float32)


@collate_batch.register_fake
def _(packed, row_ptr, lens, ld, index, pad_id, out_ld, y):
    B = len(index)
    return (packed.new_empty((B, out_ld), dtype=torch.int64), packed.new_empty((B, out_ld), dtype=torch.bool),
            packed.new_empty((B,), dtype=y.dtype) if y is not None else packed.new_empty((0,), dtype=torch.float32))


@torch.library.custom_op("gtok::collate_epoch", mutates_args=(), device_types="cuda")
def collate_epoch(packed: Tensor, row_ptr: Optional[Tensor], lens: Tensor, ld: int, order: Tensor, batch_size: int,
                  pad_id: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """gtok_collate_epoch_plan + gtok_collate_epoch: every batch of an epoch collated by one call -> (X int64 arena, attn bool arena,
    lmax int32 [batches], off int64 [batches + 1], the last two on the host): batch b = X.as_strided((B_b, lmax[b]), (lmax[b], 1), off[b])."""
    X, A, lmax, off = _ops.collate_epoch(packed, row_ptr, lens, ld, order, batch_size, pad_id)
    return X, A, torch.tensor(lmax, dtype=torch.int32), torch.tensor(off, dtype=torch.int64)


@collate_epoch.register_fake
def _(packed, row_ptr, lens, ld, order, batch_size, pad_id):
    n = order.shape[0]
    nb = -(-n // batch_size) if n else 0
    total = torch.library.get_ctx().new_dynamic_size()
    return (packed.new_empty((total,), dtype=torch.int64), packed.new_empty((total,), dtype=torch.bool),
            torch.empty((nb,), dtype=torch.int32), torch.empty((nb + 1,), dtype=torch.int64))


@torch.library.custom_op("gtok::collate_packed", mutates_args=(), device_types="cuda")
def collate_packed(packed: Tensor, row_ptr: Optional[Tensor], lens: Tensor, ld: int, index: Tensor, pad_id: int,
                   out_ld: int) -> Tuple[Tensor, Tensor]:
    return _ops.collate_packed(packed, row_ptr, lens, ld, index, pad_id, out_ld)


@collate_packed.register_fake
def _(packed, row_ptr, lens, ld, index, pad_id, out_ld):
    B = index.shape[0]
    return packed.new_empty((B, out_ld), dtype=torch.int64), packed.new_empty((B, out_ld), dtype=torch.bool)
