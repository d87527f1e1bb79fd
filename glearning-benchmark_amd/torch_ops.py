"""`torch.ops.gtok.*` — the tokenizer kernels registered as PyTorch custom ops (device_types="cuda" only:
there is deliberately no CPU implementation, a CPU tensor raises NotImplementedError).

The ops take the batched-CSR arrays as plain tensors (layout: include/gtok.h); `ops.py` holds the friendlier
GraphBatch-level wrappers, both end in the same C-ABI calls.  Fake (meta) implementations give output shapes,
so the ops trace under torch.compile / FakeTensor without touching the GPU.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops as _ops
from .csr import GraphBatch


def _batch(node_ptr, edge_ptr, rowptr, col, eorder, nattr, eattr, max_nodes, max_edges) -> GraphBatch:
    return GraphBatch(int(node_ptr.numel()) - 1, max_nodes, max_edges, node_ptr, edge_ptr, rowptr, col, eorder, nattr, eattr)


@torch.library.custom_op("gtok::sent", mutates_args=(), device_types="cuda")
def sent(node_ptr: Tensor, edge_ptr: Tensor, rowptr: Tensor, col: Tensor, nattr: Optional[Tensor],
         eattr: Optional[Tensor], query: Optional[Tensor], max_nodes: int, max_edges: int, max_num_nodes: int,
         max_len: int, ld: int, seed: int, epoch: int, labeled: bool, num_node_types: int, num_edge_types: int,
         remap_zinc: bool, pad_id: int, graph_base: int) -> Tuple[Tensor, Tensor]:
    b = _batch(node_ptr, edge_ptr, rowptr, col, None, nattr, eattr, max_nodes, max_edges)
    return _ops.sent(b, max_num_nodes, max_len, seed, epoch, labeled=labeled, num_node_types=num_node_types,
                     num_edge_types=num_edge_types, remap_zinc=remap_zinc, pad_id=pad_id, graph_base=graph_base,
                     query=query, ld=ld)


@sent.register_fake
def _(node_ptr, edge_ptr, rowptr, col, nattr, eattr, query, max_nodes, max_edges, max_num_nodes, max_len, ld, seed,
      epoch, labeled, num_node_types, num_edge_types, remap_zinc, pad_id, graph_base):
    G = node_ptr.shape[0] - 1
    return node_ptr.new_empty((G, ld), dtype=torch.int32), node_ptr.new_empty((G,), dtype=torch.int32)


@torch.library.custom_op("gtok::sent_epochs", mutates_args=(), device_types="cuda")
def sent_epochs(node_ptr: Tensor, edge_ptr: Tensor, rowptr: Tensor, col: Tensor, nattr: Optional[Tensor],
                eattr: Optional[Tensor], query: Optional[Tensor], max_nodes: int, max_edges: int, max_num_nodes: int,
                max_len: int, ld: int, seed: int, epoch: int, epochs: int, labeled: bool, num_node_types: int, num_edge_types: int,
                remap_zinc: bool, pad_id: int, graph_base: int, pad: bool, u16: bool) -> Tuple[Tensor, Tensor]:
    """gtok_sent with ABI v4's epoch_count and row flags: epochs epoch .. epoch + epochs - 1 in ONE launch ->
    (ids [epochs * G, ld] int32 - or int16 storage holding 16-bit ids when u16 -, len int32 [epochs * G]), epoch-major;
    pad=False leaves the pad tails unwritten (GTOK_SENT_NO_PAD)."""
    b = _batch(node_ptr, edge_ptr, rowptr, col, None, nattr, eattr, max_nodes, max_edges)
    ids, ln = _ops.sent(b, max_num_nodes, max_len, seed, epoch, labeled=labeled, num_node_types=num_node_types,
                        num_edge_types=num_edge_types, remap_zinc=remap_zinc, pad_id=pad_id, graph_base=graph_base,
                        query=query, ld=ld, pad=pad, epochs=epochs, u16=u16)
    return ids.reshape(-1, ld), ln.reshape(-1)


@sent_epochs.register_fake
def _(node_ptr, edge_ptr, rowptr, col, nattr, eattr, query, max_nodes, max_edges, max_num_nodes, max_len, ld, seed,
      epoch, epochs, labeled, num_node_types, num_edge_types, remap_zinc, pad_id, graph_base, pad, u16):
    rows = (node_ptr.shape[0] - 1) * max(1, epochs)
    return node_ptr.new_empty((rows, ld), dtype=torch.int16 if u16 else torch.int32), node_ptr.new_empty((rows,), dtype=torch.int32)


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


@torch.library.custom_op("gtok::unpack_rows", mutates_args=(), device_types="cuda")
def unpack_rows(packed: Tensor, row_ptr: Optional[Tensor], lens: Tensor, ld: int, pad_id: int, segment_rows: int,
                segment_stride: int) -> Tensor:
    """row_ptr None: `packed` is a [rows, ld] slab of 16- / 32-bit ids read in place (the strided form)."""
    return _ops.unpack_rows(packed, row_ptr, lens, ld, pad_id, segment_rows, segment_stride)


@unpack_rows.register_fake
def _(packed, row_ptr, lens, ld, pad_id, segment_rows, segment_stride):
    return packed.new_empty((lens.shape[0], ld), dtype=torch.int32)


@torch.library.custom_op("gtok::collate_packed", mutates_args=(), device_types="cuda")
def collate_packed(packed: Tensor, row_ptr: Optional[Tensor], lens: Tensor, ld: int, index: Tensor, pad_id: int,
                   out_ld: int) -> Tuple[Tensor, Tensor]:
    return _ops.collate_packed(packed, row_ptr, lens, ld, index, pad_id, out_ld)


@collate_packed.register_fake
def _(packed, row_ptr, lens, ld, index, pad_id, out_ld):
    B = index.shape[0]
    return packed.new_empty((B, out_ld), dtype=torch.int64), packed.new_empty((B, out_ld), dtype=torch.bool)
