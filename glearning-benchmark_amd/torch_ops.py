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
