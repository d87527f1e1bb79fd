"""Batched CSR container — the form graphs take in HBM (layout: include/gtok.h).

Built once on the host from PyG-style per-graph COO (x, edge_index, edge_attr:
what graph_data_loader/zinc_dataset_autograph.py:51-73 and
graph_data_loader/graph_token_dataset_autograph.py:338-348 hand to the
tokenizer), then resident on the device for every epoch.
"""
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from ._lib import CSR_SIMPLE_SYMMETRIC, GtokCsr


def _to_u8(a: Optional[np.ndarray]) -> Optional[np.ndarray]:
    """Type ids live in HBM as uint8; anything outside 0..254 becomes 255 ("other")."""
    if a is None:
        return None
    a = np.asarray(a).reshape(-1).astype(np.int64, copy=False)
    return np.where((a >= 0) & (a < 255), a, 255).astype(np.uint8)


def _chunk_max(counts: np.ndarray, group: int = 64) -> int:
    """max over aligned groups of `group` graphs of the summed counts."""
    counts = np.asarray(counts, dtype=np.int64)
    if counts.size == 0:
        return 0
    pad = (-counts.size) % group
    return int(np.pad(counts, (0, pad)).reshape(-1, group).sum(1).max())


def _simple_symmetric(gid_e: np.ndarray, src: np.ndarray, dst: np.ndarray, max_nodes: int) -> bool:
    """True iff there is no self-loop, no (graph, u, v) entry is listed twice and every (u, v) has its (v, u)."""
    if (src == dst).any():
        return False
    m = np.int64(max_nodes + 1)
    fwd = np.sort((gid_e * m + src) * m + dst)
    if fwd.size and (fwd[1:] == fwd[:-1]).any():
        return False
    rev = np.sort((gid_e * m + dst) * m + src)
    return bool(np.array_equal(fwd, rev))


@dataclass
class GraphBatch:
    num_graphs: int
    max_nodes: int
    max_edges: int
    node_ptr: torch.Tensor            # int32 [G+1]
    edge_ptr: torch.Tensor            # int64 [G+1]
    rowptr: torch.Tensor              # int32 [sum N + G], local per graph
    col: torch.Tensor                 # int32 [sum E]
    eorder: Optional[torch.Tensor]    # int32 [sum E] or None (identity)
    nattr: Optional[torch.Tensor]     # uint8 [sum N]
    eattr: Optional[torch.Tensor]     # uint8 [sum E]
    flags: int = 0                    # GTOK_CSR_*: properties verified on the host for the whole batch
    chunk_nodes: int = 0              # max over 64-graph groups of sum N_g / sum E_g (LDS sizing of the
    chunk_edges: int = 0              # lane-per-graph kernel); 0 = unknown
    max_degree: int = 0               # longest CSR row (an upper bound is fine); 0 = unknown
    rowptr8: Optional[torch.Tensor] = None   # uint8 mirrors of rowptr / col (device batches of small graphs:
    col8: Optional[torch.Tensor] = None      # ops.pack8, once per resident batch)
    adj_rows: Optional[torch.Tensor] = None      # adjacency bit-matrix mirror (ops.adjbits, once per resident batch):
    adj_planes: Optional[torch.Tensor] = None    # int64 [sum N, W] rows, int64 [G, 8, W] degree bit planes,
    lane_order: Optional[torch.Tensor] = None    # int32 [G] graphs in the order they are dealt to lanes
    adj_words: int = 0
    adj_max_degree: int = 0
    lane_order_len: int = 0                      # the max_len lane_order was made for (ops.sent)
    graph_ids: Optional[torch.Tensor] = None     # a batch reordered for the lane-per-graph SENT kernel (ops.lane_sorted):
    unit_ptr: Optional[torch.Tensor] = None      # int32 [G] dataset index of every slot, int32 [units + 1] slots per wave
    num_units: int = 0
    unit_info: Optional[torch.Tensor] = None     # int32 [units, 8]: per-unit descriptor (include/gtok.h: gtok_csr.unit_info)
    lane_sorted: Optional["GraphBatch"] = None   # the reordered copy of THIS batch (device batches, made on first use)
    adj_unusable: bool = False                   # gtok_csr_adjbits found a closure degree above 255: do not try again
    checked: bool = False                        # flags were established on the device (gtok_csr_check; torch_ops: raw-tensor calls)
    prepared: bool = False                       # the caller built the layouts itself (torch.ops.gtok.csr_prepare): ops.sent prepares nothing

    @property
    def device(self) -> torch.device:
        return self.col.device

    @property
    def num_nodes_total(self) -> int:
        return int(self.rowptr.numel()) - self.num_graphs

    @property
    def num_edges_total(self) -> int:
        return int(self.col.numel())

    def to(self, device) -> "GraphBatch":
        mv = lambda t: None if t is None else t.to(device, non_blocking=True)
        return GraphBatch(self.num_graphs, self.max_nodes, self.max_edges, mv(self.node_ptr), mv(self.edge_ptr),
                          mv(self.rowptr), mv(self.col), mv(self.eorder), mv(self.nattr), mv(self.eattr),
                          self.flags, self.chunk_nodes, self.chunk_edges, self.max_degree, mv(self.rowptr8), mv(self.col8),
                          mv(self.adj_rows), mv(self.adj_planes), mv(self.lane_order), self.adj_words, self.adj_max_degree,
                          0, mv(self.graph_ids), mv(self.unit_ptr), self.num_units, mv(self.unit_info))

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name != "_cs":
            object.__setattr__(self, "_cs", None)     # any change of a field drops the cached C struct

    def c_struct(self) -> GtokCsr:
        """The batch as the C ABI's gtok_csr (include/gtok.h).  Built once and kept until a field is assigned: a launch
        through ops.* costs tens of microseconds of Python, and 27 data_ptr() calls were a fifth of that."""
        cs = self.__dict__.get("_cs")
        if cs is not None:
            return cs
        p = lambda t: None if t is None else t.data_ptr()
        cs = GtokCsr(self.num_graphs, self.max_nodes, self.max_edges, self.flags, p(self.node_ptr), p(self.edge_ptr),
                     p(self.rowptr), p(self.col), p(self.eorder), p(self.nattr), p(self.eattr),
                     self.chunk_nodes, self.chunk_edges, self.max_degree, 0, p(self.rowptr8), p(self.col8),
                     p(self.adj_rows), p(self.adj_planes), p(self.lane_order), self.adj_words, self.adj_max_degree,
                     p(self.graph_ids), p(self.unit_ptr), self.num_units, 0, p(self.unit_info))
        object.__setattr__(self, "_cs", cs)
        return cs

    def node_counts(self) -> torch.Tensor:
        return self.node_ptr[1:] - self.node_ptr[:-1]

    def algorithmic_read_bytes(self, ibtt: bool, labeled: bool) -> int:
        """4(N+1) rowptr + 4E col (+4E order ids: IBTT, only when the batch stores them —
        a row-sorted edge_index needs none) + N + E attrs (labelled)."""
        n, e, g = self.num_nodes_total, self.num_edges_total, self.num_graphs
        b = 4 * (n + g) + 4 * e
        if ibtt and self.eorder is not None:
            b += 4 * e
        if labeled:
            b += n + e
        return b

    def shard(self, lo: int, hi: int) -> "GraphBatch":
        """Graphs [lo, hi) as their own batch (contiguous block sharding); works on host and device batches and
        stays on the batch's device.  Tokenize it with graph_base=lo to get the rows the whole batch would give."""
        n0, n1 = int(self.node_ptr[lo]), int(self.node_ptr[hi])
        e0, e1 = int(self.edge_ptr[lo]), int(self.edge_ptr[hi])
        sl = lambda t, a, b: None if t is None else t[a:b].clone()
        nc = (self.node_ptr[lo + 1:hi + 1] - self.node_ptr[lo:hi])
        ec = (self.edge_ptr[lo + 1:hi + 1] - self.edge_ptr[lo:hi])

        def chunk_max(c):
            if c.numel() == 0:
                return 0
            pad = (-c.numel()) % 64
            return int(torch.nn.functional.pad(c.to(torch.int64), (0, pad)).reshape(-1, 64).sum(1).max())
        return GraphBatch(hi - lo, int(nc.max()) if hi > lo else 0, int(ec.max()) if hi > lo else 0,
                          (self.node_ptr[lo:hi + 1] - n0).clone(), (self.edge_ptr[lo:hi + 1] - e0).clone(),
                          sl(self.rowptr, n0 + lo, n1 + hi), sl(self.col, e0, e1), sl(self.eorder, e0, e1),
                          sl(self.nattr, n0, n1), sl(self.eattr, e0, e1), self.flags, chunk_max(nc), chunk_max(ec),
                          self.max_degree)

    # ------------------------------------------------------------------ builders
    @staticmethod
    def from_coo(node_counts, edge_counts, src, dst, x=None, edge_attr=None, check_symmetric: bool = True) -> "GraphBatch":
        """Batched COO -> CSR.  src/dst are LOCAL node ids, edges of graph g are contiguous, in the
        order the source edge_index lists them (that order is kept in `eorder`).  check_symmetric: verify once
        whether the graphs are simple and stored in both directions (sets GTOK_CSR_SIMPLE_SYMMETRIC)."""
        node_counts = np.asarray(node_counts, dtype=np.int64).reshape(-1)
        edge_counts = np.asarray(edge_counts, dtype=np.int64).reshape(-1)
        src = np.asarray(src, dtype=np.int64).reshape(-1)
        dst = np.asarray(dst, dtype=np.int64).reshape(-1)
        G = int(node_counts.size)
        if edge_counts.size != G:
            raise ValueError("node_counts and edge_counts differ in length")
        node_ptr = np.zeros(G + 1, np.int64); np.cumsum(node_counts, out=node_ptr[1:])
        edge_ptr = np.zeros(G + 1, np.int64); np.cumsum(edge_counts, out=edge_ptr[1:])
        N, E = int(node_ptr[-1]), int(edge_ptr[-1])
        if src.size != E or dst.size != E:
            raise ValueError("src/dst length does not match edge_counts")
        if N + G >= 2 ** 31:
            raise ValueError("batch too large for int32 node offsets; shard it")
        gid_e = np.repeat(np.arange(G, dtype=np.int64), edge_counts)
        n_e = node_counts[gid_e]
        if E and ((src < 0).any() or (dst < 0).any() or (src >= n_e).any() or (dst >= n_e).any()):
            raise ValueError("edge endpoint outside [0, num_nodes)")
        grow = node_ptr[gid_e] + src                      # global row of every entry
        if E and np.all(grow[1:] >= grow[:-1]):
            perm = None                                   # already row-sorted: identity order
        else:
            perm = np.argsort(grow, kind="stable")
        take = (lambda a: a) if perm is None else (lambda a: a[perm])
        col = take(dst).astype(np.int32)
        eorder = None if perm is None else (perm - edge_ptr[gid_e[perm]]).astype(np.int32)
        cnt = np.bincount(grow, minlength=N).astype(np.int64) if E else np.zeros(N, np.int64)
        gid_n = np.repeat(np.arange(G, dtype=np.int64), node_counts)
        excl = np.cumsum(cnt) - cnt
        rowptr = np.empty(N + G, np.int32)
        rowptr[np.arange(N, dtype=np.int64) + gid_n] = (excl - edge_ptr[gid_n]).astype(np.int32)
        rowptr[node_ptr[1:] + np.arange(G, dtype=np.int64)] = edge_counts.astype(np.int32)
        nattr = _to_u8(x)
        eattr = _to_u8(edge_attr)
        if nattr is not None and nattr.size != N:
            raise ValueError("x length does not match node_counts")
        if eattr is not None:
            if eattr.size != E:
                raise ValueError("edge_attr length does not match edge_counts")
            eattr = take(eattr)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
        max_nodes = int(node_counts.max()) if G else 0
        flags = CSR_SIMPLE_SYMMETRIC if (check_symmetric and E and _simple_symmetric(gid_e, src, dst, max_nodes)) else 0
        return GraphBatch(G, max_nodes, int(edge_counts.max()) if G else 0,
                          t(node_ptr.astype(np.int32)), t(edge_ptr), t(rowptr), t(col), t(eorder), t(nattr), t(eattr),
                          flags, _chunk_max(node_counts), _chunk_max(edge_counts), int(cnt.max()) if N else 0)

    @staticmethod
    def from_coo_device(node_counts, edge_counts, src, dst, x=None, edge_attr=None, device="cuda",
                        check_symmetric: bool = True) -> "GraphBatch":
        """from_coo() with the heavy passes (row sort, row pointers, symmetry check) done by torch on the GPU:
        the same arrays, bit for bit, built in milliseconds instead of seconds for 10^8-entry corpora, and left on
        the device.  Inputs may be numpy arrays or tensors (host or device); src / dst are LOCAL node ids, the edges of
        graph g contiguous and in edge_index order."""
        dev = torch.device(device)
        t64 = lambda a: torch.as_tensor(a).reshape(-1).to(dev, dtype=torch.int64)
        nc, ec, s, d = t64(node_counts), t64(edge_counts), t64(src), t64(dst)
        G = int(nc.numel())
        if ec.numel() != G:
            raise ValueError("node_counts and edge_counts differ in length")
        node_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev); torch.cumsum(nc, 0, out=node_ptr[1:])
        edge_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev); torch.cumsum(ec, 0, out=edge_ptr[1:])
        N, E = int(node_ptr[-1]), int(edge_ptr[-1])
        if s.numel() != E or d.numel() != E:
            raise ValueError("src/dst length does not match edge_counts")
        if N + G >= 2 ** 31:
            raise ValueError("batch too large for int32 node offsets; shard it")
        gid_e = torch.repeat_interleave(torch.arange(G, device=dev), ec, output_size=E)
        n_e = nc[gid_e]
        if E and bool(((s < 0) | (d < 0) | (s >= n_e) | (d >= n_e)).any()):
            raise ValueError("edge endpoint outside [0, num_nodes)")
        grow = node_ptr[gid_e] + s                        # global row of every entry
        sorted_already = bool((grow[1:] >= grow[:-1]).all()) if E else False
        if sorted_already:
            perm = None
        else:
            perm = torch.sort(grow, stable=True).indices if E else None
        take = (lambda a: a) if perm is None else (lambda a: a[perm])
        col = take(d).to(torch.int32)
        eorder = None if perm is None else (perm - edge_ptr[gid_e[perm]]).to(torch.int32)
        cnt = torch.bincount(grow, minlength=N) if E else torch.zeros(N, dtype=torch.int64, device=dev)
        gid_n = torch.repeat_interleave(torch.arange(G, device=dev), nc, output_size=N)
        excl = torch.cumsum(cnt, 0) - cnt
        rowptr = torch.empty(N + G, dtype=torch.int32, device=dev)
        rowptr[torch.arange(N, device=dev) + gid_n] = (excl - edge_ptr[gid_n]).to(torch.int32)
        rowptr[node_ptr[1:] + torch.arange(G, device=dev)] = ec.to(torch.int32)

        def u8(a, n_expected, what):
            if a is None:
                return None
            a = torch.as_tensor(a).reshape(-1).to(dev, dtype=torch.int64)
            if a.numel() != n_expected:
                raise ValueError(f"{what} length does not match")
            return torch.where((a >= 0) & (a < 255), a, torch.full_like(a, 255)).to(torch.uint8)
        nattr = u8(x, N, "x")
        eattr = u8(edge_attr, E, "edge_attr")
        if eattr is not None:
            eattr = take(eattr)
        max_nodes = int(nc.max()) if G else 0

        def chunk_max(c):
            if c.numel() == 0:
                return 0
            pad = (-c.numel()) % 64
            return int(torch.nn.functional.pad(c, (0, pad)).reshape(-1, 64).sum(1).max())
        out = GraphBatch(G, max_nodes, int(ec.max()) if G else 0, node_ptr.to(torch.int32), edge_ptr, rowptr, col, eorder,
                         nattr, eattr, 0, chunk_max(nc), chunk_max(ec), int(cnt.max()) if N else 0)
        if check_symmetric and E and dev.type == "cuda":
            # simple + stored in both directions: verified by gtok_csr_check on the arrays just built (round 4 sorted the
            # entry list twice with torch for this: 2 x 12 M int64 keys for ZINC-full)
            from . import ops
            if ops.csr_check(out)["violations"] == 0:
                out.flags |= CSR_SIMPLE_SYMMETRIC
            out.checked = True
        return out

    @staticmethod
    def from_collated(node_counts, edge_index, edge_slices, x=None, edge_attr=None, indices=None, device=None,
                      check_symmetric: bool = True) -> "GraphBatch":
        """From COLLATED storage - the form torch_geometric's InMemoryDataset keeps a split in (`data, slices` of
        graph_token_dataset_autograph.py:407-408; torch_geometric.datasets.ZINC behind zinc_dataset_autograph.py:44
        and zinc_dataset_indexbase.py:79): `edge_index` [2, sum E] with LOCAL node ids (collate does not increment
        them), `edge_slices` [G+1] the per-graph offsets into it, `node_counts` [G], optional `x` [sum N(, 1)] and
        `edge_attr` [sum E(, 1)] concatenated the same way.  No per-graph Python: the whole split is a handful of
        array operations (on `device` when given: from_coo_device).  indices: the graphs to take, in order (a
        dataset subset, `dataset.indices()`); None = all."""
        t = lambda a: None if a is None else torch.as_tensor(a)
        nc, es, ei = t(node_counts).reshape(-1).to(torch.int64), t(edge_slices).reshape(-1).to(torch.int64), t(edge_index)
        if ei.dim() != 2 or ei.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        if es.numel() != nc.numel() + 1:
            raise ValueError("edge_slices must have one more entry than node_counts")
        ec = es[1:] - es[:-1]
        xa = None if x is None else t(x).reshape(t(x).shape[0], -1)[:, 0]
        ea = None if edge_attr is None else t(edge_attr).reshape(-1)
        src, dst = ei[0], ei[1]
        if indices is not None:
            idx = torch.as_tensor(list(indices) if not torch.is_tensor(indices) else indices, dtype=torch.int64).reshape(-1)
            nptr = torch.zeros(nc.numel() + 1, dtype=torch.int64); torch.cumsum(nc, 0, out=nptr[1:])
            nsel, esel = nc[idx], ec[idx]
            ar = lambda n: torch.arange(n, dtype=torch.int64)
            pick = lambda starts, counts: torch.repeat_interleave(starts - (torch.cumsum(counts, 0) - counts), counts) + ar(int(counts.sum()))
            e_take, n_take = pick(es[:-1][idx], esel), pick(nptr[:-1][idx], nsel)
            src, dst = src[e_take], dst[e_take]
            xa = None if xa is None else xa[n_take]
            ea = None if ea is None else ea[e_take]
            nc, ec = nsel, esel
        elif int(es[0]) != 0 or int(es[-1]) != src.numel():
            raise ValueError("edge_slices does not cover edge_index")
        if device is not None and torch.device(device).type == "cuda":
            return GraphBatch.from_coo_device(nc, ec, src, dst, xa, ea, device=device, check_symmetric=check_symmetric)
        return GraphBatch.from_coo(nc.numpy(), ec.numpy(), src.numpy(), dst.numpy(), None if xa is None else xa.numpy(),
                                   None if ea is None else ea.numpy(), check_symmetric=check_symmetric)

    @staticmethod
    def from_dataset(dataset, labeled: Optional[bool] = None, device=None) -> "GraphBatch":
        """The whole split behind a dataset object as one batch.  Collated storage (see from_collated) is read
        directly when the dataset exposes it the way torch_geometric's InMemoryDataset does - `_data` / `data` with
        `slices`, optional `_indices`, no `transform` - or through a `collated()` method (this package's datasets);
        otherwise the items are fetched one by one (from_data_list)."""
        got = collated_storage(dataset)
        if got is not None:
            if labeled is None:
                labeled = got["x"] is not None and got["edge_attr"] is not None
            return GraphBatch.from_collated(got["node_counts"], got["edge_index"], got["edge_slices"],
                                            got["x"] if labeled else None, got["edge_attr"] if labeled else None,
                                            indices=got["indices"], device=device)
        b = GraphBatch.from_data_list([dataset[i] for i in range(len(dataset))], labeled=labeled)
        return b if device is None else b.to(device)

    @staticmethod
    def from_data_list(data_list: Sequence, labeled: Optional[bool] = None) -> "GraphBatch":
        """From PyG-like objects exposing edge_index [2,E], num_nodes (or x), and optionally x / edge_attr.  The
        per-item work is attribute access only; the arrays are concatenated once."""
        if labeled is None:
            labeled = len(data_list) > 0 and getattr(data_list[0], "x", None) is not None \
                and getattr(data_list[0], "edge_attr", None) is not None
        if not len(data_list):
            z = np.zeros(0, np.int64)
            return GraphBatch.from_coo(z, z, z, z, z if labeled else None, z if labeled else None)
        eis = [d.edge_index for d in data_list]
        tens = all(map(torch.is_tensor, eis))
        if tens:
            try:                                   # the common case: every edge_index is already [2, E_g]
                ei = torch.cat(eis, dim=1)
                if ei.dim() != 2 or ei.shape[0] != 2:
                    raise RuntimeError
            except RuntimeError:
                eis = [e.reshape(2, -1) for e in eis]
                ei = torch.cat(eis, dim=1)
            ecs = np.fromiter((e.shape[-1] for e in eis), np.int64, len(eis))
            ei = ei.to(torch.int64).numpy()
        else:
            eis = [np.asarray(e, dtype=np.int64).reshape(2, -1) for e in eis]
            ecs = np.fromiter((e.shape[1] for e in eis), np.int64, len(eis))
            ei = np.concatenate(eis, axis=1)

        def count(d):
            n = getattr(d, "num_nodes", None)
            return int(n) if n is not None else int(d.x.shape[0])
        ncs = np.fromiter(map(count, data_list), np.int64, len(data_list))
        xs = eas = None
        if labeled:
            def cat_first_column(parts):           # [n] or [n, k] per item -> column 0, concatenated
                if all(map(torch.is_tensor, parts)):
                    try:
                        c = torch.cat(parts)
                    except RuntimeError:           # ranks differ from item to item
                        c = torch.cat([p.reshape(p.shape[0], -1)[:, :1] if p.dim() > 1 and p.numel() else p.reshape(-1, 1) for p in parts])
                    return (c.reshape(c.shape[0], -1)[:, 0] if c.dim() > 1 else c).to(torch.int64).numpy()
                first = lambda a: a[:, 0] if a.ndim > 1 and a.shape[1] > 0 else a.reshape(-1)[:a.shape[0] if a.ndim else 1]
                return np.concatenate([first(np.asarray(p, dtype=np.int64)) for p in parts])
            xs = cat_first_column([d.x for d in data_list])
            ea_parts = [d.edge_attr for d in data_list]
            ea_tens = all(map(torch.is_tensor, ea_parts))
            ea_len = np.fromiter((p.numel() if torch.is_tensor(p) else np.size(p) for p in ea_parts), np.int64, len(ea_parts))
            if np.array_equal(ea_len, ecs):
                if ea_tens:
                    try:
                        eas = torch.cat(ea_parts).reshape(-1).to(torch.int64).numpy()
                    except RuntimeError:
                        eas = torch.cat([p.reshape(-1) for p in ea_parts]).to(torch.int64).numpy()
                else:
                    eas = np.concatenate([np.asarray(p, dtype=np.int64).reshape(-1) for p in ea_parts])
            else:
                # zinc_dataset_indexbase.py:183: edges past len(bond_types) read as 'unknown' (attr 0); extra attrs are ignored
                eas = np.zeros(int(ecs.sum()), np.int64)
                eptr = np.concatenate([[0], np.cumsum(ecs)])
                for g, p in enumerate(ea_parts):
                    a = np.asarray(p, dtype=np.int64).reshape(-1)[:ecs[g]]
                    eas[eptr[g]:eptr[g] + a.size] = a
        return GraphBatch.from_coo(ncs, ecs, ei[0], ei[1], xs, eas)


def collated_storage(dataset):
    """The collated arrays behind `dataset`, or None when it does not expose any (then items must be fetched one by
    one).  Understood: a `collated()` method returning the dict below (this package's datasets), or
    torch_geometric's InMemoryDataset layout - `_data` (or `data`) holding x / edge_index / edge_attr concatenated
    over the split, `slices[name]` the per-graph offsets, `_indices` an optional subset - as long as no per-item
    `transform` is installed (the transform could change what an item holds).
    Returns dict(node_counts, edge_index, edge_slices, x, edge_attr, y, indices)."""
    fn = getattr(dataset, "collated", None)
    if callable(fn):
        return fn()
    if isinstance(dataset, torch.utils.data.Subset):     # a subset of a collated dataset: its storage, the indices composed
        inner = collated_storage(dataset.dataset)
        if inner is None:
            return None
        base = inner["indices"]
        pick = [int(i) for i in dataset.indices]
        inner = dict(inner)
        inner["indices"] = pick if base is None else [base[i] for i in pick]
        return inner
    try:
        slices = getattr(dataset, "slices", None)
        data = dataset.__dict__.get("_data", None) if hasattr(dataset, "__dict__") else None
        if data is None:
            data = getattr(dataset, "_data", None)
        if data is None:                          # torch_geometric < 2.3 keeps the collated Data under `data` alone
            data = dataset.__dict__.get("data", None) if hasattr(dataset, "__dict__") else None
        if data is None or slices is None or getattr(dataset, "transform", None) is not None:
            return None
        get = (lambda k: data.get(k)) if isinstance(data, dict) else (lambda k: getattr(data, k, None))
        ei = get("edge_index")
        if ei is None or "edge_index" not in slices:
            return None
        es = torch.as_tensor(slices["edge_index"]).to(torch.int64)
        G = int(es.numel()) - 1
        x = get("x")
        if x is not None and "x" in slices:
            ns = torch.as_tensor(slices["x"]).to(torch.int64)
            nc = ns[1:] - ns[:-1]
        else:
            x = None
            nn = data.get("_num_nodes") if isinstance(data, dict) else getattr(data, "_num_nodes", None)
            if nn is None:
                nn = get("num_nodes")
            nc = torch.as_tensor(nn).reshape(-1).to(torch.int64)
            if nc.numel() != G:
                return None
        ea = get("edge_attr") if "edge_attr" in slices else None
        idx = getattr(dataset, "_indices", None)
        return dict(node_counts=nc, edge_index=ei, edge_slices=es, x=x, edge_attr=ea, y=get("y"),
                    indices=None if idx is None else list(idx))
    except Exception:
        return None
