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
    lane_sorted: Optional["GraphBatch"] = None   # the reordered copy of THIS batch (device batches, made on first use)

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
                          0, mv(self.graph_ids), mv(self.unit_ptr), self.num_units)

    def c_struct(self) -> GtokCsr:
        p = lambda t: None if t is None else t.data_ptr()
        return GtokCsr(self.num_graphs, self.max_nodes, self.max_edges, self.flags, p(self.node_ptr), p(self.edge_ptr),
                       p(self.rowptr), p(self.col), p(self.eorder), p(self.nattr), p(self.eattr),
                       self.chunk_nodes, self.chunk_edges, self.max_degree, 0, p(self.rowptr8), p(self.col8),
                       p(self.adj_rows), p(self.adj_planes), p(self.lane_order), self.adj_words, self.adj_max_degree,
                       p(self.graph_ids), p(self.unit_ptr), self.num_units, 0)

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
        flags = 0
        if check_symmetric and E and not bool((s == d).any()):
            m = max_nodes + 1
            fwd = torch.sort((gid_e * m + s) * m + d).values
            if not bool((fwd[1:] == fwd[:-1]).any()):
                rev = torch.sort((gid_e * m + d) * m + s).values
                flags = CSR_SIMPLE_SYMMETRIC if bool(torch.equal(fwd, rev)) else 0

        def chunk_max(c):
            if c.numel() == 0:
                return 0
            pad = (-c.numel()) % 64
            return int(torch.nn.functional.pad(c, (0, pad)).reshape(-1, 64).sum(1).max())
        return GraphBatch(G, max_nodes, int(ec.max()) if G else 0, node_ptr.to(torch.int32), edge_ptr, rowptr, col, eorder,
                          nattr, eattr, flags, chunk_max(nc), chunk_max(ec), int(cnt.max()) if N else 0)

    @staticmethod
    def from_data_list(data_list: Sequence, labeled: Optional[bool] = None) -> "GraphBatch":
        """From PyG-like objects exposing edge_index [2,E], num_nodes (or x), and optionally x / edge_attr."""
        ncs, ecs, srcs, dsts, xs, eas = [], [], [], [], [], []
        if labeled is None:
            labeled = len(data_list) > 0 and getattr(data_list[0], "x", None) is not None \
                and getattr(data_list[0], "edge_attr", None) is not None
        for d in data_list:
            ei = np.asarray(d.edge_index, dtype=np.int64).reshape(2, -1)
            x = getattr(d, "x", None)
            n = getattr(d, "num_nodes", None)
            if n is None:
                n = int(np.asarray(x).shape[0])
            ncs.append(int(n)); ecs.append(ei.shape[1]); srcs.append(ei[0]); dsts.append(ei[1])
            if labeled:
                xa = np.asarray(x, dtype=np.int64)
                xs.append(xa.reshape(xa.shape[0], -1)[:, 0] if xa.size else xa.reshape(-1))
                ea = np.asarray(d.edge_attr, dtype=np.int64).reshape(-1)
                # zinc_dataset_indexbase.py:183: edges past len(bond_types) read as 'unknown'
                if ea.size < ei.shape[1]:
                    ea = np.concatenate([ea, np.zeros(ei.shape[1] - ea.size, np.int64)])
                eas.append(ea[:ei.shape[1]])
        cat = lambda l: np.concatenate(l) if l else np.zeros(0, np.int64)
        return GraphBatch.from_coo(ncs, ecs, cat(srcs), cat(dsts), cat(xs) if labeled else None,
                                   cat(eas) if labeled else None)
