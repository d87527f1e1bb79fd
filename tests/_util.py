"""Shared helpers for the parity tests: one synthetic corpus -> (product GraphBatch, oracle Coo)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

gtok = importlib.import_module("glearning-benchmark_amd")
gtok.build()  # no-op unless csrc/ is newer than libgtok.so (hipcc cross-compiles without a GPU)
import oracle as orc  # noqa: E402

FIXED_ZINC_VOCAB = {t: i for i, t in enumerate(
    ["<bos>", "<eos>", "<pad>", "<unk>", "<q>", "<p>", "<atom>", "<bond>", "C", "N", "O", "F", "P", "S", "Cl",
     "Br", "I", "single", "double", "triple", "aromatic", "regression"])}


def zinc_vocab(num_node_ids=40, with_fallbacks=True):
    v = dict(FIXED_ZINC_VOCAB)
    for i in range(num_node_ids):
        v[str(i)] = len(v)
    if with_fallbacks:
        v["X"] = len(v); v["unknown"] = len(v)
    return v


def both(d, labeled=True):
    x = d.get("x") if labeled else None
    ea = d.get("edge_attr") if labeled else None
    batch = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], x, ea)
    coo = orc.Coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], x, ea)
    return batch, coo


def edge_case_graphs():
    """Hand-made graphs: empty, single node, isolated nodes, self loops, duplicate and reversed-first
    edges, two components, out-of-range types (what the reference's fallbacks handle)."""
    graphs = [
        dict(n=0, e=[], x=[], a=[]),
        dict(n=1, e=[], x=[3], a=[]),
        dict(n=5, e=[], x=[0, 1, 2, 9, 27], a=[]),
        dict(n=3, e=[(2, 0), (0, 2), (1, 1), (1, 2), (0, 2)], x=[12, 5, 8], a=[3, 3, 0, 2, 1]),   # SURVEY Mol B
        dict(n=4, e=[(0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 2)], x=[0, 1, 2, 0], a=[1, 1, 2, 2, 4, 4]),  # Mol A
        dict(n=4, e=[(0, 0), (0, 1), (1, 0), (0, 1), (3, 3)], x=[0, 0, 0, 0], a=[1, 2, 3, 4, 0]),
        dict(n=6, e=[(0, 1), (1, 2), (2, 0), (3, 4), (4, 5), (5, 3)], x=[1] * 6, a=[1, 2, 3, 1, 2, 3]),
        dict(n=7, e=[(6, 5), (5, 4), (4, 3), (3, 2), (2, 1), (1, 0), (0, 6)], x=[8] * 7, a=[4] * 7),
        dict(n=2, e=[(1, 0)], x=[200, 254], a=[77]),
    ]
    k = 8
    graphs.append(dict(n=k, e=[(i, j) for i in range(k) for j in range(k) if i != j], x=list(range(k)),
                       a=[(i + j) % 6 for i in range(k) for j in range(k) if i != j]))
    return dict(node_counts=np.array([g["n"] for g in graphs]), edge_counts=np.array([len(g["e"]) for g in graphs]),
                src=np.array([u for g in graphs for u, _ in g["e"]], np.int64),
                dst=np.array([v for g in graphs for _, v in g["e"]], np.int64),
                x=np.array([t for g in graphs for t in g["x"]], np.int64),
                edge_attr=np.array([t for g in graphs for t in g["a"]], np.int64))


# ------------------------------------------------------------------------------------------------ golden fixtures
import json  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
_golden = None


def golden():
    """(arrays, meta) captured from the reference's own Python by tests/golden/make_golden.py."""
    global _golden
    if _golden is None:
        arr = dict(np.load(os.path.join(GOLDEN_DIR, "reference_vectors.npz")))
        with open(os.path.join(GOLDEN_DIR, "reference_vectors.json")) as f:
            meta = json.load(f)
        _golden = (arr, meta)
    return _golden


_golden2 = None


def golden2():
    """Round-2 vectors (reference_vectors_r2.*): GraphTokenDatasetForAutoGraph.process() items, the 1k-record
    config-1 corpus.  Same generator script, same rules."""
    global _golden2
    if _golden2 is None:
        arr = dict(np.load(os.path.join(GOLDEN_DIR, "reference_vectors_r2.npz")))
        with open(os.path.join(GOLDEN_DIR, "reference_vectors_r2.json")) as f:
            meta = json.load(f)
        _golden2 = (arr, meta)
    return _golden2


def config1_examples(task):
    """[{'text','label'[,'query_u','query_v']}] of the config-1 fixture, as the reference's loader returned them."""
    arr, meta = golden2()
    tag = "config1_" + task
    texts = bytes(arr[tag + "_texts"]).decode().split("\n")
    out = []
    for t, lab, (qu, qv) in zip(texts, meta[tag + "_labels"], meta[tag + "_queries"]):
        e = {"text": t, "label": lab}
        if qu is not None:
            e["query_u"], e["query_v"] = qu, qv
        out.append(e)
    return out


def golden_zinc_coo():
    arr, _ = golden()
    d = {k: arr["zinc_" + k] for k in ("node_counts", "edge_counts", "src", "dst", "x", "edge_attr", "y")}
    return d


class PygLike:
    """Attribute bag standing in for torch_geometric.data.Data in tests."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def zinc_data_list(d):
    import torch
    nptr = np.concatenate([[0], np.cumsum(d["node_counts"])]); eptr = np.concatenate([[0], np.cumsum(d["edge_counts"])])
    out = []
    for g in range(len(d["node_counts"])):
        n0, n1, e0, e1 = nptr[g], nptr[g + 1], eptr[g], eptr[g + 1]
        out.append(PygLike(x=torch.tensor(d["x"][n0:n1], dtype=torch.long).view(-1, 1),
                           edge_index=torch.tensor(np.stack([d["src"][e0:e1], d["dst"][e0:e1]]), dtype=torch.long).view(2, -1),
                           edge_attr=torch.tensor(d["edge_attr"][e0:e1], dtype=torch.long),
                           y=torch.tensor([float(d["y"][g])], dtype=torch.float32), num_nodes=int(n1 - n0)))
    return out


def unpad(ids2d, lens):
    return [ids2d[i, :l].tolist() for i, l in enumerate(lens)]


# ---- torch_geometric is not installed here.  What follows restates, from its documented behaviour, the
# ATTRIBUTE SEMANTICS of torch_geometric.data.Data / BaseStorage and the storage layout of InMemoryDataset that this
# package's ingestion and item marking depend on - attribute access routed through a storage mapping, names with a leading
# underscore kept out of the mapping (and out of keys()), missing attributes raising AttributeError, num_nodes inferred, copies
# made per fetch - so that the host tests meet objects that behave like the real ones rather than a plain attribute bag.
class PygStorageLike:
    def __init__(self, _parent=None, **kw):
        self.__dict__["_mapping"] = {}
        if _parent is not None:
            self._parent = _parent
        for k, v in kw.items():
            setattr(self, k, v)

    def __setattr__(self, key, value):
        if key == "_parent":
            import weakref
            self.__dict__[key] = weakref.ref(value)
        elif key[:1] == "_":
            self.__dict__[key] = value
        elif value is None:
            self._mapping.pop(key, None)
        else:
            self._mapping[key] = value

    def __getattr__(self, key):
        if key == "_mapping":
            self.__dict__["_mapping"] = {}
            return self.__dict__["_mapping"]
        try:
            return self._mapping[key]
        except KeyError:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{key}'") from None

    def __getitem__(self, key):
        return self._mapping[key]

    def keys(self):
        return list(self._mapping.keys())

    def __copy__(self):
        out = type(self).__new__(type(self))
        for k, v in self.__dict__.items():
            out.__dict__[k] = v
        out.__dict__["_mapping"] = dict(self._mapping)
        return out


class PygDataLike:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kw):
        self.__dict__["_store"] = PygStorageLike(_parent=self)
        for k, v in dict(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y, **kw).items():
            if v is not None:
                setattr(self, k, v)

    def __getattr__(self, key):
        if "_store" not in self.__dict__:
            raise RuntimeError("the 'data' object was created by an older version")
        return getattr(self._store, key)

    def __setattr__(self, key, value):
        prop = getattr(type(self), key, None)
        if prop is not None and getattr(prop, "fset", None) is not None:
            prop.fset(self, value)
        else:
            setattr(self._store, key, value)

    def __getitem__(self, key):
        return self._store[key]

    def keys(self):
        return self._store.keys()

    @property
    def num_nodes(self):
        m = self._store._mapping
        if "num_nodes" in m:
            return m["num_nodes"]
        if "x" in m:
            return int(m["x"].shape[0])
        return int(m["edge_index"].max()) + 1 if m["edge_index"].numel() else 0

    @num_nodes.setter
    def num_nodes(self, v):
        self._store._mapping["num_nodes"] = v

    def __copy__(self):
        import copy
        out = type(self).__new__(type(self))
        for k, v in self.__dict__.items():
            out.__dict__[k] = v
        out.__dict__["_store"] = copy.copy(self._store)
        out._store._parent = out
        return out

    def clone(self):
        import copy
        out = copy.copy(self)
        for k, v in list(out._store._mapping.items()):
            out._store._mapping[k] = v.clone() if hasattr(v, "clone") else copy.deepcopy(v)
        return out


class PygInMemoryLike:
    """InMemoryDataset as torch_geometric >= 2.3 lays it out: `_data` (one Data with every attribute concatenated, edge_index
    with LOCAL node ids), `slices`, `_indices`, `transform`; items separated on demand and copied per fetch; an index list or a
    slice returns a shallow copy of the dataset with `_indices` set (index_select).  legacy=True: only `data` exists (< 2.3)."""

    def __init__(self, d, legacy=False, transform=None):
        import torch
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
        data = PygDataLike(x=t(d["x"], torch.long).view(-1, 1), edge_index=torch.stack([t(d["src"], torch.long), t(d["dst"], torch.long)]),
                           edge_attr=t(d["edge_attr"], torch.long), y=t(d["y"], torch.float32))
        ptr = lambda c: torch.from_numpy(np.concatenate([[0], np.cumsum(np.asarray(c))]).astype(np.int64))
        nptr, eptr = ptr(d["node_counts"]), ptr(d["edge_counts"])
        if legacy:
            self.data = data
        else:
            self._data = data
        self.slices = {"x": nptr, "edge_index": eptr, "edge_attr": eptr, "y": torch.arange(len(d["node_counts"]) + 1)}
        self._indices, self.transform = None, transform

    def _storage(self):
        return self.__dict__.get("_data", None) or self.__dict__["data"]

    def indices(self):
        return range(int(self.slices["x"].numel()) - 1) if self._indices is None else self._indices

    def __len__(self):
        return len(self.indices())

    def get(self, g):
        import copy
        dt, s = self._storage(), self.slices
        n0, n1, e0, e1 = int(s["x"][g]), int(s["x"][g + 1]), int(s["edge_index"][g]), int(s["edge_index"][g + 1])
        return copy.copy(PygDataLike(x=dt.x[n0:n1], edge_index=dt.edge_index[:, e0:e1], edge_attr=dt.edge_attr[e0:e1], y=dt.y[g:g + 1]))

    def __getitem__(self, idx):
        import copy
        if isinstance(idx, (int, np.integer)):
            item = self.get(self.indices()[idx])
            return item if self.transform is None else self.transform(item)
        out = copy.copy(self)
        base = list(self.indices())
        out._indices = base[idx] if isinstance(idx, slice) else [base[int(i)] for i in idx]
        return out


# ------------------------------------------------------------------------------------------------ layout rules, restated
def lane_sort_reference(batch, lds_budget=10224):
    """The rule gtok_csr_lane_sort implements (include/gtok.h), restated with numpy on a HOST GraphBatch: graphs by
    descending (nodes + rows of length 1), ties in dataset order; units cut greedily over that order - at most 64 graphs,
    node and entry sums within the caps (the LDS budget split between the two in the corpus' own proportion).
    Returns dict(graph_ids, node_ptr, edge_ptr, rowptr, col, nattr, eattr, unit_ptr, unit_info, chunk_nodes, chunk_edges)."""
    G = batch.num_graphs
    node_ptr, edge_ptr = batch.node_ptr.numpy().astype(np.int64), batch.edge_ptr.numpy().astype(np.int64)
    rowptr, col = batch.rowptr.numpy(), batch.col.numpy()
    nc, ec = np.diff(node_ptr), np.diff(edge_ptr)
    leaves = np.zeros(G, np.int64)
    for g in range(G):
        rp = rowptr[node_ptr[g] + g:node_ptr[g + 1] + g + 1]
        leaves[g] = int((np.diff(rp) == 1).sum())
    perm = np.argsort(-(nc + leaves), kind="stable")
    nc2, ec2 = nc[perm], ec[perm]
    cn = np.concatenate([[0], np.cumsum(nc2)]); ce = np.concatenate([[0], np.cumsum(ec2)])
    room = (lds_budget - (64 + 16 + 8 + 8 + 8) - 4 * 15) // 2
    ratio = float(ec2.sum()) / max(1.0, float(nc2.sum()))
    ncap = max(64, int(room / (1.0 + ratio)))
    ecap = max(255, room - ncap)
    starts, i = [0], 0
    while i < G:
        j = min(i + 64, int(np.searchsorted(cn, cn[i] + ncap, side="right")) - 1, int(np.searchsorted(ce, ce[i] + ecap, side="right")) - 1)
        i = max(j, i + 1)
        starts.append(i)
    starts = np.asarray(starts, np.int64)
    rp2 = np.concatenate([rowptr[node_ptr[g] + g:node_ptr[g + 1] + g + 1] for g in perm]) if G else rowptr[:0]
    col2 = np.concatenate([col[edge_ptr[g]:edge_ptr[g + 1]] for g in perm]) if G else col[:0]
    take_n = lambda a: None if a is None else np.concatenate([a.numpy()[node_ptr[g]:node_ptr[g + 1]] for g in perm])
    take_e = lambda a: None if a is None else np.concatenate([a.numpy()[edge_ptr[g]:edge_ptr[g + 1]] for g in perm])
    info = np.empty((starts.size - 1, 8), np.int32)
    info[:, 0], info[:, 1] = starts[:-1], starts[1:]
    info[:, 2], info[:, 3] = cn[starts[:-1]], cn[starts[1:]]
    info[:, 4:8] = np.stack([ce[starts[:-1]], ce[starts[1:]]], 1).astype(np.int64).view(np.int32).reshape(-1, 4)
    return dict(graph_ids=perm.astype(np.int32), node_ptr=cn.astype(np.int32), edge_ptr=ce.astype(np.int64), rowptr=rp2, col=col2,
                nattr=take_n(batch.nattr), eattr=take_e(batch.eattr), unit_ptr=starts.astype(np.int32), unit_info=info,
                chunk_nodes=int((cn[starts[1:]] - cn[starts[:-1]]).max()), chunk_edges=int((ce[starts[1:]] - ce[starts[:-1]]).max()))
