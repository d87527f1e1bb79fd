"""graph-token JSON -> graphs for the SENT tokenizer — mirror of the reference's
graph_data_loader/graph_token_dataset_autograph.py (parsers :14-158, dataset :161-408).

No torch_geometric here: items are light `Data` objects with the fields the trainer reads
(edge_index [2,E] long, y, num_nodes, query_u, query_v) and `graph_batch()` is the batched CSR the
kernels consume.  Edge lists keep the text order, one direction per undirected edge, like the reference.
"""
import json
import os
import random
from glob import glob
from typing import List, Optional, Tuple

import numpy as np
import torch

from ._root import root as _root

GraphBatch = _root().GraphBatch
_tag_item = _root().rows.tag_item

_STOP = ("<q>", "<p>", "<eos>")


class Data:
    """Stand-in for torch_geometric.data.Data: plain attribute bag."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        return "Data(" + ", ".join(f"{k}={getattr(v, 'shape', v)}" for k, v in self.__dict__.items()) + ")"


def parse_graph_from_text(text: str) -> Tuple[List[int], List[Tuple[int, int]]]:
    """'... u v <e> ... <n> a b c' -> (nodes, edges).  Edges are `int int <e>` triples scanned left to right;
    the scan ends at '<n>', whose integer list (up to <q>/<p>/<eos> or a non-integer) is the node list."""
    toks = text.split()
    nodes: List[int] = []
    edges: List[Tuple[int, int]] = []
    i, n = 0, len(toks)
    while i < n:
        if i + 2 < n and toks[i + 2] == "<e>":
            try:
                edges.append((int(toks[i]), int(toks[i + 1])))
                i += 3
            except ValueError:
                i += 1
            continue
        if toks[i] == "<n>" and i + 1 < n:
            for t in toks[i + 1:]:
                if t in _STOP:
                    break
                try:
                    nodes.append(int(t))
                except ValueError:
                    break
            break
        i += 1
    return nodes, edges


def parse_query_nodes_from_text(text: str) -> Optional[Tuple[int, int]]:
    toks = text.split()
    for i in range(len(toks) - 3):
        if toks[i] == "<q>" and toks[i + 1] == "shortest_distance":
            try:
                return int(toks[i + 2]), int(toks[i + 3])
            except ValueError:
                pass
    return None


def parse_label_from_text(text: str, task: str = "cycle_check") -> Optional[int]:
    """'<p> yes|no' -> 1|0, '<p> lenK' -> K-1, '<p> INF' -> None (reference :80-113)."""
    toks = text.split()
    for i in range(len(toks) - 1):
        if toks[i] != "<p>":
            continue
        lab = toks[i + 1].upper()
        if lab in ("YES", "NO"):
            return int(lab == "YES")
        if lab.startswith("LEN"):
            try:
                return int(lab[3:]) - 1
            except ValueError:
                pass
        if lab in ("INF", "INFINITY"):
            return None
    return None


def parse_graph_from_json(record: dict, task: str = "cycle_check"):
    """(edges, num_nodes, label): explicit 'nodes'/'edges'/'label' fields win, the text fills the gaps;
    num_nodes = max(node list)+1, else max endpoint+1 (reference :116-158)."""
    nodes = record.get("nodes", [])
    edges = record.get("edges", [])
    text = record.get("text", "")
    if not edges and text:
        nodes, edges = parse_graph_from_text(text)
    label = record.get("label")
    if label is None and text:
        label = parse_label_from_text(text, task=task)
    if nodes:
        num_nodes = max(nodes) + 1
    elif edges:
        num_nodes = max(max(s, t) for s, t in edges) + 1
    else:
        num_nodes = 0
    return edges, num_nodes, label


def parse_texts_on_device(texts: List[str], device="cuda"):
    """[(edges, num_nodes, label, query)] for graph-token texts - what parse_graph_from_json / parse_query_nodes_from_text
    give for records that carry only `text` - with the parsing done by ONE pair of launches (gtok_parse_graph_text)
    instead of a Python loop per token.  Texts outside the canonical grammar (status != 0: rare, hand-edited files) go
    through the host parsers above, so the result equals the host path for every input."""
    ops = _root().ops
    tb, tp = ops.pack_texts(texts)
    r = ops.parse_graph_texts(tb.to(device), tp)
    st = r["status"].cpu().numpy(); nn = r["num_nodes"].cpu().numpy(); lab = r["label"].cpu().numpy()
    q = r["query"].cpu().numpy(); ep = r["edge_ptr"].cpu().numpy()
    src, dst = r["src"].cpu().numpy(), r["dst"].cpu().numpy()
    out = []
    for g, text in enumerate(texts):
        if st[g] != 0:
            edges, n, label = parse_graph_from_json({"text": text})
            out.append((edges, n, label, parse_query_nodes_from_text(text)))
            continue
        edges = list(zip(src[ep[g]:ep[g + 1]].tolist(), dst[ep[g]:ep[g + 1]].tolist()))
        out.append((edges, int(nn[g]), None if lab[g] == ops.NO_LABEL else int(lab[g]),
                    None if q[g, 0] < 0 else (int(q[g, 0]), int(q[g, 1]))))
    return out


class GraphTokenDatasetForAutoGraph:
    """Same constructor arguments, sampling rules and cache key as the reference class (:161-408).  The graphs are
    held in COLLATED form - `(data, slices)` in the shape InMemoryDataset.collate gives them: attributes
    concatenated, `slices[name]` the per-item offsets - which is what the batched CSR is built from without any
    per-item work; items are views made on demand.  Cache: `<root>/processed/<key>/data_gtok.pt` with the reference's
    key (:233-253) but a file name of its own, holding the collated pair as plain dicts of tensors (a pickled
    torch_geometric `Data` cannot be written or read without torch_geometric).  The reference's own `data.pt` in the
    same directory is neither read nor written: each side only ever sees its own file."""

    CACHE_NAME = "data_gtok.pt"
    DEVICE_PARSE_MIN = 256        # fewer text records than this: the host parsers are as fast as two launches

    def __init__(self, root: str, task: str = "cycle_check", algorithm=None, split: str = "train",
                 use_split_tasks_dirs: bool = True, seed: int = 0, num_graphs: Optional[int] = None,
                 num_pairs_per_graph: Optional[int] = None, transform=None, pre_transform=None, pre_filter=None,
                 use_cache: bool = True):
        self.task = task
        self.algorithms = [algorithm] if isinstance(algorithm, str) else (["er"] if algorithm is None else list(algorithm))
        self.algorithm = self.algorithms[0]
        self.split, self.use_split_tasks_dirs, self.seed = split, use_split_tasks_dirs, seed
        self.num_graphs, self.num_pairs_per_graph = num_graphs, num_pairs_per_graph
        self._root, self.transform, self.pre_transform, self.pre_filter = root, transform, pre_transform, pre_filter
        self._batches = {}
        self._items = {}
        self._coll = self._load_cache() if use_cache else None
        if self._coll is None:
            self._coll = self._process_collated()  # text records parsed on the device, no object per record (None: not applicable)
            self._plain = self._coll is not None
            if self._plain and use_cache:
                self._save_cache()
        if self._coll is None:
            data_list = self.process()
            self._plain = all(type(d) is Data and set(d.__dict__) <= {"edge_index", "y", "num_nodes", "query_u", "query_v"}
                              for d in data_list)
            if self._plain:                       # the collated pair is the storage; items are made from it on demand
                self._coll = self.collate(data_list)
                if use_cache:
                    self._save_cache()
            else:                                 # a pre_transform added fields: keep the objects it returned
                self._coll = None
                self._items = dict(enumerate(data_list))
                self._len = len(data_list)
        if self._coll is not None:
            self._len = int(self._coll[0]["num_nodes"].numel())
            self._es = self._coll[1]["edge_index"].tolist()

    # ---- reference :218-253
    @property
    def raw_dir(self) -> str:
        return self._split_dir(self.algorithm)

    @property
    def processed_dir(self) -> str:
        name = f"autograph_{self.task}_{'+'.join(sorted(self.algorithms))}_{self.split}"
        if self.use_split_tasks_dirs:
            name += "_split"
        if self.num_graphs is not None:
            name += f"_ng{self.num_graphs}"
        if self.num_pairs_per_graph is not None:
            name += f"_np{self.num_pairs_per_graph}"
        return os.path.join(self._root, "processed", name)

    @property
    def processed_file_names(self) -> List[str]:
        return ["data.pt"]

    @property
    def processed_paths(self) -> List[str]:
        return [os.path.join(self.processed_dir, n) for n in self.processed_file_names]

    @staticmethod
    def collate(data_list):
        """(data, slices) as InMemoryDataset.collate lays them out: edge_index concatenated along dim 1, y along
        dim 0, integer attributes as one tensor; slices[name][i]:slices[name][i+1] is item i's part."""
        n = len(data_list)
        ecount = torch.tensor([0] + [int(d.edge_index.shape[1]) for d in data_list], dtype=torch.long)
        has_q = torch.tensor([hasattr(d, "query_u") and hasattr(d, "query_v") for d in data_list], dtype=torch.bool)
        data = {
            "edge_index": torch.cat([d.edge_index for d in data_list], dim=1) if n else torch.empty((2, 0), dtype=torch.long),
            "y": torch.cat([d.y for d in data_list]) if n else torch.empty((0,), dtype=torch.long),
            "num_nodes": torch.tensor([int(d.num_nodes) for d in data_list], dtype=torch.long),
            "query_u": torch.tensor([int(getattr(d, "query_u", -1)) for d in data_list], dtype=torch.long),
            "query_v": torch.tensor([int(getattr(d, "query_v", -1)) for d in data_list], dtype=torch.long),
            "has_query": has_q,
        }
        one = torch.arange(n + 1, dtype=torch.long)
        slices = {"edge_index": torch.cumsum(ecount, 0), "y": one, "num_nodes": one, "query_u": one, "query_v": one,
                  "has_query": one}
        return data, slices

    @property
    def cache_path(self) -> str:
        return os.path.join(self.processed_dir, self.CACHE_NAME)

    def _item(self, i: int) -> Data:
        data, _ = self._coll
        a, b = self._es[i], self._es[i + 1]
        # copies, as torch_geometric's InMemoryDataset.get hands out: an in-place transform or a trainer that writes into an item
        # must not reach the collated arrays graph_batch() / collated() / the data_gtok.pt cache are built from
        d = Data(edge_index=data["edge_index"][:, a:b].clone(), y=data["y"][i:i + 1].clone(), num_nodes=int(data["num_nodes"][i]))
        if bool(data["has_query"][i]):
            d.query_u, d.query_v = int(data["query_u"][i]), int(data["query_v"][i])
        return d

    def _load_cache(self):
        path = self.cache_path
        if not os.path.exists(path) or self.pre_transform is not None or self.pre_filter is not None:
            return None
        try:
            data, slices = torch.load(path, weights_only=True)
            if not {"edge_index", "y", "num_nodes", "query_u", "query_v", "has_query"} <= set(data) or "edge_index" not in slices:
                return None
            return data, slices
        except Exception:       # unreadable / truncated: process again
            return None

    def _save_cache(self) -> None:
        if self.pre_transform is not None or self.pre_filter is not None:
            return
        try:
            os.makedirs(self.processed_dir, exist_ok=True)
            torch.save(self._coll, self.cache_path)
        except OSError:         # read-only data tree: stay in memory
            pass

    def _split_dir(self, algo: str) -> str:
        if self.use_split_tasks_dirs:
            top = "tasks_test" if self.split in ("val", "test") else "tasks_train"
            base = os.path.join(self._root, top, self.task, algo)
        else:
            base = os.path.join(self._root, "tasks", self.task, algo)
        d = os.path.join(base, self.split)
        if self.split == "val" and self.use_split_tasks_dirs and not glob(os.path.join(d, "*.json")):
            d = os.path.join(base, "test")
        return d

    def _make(self, edges, num_nodes, label, query):
        ei = torch.tensor(edges, dtype=torch.long).t().contiguous() if len(edges) else torch.empty((2, 0), dtype=torch.long)
        d = Data(edge_index=ei, y=torch.tensor([label], dtype=torch.long), num_nodes=num_nodes)
        if query is not None:
            d.query_u, d.query_v = query
        if self.pre_filter is not None and not self.pre_filter(d):
            return None
        return self.pre_transform(d) if self.pre_transform is not None else d

    def _list_files(self) -> List[str]:
        """The split's JSON files in the reference's order, with its per-algorithm sampling (:300-318)."""
        files: List[str] = []
        for algo in self.algorithms:
            pattern = os.path.join(self._split_dir(algo), "*.json")
            af = sorted(glob(pattern))
            if self.num_graphs is not None and len(af) > self.num_graphs:
                total = len(af)
                af = sorted(random.Random(self.seed + hash(algo) % 10000).sample(af, self.num_graphs))
                print(f"  [{algo}] Sampled {self.num_graphs}/{total} graph files")
            files += af
        if not files:
            raise RuntimeError(f"No JSON files found for algorithms {self.algorithms}. "
                               f"Did you run the graph-token task generator?")
        return files

    def _process_collated(self):
        """process() and collate() in one go, without a Data object (and a tuple list) per record: the text records of all
        files are parsed by one pair of launches and the collated arrays - which are this class's storage - are cut straight
        out of the parser's edge arrays; records the device does not take (explicit `edges` / `nodes`, texts outside the
        canonical grammar) are parsed on the host and spliced in.  Same records, same order, same per-file pair sampling as
        process() (tests/test_gpu_boundary.py compares item for item).  None when this route does not apply: no GPU, too
        few text records to pay for two launches, or a pre_filter / pre_transform that needs the objects."""
        if self.pre_filter is not None or self.pre_transform is not None or not torch.cuda.is_available():
            return None
        files = self._list_files()
        per_file: List[list] = []
        texts: List[str] = []
        for fp in files:
            with open(fp, "r") as f:
                content = json.load(f)
            recs = content if isinstance(content, list) else [content]
            per_file.append(recs)
            for r in recs:
                if isinstance(r, dict) and not r.get("edges") and not r.get("nodes") and isinstance(r.get("text"), str) and r["text"]:
                    texts.append(r["text"])
        if len(texts) < self.DEVICE_PARSE_MIN or not all(t.isascii() for t in texts):
            return None
        print(f"[GraphTokenDatasetForAutoGraph] Processing {len(files)} graph files from {len(self.algorithms)} algorithm(s)")
        ops = _root().ops
        tb, tp = ops.pack_texts(texts)
        res = ops.parse_graph_texts(tb.to(torch.device("cuda", torch.cuda.current_device())), tp)
        st = res["status"].cpu().tolist(); nn = res["num_nodes"].cpu().tolist(); lab = res["label"].cpu().tolist()
        qn = res["query"].cpu().tolist(); ep = res["edge_ptr"].cpu().tolist()
        src, dst = res["src"].cpu().numpy(), res["dst"].cpu().numpy()
        extra_s: List[int] = []
        extra_d: List[int] = []
        base = int(src.shape[0])                      # host-parsed edges live behind the device's
        kept: List[tuple] = []                        # (first edge, edges, num_nodes, label, query)
        rng = random.Random(self.seed)
        pairs_mode = self.task == "shortest_path" and self.num_pairs_per_graph is not None
        sp = self.task == "shortest_path"
        b = 0
        for recs in per_file:
            parsed = []
            for r in recs:
                text = r.get("text", "")
                in_bulk = isinstance(r, dict) and not r.get("edges") and not r.get("nodes") and isinstance(text, str) and text
                if in_bulk and st[b] == 0:
                    start, cnt, n = ep[b], ep[b + 1] - ep[b], nn[b]
                    label = r.get("label")
                    if label is None:
                        label = None if lab[b] == ops.NO_LABEL else lab[b]
                    q = (qn[b][0], qn[b][1]) if (sp and qn[b][0] >= 0) else None
                    b += 1
                else:
                    if in_bulk:                       # outside the canonical grammar: the host parsers take it
                        b += 1
                        edges, n, tlabel = parse_graph_from_json({"text": text})
                        label = r.get("label")
                        if label is None:
                            label = tlabel
                    else:
                        edges, n, label = parse_graph_from_json(r, task=self.task)
                    q = parse_query_nodes_from_text(text) if (sp and text) else None
                    start, cnt = base + len(extra_s), len(edges)
                    extra_s += [e[0] for e in edges]
                    extra_d += [e[1] for e in edges]
                if n == 0 or label is None:
                    continue
                if pairs_mode and q is None:
                    continue
                parsed.append((start, cnt, n, label, q))
            if pairs_mode and len(parsed) > self.num_pairs_per_graph:
                parsed = rng.sample(parsed, self.num_pairs_per_graph)
            kept += parsed
        n_items = len(kept)
        starts = np.fromiter((k[0] for k in kept), np.int64, n_items)
        counts = np.fromiter((k[1] for k in kept), np.int64, n_items)
        es = np.zeros(n_items + 1, np.int64)
        np.cumsum(counts, out=es[1:])
        all_s = np.concatenate([src.astype(np.int64), np.asarray(extra_s, np.int64)])
        all_d = np.concatenate([dst.astype(np.int64), np.asarray(extra_d, np.int64)])
        idx = np.repeat(starts - es[:-1], counts) + np.arange(int(es[-1]), dtype=np.int64)
        has_q = [k[4] is not None for k in kept]
        data = {
            "edge_index": torch.from_numpy(np.stack([all_s[idx], all_d[idx]])) if n_items else torch.empty((2, 0), dtype=torch.long),
            "y": torch.tensor([k[3] for k in kept], dtype=torch.long) if n_items else torch.empty((0,), dtype=torch.long),
            "num_nodes": torch.tensor([int(k[2]) for k in kept], dtype=torch.long),
            "query_u": torch.tensor([int(k[4][0]) if k[4] is not None else -1 for k in kept], dtype=torch.long),
            "query_v": torch.tensor([int(k[4][1]) if k[4] is not None else -1 for k in kept], dtype=torch.long),
            "has_query": torch.tensor(has_q, dtype=torch.bool),
        }
        one = torch.arange(n_items + 1, dtype=torch.long)
        slices = {"edge_index": torch.from_numpy(es), "y": one, "num_nodes": one, "query_u": one, "query_v": one, "has_query": one}
        print(f"[GraphTokenDatasetForAutoGraph] Processed {n_items} data samples")
        return data, slices

    def process(self) -> List[Data]:
        files = self._list_files()
        print(f"[GraphTokenDatasetForAutoGraph] Processing {len(files)} graph files from {len(self.algorithms)} algorithm(s)")
        out: List[Data] = []
        rng = random.Random(self.seed)
        pairs_mode = self.task == "shortest_path" and self.num_pairs_per_graph is not None
        # pass 1: read the files.  Records that carry their graph only as `text` (what the graph-token generator writes)
        # are parsed in bulk on the device - one pair of launches instead of a Python loop over every token - when a GPU
        # is there; explicit `edges` / `nodes` fields, texts outside the canonical grammar and GPU-less runs (host-side
        # tests) go through the host parsers, which give the same tuples (tests/test_gpu_golden.py).
        per_file: List[list] = []
        bulk_texts: List[str] = []
        for fp in files:
            with open(fp, "r") as f:
                content = json.load(f)
            recs = content if isinstance(content, list) else [content]
            per_file.append(recs)
            for r in recs:
                if isinstance(r, dict) and not r.get("edges") and not r.get("nodes") and isinstance(r.get("text"), str) and r["text"]:
                    bulk_texts.append(r["text"])
        bulk = None
        if len(bulk_texts) >= self.DEVICE_PARSE_MIN and torch.cuda.is_available() and all(t.isascii() for t in bulk_texts):
            bulk = iter(parse_texts_on_device(bulk_texts, device=torch.device("cuda", torch.cuda.current_device())))
        for recs in per_file:
            parsed = []
            for r in recs:
                text = r.get("text", "")
                in_bulk = isinstance(r, dict) and not r.get("edges") and not r.get("nodes") and isinstance(text, str) and text
                if bulk is not None and in_bulk:
                    edges, n, tlabel, q = next(bulk)
                    label = r.get("label")
                    if label is None:
                        label = tlabel
                    if self.task != "shortest_path":
                        q = None
                else:
                    edges, n, label = parse_graph_from_json(r, task=self.task)
                    q = parse_query_nodes_from_text(text) if (self.task == "shortest_path" and text) else None
                if n == 0 or label is None:
                    continue
                if pairs_mode and q is None:
                    continue
                parsed.append((edges, n, label, q))
            if pairs_mode and len(parsed) > self.num_pairs_per_graph:
                parsed = rng.sample(parsed, self.num_pairs_per_graph)
            for p in parsed:
                d = self._make(*p)
                if d is not None:
                    out.append(d)
        print(f"[GraphTokenDatasetForAutoGraph] Processed {len(out)} data samples")
        return out

    def __len__(self):
        return self._len

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return [self[i] for i in range(*idx.indices(len(self)))]
        i = idx + self._len if idx < 0 else idx
        if not 0 <= i < self._len:
            raise IndexError(idx)
        d = self._items.get(i)
        if d is None:                             # made on demand from the collated arrays, not kept: no second copy of the split grows behind the caller
            d = self._item(i)
        if self.transform is not None:            # the batch below would not see what a transform does: no mark
            return self.transform(d)
        return _tag_item(self, i, d)              # lets Graph2TrailTokenizer tokenize the item's whole split at once

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def collated(self):
        """The split's arrays for GraphBatch.from_collated (csr.collated_storage picks this up), or None."""
        if self._coll is None or self.transform is not None:
            return None
        data, slices = self._coll
        return dict(node_counts=data["num_nodes"], edge_index=data["edge_index"], edge_slices=slices["edge_index"],
                    x=None, edge_attr=None, y=data["y"], indices=None)

    def graph_batch(self, device=None, labeled: bool = False) -> GraphBatch:
        """The split as one batched CSR (built once per device), straight from the collated arrays."""
        key = None if device is None else str(device)
        if key not in self._batches:
            self._batches[key] = GraphBatch.from_dataset(self, labeled=False, device=device)
        return self._batches[key]

    def queries(self) -> Optional[np.ndarray]:
        """[G,2] (query_u, query_v) when every item carries a query, else None."""
        if self._coll is not None:
            data = self._coll[0]
            if not len(self) or not bool(data["has_query"].all()):
                return None
            return torch.stack([data["query_u"], data["query_v"]], 1).to(torch.int32).numpy()
        items = [self._items[i] for i in range(len(self))]
        if not items or not all(hasattr(d, "query_u") for d in items):
            return None
        return np.array([[d.query_u, d.query_v] for d in items], np.int32)
