"""IBTT text -> ids path: same public names as the reference's graph_data_loader/data_loader.py.

TokenDataset tokenizes on the GPU (gtok_text_to_ids); corpus plumbing (load_examples & co.) and vocab
construction stay on the host, written to return exactly what the reference returns
(tests/test_host_golden.py compares them with vectors captured from the reference's own code).
"""
import json
import os
import random
from collections import Counter
from collections.abc import Sequence
from glob import glob
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from ._root import root as _root

_ops = _root().ops
GtokError = _root().GtokError

SPECIAL = ["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no"]

_POSITIVE = ("yes", "true", "connected", "reachable")
_NEGATIVE = ("no", "false", "disconnected", "unreachable")


# ------------------------------------------------------------------------------------------------ parsers
def parse_yes_no_from_text(text: str) -> Optional[int]:
    """Last yes/no token of the text, case-insensitive (reference :12-17)."""
    for tok in reversed(text.split()):
        low = tok.lower()
        if low in ("yes", "no"):
            return int(low == "yes")
    return None


def parse_distance_label_from_text(text: str) -> Optional[int]:
    """'<p> lenK' -> K-1; INF / INFINITY / <EOS> right after <p> -> None (reference :19-40)."""
    toks = text.split()
    for i in range(len(toks) - 1):
        if toks[i] != "<p>":
            continue
        lab = toks[i + 1].upper()
        if lab in ("INF", "INFINITY", "<EOS>"):
            return None
        if lab.startswith("LEN"):
            try:
                return int(lab[3:]) - 1
            except ValueError:
                pass
    return None


def parse_query_nodes_from_text(text: str) -> Optional[Tuple[int, int]]:
    """'<q> shortest_distance u v' -> (u, v) (reference :42-55)."""
    toks = text.split()
    for i in range(len(toks) - 3):
        if toks[i] == "<q>" and toks[i + 1] == "shortest_distance":
            try:
                return int(toks[i + 2]), int(toks[i + 3])
            except ValueError:
                pass
    return None


def _from_text(text: str, task: str):
    if task == "shortest_path":
        return parse_distance_label_from_text(text), parse_query_nodes_from_text(text)
    return parse_yes_no_from_text(text), None


def _extract_text_and_label(rec: Any, task: str = "cycle_check"):
    """(text, label, query_nodes) of one record: str, dict or flat list (reference :57-110)."""
    if isinstance(rec, str):
        t = rec.strip()
        return (t,) + _from_text(t, task)
    if isinstance(rec, list):
        if not all(isinstance(x, (str, int)) for x in rec):
            return None, None, None
        t = " ".join(map(str, rec))
        return (t,) + _from_text(t, task)
    if not isinstance(rec, dict):
        return None, None, None
    text = rec.get("text") or rec.get("sequence")
    if text is None and isinstance(rec.get("tokens"), (list, tuple)):
        text = " ".join(map(str, rec["tokens"]))
    lab, query = rec.get("label"), None
    if task == "shortest_path":
        if not isinstance(lab, int) and isinstance(text, str):
            lab, query = _from_text(text, task)
    else:
        if isinstance(lab, str):
            low = lab.lower().strip()
            lab = 1 if low in _POSITIVE else 0 if low in _NEGATIVE else None
        elif isinstance(lab, (int, bool)):
            lab = int(bool(lab))
        if lab is None and isinstance(text, str):
            lab = parse_yes_no_from_text(text)
    return (text.strip() if isinstance(text, str) else None), lab, query


def _records_of_file(path: str) -> Iterator[Any]:
    """Whole-file JSON, else JSON lines, else raw lines (yielded as str)."""
    with open(path, "r") as f:
        raw = f.read().strip()
    if not raw:
        return
    try:
        obj = json.loads(raw)
    except json.JSONDecodeError:
        for line in raw.splitlines():
            line = line.strip()
            if not line:
                continue
            try:
                obj = json.loads(line)
            except json.JSONDecodeError:
                yield ("raw", line)
                continue
            yield from (("json", r) for r in (obj if isinstance(obj, list) else [obj]))
        return
    yield from (("json", r) for r in (obj if isinstance(obj, list) else [obj]))


def load_examples(path_glob: str, task: str = "cycle_check", data_fraction: float = 1.0, seed: int = 0,
                  num_graphs: Optional[int] = None, num_pairs_per_graph: Optional[int] = None) -> List[Dict[str, Any]]:
    """graph-token JSON files -> [{'text','label'[,'query_u','query_v']}] (reference :112-245): optional
    per-file sampling of `num_graphs` files and, for shortest_path, `num_pairs_per_graph` records per file,
    both with random.Random(seed)."""
    files = sorted(glob(path_glob))
    if num_graphs is not None and len(files) > num_graphs:
        total = len(files)
        files = sorted(random.Random(seed).sample(files, num_graphs))
        print(f"[load_examples] Sampled {num_graphs}/{total} graph files")
    out: List[Dict[str, Any]] = []

    def entry(kind, rec, need_query):
        if kind == "raw":
            if task == "shortest_path":
                lab, q = _from_text(rec, task)
            else:
                lab, q = parse_yes_no_from_text(rec), None
            t = rec
            if need_query and not q:
                return None
        else:
            t, lab, q = _extract_text_and_label(rec, task=task)
            if not t or (need_query and q is None):
                return None
        e = {"text": t, "label": lab}
        if q:
            e["query_u"], e["query_v"] = q
        return e

    if task == "shortest_path" and num_pairs_per_graph is not None:
        rng = random.Random(seed)
        for fp in files:
            per_file = [e for e in (entry(k, r, True) for k, r in _records_of_file(fp)) if e is not None]
            out.extend(rng.sample(per_file, num_pairs_per_graph) if len(per_file) > num_pairs_per_graph else per_file)
        print(f"[load_examples] Loaded {len(out)} pairs from {len(files)} graphs (target: {num_pairs_per_graph} pairs/graph)")
        return out
    for fp in files:
        out.extend(e for e in (entry(k, r, False) for k, r in _records_of_file(fp)) if e is not None)
    if num_graphs is None and data_fraction < 1.0 and out:
        out = random.Random(seed).sample(out, max(1, int(len(out) * data_fraction)))
    return out


def _split_dir(root: str, task: str, algo: str, split: str, use_split_tasks_dirs: bool) -> str:
    if use_split_tasks_dirs:
        base = os.path.join(root, "tasks_test" if split in ("val", "test") else "tasks_train", task, algo)
    else:
        base = os.path.join(root, "tasks", task, algo)
    d = os.path.join(base, split)
    if split == "val" and not glob(os.path.join(d, "*.json")):
        d = os.path.join(base, "test")
    return d


def load_examples_multi_algorithm(root: str, task: str, algorithms: List[str], split: str,
                                  use_split_tasks_dirs: bool = True, seed: int = 0, num_graphs: Optional[int] = None,
                                  num_pairs_per_graph: Optional[int] = None) -> List[Dict[str, Any]]:
    """Concatenate load_examples over algorithms; per-algorithm seed = seed + hash(algo) % 10000 exactly as the
    reference (:588-633) — str hashes are PYTHONHASHSEED dependent there too."""
    allx: List[Dict[str, Any]] = []
    for algo in algorithms:
        pattern = os.path.join(_split_dir(root, task, algo, split, use_split_tasks_dirs), "*.json")
        ex = load_examples(pattern, task=task, seed=seed + hash(algo) % 10000, num_graphs=num_graphs,
                           num_pairs_per_graph=num_pairs_per_graph)
        print(f"  [{algo}] Loaded {len(ex)} examples")
        allx.extend(ex)
    print(f"[load_examples_multi_algorithm] Total: {len(allx)} examples from {len(algorithms)} algorithms")
    return allx


def resolve_split_globs(root: str, task: str, algorithm: str, use_split_tasks_dirs: bool = True):
    """(train, val, test) glob patterns of one algorithm (reference :499-520)."""
    a = [os.path.join(root, top, task, algorithm, sp, "*.json")
         for top, sp in (("tasks_train", "train"), ("tasks_test", "val"), ("tasks_test", "test"))]
    b = [os.path.join(root, "tasks", task, algorithm, sp, "*.json") for sp in ("train", "val", "test")]
    pick = a if (use_split_tasks_dirs and glob(a[0])) or not glob(b[0]) else b
    train_glob, val_glob, test_glob = pick
    if not glob(val_glob):
        val_glob = a[2] if use_split_tasks_dirs else b[2]
    return train_glob, val_glob, test_glob


# ------------------------------------------------------------------------------------------------ vocab
def build_vocab_from_texts(texts: List[str], min_freq: int = 1, max_tokens: Optional[int] = None):
    """SPECIAL first, then tokens by descending corpus frequency, ties by first appearance; stop below
    min_freq or once `max_tokens` ids exist (reference :451-463)."""
    counts: Counter = Counter()
    for t in texts:
        counts.update(t.split())
    vocab = {tok: i for i, tok in enumerate(SPECIAL)}
    for tok, c in counts.most_common():
        if tok in vocab:
            continue
        if c < min_freq:
            break
        vocab[tok] = len(vocab)
        if max_tokens and len(vocab) >= max_tokens:
            break
    return vocab, {i: t for t, i in vocab.items()}


def build_vocab_from_texts_on_device(texts: List[str], min_freq: int = 1, max_tokens: Optional[int] = None, device=None,
                                     capacity: int = 1 << 16):
    """build_vocab_from_texts (reference :451-463) with the corpus pass - every token of every text counted, first
    appearances kept - done by ONE launch over the packed texts (gtok_vocab_stats_text); the host only walks the
    DISTINCT tokens, already in Counter.most_common order.  Same (vocab, inverse) as build_vocab_from_texts."""
    if device is None:
        if not torch.cuda.is_available():
            raise GtokError("build_vocab_from_texts_on_device needs a GPU; build_vocab_from_texts is the host function")
        device = torch.device("cuda", torch.cuda.current_device())
    vocab = {tok: i for i, tok in enumerate(SPECIAL)}
    if texts:
        blob, ptr = _ops.pack_texts(texts)
        blob = blob.to(device)
        table = _ops.vocab_stats_text(blob, ptr, capacity)
        for tok, c, _ in _ops.text_stats_entries(table, blob):
            if tok in vocab:
                continue
            if c < min_freq:
                break
            vocab[tok] = len(vocab)
            if max_tokens and len(vocab) >= max_tokens:
                break
    return vocab, {i: t for t, i in vocab.items()}


def vocab_from_stats(count, first, node_counts, edge_counts, task: Optional[str] = None, label_tokens=None,
                     query_nodes=None, min_freq: int = 1, max_tokens: Optional[int] = None, graph_base: int = 0):
    """build_vocab_from_texts (reference :451-463) for a graph-token corpus that was never rendered as text:
    `count` / `first` are the node-id token statistics (ops.vocab_stats_synth on the device, or the oracle's
    restatement), the handful of other tokens (task name, labels) are counted here from per-graph arrays.
    Texts are `<bos> u v <e> ... <n> 0 .. N-1 <q> TASK [qu qv] <p> LABEL <eos>` (docs/synthetic_data.md:46-68).
    Returns (vocab, inverse) equal to build_vocab_from_texts(texts, min_freq, max_tokens) on those texts."""
    import numpy as np
    count = np.asarray(count, dtype=np.int64); first = np.asarray(first, dtype=np.int64)
    nc = np.asarray(node_counts, dtype=np.int64); ec = np.asarray(edge_counts, dtype=np.int64)
    G = int(nc.size)
    entries = []   # (token, count, first-seen key)
    for i in np.nonzero(count)[0]:
        entries.append((str(int(i)), int(count[i]), int(first[i])))
    gkey = (np.arange(G, dtype=np.int64) + graph_base) << 32
    q_pos = 2 + 3 * ec + nc                       # position of <q> in graph g's text
    if task is not None and G:
        entries.append((task, G, int(gkey[0] + q_pos[0] + 1)))
    nq = np.zeros(G, np.int64)
    if query_nodes is not None:
        nq = (np.asarray(query_nodes).reshape(G, 2) >= 0).sum(1)
    if label_tokens is not None:
        lab = np.asarray(label_tokens, dtype=object)
        has = np.array([t is not None for t in lab], bool)
        toks, first_idx, cnts = np.unique(lab[has].astype(str), return_index=True, return_counts=True)
        where = np.nonzero(has)[0][first_idx]
        for t, g, c in zip(toks, where, cnts):   # LABEL sits after <q> [TASK] [qu qv] <p>
            entries.append((str(t), int(c), int(gkey[g] + q_pos[g] + (1 if task is not None else 0) + nq[g] + 2)))
    # structure tokens are all in SPECIAL already; Counter.most_common = count descending, ties by first appearance
    entries.sort(key=lambda r: (-r[1], r[2]))
    vocab = {tok: i for i, tok in enumerate(SPECIAL)}
    for tok, c, _ in entries:
        if tok in vocab:
            continue
        if c < min_freq:
            break
        vocab[tok] = len(vocab)
        if max_tokens and len(vocab) >= max_tokens:
            break
    return vocab, {i: t for t, i in vocab.items()}


def build_vocab_from_graphs(batch, num_ids: int, task: Optional[str] = None, label_tokens=None, query_nodes=None,
                            min_freq: int = 1, max_tokens: Optional[int] = None):
    """build_vocab_from_texts without the texts: node-id token statistics on the device (one launch over the
    CSR-resident corpus), then vocab_from_stats.  `batch` is a device GraphBatch; query_nodes int32 [G, 2] or None."""
    import numpy as np
    q = None
    if query_nodes is not None:
        q = torch.as_tensor(np.asarray(query_nodes, dtype=np.int32).reshape(-1, 2))
    count, first = _ops.vocab_stats_synth(batch, num_ids, q)
    nc = (batch.node_ptr[1:] - batch.node_ptr[:-1]).cpu().numpy()
    ec = (batch.edge_ptr[1:] - batch.edge_ptr[:-1]).cpu().numpy()
    return vocab_from_stats(count.cpu().numpy(), first.cpu().numpy(), nc, ec, task, label_tokens, query_nodes,
                            min_freq, max_tokens)


# ------------------------------------------------------------------------------------------------ dataset
class _Rows(Sequence):
    """`seqs` / `labels` of a TokenDataset: the reference keeps a Python list of tensors (:483-484); a quarter of a million
    tensor objects cost more to create than the whole corpus costs to tokenize, so this sequence makes row i - a view of
    the one host buffer - when it is asked for (in the DataLoader worker that asks, not up front in the parent)."""

    def __init__(self, buf: torch.Tensor, start, count):
        self._buf, self._start, self._count = buf, start, count

    @property
    def buffer(self) -> torch.Tensor:
        """The one host buffer the rows are views of (int64, rows back to back)."""
        return self._buf

    def __len__(self):
        return len(self._count)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self._count)
        s = self._start[i]
        return self._buf[s:s + self._count[i]]


class _Scalars(Sequence):
    """One 0-dim view per element of a 1-D tensor, made on access (`labels` of a TokenDataset)."""

    def __init__(self, t: torch.Tensor):
        self._t = t

    def __len__(self):
        return int(self._t.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return list(self._t[i].unbind(0))
        return self._t[i]


class TokenDataset(Dataset):
    """Eager text -> ids, like the reference (:465-486): the whole corpus goes through ONE gtok_text_to_ids
    launch in `__init__`.  Two copies of the result are kept: the int32 slab on the device (`ids`, `lens`, read by
    `device_batches`, the no-copy path) and, copied back once right after the launch, `seqs` / `labels` as CPU
    int64 tensors — what the reference stores, as sequences that make element i (a view of ONE host buffer) on access.  `__getitem__` only touches the CPU copies, so the object can be
    handed to `DataLoader(num_workers=2)` workers (trainer/train_ibtt.py:395-402, configs/ibtt_*.yaml): forked
    workers never see a device tensor, and pickling (spawned workers) drops the device members."""

    _DEVICE_MEMBERS = ("ids", "lens", "_table")

    def __init__(self, examples, vocab, max_len=512, strip_label=True, require_label=True, device=None):
        self.vocab, self.max_len = vocab, max_len
        if device is None:
            if not torch.cuda.is_available():
                raise GtokError("TokenDataset tokenizes on the GPU and no GPU is visible; there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        texts, labels = [], []
        for ex in examples:
            text, label = ex["text"], ex.get("label")
            if label is None and isinstance(text, str):
                label = parse_yes_no_from_text(text)
            if require_label and label is None:
                continue
            texts.append(text)
            labels.append(int(label) if label is not None else 0)       # int() truncates ZINC's float labels
        self._y = torch.tensor(labels, dtype=torch.long)
        self.labels = _Scalars(self._y)                                    # 0-dim views of one tensor, made on access
        if texts:
            blob, ptr = _ops.pack_texts(texts)
            self._table = _ops.VocabTable(vocab, self.device)
            self.ids, self.lens = _ops.text_to_ids(blob.to(self.device), ptr, self._table, max_len, strip_label)
            # host copy, made here in the parent process: ONE packed int64 buffer (rows back to back, no padding) crosses
            # to the host, row i is a view of it cut at its length - what the reference stores as one tensor per example
            rows = _root().rows.EpochRows(self.ids, self.lens, pin=False)   # pageable: forked DataLoader workers read it
            self._lens_h = torch.tensor(rows.count, dtype=torch.int32)
            self._starts_h = torch.tensor(rows.start, dtype=torch.int64)
            self.seqs = _Rows(rows.tokens, rows.start, rows.count)
        else:
            self._table = None
            self.ids = torch.empty((0, 4), dtype=torch.int32, device=self.device)
            self.lens = torch.empty((0,), dtype=torch.int32, device=self.device)
            self._lens_h = torch.empty((0,), dtype=torch.int32)
            self._starts_h = torch.empty((0,), dtype=torch.int64)
            self.seqs = _Rows(torch.empty(0, dtype=torch.int64), [], [])

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in self._DEVICE_MEMBERS:      # a pickled copy (spawned worker) serves __getitem__ only
            state[k] = None
        return state

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, idx):       # CPU tensors only: safe in DataLoader workers
        return self.seqs[idx], self.labels[idx]

    def __getitems__(self, indices):
        """Batch-level fetch (torch.utils.data.DataLoader calls it instead of __getitem__ when a dataset has one): the
        batch's rows as ONE object that `collate` turns into (X, attn, Y) with a handful of vectorised gathers over the
        packed host buffer - no per-item tensor, works in DataLoader workers (host memory only).  Any other collate
        function iterates it and gets __getitem__'s (ids, label) pairs."""
        return RowBatch(self, indices)

    def device_batches(self, batch_size: int, shuffle: bool = False, generator: Optional[torch.Generator] = None,
                       pad_id: Optional[int] = None):
        """Yield (X int64 [B,L], attn bool [B,L], Y int64 [B]) on the device: gtok_collate over the slab."""
        pad_id = self.vocab["<pad>"] if pad_id is None else pad_id
        n = len(self)
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        if self.ids is None:
            raise GtokError("device_batches on an unpickled TokenDataset copy: the device slab stays with the parent")
        lens_h = self._lens_h
        y = self._y.to(self.device)
        for s in range(0, n, batch_size):
            idx = order[s:s + batch_size]
            X, A = _ops.collate(self.ids, self.lens, idx, pad_id, int(lens_h[idx].max()))
            yield X, A, y[idx.to(self.device)]


class RowBatch(Sequence):
    """What TokenDataset.__getitems__ returns: the indices of a batch, collated in one go by `collate` (below); a
    sequence of __getitem__'s (ids, label) pairs for any other collate function."""

    def __init__(self, ds, indices):
        self._ds, self._idx = ds, list(indices)

    def __len__(self):
        return len(self._idx)

    def __getitem__(self, i):
        return self._ds[self._idx[i]]

    def collate(self, pad_id: int):
        # numpy on purpose: a [128, ~200] batch is below torch's intra-op threading threshold for some of these ops and above
        # it for others, and on a box whose cgroup grants 16 CPUs of a 256-thread host a threaded gather costs milliseconds
        ds = self._ds
        idx = np.asarray(self._idx, dtype=np.int64)
        buf = ds.seqs.buffer.numpy()
        lens = ds._lens_h.numpy()[idx].astype(np.int64)
        L = int(lens.max()) if idx.size else 0
        pos = np.arange(L, dtype=np.int64)
        attn = pos[None, :] < lens[:, None]
        if buf.size:
            src = np.minimum(ds._starts_h.numpy()[idx][:, None] + pos[None, :], buf.size - 1)
            X = np.where(attn, buf[src], np.int64(pad_id))
        else:
            X = np.full((idx.size, L), pad_id, np.int64)
        return torch.from_numpy(X), torch.from_numpy(attn), ds._y[torch.from_numpy(idx)]


def collate(batch, pad_id: int):
    """[(ids, label)] -> (X int64 [B,L] padded with pad_id, attn bool [B,L], Y int64 [B]) (reference :488-497).
    A batch fetched through TokenDataset.__getitems__ is collated without touching an item."""
    if isinstance(batch, RowBatch):
        return batch.collate(pad_id)
    xs, ys = zip(*batch)
    lens = torch.tensor([x.size(0) for x in xs])
    L = int(lens.max())
    X = torch.full((len(xs), L), pad_id, dtype=torch.long)
    for i, x in enumerate(xs):
        X[i, :x.size(0)] = x
    attn = torch.arange(L)[None, :] < lens[:, None]
    return X, attn, torch.tensor([int(y) for y in ys], dtype=torch.long)


# ------------------------------------------------------------------------------------------------ misc
def _report_classes(task: str, max_label: int) -> int:
    n = max_label + 1
    print(f"AUTO-DETERMINED NUM_CLASSES  task={task}  num_classes={n}")
    return n


def determine_num_classes(examples: List[Dict[str, Any]], task: str) -> int:
    """2 for cycle_check, 1 for zinc (regression), else max int label + 1 (reference :636-685)."""
    if task == "cycle_check":
        return 2
    if task == "zinc":
        return 1
    labs = [e.get("label") for e in examples]
    return _report_classes(task, max([l for l in labs if l is not None and isinstance(l, int)], default=-1))


def determine_num_classes_pyg(dataset, task: str) -> int:
    """Same rule over Data-like objects with `.y` (reference :688-738)."""
    if task == "cycle_check":
        return 2
    if task == "zinc":
        return 1
    m = -1
    for i in range(len(dataset)):
        d = dataset[i]
        if hasattr(d, "y"):
            m = max(m, d.y.item())
    return _report_classes(task, m)
