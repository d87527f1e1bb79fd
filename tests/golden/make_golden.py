"""Generate tests/golden/reference_vectors.{npz,json} by RUNNING THE REFERENCE'S OWN PYTHON.

Runs only in the build container (needs /root/reference; nothing here travels to the GPU box except
the data files it writes):

    PYTHONHASHSEED=0 PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's third-party imports that are absent offline (torch_geometric, wandb, autograph,
seaborn) are satisfied with empty stand-in modules; no reference function on the tokenizer path
touches them.  PYTHONHASHSEED=0 pins the `set` iteration order that decides the
ZINC dynamic-token ids (trainer/train_ibtt.py:364-372); the resulting vocab is stored
in the fixture and is an INPUT of every parity test.

What is captured (inputs and the reference's outputs only):
  zinc_*    ZINCTokenizationDataset.__getitem__ texts/labels  (zinc_dataset_indexbase.py:143-227)
            -> fixed+dynamic vocab (train_ibtt.py:361-372) -> TokenDataset -> collate
  synth_*   load_examples on graph-token JSON files -> build_vocab_from_texts -> TokenDataset -> collate,
            plus parse_graph_from_json / parse_query_nodes_from_text (graph_token_dataset_autograph.py)
  agtt_*    TokenizedGraphDataset.remap_zinc_tokens / __getitem__ query append / collate_fn
            (trainer/train_agtt.py:171-302) driven by a stub tokenizer that replays given token tensors
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0"
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tests"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Tok:
    pad = 5


class InMemoryDataset:
    """torch_geometric is absent; this is the base class's part in GraphTokenDatasetForAutoGraph
    (graph_token_dataset_autograph.py:209-210, :407-408): keep the four constructor arguments, run process(), whose
    last two lines hand the item list to collate() and torch.save() — captured below instead of written as a PyG
    pickle (agds_section patches torch.save / torch.load for the duration of the call)."""

    def __init__(self, root=None, transform=None, pre_transform=None, pre_filter=None):
        self.root, self.transform, self.pre_transform, self.pre_filter = root, transform, pre_transform, pre_filter
        self.process()

    @property
    def processed_paths(self):
        return [os.path.join(self.processed_dir, n) for n in self.processed_file_names]

    @staticmethod
    def collate(data_list):
        return list(data_list), {"items": len(data_list)}


_mod("torch_geometric"); _mod("torch_geometric.datasets", ZINC=type("ZINC", (), {}))
_mod("torch_geometric.data", Data=Data, InMemoryDataset=InMemoryDataset)
_mod("wandb"); _mod("seaborn")
_mod("autograph"); _mod("autograph.datamodules"); _mod("autograph.datamodules.data")
_mod("autograph.datamodules.data.tokenizer", Graph2TrailTokenizer=_Tok)

import importlib  # noqa: E402

import graph_data_loader as gdl  # noqa: E402  (the reference package)
from graph_data_loader.graph_token_dataset_autograph import (  # noqa: E402
    parse_graph_from_json, parse_label_from_text, parse_query_nodes_from_text)
from graph_data_loader.zinc_dataset_indexbase import ZINCTokenizationDataset  # noqa: E402

assert gdl.__file__.startswith(REF)
import trainer.train_agtt as ref_agtt  # noqa: E402

gtok = importlib.import_module("glearning-benchmark_amd")   # only for the synthetic INPUT generators
from _util import edge_case_graphs  # noqa: E402

arrays, meta = {}, {}


def pad2d(seqs, fill=-1):
    L = max((len(s) for s in seqs), default=0)
    out = np.full((len(seqs), max(L, 1)), fill, np.int64)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = np.asarray(s, np.int64)
    return out


# ------------------------------------------------------------------------------------------------ zinc
def zinc_section():
    parts = [gtok.synth.zinc_like(40, seed=101), gtok.synth.zinc_like(24, seed=102, coalesced=False)]
    ec = edge_case_graphs()
    keep = ec["node_counts"] > 0                     # PyG never yields an empty molecule
    # drop graph 0 (n=0, no edges): its slices are empty so the concatenated arrays are unchanged
    ec = dict(node_counts=ec["node_counts"][keep], edge_counts=ec["edge_counts"][keep], src=ec["src"], dst=ec["dst"],
              x=ec["x"], edge_attr=ec["edge_attr"], y=np.linspace(-3.0, 3.0, int(keep.sum())).astype(np.float32))
    parts.append(ec)
    cat = lambda k: np.concatenate([np.asarray(p[k]) for p in parts])
    d = {k: cat(k) for k in ("node_counts", "edge_counts", "src", "dst", "x", "edge_attr", "y")}
    d["y"][:6] = [-2.1049, 0.5, -0.004, 12.345, -0.996, 1.005]   # label-format corner cases (:192)
    nptr = np.concatenate([[0], np.cumsum(d["node_counts"])]); eptr = np.concatenate([[0], np.cumsum(d["edge_counts"])])
    datas = []
    for g in range(d["node_counts"].size):
        n0, n1, e0, e1 = nptr[g], nptr[g + 1], eptr[g], eptr[g + 1]
        datas.append(Data(x=torch.tensor(d["x"][n0:n1], dtype=torch.long).view(-1, 1),
                          edge_index=torch.tensor(np.stack([d["src"][e0:e1], d["dst"][e0:e1]]), dtype=torch.long).view(2, -1),
                          edge_attr=torch.tensor(d["edge_attr"][e0:e1], dtype=torch.long),
                          y=torch.tensor([float(d["y"][g])], dtype=torch.float32)))
    for k, v in d.items():
        arrays["zinc_" + k] = v
    for max_len in (1024, 48):
        ds = ZINCTokenizationDataset.__new__(ZINCTokenizationDataset)
        ds.max_len, ds.split, ds.zinc_dataset = max_len, "train", datas
        ex = [ds[i] for i in range(len(ds))]
        # trainer/train_ibtt.py:361-372
        vocab, _ = gdl.build_fixed_zinc_vocab()
        dyn = set()
        for e in ex:
            for t in e["text"].split():
                if t not in vocab:
                    dyn.add(t)
        vocab = gdl.extend_vocab_with_dynamic_tokens(vocab, list(dyn))
        td = gdl.TokenDataset(ex, vocab, max_len)
        tag = f"zinc_L{max_len}"
        meta[tag + "_texts"] = [e["text"] for e in ex]
        meta[tag + "_labels"] = [e["label"] for e in ex]
        meta[tag + "_graph_ids"] = [e["graph_id"] for e in ex]
        meta[tag + "_vocab"] = list(vocab.items())
        arrays[tag + "_ids"] = pad2d([s.tolist() for s in td.seqs])
        arrays[tag + "_len"] = np.array([s.numel() for s in td.seqs], np.int64)
        arrays[tag + "_y"] = np.array([int(t) for t in td.labels], np.int64)
        X, A, Y = gdl.collate([td[i] for i in range(16)], vocab["<pad>"])
        arrays[tag + "_collate_X"], arrays[tag + "_collate_A"], arrays[tag + "_collate_Y"] = X.numpy(), A.numpy(), Y.numpy()


# ------------------------------------------------------------------------------------------------ synthetic
def synth_section():
    for task in ("cycle_check", "shortest_path"):
        g = gtok.synth.graph_token_like(60, seed=1234, task=task)
        with tempfile.TemporaryDirectory() as tmp:
            # graph-token layout: tasks_train/<task>/<algo>/train/<i>.json, one record list per file
            for i, (txt, alg) in enumerate(zip(g["texts"], g["algorithms"])):
                dd = os.path.join(tmp, "tasks_train", task, alg, "train")
                os.makedirs(dd, exist_ok=True)
                with open(os.path.join(dd, f"{i:04d}.json"), "w") as f:
                    json.dump([{"text": txt}], f)
            ex = []
            for alg in sorted(set(g["algorithms"])):
                ex += gdl.load_examples(os.path.join(tmp, "tasks_train", task, alg, "train", "*.json"), task=task)
        vocab, _ = gdl.build_vocab_from_texts([e["text"] for e in ex], max_tokens=600)
        small, _ = gdl.build_vocab_from_texts([e["text"] for e in ex], max_tokens=40)
        tag = "synth_" + task
        meta[tag + "_examples"] = ex
        meta[tag + "_vocab"] = list(vocab.items())
        meta[tag + "_vocab40"] = list(small.items())
        for vname, v in (("", vocab), ("_v40", small)):
            for max_len in (600, 64):
                td = gdl.TokenDataset(ex, v, max_len)
                arrays[f"{tag}{vname}_L{max_len}_ids"] = pad2d([s.tolist() for s in td.seqs])
                arrays[f"{tag}{vname}_L{max_len}_len"] = np.array([s.numel() for s in td.seqs], np.int64)
                arrays[f"{tag}{vname}_L{max_len}_y"] = np.array([int(t) for t in td.labels], np.int64)
        td = gdl.TokenDataset(ex, vocab, 600)
        X, A, Y = gdl.collate([td[i] for i in range(min(16, len(td)))], vocab["<pad>"])
        arrays[tag + "_collate_X"], arrays[tag + "_collate_A"], arrays[tag + "_collate_Y"] = X.numpy(), A.numpy(), Y.numpy()
        parsed = []
        for e in ex:
            edges, n, lab = parse_graph_from_json({"text": e["text"]}, task=task)
            q = parse_query_nodes_from_text(e["text"])
            parsed.append(dict(edges=[list(map(int, p)) for p in edges], num_nodes=int(n), label=lab,
                               query=None if q is None else list(q), label_from_text=parse_label_from_text(e["text"], task)))
        meta[tag + "_parsed"] = parsed
    # hand-written records exercising the parser branches (graph_token_dataset_autograph.py:14-158)
    recs = [
        {"text": "<bos> 0 1 <e> 1 2 <e> 0 2 <e> 2 3 <e> <n> 0 1 2 3 <q> shortest_distance 0 3 <p> len2 <eos>"},
        {"text": "<bos> 0 1 <e> 1 2 <e> <q> has_cycle <p> no <eos>"},
        {"text": "<bos> <n> 0 1 2 <q> has_cycle <p> NO <eos>"},
        {"text": "<bos> 0 1 <e> <n> 0 1 5 <q> shortest_distance 0 5 <p> INF <eos>"},
        {"text": "<bos> 3 4 <e> x y <e> 4 5 <e> <n> 3 4 5 <q> has_cycle <p> yes <eos>", "label": 7},
        {"nodes": [0, 1, 2, 9], "edges": [[0, 1], [1, 2]], "text": "<q> has_cycle <p> no"},
        {"text": ""},
    ]
    meta["parser_records"] = recs
    meta["parser_out"] = []
    for r in recs:
        for task in ("cycle_check", "shortest_path"):
            edges, n, lab = parse_graph_from_json(r, task=task)
            meta["parser_out"].append(dict(task=task, edges=[list(map(int, p)) for p in edges], num_nodes=int(n), label=lab,
                                           query=(lambda q: None if q is None else list(q))(parse_query_nodes_from_text(r.get("text", "")))))
    # vocab frequency ties / min_freq / max_tokens (data_loader.py:451-463)
    texts = ["b a c a", "c b d", "e e e f", "<bos> yes zz"]
    meta["vocab_cases"] = [dict(texts=texts, min_freq=mf, max_tokens=mt,
                                vocab=list(gdl.build_vocab_from_texts(texts, min_freq=mf, max_tokens=mt)[0].items()))
                           for mf, mt in ((1, None), (2, None), (1, 11), (1, 9), (3, 600))]


# ------------------------------------------------------------------------------------------------ agtt glue
class StubTokenizer:
    """Replays prepared token tensors; carries the attributes TokenizedGraphDataset reads."""
    pad = 5

    def __init__(self, seqs, idx_offset, max_num_nodes, num_node_types):
        self.seqs, self.i = seqs, 0
        self.idx_offset = idx_offset
        self.node_idx_offset = idx_offset + max_num_nodes
        self.edge_idx_offset = self.node_idx_offset + num_node_types

    def __call__(self, data):
        s = self.seqs[self.i % len(self.seqs)]
        self.i += 1
        return s.clone()


def agtt_section():
    rng = np.random.default_rng(7)
    idx_off, max_nodes, ntypes = 6, 37, 9
    top = idx_off + max_nodes + ntypes + 12
    seqs = [torch.arange(0, top + 40, dtype=torch.long)]                      # every branch incl. both fallbacks
    seqs += [torch.tensor(rng.integers(0, top + 5, int(rng.integers(1, 90))), dtype=torch.long) for _ in range(23)]
    datas = [Data(y=torch.tensor([float(rng.normal())]), num_nodes=int(rng.integers(5, 38))) for _ in seqs]
    ds = ref_agtt.TokenizedGraphDataset(datas, StubTokenizer(seqs, idx_off, max_nodes, ntypes), task="zinc",
                                        remap_to_fixed_vocab=True)
    items = [ds[i] for i in range(len(ds))]
    arrays["agtt_remap_in"] = pad2d([s.tolist() for s in seqs])
    arrays["agtt_remap_len"] = np.array([s.numel() for s in seqs], np.int64)
    arrays["agtt_remap_out"] = pad2d([it[0].tolist() for it in items])
    meta["agtt_remap_offsets"] = [idx_off, idx_off + max_nodes, idx_off + max_nodes + ntypes]
    X, A, Y, dl = ref_agtt.collate_fn(items[:16])
    arrays["agtt_zinc_collate_X"], arrays["agtt_zinc_collate_A"] = X.numpy(), A.numpy()
    arrays["agtt_zinc_collate_Y"] = Y.numpy()
    meta["agtt_zinc_collate_Y_dtype"] = str(Y.dtype)
    meta["agtt_zinc_labels"] = [it[2] for it in items[:16]]
    # shortest_path: query append after the trail, int labels
    nn = 49
    seqs2 = [torch.tensor(rng.integers(0, idx_off + nn, int(rng.integers(2, 60))), dtype=torch.long) for _ in range(20)]
    datas2 = []
    for i in range(len(seqs2)):
        n = int(rng.integers(5, nn + 1))
        dd = Data(y=torch.tensor([int(rng.integers(0, 6))], dtype=torch.long), num_nodes=n)
        if i % 5 != 4:                                   # a sample without query fields stays un-appended
            dd.query_u, dd.query_v = int(rng.integers(0, n)), int(rng.integers(0, n))
        datas2.append(dd)
    ds2 = ref_agtt.TokenizedGraphDataset(datas2, StubTokenizer(seqs2, idx_off, nn, 0), task="shortest_path",
                                         remap_to_fixed_vocab=False)
    items2 = [ds2[i] for i in range(len(ds2))]
    arrays["agtt_sp_in"] = pad2d([s.tolist() for s in seqs2])
    arrays["agtt_sp_in_len"] = np.array([s.numel() for s in seqs2], np.int64)
    arrays["agtt_sp_num_nodes"] = np.array([d.num_nodes for d in datas2], np.int64)
    arrays["agtt_sp_query"] = np.array([[getattr(d, "query_u", -1), getattr(d, "query_v", -1)] for d in datas2], np.int64)
    arrays["agtt_sp_out"] = pad2d([it[0].tolist() for it in items2])
    arrays["agtt_sp_out_len"] = np.array([it[0].numel() for it in items2], np.int64)
    arrays["agtt_sp_labels"] = np.array([it[2] for it in items2], np.int64)
    X, A, Y, dl = ref_agtt.collate_fn(items2[:16])
    arrays["agtt_sp_collate_X"], arrays["agtt_sp_collate_A"], arrays["agtt_sp_collate_Y"] = X.numpy(), A.numpy(), Y.numpy()
    meta["agtt_sp_collate_Y_dtype"] = str(Y.dtype)


MISC_FILES = {
    "a.json": '{"text": "<bos> 0 1 <e> <n> 0 1 <q> shortest_distance 0 1 <p> len1 <eos>"}\n'
              '<bos> 0 1 <e> <n> 0 1 <q> shortest_distance 1 0 <p> len1 <eos>\n\n',
    "b.json": json.dumps([{"tokens": ["<bos>", 0, 1, "<e>", "<q>", "has_cycle", "<p>", "no"]},
                          {"text": "x", "label": "Reachable"}, ["<p>", "yes"], {"nope": 1},
                          {"sequence": " <bos> 0 1 <e> <q> shortest_distance 0 1 <p> INF <eos> ", "label": None},
                          {"text": "<q> shortest_distance 3 4 <p> len2", "label": 5}, "plain yes"]),
    "c.json": "",
    "d.json": json.dumps({"text": "<bos> 1 2 <e> <q> shortest_distance 1 2 <p> len1 <eos>"}),
    "e.json": "\n".join(json.dumps({"text": f"<bos> 0 {k} <e> <q> shortest_distance 0 {k} <p> len{k} <eos>"}) for k in range(1, 7)),
}
MISC_CALLS = [dict(task="cycle_check"), dict(task="shortest_path"), dict(task="shortest_path", num_pairs_per_graph=2, seed=3),
              dict(task="shortest_path", num_pairs_per_graph=1, seed=0), dict(task="cycle_check", num_graphs=2, seed=1),
              dict(task="shortest_path", num_graphs=3, seed=5, num_pairs_per_graph=3), dict(task="cycle_check", data_fraction=0.5, seed=2)]


def loader_section():
    with tempfile.TemporaryDirectory() as tmp:
        for name, body in MISC_FILES.items():
            with open(os.path.join(tmp, name), "w") as f:
                f.write(body)
        meta["loader_misc_files"] = MISC_FILES
        meta["loader_misc_calls"] = MISC_CALLS
        meta["loader_misc_out"] = [gdl.load_examples(os.path.join(tmp, "*.json"), **kw) for kw in MISC_CALLS]


def vocab_section():
    v, itos = gdl.build_fixed_zinc_vocab()
    meta["fixed_zinc_vocab"] = list(v.items())
    meta["special"] = list(gdl.SPECIAL)
    meta["atom_ids"] = [gdl.get_atom_type_id(i) for i in range(9)]
    meta["bond_ids"] = [gdl.get_bond_type_id(i) for i in range(1, 5)]
    meta["zinc_num_types"] = list(gdl.get_zinc_num_types())
    m = []
    for t in range(0, 70):
        for kind in ((False, False), (True, False), (False, True)):
            try:
                r = gdl.map_autograph_token_to_fixed_id(t, 43, 52, is_node_type=kind[0], is_edge_type=kind[1])
            except ValueError:
                r = "ValueError"
            m.append([t, int(kind[0]), int(kind[1]), r])
    meta["map_autograph_token"] = m


# ------------------------------------------------------------------------------------------------ round 2
# Written to reference_vectors_r2.{npz,json} so that the round-1 files stay byte-identical.
arrays2, meta2 = {}, {}

AGDS_CASES = [
    dict(task="cycle_check", algorithm=["er", "ba", "path"], split="train"),
    dict(task="cycle_check", algorithm=["er", "sbm"], split="val", num_graphs=3, seed=5),          # val -> test fallback
    dict(task="cycle_check", algorithm=["sbm", "ba", "er"], split="test", num_graphs=2, seed=11),
    dict(task="shortest_path", algorithm=["er", "ba"], split="train", num_pairs_per_graph=2, seed=1),
    dict(task="shortest_path", algorithm="er", split="test", num_graphs=4, num_pairs_per_graph=1, seed=0),
    dict(task="shortest_path", algorithm=["star", "complete"], split="train"),                     # regular path, INF skipped
    dict(task="shortest_path", algorithm=None, split="train", num_graphs=5, seed=7),               # default ['er']
    dict(task="cycle_check", algorithm=["path", "star"], split="train", use_split_tasks_dirs=False),
]


def _item(d):
    return dict(edge_index=d.edge_index.tolist(), edge_index_shape=list(d.edge_index.shape), y=d.y.tolist(),
                y_dtype=str(d.y.dtype), num_nodes=int(d.num_nodes), query_u=getattr(d, "query_u", None),
                query_v=getattr(d, "query_v", None))


def agds_section():
    """GraphTokenDatasetForAutoGraph.process() (graph_token_dataset_autograph.py:259-408) over a small graph-token
    tree: per-algorithm file sampling seeded with seed + hash(algo) % 10000 (PYTHONHASHSEED=0 here and in the
    test), num_pairs_per_graph sampling, val -> test fallback, INF / unlabeled records skipped, query fields."""
    from graph_data_loader.graph_token_dataset_autograph import GraphTokenDatasetForAutoGraph
    algs = ("er", "ba", "sbm", "path", "star", "complete")
    tree = {}
    tree.update(gtok.synth.graph_token_tree(6, seed=21, task="cycle_check", algorithms=algs, min_nodes=4, max_nodes=12))
    tree.update(gtok.synth.graph_token_tree(6, seed=22, task="shortest_path", algorithms=algs, min_nodes=4, max_nodes=12,
                                            pairs_per_graph=5))
    tree.update(gtok.synth.graph_token_tree(3, seed=23, task="cycle_check", algorithms=("path", "star"), splits=("train",),
                                            min_nodes=4, max_nodes=9, use_split_tasks_dirs=False))
    # hand-written records: a single-dict file, explicit fields, a record without label, an empty graph
    tree["tasks_train/cycle_check/er/train/zz_single.json"] = {"text": "<bos> 0 1 <e> 1 2 <e> 2 0 <e> <n> 0 1 2 <q> has_cycle <p> yes <eos>"}
    tree["tasks_train/cycle_check/er/train/zz_mixed.json"] = [
        {"nodes": [0, 1, 2, 5], "edges": [[0, 1], [1, 2]], "label": 0},
        {"text": "<bos> 0 1 <e> <n> 0 1 <q> has_cycle <p> maybe <eos>"},
        {"text": "<bos> <q> has_cycle <p> no <eos>"},
        {"text": "<bos> <n> 0 1 2 <q> has_cycle <p> no <eos>"}]
    saved = {}
    real_save, real_load = torch.save, torch.load
    torch.save = lambda obj, path, *a, **k: saved.__setitem__(path, obj)
    torch.load = lambda path, *a, **k: saved[path]
    out = []
    try:
        with tempfile.TemporaryDirectory() as tmp:
            gtok.synth.write_tree(tmp, tree)
            for kw in AGDS_CASES:
                ds = GraphTokenDatasetForAutoGraph(tmp, **kw)
                items, slices = ds.data, ds.slices
                out.append(dict(kwargs=kw, processed_dir=os.path.relpath(ds.processed_dir, tmp),
                                raw_dir=os.path.relpath(ds.raw_dir, tmp), items=[_item(d) for d in items]))
            try:
                GraphTokenDatasetForAutoGraph(tmp, task="cycle_check", algorithm=["nope"], split="train")
                err = None
            except RuntimeError as e:
                err = str(e)
    finally:
        torch.save, torch.load = real_save, real_load
    meta2["agds_tree"] = tree
    meta2["agds_cases"] = out
    meta2["agds_missing_error"] = err


def config1_section():
    """BASELINE config 1 at the size SURVEY.md section 8d names: ~1,000 graph-token records per task (6 families x 167
    graphs, 10-49 nodes, seed 1234) -> load_examples_multi_algorithm -> build_vocab_from_texts(max_tokens=600) ->
    TokenDataset(max_len=600) -> collate of the first batch of 128 (train_ibtt.py:263-295, :391-402)."""
    algs = ["er", "ba", "sbm", "path", "star", "complete"]
    for task in ("cycle_check", "shortest_path"):
        tree = gtok.synth.graph_token_tree(167, seed=1234, task=task, algorithms=algs, splits=("train",))
        with tempfile.TemporaryDirectory() as tmp:
            gtok.synth.write_tree(tmp, tree)
            ex = gdl.load_examples_multi_algorithm(tmp, task, algs, "train", seed=0)
        vocab, _ = gdl.build_vocab_from_texts([e["text"] for e in ex], max_tokens=600)
        td = gdl.TokenDataset(ex, vocab, 600)
        tag = "config1_" + task
        arrays2[tag + "_texts"] = np.frombuffer("\n".join(e["text"] for e in ex).encode(), np.uint8)
        meta2[tag + "_labels"] = [e["label"] for e in ex]
        meta2[tag + "_queries"] = [[e.get("query_u"), e.get("query_v")] for e in ex]
        meta2[tag + "_vocab"] = list(vocab.items())
        ids = pad2d([s.tolist() for s in td.seqs])
        assert ids.max() < 32768
        arrays2[tag + "_ids"] = ids.astype(np.int16)
        arrays2[tag + "_len"] = np.array([s.numel() for s in td.seqs], np.int32)
        arrays2[tag + "_y"] = np.array([int(t) for t in td.labels], np.int32)
        X, A, Y = gdl.collate([td[i] for i in range(128)], vocab["<pad>"])
        arrays2[tag + "_collate_X"], arrays2[tag + "_collate_A"], arrays2[tag + "_collate_Y"] = \
            X.numpy().astype(np.int16), A.numpy(), Y.numpy().astype(np.int32)


if __name__ == "__main__":
    zinc_section(); synth_section(); agtt_section(); loader_section(); vocab_section()
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **arrays)
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
        json.dump(meta, f)
    agds_section(); config1_section()
    np.savez_compressed(os.path.join(HERE, "reference_vectors_r2.npz"), **arrays2)
    with open(os.path.join(HERE, "reference_vectors_r2.json"), "w") as f:
        json.dump(meta2, f)
    print("round 2:", len(arrays2), "arrays and", len(meta2), "json entries;",
          os.path.getsize(os.path.join(HERE, "reference_vectors_r2.npz")) // 1024, "KiB npz,",
          os.path.getsize(os.path.join(HERE, "reference_vectors_r2.json")) // 1024, "KiB json")
    print("wrote", len(arrays), "arrays and", len(meta), "json entries;",
          os.path.getsize(os.path.join(HERE, "reference_vectors.npz")) // 1024, "KiB npz,",
          os.path.getsize(os.path.join(HERE, "reference_vectors.json")) // 1024, "KiB json")
