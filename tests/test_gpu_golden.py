"""GPU: the product (HIP kernels behind the reference-shaped classes) against golden vectors captured from
the reference's own Python — the drop-in claim, end to end."""
import numpy as np
import pytest
import torch

from _util import PygLike, config1_examples, golden, golden2, golden_zinc_coo, gtok, unpad, zinc_data_list

pytestmark = pytest.mark.gpu
gdl = gtok.graph_data_loader
DEV = "cuda:0"


def _rows(ids, ln):
    return unpad(ids.cpu().numpy(), ln.cpu().numpy())


@pytest.mark.parametrize("max_len", [1024, 48])
def test_zinc_ibtt_both_routes_equal_reference(max_len):
    arr, meta = golden()
    vocab = dict(meta[f"zinc_L{max_len}_vocab"])
    want = unpad(arr[f"zinc_L{max_len}_ids"], arr[f"zinc_L{max_len}_len"])
    ds = gdl.ZINCTokenizationDataset(split="train", max_len=max_len, zinc_dataset=zinc_data_list(golden_zinc_coo()))
    # route 1 (what train_ibtt.py does): strings -> TokenDataset (GPU text kernel)
    ex = [ds[i] for i in range(len(ds))]
    td = gdl.TokenDataset(ex, vocab, max_len)
    assert [s.tolist() for s in td.seqs] == want
    assert [int(y) for y in td.labels] == arr[f"zinc_L{max_len}_y"].tolist()
    s0, y0 = td[0]
    assert s0.dtype == torch.long and y0.dtype == torch.long and y0.dim() == 0
    # route 2: CSR -> ids, no strings
    ids, ln = ds.tokenize(vocab, device=DEV)
    assert _rows(ids, ln) == want
    # reference collate on the items, and the device collate over the slab
    X, A, Y = gdl.collate([td[i] for i in range(16)], vocab["<pad>"])
    assert np.array_equal(X.numpy(), arr[f"zinc_L{max_len}_collate_X"]) and np.array_equal(Y.numpy(), arr[f"zinc_L{max_len}_collate_Y"])
    Xd, Ad, Yd = next(td.device_batches(16))
    assert Xd.is_cuda and torch.equal(Xd.cpu(), X) and torch.equal(Ad.cpu(), A) and torch.equal(Yd.cpu(), Y)


@pytest.mark.parametrize("task", ["cycle_check", "shortest_path"])
def test_synthetic_token_dataset_equals_reference(task):
    arr, meta = golden()
    tag = "synth_" + task
    ex = meta[tag + "_examples"]
    for vname, vkey in (("", "_vocab"), ("_v40", "_vocab40")):
        vocab = dict(meta[tag + vkey])
        for max_len in (600, 64):
            td = gdl.TokenDataset(ex, vocab, max_len)
            assert [s.tolist() for s in td.seqs] == unpad(arr[f"{tag}{vname}_L{max_len}_ids"], arr[f"{tag}{vname}_L{max_len}_len"])
            assert [int(y) for y in td.labels] == arr[f"{tag}{vname}_L{max_len}_y"].tolist()
    td = gdl.TokenDataset(ex, dict(meta[tag + "_vocab"]), 600)
    X, A, Y = gdl.collate([td[i] for i in range(16)], 0)
    assert np.array_equal(X.numpy(), arr[tag + "_collate_X"]) and np.array_equal(A.numpy(), arr[tag + "_collate_A"])
    assert np.array_equal(Y.numpy(), arr[tag + "_collate_Y"])
    # the graph route: parse the same records into graphs, emit the grammar on the GPU, same ids
    vocab = dict(meta[tag + "_vocab"])
    kept = [e for e in ex if e["label"] is not None]
    graphs = [gdl.parse_graph_from_json({"text": e["text"]}, task=task) for e in kept]
    datas = [PygLike(edge_index=np.array(g[0], np.int64).reshape(-1, 2).T, num_nodes=g[1]) for g in graphs]
    batch = gtok.GraphBatch.from_data_list(datas, labeled=False).to(DEV)
    pad = vocab["<pad>"]
    q = np.zeros((len(kept), 4), np.int32)
    for i, e in enumerate(kept):
        toks = e["text"].split()
        qt = toks[toks.index("<q>") + 1: toks.index("<p>")]
        q[i, 0] = len(qt); q[i, 1:1 + len(qt)] = [vocab.get(t, pad) for t in qt]
    ids, ln = gtok.ops.ibtt_synth(batch, gtok.ops.synth_lut(vocab, 64), torch.from_numpy(q), 600, pad)
    assert _rows(ids, ln) == unpad(arr[f"{tag}_L600_ids"], arr[f"{tag}_L600_len"])


class _Replay:
    """Stub tokenizer replaying the fixture's token tensors (what make_golden.py fed the reference)."""
    pad = 5

    def __init__(self, seqs, idx_offset, max_nodes, ntypes):
        self.seqs, self.i, self.idx_offset = seqs, 0, idx_offset
        self.node_idx_offset = idx_offset + max_nodes
        self.edge_idx_offset = self.node_idx_offset + ntypes
        self.labeled_graph = ntypes > 0

    def __call__(self, data):
        s = self.seqs[self.i % len(self.seqs)]
        self.i += 1
        return s.clone()


def test_agtt_dataset_glue_equals_reference():
    arr, meta = golden()
    seqs = [torch.tensor(r) for r in unpad(arr["agtt_remap_in"], arr["agtt_remap_len"])]
    datas = [PygLike(y=torch.tensor([0.5]), num_nodes=5) for _ in seqs]
    ds = gtok.agtt.TokenizedGraphDataset(datas, _Replay(seqs, 6, 37, 9), task="zinc", remap_to_fixed_vocab=True, device=DEV)
    items = [ds[i] for i in range(len(ds))]
    assert [it[0].tolist() for it in items] == unpad(arr["agtt_remap_out"], arr["agtt_remap_len"])
    assert all(it[0].dtype == torch.long and it[1].dtype == torch.bool and it[1].all() for it in items)
    assert isinstance(items[0][2], float)
    # shortest_path query append, including the items that carry no query fields
    seqs2 = [torch.tensor(r) for r in unpad(arr["agtt_sp_in"], arr["agtt_sp_in_len"])]
    datas2 = []
    for n, (u, v), y in zip(arr["agtt_sp_num_nodes"], arr["agtt_sp_query"], arr["agtt_sp_labels"]):
        d = PygLike(y=torch.tensor([int(y)]), num_nodes=int(n))
        if u >= 0:
            d.query_u, d.query_v = int(u), int(v)
        datas2.append(d)
    ds2 = gtok.agtt.TokenizedGraphDataset(datas2, _Replay(seqs2, 6, 49, 0), task="shortest_path", device=DEV)
    items2 = [ds2[i] for i in range(len(ds2))]
    assert [it[0].tolist() for it in items2] == unpad(arr["agtt_sp_out"], arr["agtt_sp_out_len"])
    X, A, Y, dl = gtok.agtt.collate_fn(items2[:16])
    assert np.array_equal(X.numpy(), arr["agtt_sp_collate_X"]) and np.array_equal(A.numpy(), arr["agtt_sp_collate_A"])
    assert np.array_equal(Y.numpy(), arr["agtt_sp_collate_Y"]) and len(dl) == 16


def test_agtt_end_to_end_with_the_gpu_tokenizer():
    """The train_agtt.py flow with our Graph2TrailTokenizer: per-item fetches serve the epoch slab, a second
    fetch of the same item gives a NEW trail, ids stay inside the fixed-vocab ranges, device batches == collate_fn."""
    Graph2TrailTokenizer = gtok.Graph2TrailTokenizer
    datas = zinc_data_list(golden_zinc_coo())
    src = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=datas)
    tok = Graph2TrailTokenizer(dataset_names=[], max_length=1024, truncation_length=1024, labeled_graph=True, undirected=True)
    max_nodes = max(d.num_nodes for d in src)
    tok.set_num_nodes(max_nodes); tok.set_num_node_and_edge_types(*gdl.get_zinc_num_types())
    ds = gtok.agtt.TokenizedGraphDataset(src, tok, task="zinc", remap_to_fixed_vocab=True, device=DEV)
    first = [ds[i] for i in range(len(ds))]
    vocab_size = 22 + max_nodes + 100                                    # train_agtt.py:561
    for t, m, y, d in first:
        assert t[0] == 0 and t[-1] == 1 and m.all() and isinstance(y, float)
        if int(d.x.max()) <= 27 and (d.edge_attr.numel() == 0 or int(d.edge_attr.max()) <= 4):   # real ZINC ranges
            assert int(t.max()) < vocab_size
    assert ds._epoch == 0
    again = [ds[i][0] for i in range(len(ds))]                            # second pass over the data: epoch 1
    assert ds._epoch == 1 and any(not torch.equal(a, f[0]) for a, f in zip(again, first))
    X, A, Y, dl = gtok.agtt.collate_fn(first[:16])
    ds3 = gtok.agtt.TokenizedGraphDataset(src, tok, task="zinc", remap_to_fixed_vocab=True, device=DEV)
    Xd, Ad, Yd, dld = next(ds3.device_batches(16, epoch=0))
    assert torch.equal(Xd.cpu(), X) and torch.equal(Ad.cpu(), A) and torch.equal(Yd.cpu(), Y) and Yd.dtype == torch.float32
    # single-item call site (tokenizer(data)): a valid SENT of that molecule
    one = tok(datas[3])
    assert one.dtype == torch.long and one[0] == 0 and one[-1] == 4


def _collate_with(pad):
    def f(b):
        return gdl.collate(b, pad)
    return f


@pytest.mark.parametrize("workers", [2, 0])
def test_token_dataset_through_dataloader_workers(workers):
    """The reference call site (trainer/train_ibtt.py:395-402, configs/ibtt_*.yaml num_workers: 2): the dataset is
    built in the parent (GPU launch + one copy back), worker PROCESSES then index it and run the reference's
    collate.  First batch == the reference's golden collate, every batch == collate over the golden rows."""
    from torch.utils.data import DataLoader
    arr, meta = golden()
    cases = [("zinc_L1024", dict(meta["zinc_L1024_vocab"]), None, 1024, "<pad>")]
    for task in ("cycle_check", "shortest_path"):
        cases.append(("synth_" + task, dict(meta[f"synth_{task}_vocab"]), meta[f"synth_{task}_examples"], 600, "<pad>"))
    for tag, vocab, ex, max_len, padtok in cases:
        if ex is None:
            ds = gdl.ZINCTokenizationDataset(split="train", max_len=max_len, zinc_dataset=zinc_data_list(golden_zinc_coo()))
            ex = [ds[i] for i in range(len(ds))]
            ids_key, len_key = f"{tag}_ids", f"{tag}_len"
        else:
            ids_key, len_key = f"{tag}_L{max_len}_ids", f"{tag}_L{max_len}_len"
        td = gdl.TokenDataset(ex, vocab, max_len)
        pad = vocab[padtok]
        dl = DataLoader(td, batch_size=16, shuffle=False, num_workers=workers, collate_fn=_collate_with(pad))
        want_rows = unpad(arr[ids_key], arr[len_key])
        seen = 0
        for b, (X, A, Y) in enumerate(dl):
            rows = want_rows[seen:seen + X.shape[0]]
            L = max(len(r) for r in rows)
            assert X.dtype == torch.long and A.dtype == torch.bool and Y.dtype == torch.long and X.shape[1] == L
            for i, r in enumerate(rows):
                assert X[i, :len(r)].tolist() == r and (X[i, len(r):] == pad).all()
                assert A[i].tolist() == [True] * len(r) + [False] * (L - len(r))
            if b == 0:
                assert np.array_equal(X.numpy(), arr[tag + "_collate_X"]) and np.array_equal(A.numpy(), arr[tag + "_collate_A"])
                assert np.array_equal(Y.numpy(), arr[tag + "_collate_Y"])
            seen += X.shape[0]
        assert seen == len(td) == len(want_rows)


def test_token_dataset_pickles_without_device_members():
    """Spawned DataLoader workers get a pickled copy: it must carry the CPU rows and no device tensor."""
    import pickle
    arr, meta = golden()
    ex = meta["synth_cycle_check_examples"]
    td = gdl.TokenDataset(ex, dict(meta["synth_cycle_check_vocab"]), 600)
    cp = pickle.loads(pickle.dumps(td))
    assert cp.ids is None and cp.lens is None and len(cp) == len(td)
    assert all(torch.equal(a, b) and not a.is_cuda for a, b in zip(cp.seqs, td.seqs))
    with pytest.raises(gtok.GtokError):
        next(cp.device_batches(4))


def test_agtt_items_through_dataloader_no_workers():
    """trainer/train_agtt.py:599-607 (configs/agtt_*.yaml num_workers: 0): DataLoader + collate_fn over the
    GPU-tokenized items == collate_fn over the items fetched directly for the same epoch."""
    from torch.utils.data import DataLoader
    datas = zinc_data_list(golden_zinc_coo())
    src = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=datas)
    tok = gtok.Graph2TrailTokenizer(dataset_names=[], max_length=1024, truncation_length=1024, labeled_graph=True, undirected=True)
    tok.set_num_nodes(max(d.num_nodes for d in src)); tok.set_num_node_and_edge_types(*gdl.get_zinc_num_types())
    a = gtok.agtt.TokenizedGraphDataset(src, tok, task="zinc", remap_to_fixed_vocab=True, device=DEV)
    b = gtok.agtt.TokenizedGraphDataset(src, tok, task="zinc", remap_to_fixed_vocab=True, device=DEV)
    direct = [b[i] for i in range(len(b))]
    seen = 0
    for X, A, Y, dl in DataLoader(a, batch_size=16, shuffle=False, num_workers=0, collate_fn=gtok.agtt.collate_fn):
        Xw, Aw, Yw, _ = gtok.agtt.collate_fn(direct[seen:seen + X.shape[0]])
        # (round 4: the loader fetches whole batches through __getitems__, collated on the device - the tensors the trainer
        # moves `.to(device)` anyway, train_agtt.py:309)
        assert X.device.type == "cuda" and torch.equal(X.cpu(), Xw) and torch.equal(A.cpu(), Aw) and torch.equal(Y.cpu(), Yw) and len(dl) == X.shape[0]
        seen += X.shape[0]
    assert seen == len(a)
    # worker processes would have to launch kernels: refused with a clear message instead of a HIP re-init failure
    with pytest.raises(Exception, match="num_workers"):
        next(iter(DataLoader(a, batch_size=4, num_workers=1, collate_fn=gtok.agtt.collate_fn)))


@pytest.mark.parametrize("task", ["cycle_check", "shortest_path"])
def test_config1_token_dataset_equals_reference(task):
    """BASELINE config 1 (1,002 graph-token records per task) through the product: the reference
    serves this configuration on the CPU; the product has no CPU path and serves it with the same kernels
    (DESIGN.md section 2).  TokenDataset rows, labels, first 128-row batch via DataLoader(num_workers=2) and via
    device_batches == the reference's."""
    from torch.utils.data import DataLoader
    arr, meta = golden2()
    tag = "config1_" + task
    vocab = dict(meta[tag + "_vocab"])
    td = gdl.TokenDataset(config1_examples(task), vocab, 600)
    want = unpad(arr[tag + "_ids"].astype(np.int64), arr[tag + "_len"])
    assert [s.tolist() for s in td.seqs] == want
    assert [int(y) for y in td.labels] == arr[tag + "_y"].tolist()
    X, A, Y = next(iter(DataLoader(td, batch_size=128, shuffle=False, num_workers=2, collate_fn=_collate_with(vocab["<pad>"]))))
    assert np.array_equal(X.numpy(), arr[tag + "_collate_X"].astype(np.int64)) and np.array_equal(A.numpy(), arr[tag + "_collate_A"])
    assert np.array_equal(Y.numpy(), arr[tag + "_collate_Y"].astype(np.int64))
    Xd, Ad, Yd = next(td.device_batches(128))
    assert torch.equal(Xd.cpu(), X) and torch.equal(Ad.cpu(), A) and torch.equal(Yd.cpu(), Y)
