"""GPU: throughput-critical behaviour of the drop-in boundary (what `train.py --model agtt/ibtt` sees).

 * the reference's per-item AGTT loop (trainer/train_agtt.py:246-273, restated below) over this package's dataset
   classes costs ONE gtok_sent launch per split and epoch, and yields exactly the rows of the batched call;
 * a split held in collated storage (torch_geometric's InMemoryDataset layout: zinc_dataset_autograph.py:44) reaches
   the device as a batched CSR without any per-item work - ZINC-full in well under a second.
"""
import time

import numpy as np
import pytest
import torch

from _util import both, gtok, orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
gdl = gtok.graph_data_loader


def _reference_getitem(pyg_dataset, tokenizer, idx, remap, task="zinc"):
    """trainer/train_agtt.py:246-273, restated: fetch, tokenize, (remap), (query tail), mask, label."""
    data = pyg_dataset[idx]
    tokens = tokenizer(data)
    if remap:
        io, no, eo = tokenizer.idx_offset, tokenizer.node_idx_offset, tokenizer.edge_idx_offset
        tokens = torch.from_numpy(orc.remap_zinc(tokens.numpy().astype(np.int32)[None, :], np.array([tokens.numel()], np.int32), io, no, eo)[0]).long()
    if task == "shortest_path" and hasattr(data, "query_u") and hasattr(data, "query_v"):
        off = tokenizer.idx_offset
        tokens = torch.cat([tokens, torch.tensor([off + data.num_nodes, off + data.query_u, off + data.query_v], dtype=torch.long)])
    return tokens, torch.ones(tokens.size(0), dtype=torch.bool), data.y.item(), data


def _zinc_tokenizer(max_nodes):
    tok = gtok.Graph2TrailTokenizer(dataset_names=[], max_length=1024, truncation_length=1024, labeled_graph=True, undirected=True,
                                    seed=5, device=DEV)
    tok.set_num_nodes(max_nodes); tok.set_num_node_and_edge_types(*gdl.get_zinc_num_types())
    return tok


def test_reference_per_item_loop_is_one_launch_per_k_epochs():
    """The reference's per-item loop over this package's dataset: ONE gtok_sent launch serves K epochs of the split
    (K = tokenizer.epochs_for(G) = 32 for the 12 k split: trails depend on (seed, epoch, graph) only), every epoch's rows
    equal to the single-epoch batched call and to the oracle."""
    G = 12000                                                            # BASELINE config 2
    d = gtok.synth.zinc_like(G, seed=40)
    batch, coo = both(d)
    pyg = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=gtok.synth.InMemoryLike(d))
    tok = _zinc_tokenizer(37)
    order = torch.randperm(G, generator=torch.Generator().manual_seed(3)).tolist()      # shuffle=True loader
    for epoch in range(2):
        got = {}
        for i in order:
            tokens, mask, label, data = _reference_getitem(pyg, tok, i, remap=True)
            got[i] = tokens
            assert mask.all() and mask.numel() == tokens.numel() and label == pytest.approx(float(d["y"][i]))
        assert tok.epochs_for(G) == 32 and tok.launches == 1, "one gtok_sent launch per K epochs, however the items are fetched"
        ids, ln = tok.tokenize_batch(batch.to(DEV), epoch=epoch, remap_zinc=True)       # the batched call, fused remap
        ref, rln = orc.sent(coo, 37, 1024, 5, epoch, ld=ids.shape[1], labeled=True, num_node_types=9, num_edge_types=4,
                            remap_zinc=True, nthreads=8)
        tok.launches -= 1
        assert np.array_equal(ids.cpu().numpy(), ref) and np.array_equal(ln.cpu().numpy(), rln)
        for i in range(G):
            assert got[i].dtype == torch.long and np.array_equal(got[i].numpy(), ref[i, :rln[i]]), (epoch, i)
    # an object the datasets did not hand out is tokenized on its own (one small launch), still a valid SENT
    loose = gtok.synth.InMemoryLike(d)[11]
    before = tok.launches
    one = tok(loose)
    assert tok.launches == before + 1 and one[0] == 0 and one[-1] == 4
    # fetching one item of a finished epoch again starts the next epoch (a new random trail) - a slice of the same launch
    before = tok.launches
    t1 = tok(pyg[0]); t2 = tok(pyg[1])
    assert tok.launches == before
    ids, ln = tok.tokenize_batch(batch.to(DEV), epoch=2)
    assert torch.equal(t1, ids[0, :int(ln[0])].cpu().long()) and torch.equal(t2, ids[1, :int(ln[1])].cpu().long())
    # with an epoch per launch (epochs_per_launch=1) every epoch is a launch of its own, same rows
    tok1 = _zinc_tokenizer(37); tok1.epochs_per_launch = 1
    for epoch in range(2):
        rows = [tok1(pyg[i]) for i in range(0, G, 97)]
        assert tok1.launches == epoch + 1
        ids, ln = tok.tokenize_batch(batch.to(DEV), epoch=epoch)
        assert all(torch.equal(r, ids[i, :int(ln[i])].cpu().long()) for r, i in zip(rows, range(0, G, 97)))


def test_three_splits_share_a_tokenizer_and_the_swapped_in_dataset_serves_packed_rows():
    """train_agtt.py:594-607: train / val / test datasets around ONE tokenizer.  Each split keeps its own epoch; the
    one-line-swap class (agtt.TokenizedGraphDataset) gives the same rows through its own packed host copy."""
    ds = [gtok.synth.zinc_like(n, seed=60 + k) for k, n in enumerate((3000, 700, 500))]
    pygs = [gdl.ZINCDatasetForAutoGraph(split=s, zinc_dataset=gtok.synth.InMemoryLike(d)) for s, d in zip(("train", "val", "test"), ds)]
    tok = _zinc_tokenizer(37)
    for rounds, pyg in zip((2, 1, 1), pygs):
        for _ in range(rounds):
            for i in range(len(pyg)):
                tok(pyg[i])
    assert tok.launches == 3          # one launch per split: the second epoch of the train split is a slice of the first launch
    for pyg, d in zip(pygs, ds):
        _, coo = both(d)
        fast = gtok.agtt.TokenizedGraphDataset(pyg, tok, task="zinc", remap_to_fixed_vocab=True, device=DEV)
        before = tok.launches
        items = [fast[i] for i in range(len(fast))]
        assert tok.launches == before + 1
        ref, rln = orc.sent(coo, 37, 1024, 5, 0, ld=256, labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True, nthreads=8)
        for i, (t, m, y, data) in enumerate(items):
            assert np.array_equal(t.numpy(), ref[i, :rln[i]]) and m.all() and isinstance(y, float)
        again = fast[0][0]                                               # second fetch: next epoch's trail, from the same launch
        assert tok.launches == before + 1 and fast._epoch == 1
        ref1, rln1 = orc.sent(coo, 37, 1024, 5, 1, ld=256, labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True, nthreads=8)
        assert np.array_equal(again.numpy(), ref1[0, :rln1[0]])


def test_graph_token_split_with_queries_through_the_reference_loop(tmp_path):
    tree = gtok.synth.graph_token_tree(30, seed=9, task="shortest_path")
    gtok.synth.write_tree(str(tmp_path), tree)
    pyg = gdl.GraphTokenDatasetForAutoGraph(root=str(tmp_path), task="shortest_path", algorithm=["er", "ba", "sbm"], split="train",
                                            num_pairs_per_graph=3)
    tok = gtok.Graph2TrailTokenizer(dataset_names=[], max_length=600, truncation_length=600, labeled_graph=False, seed=2, device=DEV)
    tok.set_num_nodes(max(dd.num_nodes for dd in pyg))
    rows = [_reference_getitem(pyg, tok, i, remap=False, task="shortest_path")[0] for i in range(len(pyg))]
    assert tok.launches == 1 and len(rows) > 50
    b = pyg.graph_batch(device=DEV)
    ids, ln = tok.tokenize_batch(b, epoch=0, query=torch.as_tensor(pyg.queries()))
    for i, r in enumerate(rows):
        assert np.array_equal(r.numpy(), ids[i, :int(ln[i])].cpu().numpy()), i
    fast = gtok.agtt.TokenizedGraphDataset(pyg, tok, task="shortest_path", device=DEV)
    assert all(torch.equal(fast[i][0], rows[i]) for i in range(len(pyg)))
    # the same split behind the stock DataLoader (batch-level fetch): rows incl. their query tails, int64 labels, and the
    # `data_list[0].num_nodes` the model computes its <q> id from (train_agtt.py:127-133)
    from torch.utils.data import DataLoader
    fast2 = gtok.agtt.TokenizedGraphDataset(pyg, tok, task="shortest_path", device=DEV)
    seen = 0
    for X, A, Y, datas in DataLoader(fast2, batch_size=32, shuffle=False, num_workers=0, collate_fn=gtok.agtt.collate_fn):
        for b in range(X.shape[0]):
            r = rows[seen + b]
            assert torch.equal(X[b, :r.numel()].cpu(), r) and bool(A[b, :r.numel()].all()) and not bool(A[b, r.numel():].any())
            assert int(Y[b]) == pyg[seen + b].y.item()
        assert Y.dtype == torch.long and datas[0].num_nodes == pyg[seen].num_nodes and int(X[0, rows[seen].numel() - 3]) == tok.idx_offset + datas[0].num_nodes
        seen += X.shape[0]
    assert seen == len(pyg)


def test_zinc_full_ingestion_from_collated_storage_under_a_second():
    G = 249456
    d = gtok.synth.zinc_like(G, seed=1000)
    ds = gtok.synth.InMemoryLike(d)
    gtok.GraphBatch.from_dataset(gtok.synth.InMemoryLike(gtok.synth.zinc_like(2000, seed=1)), device=DEV)   # torch kernels warmed up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = gtok.GraphBatch.from_dataset(ds, device=DEV)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
    for k in ("node_ptr", "edge_ptr", "rowptr", "col", "nattr", "eattr"):
        assert torch.equal(getattr(b, k).cpu(), getattr(ref, k)), k
    assert b.eorder is None and (b.flags, b.max_nodes, b.max_edges, b.chunk_nodes, b.chunk_edges, b.max_degree) == \
        (ref.flags, ref.max_nodes, ref.max_edges, ref.chunk_nodes, ref.chunk_edges, ref.max_degree)
    print(f"ZINC-full ingestion from collated storage: {dt:.3f} s")
    assert dt < 1.0, f"{dt:.3f} s"


@pytest.mark.parametrize("max_len", [1024, 48, 2])
def test_zinc_strings_rendered_on_the_device_equal_the_reference(max_len):
    """ZINCTokenizationDataset.__getitem__ strings (zinc_dataset_indexbase.py:143-227; train_ibtt.py:229-235 fetches
    every item of every split): rendered for the whole split by gtok_ibtt_zinc + gtok_ids_to_text, equal to the texts
    the reference itself produced (golden) and to the per-item Python restatement on corpora with every fallback."""
    from _util import golden, golden_zinc_coo, zinc_data_list, edge_case_graphs
    _, meta = golden()
    ds = gdl.ZINCTokenizationDataset(split="train", max_len=max_len, zinc_dataset=zinc_data_list(golden_zinc_coo()))
    items = [ds[i] for i in range(len(ds))]
    assert ds._texts is not None, "the split was rendered in one go"
    if max_len in (1024, 48):
        assert [it["text"] for it in items] == meta[f"zinc_L{max_len}_texts"]
        assert [it["label"] for it in items] == meta[f"zinc_L{max_len}_labels"]
        assert [it["graph_id"] for it in items] == meta[f"zinc_L{max_len}_graph_ids"]
    assert [it["text"] for it in items] == [ds._item(i)["text"] for i in range(len(ds))]
    for d in (gtok.synth.zinc_like(3000, seed=5), gtok.synth.zinc_like(500, seed=6, coalesced=False), edge_case_graphs()):
        d = dict(d)
        d.setdefault("y", np.linspace(-3.0, 3.0, len(d["node_counts"])).astype(np.float32))
        d["y"][:3] = [-0.004, 0.005, -12.345]                       # val_neg0_00, rounding, sign
        for src in (zinc_data_list(d), gtok.synth.InMemoryLike(d)):
            z = gdl.ZINCTokenizationDataset(split="val", max_len=max_len, zinc_dataset=src)
            got = [z[i] for i in range(len(z))]
            want = [z._item(i) for i in range(len(z))]
            assert z._texts is not None and got == want


@pytest.mark.parametrize("task", ["cycle_check", "shortest_path"])
def test_process_parses_text_records_on_the_device(task, tmp_path):
    """GraphTokenDatasetForAutoGraph.process() (graph_token_dataset_autograph.py:259-408) with the text records of all
    files parsed by one pair of launches: item for item what the host parsers give - including hand-edited texts outside
    the canonical grammar, records with explicit fields, INF / unlabeled / empty ones and the per-file pair sampling."""
    import json, os
    tree = gtok.synth.graph_token_tree(90, seed=21, task=task, algorithms=("er", "ba", "sbm", "path"), splits=("train",),
                                       pairs_per_graph=4 if task == "shortest_path" else 1)
    keys = sorted(tree)
    tree[keys[0]][0] = {"text": tree[keys[0]][0]["text"].replace("<n>", "<n> x"), "label": 1}        # not canonical: host fallback
    tree[keys[1]][0] = {"edges": [[0, 1], [1, 2]], "nodes": [0, 1, 2], "label": 0, "text": tree[keys[1]][0]["text"]}
    tree[keys[2]][0] = {"text": "<bos> <n> <q> has_cycle <p> yes <eos>"}                                # no nodes: skipped
    tree[keys[3]][0] = {"text": tree[keys[3]][0]["text"].rsplit("<p>", 1)[0] + "<p> maybe <eos>"}      # no label: skipped
    gtok.synth.write_tree(str(tmp_path), tree)
    G = gdl.GraphTokenDatasetForAutoGraph
    kw = dict(root=str(tmp_path), task=task, algorithm=["er", "ba", "sbm", "path"], split="train", use_cache=False,
              num_pairs_per_graph=2 if task == "shortest_path" else None)
    calls = []
    real = gtok.ops.parse_graph_texts
    gtok.ops.parse_graph_texts = lambda *a, **k: (calls.append(int(a[1].numel()) - 1), real(*a, **k))[1]
    try:
        dev_ds = G(**kw)                                           # collated storage cut straight out of the parser's arrays
        obj_ds = G(**kw, pre_transform=lambda d: d)                # the route that needs objects: device parse, then one Data per record
        G.DEVICE_PARSE_MIN = 10 ** 9
        host_ds = G(**kw)
    finally:
        G.DEVICE_PARSE_MIN = 256
        gtok.ops.parse_graph_texts = real
    assert len(calls) == 2 and calls[0] == calls[1] > 200, "the device parser took the text records of every file in one call"
    assert len(dev_ds) == len(host_ds) == len(obj_ds) > 100
    for a, b in zip(obj_ds, host_ds):
        assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.y, b.y) and a.num_nodes == b.num_nodes
        assert (getattr(a, "query_u", None), getattr(a, "query_v", None)) == (getattr(b, "query_u", None), getattr(b, "query_v", None))
    hc = G.collate(list(host_ds))                                  # and the collated pair itself, array for array
    for k in hc[0]:
        assert torch.equal(dev_ds._coll[0][k], hc[0][k]) and dev_ds._coll[0][k].dtype == hc[0][k].dtype, k
    for k in hc[1]:
        assert torch.equal(dev_ds._coll[1][k], hc[1][k]), k
    for a, b in zip(dev_ds, host_ds):
        assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.y, b.y) and a.num_nodes == b.num_nodes
        assert (getattr(a, "query_u", None), getattr(a, "query_v", None)) == (getattr(b, "query_u", None), getattr(b, "query_v", None))


def test_text_to_ids_with_a_vocab_too_large_for_lds():
    """The vocab the ZINC trainer really builds (train_ibtt.py:361-372) carries one `val_x_xx` token per distinct label -
    thousands of keys, none of which is ever emitted: too large for the workgroup's LDS copy, so the kernel probes the
    global table for long tokens while the few dozen short keys that do occur are matched through the LDS short-key table
    (filled shortest keys first until half full; a miss there is not final).  Rows == the oracle's, also with a vocab id of -1
    and with tokens that are in no table."""
    d = gtok.synth.zinc_like(4000, seed=77)
    ds = gdl.ZINCTokenizationDataset(split="train", max_len=1024, zinc_dataset=gtok.synth.InMemoryLike(d))
    texts = [ds[i]["text"] for i in range(len(ds))]
    vocab, _ = gdl.build_fixed_zinc_vocab()
    dyn = []
    for t in texts:
        for w in t.split():
            if w not in vocab and w not in dyn[-50:]:
                dyn.append(w)
    vocab = gdl.extend_vocab_with_dynamic_tokens(vocab, dict.fromkeys(dyn))
    for k in range(3000):                                   # far more label tokens than the texts at hand carry
        vocab.setdefault(f"val_{k}_{k % 100:02d}", len(vocab))
    vocab["7"] = -1                                          # an id of -1 stays out of the short-key table
    vocab.pop("single")                                      # a frequent token that is in no table: pad id
    assert len(vocab) > 3000
    blob, ptr = gtok.ops.pack_texts(texts)
    table = gtok.ops.VocabTable(vocab, DEV)
    assert table.capacity > 1024
    for max_len, strip in ((1024, True), (1024, False), (40, True)):
        ids, ln = gtok.ops.text_to_ids(blob.to(DEV), ptr, table, max_len, strip, ld=256)
        ref, rln = orc.text_to_ids(texts, vocab, max_len, 256, strip_label=strip, nthreads=8)
        assert np.array_equal(ln.cpu().numpy(), rln) and np.array_equal(ids.cpu().numpy(), ref), (max_len, strip)


def test_text_to_ids_adopts_left_out_short_keys_while_splitting():
    """With a large vocab the short-key table of a workgroup is filled at set-up only part of the way (keys of up to 7 bytes
    first, 8..11-byte ones while it is under a quarter full); a left-out key that does occur in the texts is adopted into the
    table by the first look-up that finds it the slow way.  Texts that use MORE distinct 8..11-byte keys than the table can
    ever hold, the same key many times per step, ids of -1 and -2 (never in the table: -2 + 1 is the busy mark), a token that
    is a single NUL byte (an all-zero key - what a slot mid-write looks like), unknown tokens and labels at the very end of a
    text (the last 12 bytes go through the same path now): rows == the oracle's."""
    rng = np.random.default_rng(31)
    vocab = {"<pad>": 0, "<bos>": 1, "<p>": 2, "<eos>": 3, "regression": 4, "aromatic": 5, "\0": 6}
    for k in range(5000):
        vocab.setdefault(f"key_{k:05d}", len(vocab))          # 9 bytes each: far more than 2048 slots take
    for k in range(60):
        vocab.setdefault(str(k), len(vocab))
    vocab["key_00007"] = -1
    vocab["key_00008"] = -2
    vocab["11"] = -2
    names = list(vocab)
    texts = []
    for t in range(3000):
        hot = [names[int(j)] for j in rng.integers(0, len(names), 6)]
        toks = ["<bos>"] + [hot[int(j)] if rng.random() < 0.7 else names[int(rng.integers(0, len(names)))]
                            for j in rng.integers(0, 6, int(rng.integers(0, 160)))]
        toks += ["regression", "nokey_123", "aromatic", "\0", "regression"]
        toks += ["<p>", f"key_{t % 5000:05d}"] + (["<eos>"] if t % 3 else [])
        texts.append(" ".join(toks) + (" " if t % 5 == 0 else ""))
    texts += ["key_00009", "\0", "\0 \0", "<p>", "key_00008", ""]
    blob, ptr = gtok.ops.pack_texts(texts)
    table = gtok.ops.VocabTable(vocab, DEV)
    assert table.capacity > 1024
    for max_len, strip in ((256, False), (256, True), (33, False)):
        ids, ln = gtok.ops.text_to_ids(blob.to(DEV), ptr, table, max_len, strip, ld=256)
        ref, rln = orc.text_to_ids(texts, vocab, max_len, 256, strip_label=strip, nthreads=8)
        assert np.array_equal(ln.cpu().numpy(), rln) and np.array_equal(ids.cpu().numpy(), ref), (max_len, strip)


def test_stock_dataloader_fetches_whole_batches_through_getitems():
    """trainer/train_agtt.py:599-607 unchanged - `DataLoader(ds, batch_size, shuffle, num_workers=0, collate_fn=collate_fn)` -
    over the one-line-swap class: the loader hands its index list to __getitems__, the batch is collated on the device and
    collate_fn passes it through.  Batches equal what per-item __getitem__ + collate_fn give for the same epoch (shuffle=True
    and False); a second pass over the loader is the next epoch; a collate function that knows nothing about it (the
    reference's own) still gets per-item tuples."""
    from torch.utils.data import DataLoader
    G = 3000
    d = gtok.synth.zinc_like(G, seed=77)
    _, coo = both(d)
    pyg = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=gtok.synth.InMemoryLike(d))
    agtt = gtok.agtt
    ds = agtt.TokenizedGraphDataset(pyg, _zinc_tokenizer(37), task="zinc", remap_to_fixed_vocab=True, device=DEV)
    refs = [orc.sent(coo, 37, 1024, 5, e, ld=256, labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True, nthreads=8) for e in range(3)]

    def check(X, A, Y, datas, idx, epoch):
        ref, rln = refs[epoch]
        L = int(rln[idx].max())
        assert X.dtype == torch.long and A.dtype == torch.bool and tuple(X.shape) == (len(idx), L) and X.device.type == "cuda"
        Xh, Ah = X.cpu().numpy(), A.cpu().numpy()
        for b, i in enumerate(idx):
            n = int(rln[i])
            assert np.array_equal(Xh[b, :n], ref[i, :n]) and (Xh[b, n:] == 5).all() and Ah[b, :n].all() and not Ah[b, n:].any()
        assert Y.dtype == torch.float32 and np.allclose(Y.cpu().numpy(), d["y"][idx]) and len(datas) == len(idx)
        assert datas[0].num_nodes == int(d["node_counts"][idx[0]])
    dl = DataLoader(ds, batch_size=128, shuffle=True, num_workers=0, collate_fn=agtt.collate_fn, generator=torch.Generator().manual_seed(11))
    launches = ds.tokenizer.launches
    y32 = d["y"].astype(np.float32)
    assert np.unique(y32).size == G                          # rows of a shuffled batch are identified by their label
    by_label = {float(v): i for i, v in enumerate(y32)}
    for epoch in range(2):                                   # two passes: two epochs, one launch (K epochs per launch)
        seen = set()
        for X, A, Y, datas in dl:
            idx = np.asarray([by_label[float(v)] for v in Y.cpu().numpy()])
            check(X, A, Y, datas, idx, epoch)
            seen.update(idx.tolist())
        assert len(seen) == G
    assert ds.tokenizer.launches == launches + 1
    # shuffle=False (val / test loaders): dataset order, next epoch
    dl2 = DataLoader(ds, batch_size=100, shuffle=False, num_workers=0, collate_fn=agtt.collate_fn)
    for k, (X, A, Y, datas) in enumerate(dl2):
        check(X, A, Y, datas, np.arange(k * 100, min(G, (k + 1) * 100)), 2)
    # a collate function that does not know CollatedBatch (the reference's own, restated) gets per-item tuples of the same epoch
    def reference_collate(batch):
        toks, masks, labels, datas = zip(*batch)
        L = max(t.size(0) for t in toks)
        X = torch.full((len(toks), L), 5, dtype=torch.long); Am = torch.zeros((len(toks), L), dtype=torch.bool)
        for i, (t, m) in enumerate(zip(toks, masks)):
            X[i, :t.size(0)] = t; Am[i, :t.size(0)] = m
        return X, Am, torch.tensor(labels, dtype=torch.float if isinstance(labels[0], float) else torch.long), list(datas)
    ds3 = agtt.TokenizedGraphDataset(pyg, _zinc_tokenizer(37), task="zinc", remap_to_fixed_vocab=True, device=DEV)
    for k, (X, A, Y, datas) in enumerate(DataLoader(ds3, batch_size=64, shuffle=False, num_workers=0, collate_fn=reference_collate)):
        assert X.device.type == "cpu"
        check(X.cuda(), A.cuda(), Y.cuda(), datas, np.arange(k * 64, min(G, (k + 1) * 64)), 0)
        if k == 3:
            break


def test_device_batches_collate_the_epoch_in_one_call_and_the_slab_follows_the_tokenizer():
    """agtt.TokenizedGraphDataset.device_batches: every batch of the epoch comes out of ONE gtok_collate_epoch call (views of
    the epoch's arenas) and equals gtok_collate_packed per batch == the oracle's rows; shuffle and a short last batch included.
    ADVICE r4: a tokenizer reconfigured between epochs (seed) must not be served trails of the cached K-epoch slab; a batch
    with a duplicated index goes through the per-item path."""
    from torch.utils.data import DataLoader
    G = 2500
    d = gtok.synth.zinc_like(G, seed=78)
    _, coo = both(d)
    pyg = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=gtok.synth.InMemoryLike(d))
    agtt = gtok.agtt
    ds = agtt.TokenizedGraphDataset(pyg, _zinc_tokenizer(37), task="zinc", remap_to_fixed_vocab=True, device=DEV)
    okw = dict(ld=256, labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True, nthreads=8)
    for epoch, shuffle in ((4, False), (5, True)):
        ref, rln = orc.sent(coo, 37, 1024, 5, epoch, **okw)
        gen = torch.Generator().manual_seed(3)
        order = torch.randperm(G, generator=torch.Generator().manual_seed(3)).numpy() if shuffle else np.arange(G)
        seen = 0
        for k, (X, A, Y, datas) in enumerate(ds.device_batches(96, epoch=epoch, shuffle=shuffle, generator=gen)):
            idx = order[k * 96:(k + 1) * 96]
            L = int(rln[idx].max())
            assert tuple(X.shape) == (len(idx), L) and X.dtype == torch.long and A.dtype == torch.bool and X.is_cuda
            Xh, Ah = X.cpu().numpy(), A.cpu().numpy()
            want = np.where(np.arange(L)[None, :] < rln[idx][:, None], ref[idx][:, :L], 5)
            assert np.array_equal(Xh, want) and np.array_equal(Ah, np.arange(L)[None, :] < rln[idx][:, None])
            assert np.allclose(Y.cpu().numpy(), d["y"][idx]) and len(datas) == len(idx) and datas[1].num_nodes == int(d["node_counts"][idx[1]])
            seen += len(idx)
        assert seen == G and k == (G - 1) // 96
    # the raw op against gtok_collate_packed, int32 slab + explicit row_ptr too
    ids, ln = ds.tokenize_epoch(7)
    order_d = torch.randperm(G, generator=torch.Generator().manual_seed(9)).to(DEV)
    ptr = gtok.ops.row_offsets(ln, ids.shape[1])
    packed, _ = gtok.ops.pack_rows(ids, ln, ptr, elem_bytes=2)
    for src, rp in ((ids, None), (packed, ptr)):
        X, A, lmax, off = gtok.ops.collate_epoch(src, rp, ln, ids.shape[1], order_d, 128, 5)
        for b in (0, 7, len(lmax) - 1):
            sel = order_d[b * 128:(b + 1) * 128]
            Xb, Ab = gtok.ops.collate_packed(src, rp, ln, ids.shape[1], sel, 5, lmax[b])
            B = sel.numel()
            assert torch.equal(X.as_strided((B, lmax[b]), (lmax[b], 1), off[b]), Xb) and torch.equal(A.as_strided((B, lmax[b]), (lmax[b], 1), off[b]), Ab)
        assert off[-1] == sum(min(128, G - 128 * b) * lmax[b] for b in range(len(lmax)))
    # ADVICE: the cached K-epoch slab follows the tokenizer's configuration
    a0, l0 = ds.tokenize_epoch_u16(20)
    a0, l0 = a0.clone(), l0.clone()
    ds.tokenizer.seed = 6
    a1, l1 = ds.tokenize_epoch_u16(21)                 # inside the K epochs of the cached launch, but the seed changed
    r1, rl1 = orc.sent(coo, 37, 1024, 6, 21, **dict(okw, ld=a1.shape[1]))
    assert np.array_equal(l1.cpu().numpy(), rl1)
    got = a1.cpu().numpy().view(np.uint16).astype(np.int32)
    inside = np.arange(a1.shape[1])[None, :] < rl1[:, None]
    assert np.array_equal(np.where(inside, got, 0), np.where(inside, r1, 0))
    ds.tokenizer.seed = 5
    # a batch with a duplicated index: per-item path (two fetches of an item are two trails), no exception
    out = ds.__getitems__([3, 3, 9])
    assert isinstance(out, list) and len(out) == 3 and out[0][0].dtype == torch.long
    # the package's batch sampler: one permutation per epoch cut into index lists - every item once, short last batch, drop_last
    bsamp = agtt.EpochBatchSampler(G, 96, shuffle=True, generator=torch.Generator().manual_seed(4))
    lists = list(bsamp)
    assert len(lists) == len(bsamp) == -(-G // 96) and sorted(i for l in lists for i in l) == list(range(G)) and len(lists[-1]) == G % 96
    assert len(list(agtt.EpochBatchSampler(G, 96, drop_last=True))) == G // 96 and list(agtt.EpochBatchSampler(5, 2))[-1] == [4]
    seen = 0
    for X, A, Y, datas in DataLoader(ds, batch_sampler=agtt.EpochBatchSampler(G, 96, shuffle=True), num_workers=0, collate_fn=agtt.collate_fn):
        assert X.is_cuda and X.shape[0] == len(datas) == Y.shape[0] and abs(float(Y[0]) - float(datas[0].y)) < 1e-6
        seen += X.shape[0]
    assert seen == G
    # the sampler that announces its batches (dataset=ds): the whole epoch is collated by ONE call when its first batch is asked for,
    # every batch is three views - same tensors as the per-batch route over the same epoch's rows, labels and Data objects included
    gen = torch.Generator().manual_seed(11)
    bs_plan = agtt.EpochBatchSampler(G, 96, shuffle=True, generator=gen, dataset=ds)
    loader = DataLoader(ds, batch_sampler=bs_plan, num_workers=0, collate_fn=agtt.collate_fn)
    epochs_seen = []
    for rep in range(2):
        got = [(X, A, Y, float(datas[0].y), len(datas)) for X, A, Y, datas in loader]
        lists = ds._plan["lists"]
        assert len(got) == len(lists) == -(-G // 96) and sorted(i for l in lists for i in l) == list(range(G))
        epochs_seen.append(ds._epoch)
        ld_ = ds._ids.shape[1]
        for (X, A, Y, y0, n), l in zip(got, lists):
            idx = torch.tensor(l, device=DEV)
            lmax = int(ds._lens[idx].clamp(max=ld_).max())
            rX, rA = gtok.ops.collate_packed(ds._ids, None, ds._lens, ld_, idx, 5, lmax)
            assert n == len(l) and torch.equal(X, rX) and torch.equal(A, rA) and torch.equal(Y, ds._labels_on_device()[idx]) and abs(y0 - float(Y[0])) < 1e-6
    assert epochs_seen[1] == epochs_seen[0] + 1                      # a new plan = the next epoch's trails
    # a list that is not the plan's own object goes the per-batch way (and, its rows being served already, starts the next epoch)
    cb = ds.__getitems__(list(ds._plan["lists"][0]))
    assert isinstance(cb, agtt.CollatedBatch) and ds._epoch == epochs_seen[1] + 1
    assert len(list(agtt.EpochBatchSampler(G, 96, drop_last=True, dataset=ds))) == G // 96
    short = agtt.EpochBatchSampler(ds, 96, shuffle=True)                 # the dataset in place of its length
    assert short.dataset is ds and short.n == G and len(short) == -(-G // 96)
    # an abandoned plan: the loader is dropped after two batches, the next iteration plans again
    it = iter(loader)
    next(it); next(it)
    del it
    assert sum(X.shape[0] for X, _, _, _ in loader) == G
    # (pin_memory=True is not supported over device batches - the loader pins what collate_fn returns, and CUDA tensors cannot be
    # pinned; the reference's loaders do not set it, INTEGRATION.md says so.)  The fetcher's own object passes a pin request through:
    cb = ds.__getitems__([1, 2, 5])
    assert cb.pin_memory() is cb and agtt.collate_fn(cb)[0].is_cuda


def test_token_dataset_batches_through_getitems_equal_per_item_collate():
    """IBTT: `DataLoader(TokenDataset, batch_size, shuffle, num_workers=2, collate_fn=lambda b: collate(b, pad_id))`
    (trainer/train_ibtt.py:399-402) - batches built from the packed host buffer in one go equal the per-item collate."""
    from torch.utils.data import DataLoader
    g = gtok.synth.graph_token_like(700, seed=14, task="cycle_check")
    ex = [{"text": t, "label": int(l)} for t, l in zip(g["texts"], g["labels"])]
    vocab, _ = gdl.build_vocab_from_texts([e["text"] for e in ex], max_tokens=600)
    td = gdl.TokenDataset(ex, vocab, max_len=600, device=DEV)
    pad = vocab["<pad>"]
    per_item = [gdl.collate([td[i] for i in range(s, min(len(td), s + 50))], pad) for s in range(0, len(td), 50)]
    for workers in (0, 2):
        dl = DataLoader(td, batch_size=50, shuffle=False, num_workers=workers, collate_fn=lambda b: gdl.collate(b, pad))
        for (X, A, Y), (rX, rA, rY) in zip(dl, per_item):
            assert torch.equal(X, rX) and torch.equal(A, rA) and torch.equal(Y, rY)


@pytest.mark.parametrize("legacy", [False, True])
def test_agtt_flow_over_a_dataset_with_torch_geometric_semantics(legacy):
    """The whole AGTT boundary on objects that behave like torch_geometric's (tests/_util.py: PygDataLike / PygInMemoryLike -
    attribute access through a storage mapping, a copy per fetch, `_data` or, before 2.3, `data`): the reference's per-item
    loop and the stock DataLoader over agtt.TokenizedGraphDataset give the oracle's rows with one launch per K epochs, items
    the torch_geometric dataset hands out itself are tokenized on their own."""
    from torch.utils.data import DataLoader
    from _util import PygInMemoryLike
    G = 3000
    d = gtok.synth.zinc_like(G, seed=41)
    batch, coo = both(d)
    pyg = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=PygInMemoryLike(d, legacy=legacy))
    tok = _zinc_tokenizer(37)
    got = [_reference_getitem(pyg, tok, i, remap=True) for i in range(G)]
    assert tok.launches == 1
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ref, rln = orc.sent(coo, 37, 1024, 5, 0, ld=gtok.ops.sent_safe_ld(batch, True, 1024), nthreads=8, **kw)
    for i, (tokens, mask, label, data) in enumerate(got):
        assert np.array_equal(tokens.numpy(), ref[i, :rln[i]]) and label == pytest.approx(float(d["y"][i])) and data.num_nodes == int(d["node_counts"][i])
    loose = pyg.zinc_dataset[5]                                    # not handed out by this package's class: a launch of its own
    before = tok.launches
    one = tok(loose)
    assert tok.launches == before + 1 and one[0] == 0 and one[-1] == 4
    ds = gtok.agtt.TokenizedGraphDataset(pyg, _zinc_tokenizer(37), task="zinc", remap_to_fixed_vocab=True)
    seen = 0
    for X, A, Y, data_list in DataLoader(ds, batch_size=128, shuffle=False, num_workers=0, collate_fn=gtok.agtt.collate_fn):
        B = X.shape[0]
        for b in (0, B - 1):
            i = seen + b
            n = int(A[b].sum())
            assert n == rln[i] and np.array_equal(X[b, :n].cpu().numpy(), ref[i, :n]) and float(Y[b]) == pytest.approx(float(d["y"][i]))
            assert data_list[b].num_nodes == int(d["node_counts"][i])
        seen += B
    assert seen == G and ds.tokenizer.launches == 1


def test_ibtt_strings_over_a_dataset_with_torch_geometric_semantics():
    """ZINCTokenizationDataset over the same kind of objects: the strings rendered for the whole split on the device (serialiser +
    gtok_zinc_text_tails + gtok_ids_to_text) equal the restated per-item Python on every item, at a max_len that cuts some texts."""
    from _util import PygInMemoryLike
    d = gtok.synth.zinc_like(800, seed=43)
    for max_len in (2048, 60):
        ds = gdl.ZINCTokenizationDataset(split="val", zinc_dataset=PygInMemoryLike(d), max_len=max_len)
        items = [ds[i] for i in range(len(ds))]
        assert ds._texts is not None, "rendered in bulk on the first fetch"
        for i in range(0, len(ds), 7):
            want = ds._item(i)
            assert items[i]["text"] == want["text"] and items[i]["label"] == pytest.approx(want["label"]) and items[i]["graph_id"] == want["graph_id"]
        if max_len == 60:
            assert any(len(it["text"].split()) == 60 for it in items)
