"""GPU, at BASELINE.json's full single-GPU sizes: the ZINC-full-shaped corpus (249,456 molecules) and a batch of
large graph-token-shaped graphs (10-256 nodes).  The multi-threaded C oracle is fast enough to check every row
bit for bit; on top of that the size-independent properties of the domain (determinism, new trails per epoch,
shard invariance through the global graph index, losslessness of a sample, closed-form IBTT lengths)."""
import numpy as np
import pytest
import torch

from _util import both, gtok, orc, zinc_vocab

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ZINC_FULL = 249456
THREADS = max(1, min(16, orc.num_threads()))


@pytest.fixture(scope="module")
def zinc_full():
    d = gtok.synth.zinc_like(ZINC_FULL, seed=1000)
    batch, coo = both(d)
    return d, batch, batch.to(DEV), coo


def test_sent_zinc_full_bit_exact_and_properties(zinc_full):
    d, host, dev, coo = zinc_full
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ld = 192
    ids, ln = gtok.ops.sent(dev, 37, 1024, 7, 3, ld=ld, **kw)
    ref, rln = orc.sent(coo, 37, 1024, 7, 3, ld=ld, nthreads=THREADS, **kw)
    assert int(ln.max()) <= ld, "slab too narrow for this epoch: the comparison below would be vacuous"
    assert np.array_equal(ln.cpu().numpy(), rln)
    assert np.array_equal(ids.cpu().numpy(), ref)
    # properties on the device tensors
    rows = torch.arange(ZINC_FULL, device=DEV)
    assert bool((ids[:, 0] == 0).all())                                   # <bos>
    assert bool((ids[rows, (ln - 1).long()] == 1).all())                  # <eos> after the remap
    pad_mask = torch.arange(ld, device=DEV)[None, :] >= ln[:, None]
    assert bool((ids[pad_mask] == 5).all())                               # Graph2TrailTokenizer.pad beyond the length
    assert int(ids.max()) < 22 + 37 + 100                                 # vocab_size of train_agtt.py:561 (x <= 27)
    # determinism, a fresh trail per epoch, shard invariance through graph_base
    again, aln = gtok.ops.sent(dev, 37, 1024, 7, 3, ld=ld, **kw)
    assert torch.equal(again, ids) and torch.equal(aln, ln)
    nxt, _ = gtok.ops.sent(dev, 37, 1024, 7, 4, ld=ld, **kw)
    assert float((nxt != ids).any(1).float().mean()) > 0.99
    lo, hi = 100000, 180000
    part, pln = gtok.ops.sent(host.shard(lo, hi).to(DEV), 37, 1024, 7, 3, graph_base=lo, ld=ld, **kw)
    assert torch.equal(part, ids[lo:hi]) and torch.equal(pln, ln[lo:hi])


def test_sent_zinc_full_every_row_decodes_to_its_molecule(zinc_full):
    """Round trip at corpus scale: every one of the 249,456 un-remapped rows the GPU wrote is decoded back (C decoder,
    visit order replayed by the oracle) and equals its input molecule exactly - atoms, bonds, atom and bond types, no
    bond twice, none missing.  Independent of the bit-for-bit comparison above: it checks the SENT spec, not the port."""
    d, host, dev, coo = zinc_full
    ids, ln = gtok.ops.sent(dev, 37, 1024, 11, 0, labeled=True, num_node_types=28, num_edge_types=6, ld=192)
    assert int(ln.max()) <= 192
    st = orc.sent_roundtrip(coo, ids.cpu().numpy(), ln.cpu().numpy(), 37, 1024, 11, 0, labeled=True, num_node_types=28,
                            nthreads=THREADS)
    assert not st.any(), (int((st != 0).sum()), st[st != 0][:10])
    assert bool((ids[torch.arange(ZINC_FULL, device=DEV), (ln - 1).long()] == 4).all())     # every row is complete


def test_sent_zinc_full_lossless_sample(zinc_full):
    d, host, dev, coo = zinc_full
    ids, ln = gtok.ops.sent(dev, 37, 1024, 11, 0, labeled=True, num_node_types=28, num_edge_types=6, ld=192)
    pick = np.random.default_rng(0).choice(ZINC_FULL, 1500, replace=False)
    h, hl = ids[torch.from_numpy(pick).to(DEV)].cpu().numpy(), ln.cpu().numpy()[pick]
    for r, g in enumerate(pick):
        n, edges, ntypes, etypes, used = orc.sent_decode(h[r, :hl[r]].tolist(), 37, True, 28)
        e0, e1, n0 = coo.edge_ptr[g], coo.edge_ptr[g + 1], coo.node_ptr[g]
        assert used == hl[r] and n == coo.node_counts[g] and len(edges) == (e1 - e0) // 2
        assert sorted(ntypes.values()) == sorted(int(v) for v in coo.x[n0:n0 + n])
        assert sorted(etypes.values()) == sorted(int(v) for v in coo.edge_attr[e0:e1][coo.src[e0:e1] < coo.dst[e0:e1]])


def test_ibtt_zinc_full_bit_exact_and_lengths(zinc_full):
    d, host, dev, coo = zinc_full
    vocab = zinc_vocab(40)
    lut = gtok.ops.zinc_lut(vocab, 40)
    ids, ln = gtok.ops.ibtt_zinc(dev, lut, 1024, vocab["<pad>"], ld=240)
    ref, rln = orc.ibtt_zinc(coo, lut.numpy(), 1024, vocab["<pad>"], 240, nthreads=THREADS)
    assert int(ln.max()) <= 240
    assert np.array_equal(ln.cpu().numpy(), rln) and np.array_equal(ids.cpu().numpy(), ref)
    # closed form: 4 + 2N + 4B ids (B undirected bonds; the synthetic molecules list both directions once)
    want = 4 + 2 * d["node_counts"] + 4 * (d["edge_counts"] // 2)
    assert np.array_equal(ln.cpu().numpy(), want)
    # idempotence / determinism
    again, aln = gtok.ops.ibtt_zinc(dev, lut, 1024, vocab["<pad>"], ld=240)
    assert torch.equal(again, ids) and torch.equal(aln, ln)


def test_large_graphs_bit_exact():
    """BASELINE config 5 shape: 10..256 nodes, sparsity 0.1-0.2, max_len 600 (most trails are cut there)."""
    d = gtok.synth.er_batch_device(4096, torch.device(DEV), seed=1000)
    batch, coo = both(d, False)
    dev = batch.to(DEV)
    ids, ln = gtok.ops.sent(dev, 256, 600, 3, 1, ld=600)
    ref, rln = orc.sent(coo, 256, 600, 3, 1, ld=600, nthreads=THREADS)
    assert np.array_equal(ln.cpu().numpy(), rln) and np.array_equal(ids.cpu().numpy(), ref)
    assert float((ln == 600).float().mean()) > 0.5, "the truncation path must be hot in this configuration"
    # round trip: rows cut at max_len decode to a part of their graph, the others to all of it
    st = orc.sent_roundtrip(coo, ids.cpu().numpy(), ln.cpu().numpy(), 256, 600, 3, 1, nthreads=THREADS)
    assert not st.any(), (int((st != 0).sum()), st[st != 0][:10])
    vocab = {t: i for i, t in enumerate(["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no", "has_cycle"]
                                        + [str(i) for i in range(256)])}
    lut = gtok.ops.synth_lut(vocab, 256)
    q = np.zeros((4096, 4), np.int32); q[:, 0] = 1; q[:, 1] = vocab["has_cycle"]
    ids2, ln2 = gtok.ops.ibtt_synth(dev, lut, torch.from_numpy(q), 600, 0, ld=600)
    ref2, rln2 = orc.ibtt_synth(coo, lut.numpy(), q, 600, 0, 600, nthreads=THREADS)
    assert np.array_equal(ln2.cpu().numpy(), rln2) and np.array_equal(ids2.cpu().numpy(), ref2)
    want = np.minimum(600, 1 + 3 * d["edge_counts"] + 1 + d["node_counts"] + 1 + 1 + 1)
    assert np.array_equal(ln2.cpu().numpy(), want)


def test_packed_rows_and_strings_at_zinc_full_size(zinc_full):
    """Size-independent properties of round 3's formats on the full corpus: pack -> unpack is the identity on everything
    inside the lengths (both widths, padded and GTOK_SENT_NO_PAD slabs), the packed form is the token count rounded per row,
    the host view of an epoch serves exactly the slab's rows, and the device-rendered strings of the whole split equal the
    per-item Python on a sample and re-tokenize (text -> ids) to the ids the string-free route gives."""
    d, host, dev, coo = zinc_full
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ld = 192
    ids, ln = gtok.ops.sent(dev, 37, 1024, 7, 3, ld=ld, **kw)
    raw, rln = gtok.ops.sent(dev, 37, 1024, 7, 3, ld=ld, pad=False, **kw)
    assert torch.equal(ln, rln)
    ptr = gtok.ops.row_offsets(ln, ld)
    assert int(ptr[-1]) == int(((ln.long() + 7) // 8 * 8).sum())
    for eb in (2, 4):
        a, _ = gtok.ops.pack_rows(ids, ln, ptr, elem_bytes=eb)
        b, _ = gtok.ops.pack_rows(raw, ln, ptr, elem_bytes=eb)           # the unpadded slab packs to the same rows
        assert torch.equal(gtok.ops.unpack_rows(a, ptr, ln, ld, 5), ids)
        assert torch.equal(gtok.ops.unpack_rows(b, ptr, ln, ld, 5), ids)
    rows = gtok.rows.EpochRows(raw, ln)
    ids_h, ln_h = ids.cpu(), ln.cpu().tolist()
    for i in list(range(0, ZINC_FULL, 997)) + [ZINC_FULL - 1]:
        assert torch.equal(rows.row(i), ids_h[i, :ln_h[i]].long())
    assert rows.take(5) is not None and rows.take(5) is None           # once per epoch
    # strings of the whole split
    gdl = gtok.graph_data_loader
    ds = gdl.ZINCTokenizationDataset(split="train", max_len=1024, zinc_dataset=gtok.synth.InMemoryLike(d))
    texts, labels = ds.render_all(DEV)
    assert len(texts) == ZINC_FULL
    for i in list(range(0, ZINC_FULL, 4999)) + [ZINC_FULL - 1]:
        want = ds._item(i)
        assert texts[i] == want["text"] and labels[i] == want["label"]
    vocab = zinc_vocab(40)
    S = 20000
    td = gdl.TokenDataset([{"text": t, "label": y} for t, y in zip(texts[:S], labels[:S])], vocab, 1024, device=DEV)
    direct, dln = ds.tokenize(vocab, 1024, device=DEV)
    dl = dln.cpu().tolist()
    dh = direct.cpu()
    for i in range(0, S, 97):
        assert torch.equal(td.seqs[i], dh[i, :dl[i]].long())
    # every text of the split through the text kernel with the vocab the trainer really builds (fixed table + node ids + one
    # val_x_xx token per distinct label: too large for the LDS copy, so the short-key table is filled in part and adopts the
    # rest on use) == the string-free route over the same molecules, row for row
    vocab_dyn = gdl.build_zinc_vocab_on_device(texts, device=DEV)
    assert len(vocab_dyn) > 700, len(vocab_dyn)
    td_all = gdl.TokenDataset([{"text": t, "label": y} for t, y in zip(texts, labels)], vocab_dyn, 1024, device=DEV)
    g_ids, g_len = ds.tokenize(vocab_dyn, 1024, device=DEV)
    assert torch.equal(td_all.lens, g_len)
    w = min(int(td_all.ids.shape[1]), int(g_ids.shape[1]))
    assert int(g_len.max()) <= w
    keep = torch.arange(w, device=g_len.device)[None, :] < g_len[:, None]
    assert torch.equal(td_all.ids[:, :w][keep], g_ids[:, :w][keep])
    assert torch.equal(td_all.seqs[ZINC_FULL - 1], g_ids[ZINC_FULL - 1, :int(g_len[-1])].long().cpu())


def test_sent_zinc_full_sixteen_epochs_per_launch_as_16_bit_rows(zinc_full):
    """What an epoch of the dataset classes runs at ZINC-full size (tokenizer.epochs_for(249,456, ld) = 16 epochs per launch, 16-bit
    rows, no padding; ~4 M walks in sixteen rounds of resident waves, the (unit, epoch) pairs epoch-major): every epoch slice
    bit-exact against the oracle inside the row lengths, and every un-remapped row of the first, a middle and the last slice
    decodes back to its molecule."""
    d, host, dev, coo = zinc_full
    K, ld = 16, 192
    tok = gtok.Graph2TrailTokenizer(dataset_names=[], max_length=1024, labeled_graph=True)
    assert tok.epochs_for(ZINC_FULL, ld) == K == tok.epochs_for(ZINC_FULL) and tok.epochs_for(125000, 1040) == 16 and tok.epochs_for(12000, 208) == 32
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ids, ln = gtok.ops.sent(dev, 37, 1024, 7, 100, ld=ld, epochs=K, u16=True, pad=False, **kw)
    assert ids.dtype == torch.int16 and tuple(ids.shape) == (K, ZINC_FULL, ld) and int(ln.max()) <= ld
    inside = None
    for e in range(K):
        ref, rln = orc.sent(coo, 37, 1024, 7, 100 + e, ld=ld, nthreads=THREADS, **kw)
        assert np.array_equal(ln[e].cpu().numpy(), rln), e
        got = ids[e].cpu().numpy().view(np.uint16)
        inside = np.arange(ld)[None, :] < rln[:, None]
        assert np.array_equal(np.where(inside, got, 0), np.where(inside, ref, 0)), e
    del ids, ln
    # the round trip on the un-remapped flavour (the decoder reads raw SENT ids), all sixteen epochs in one launch
    raw, rl = gtok.ops.sent(dev, 37, 1024, 11, 40, labeled=True, num_node_types=28, num_edge_types=6, ld=ld, epochs=K, u16=True)
    for e in (0, 7, K - 1):
        rows = raw[e].cpu().numpy().view(np.uint16).astype(np.int32)
        st = orc.sent_roundtrip(coo, rows, rl[e].cpu().numpy(), 37, 1024, 11, 40 + e, labeled=True, num_node_types=28, nthreads=THREADS)
        assert not st.any(), (e, int((st != 0).sum()))
    assert bool((raw.view(K * ZINC_FULL, ld)[torch.arange(K * ZINC_FULL, device=DEV), (rl.view(-1) - 1).long()] == 4).all())


def test_large_graphs_eight_epochs_per_launch():
    """BASELINE config 5 shape through the bit-matrix lane kernel with 8 epochs per launch (what the dataset classes run for a
    125 k-graph shard; here 24 k graphs: pairs beyond the first round come from the ticket counters): 16-bit rows, every slice
    == the oracle, rows cut at max_len decode to a part of their graph."""
    G, K = 24000, 8
    d = gtok.synth.er_batch_device(G, torch.device(DEV), seed=77)
    batch, coo = both(d, False)
    dev = batch.to(DEV)
    assert gtok.ops.sent_kernel_name(dev, 256, 600, epochs=K) in ("sent_blane_kernel<W=4>", "sent_lds_kernel<W=4>")
    ids, ln = gtok.ops.sent(dev, 256, 600, 3, 5, ld=608, epochs=K, u16=True)
    assert gtok.ops.sent_kernel_name(dev, 256, 600, epochs=K) == "sent_blane_kernel<W=4>"      # (the mirror exists now)
    for e in (0, 3, K - 1):
        ref, rln = orc.sent(coo, 256, 600, 3, 5 + e, ld=608, nthreads=THREADS)
        assert np.array_equal(ln[e].cpu().numpy(), rln) and np.array_equal(ids[e].cpu().numpy().view(np.uint16), ref), e
        st = orc.sent_roundtrip(coo, ref, rln, 256, 600, 3, 5 + e, nthreads=THREADS)
        assert not st.any()


def test_config5_share_at_the_epochs_per_launch_the_dataset_classes_use():
    """A 125,000-graph share of BASELINE config 5 (10..256 nodes, max_len 600) the way the dataset classes tokenize it:
    epochs_for(125000, 608) = 28 epochs in one gtok_sent launch (the 4 GiB slab bound), 16-bit rows without padding, 3.5 M walks
    through sent_blane_kernel with every pair beyond the first round drawn from the ticket counters - first, a middle and the last
    slice bit-exact against the oracle inside the row lengths, lengths of every slice equal, rows cut at max_len decode to a part
    of their graph."""
    G, ld = 125000, 608
    tok = gtok.Graph2TrailTokenizer(dataset_names=[], max_length=600, labeled_graph=False)
    K = tok.epochs_for(G, ld)
    assert K == 28
    d = gtok.synth.er_batch_device(G, torch.device(DEV), seed=1000)
    batch, coo = both(d, False)
    dev = batch.to(DEV)
    ids, ln = gtok.ops.sent(dev, 256, 600, 5, 20, ld=ld, epochs=K, u16=True, pad=False)
    assert gtok.ops.sent_kernel_name(dev, 256, 600, epochs=K) == "sent_blane_kernel<W=4>" and tuple(ids.shape) == (K, G, ld)
    for e in (0, 13, K - 1):
        ref, rln = orc.sent(coo, 256, 600, 5, 20 + e, ld=ld, nthreads=THREADS)
        assert np.array_equal(ln[e].cpu().numpy(), rln), e
        got = ids[e].cpu().numpy().view(np.uint16)
        inside = np.arange(ld)[None, :] < rln[:, None]
        assert np.array_equal(np.where(inside, got, 0), np.where(inside, ref, 0)), e
        if e == K - 1:
            st = orc.sent_roundtrip(coo, ref, rln, 256, 600, 5, 20 + e, nthreads=THREADS)
            assert not st.any()
    assert int(ln.min()) >= 2 and int(ln.max()) <= 600
