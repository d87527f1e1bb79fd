"""GPU parity: every HIP entry point, called through the C ABI, against the CPU oracle — bit-exact
(integer work).  Sizes are what the oracle finishes in seconds; corpus-scale properties live in
test_gpu_properties.py."""
import os
import numpy as np
import pytest
import torch

from _util import both, edge_case_graphs, gtok, orc, zinc_vocab

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cmp(ids, ln, ref_ids, ref_ln, what):
    ids, ln = ids.cpu().numpy(), ln.cpu().numpy()
    assert np.array_equal(ln, ref_ln), f"{what}: lengths differ at {np.nonzero(ln != ref_ln)[0][:5]}"
    bad = np.nonzero((ids != ref_ids).any(1))[0]
    if bad.size:
        r = bad[0]
        c = int(np.nonzero(ids[r] != ref_ids[r])[0][0])
        lo = max(0, c - 12)
        raise AssertionError(f"{what}: {bad.size} rows differ, first row {r} col {c}: "
                             f"{ids[r, lo:c + 8].tolist()} vs {ref_ids[r, lo:c + 8].tolist()}")


@pytest.mark.parametrize("coalesced", [True, False])
@pytest.mark.parametrize("max_len,ld", [(1024, None), (64, 64), (10, 12), (1, 4), (0, 4), (1024, 40)])
def test_ibtt_zinc(coalesced, max_len, ld):
    d = gtok.synth.zinc_like(3000, seed=3, coalesced=coalesced)
    batch, coo = both(d)
    vocab = zinc_vocab(40)
    lut = gtok.ops.zinc_lut(vocab, 40)
    ids, ln = gtok.ops.ibtt_zinc(batch.to(DEV), lut, max_len, vocab["<pad>"], ld=ld)
    ref, rln = orc.ibtt_zinc(coo, lut.numpy(), max_len, vocab["<pad>"], ids.shape[1])
    _cmp(ids, ln, ref, rln, "ibtt_zinc")


def test_ibtt_zinc_edge_cases_and_missing_vocab():
    d = edge_case_graphs()
    batch, coo = both(d)
    vocab = zinc_vocab(4, with_fallbacks=False)      # most node ids, 'X', 'unknown' fall back to <pad>
    lut = gtok.ops.zinc_lut(vocab, 4)
    for max_len in (1024, 7, 3):
        ids, ln = gtok.ops.ibtt_zinc(batch.to(DEV), lut, max_len, vocab["<pad>"])
        ref, rln = orc.ibtt_zinc(coo, lut.numpy(), max_len, vocab["<pad>"], ids.shape[1])
        _cmp(ids, ln, ref, rln, f"ibtt_zinc edge max_len={max_len}")


SENT_CASES = [
    dict(name="zinc_labeled", gen=lambda: gtok.synth.zinc_like(4000, seed=5), labeled=True, nn=37),
    dict(name="zinc_uncoalesced", gen=lambda: gtok.synth.zinc_like(1000, seed=6, coalesced=False), labeled=True, nn=37),
    dict(name="zinc_unlabeled", gen=lambda: gtok.synth.zinc_like(2000, seed=7), labeled=False, nn=40),
    dict(name="graph_token", gen=lambda: gtok.synth.graph_token_like(600, seed=8, with_text=False), labeled=False, nn=49),
    dict(name="er_128", gen=lambda: gtok.synth.er_batch(150, seed=9, min_nodes=65, max_nodes=128), labeled=False, nn=128),
    dict(name="er_256", gen=lambda: gtok.synth.er_batch(100, seed=10, min_nodes=10, max_nodes=256), labeled=False, nn=256),
    dict(name="er_400", gen=lambda: gtok.synth.er_batch(24, seed=11, min_nodes=300, max_nodes=400,
                                                        min_sparsity=0.02, max_sparsity=0.05), labeled=False, nn=400),
    dict(name="edge_cases", gen=edge_case_graphs, labeled=True, nn=8),
    dict(name="edge_cases_unlabeled", gen=edge_case_graphs, labeled=False, nn=8),
]


@pytest.mark.parametrize("case", SENT_CASES, ids=[c["name"] for c in SENT_CASES])
@pytest.mark.parametrize("max_len", [1024, 48])
def test_sent(case, max_len):
    d = case["gen"]()
    batch, coo = both(d, case["labeled"])
    kw = dict(labeled=case["labeled"], num_node_types=28 if case["labeled"] else 0,
              num_edge_types=5 if case["labeled"] else 0)
    dbatch = batch.to(DEV)
    for seed, epoch, base in ((0, 0, 0), (12345678901234567, 3, 10 ** 10)):
        ids, ln = gtok.ops.sent(dbatch, case["nn"], max_len, seed, epoch, graph_base=base, **kw)
        ref, rln = orc.sent(coo, case["nn"], max_len, seed, epoch, graph_base=base, ld=ids.shape[1], **kw)
        _cmp(ids, ln, ref, rln, f"sent {case['name']} seed={seed}")


def test_sent_remap_query_and_narrow_slab():
    d = gtok.synth.zinc_like(1500, seed=21)
    batch, coo = both(d)
    dbatch = batch.to(DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4)
    ids, ln = gtok.ops.sent(dbatch, 37, 1024, 99, 1, remap_zinc=True, **kw)
    ref, rln = orc.sent(coo, 37, 1024, 99, 1, remap_zinc=True, ld=ids.shape[1], **kw)
    _cmp(ids, ln, ref, rln, "sent+remap")
    # stand-alone remap kernel == fused remap
    raw, rl = gtok.ops.sent(dbatch, 37, 1024, 99, 1, **kw)
    re = gtok.ops.remap_zinc(raw, rl, 6, 6 + 37, 6 + 37 + 9)
    mask = (torch.arange(raw.shape[1], device=DEV)[None, :] < rl[:, None])
    assert torch.equal(torch.where(mask, re, ids), ids)
    # query append on unlabelled graphs (shortest_path), incl. a slab narrower than some rows
    g = gtok.synth.graph_token_like(500, seed=22, task="shortest_path", with_text=False)
    b2, c2 = both(g, False)
    q = np.array([qq if qq is not None else (0, 0) for qq in g["queries"]], np.int32)
    for ld in (None, 40):
        ids, ln = gtok.ops.sent(b2.to(DEV), 49, 600, 5, 2, query=torch.from_numpy(q), ld=ld)
        ref, rln = orc.sent(c2, 49, 600, 5, 2, query=q, ld=ids.shape[1])
        _cmp(ids, ln, ref, rln, f"sent+query ld={ld}")
        if ld is not None:
            assert (ln.cpu().numpy() > ld).any(), "test should exercise the too-narrow-slab signal"


def test_sent_is_shard_invariant():
    d = gtok.synth.zinc_like(1001, seed=30)
    batch, _ = both(d)
    full, fl = gtok.ops.sent(batch.to(DEV), 37, 256, 4, 0, labeled=True, num_node_types=28, num_edge_types=5, ld=256)
    parts = []
    for lo, hi in ((0, 334), (334, 668), (668, 1001)):
        ids, ln = gtok.ops.sent(batch.shard(lo, hi).to(DEV), 37, 256, 4, 0, labeled=True, num_node_types=28,
                                num_edge_types=5, graph_base=lo, ld=256)
        parts.append(ids)
    assert torch.equal(torch.cat(parts), full)


def test_ibtt_synth_and_text():
    g = gtok.synth.graph_token_like(800, seed=40, task="shortest_path")
    batch, coo = both(g, False)
    from collections import Counter
    cnt = Counter(t for s in g["texts"] for t in s.split())
    vocab = {t: i for i, t in enumerate(["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no"])}
    for t, _ in cnt.most_common():
        if t not in vocab and len(vocab) < 45:       # small vocab: some node ids fall back to <pad>
            vocab[t] = len(vocab)
    pad = vocab["<pad>"]
    lut = gtok.ops.synth_lut(vocab, 64)
    q = np.zeros((batch.num_graphs, 4), np.int32)
    for i, qq in enumerate(g["queries"]):
        q[i] = (3, vocab.get("shortest_distance", pad), vocab.get(str(qq[0]), pad), vocab.get(str(qq[1]), pad))
    for max_len in (600, 100):
        ids, ln = gtok.ops.ibtt_synth(batch.to(DEV), lut, torch.from_numpy(q), max_len, pad)
        ref, rln = orc.ibtt_synth(coo, lut.numpy(), q, max_len, pad, ids.shape[1])
        _cmp(ids, ln, ref, rln, "ibtt_synth")
        # the same ids must come out of the generic text path (TokenDataset semantics)
        tb, tp = gtok.ops.pack_texts(g["texts"])
        table = gtok.ops.VocabTable(vocab, DEV)
        tids, tln = gtok.ops.text_to_ids(tb.to(DEV), tp, table, max_len, ld=ids.shape[1])
        tref, trln = orc.text_to_ids(g["texts"], vocab, max_len, ids.shape[1])
        _cmp(tids, tln, tref, trln, "text_to_ids")
        _cmp(tids, tln, ref, rln, "text_to_ids vs ibtt_synth")


def test_text_to_ids_odd_whitespace_and_no_label():
    texts = ["", "   ", "<bos> 1 2\t<e>\n\n3  4 <e> <n> 1 2 3 4 <q> has_cycle <p> yes <eos>", "<p>", "a<p> <p>x <p> tail",
             "x" * 300 + " y " + "z" * 70, "\x1c<bos>\x1f1\x0b2\x0c<e>\r", " ".join(str(i) for i in range(700))]
    vocab = {"<pad>": 0, "<bos>": 1, "<e>": 2, "<n>": 3, "<q>": 4, "<p>": 5, "<eos>": 6, "yes": 7, "no": 8,
             "1": 9, "2": 10, "3": 11, "has_cycle": 12, "x" * 300: 13, "z" * 70: 14, "699": 15, "a<p>": 16}
    table = gtok.ops.VocabTable(vocab, DEV)
    tb, tp = gtok.ops.pack_texts(texts)
    for strip in (True, False):
        for max_len in (600, 5):
            ids, ln = gtok.ops.text_to_ids(tb.to(DEV), tp, table, max_len, strip_label=strip)
            ref, rln = orc.text_to_ids(texts, vocab, max_len, ids.shape[1], strip_label=strip)
            _cmp(ids, ln, ref, rln, f"text strip={strip} max_len={max_len}")


def test_collate():
    d = gtok.synth.zinc_like(300, seed=50)
    batch, _ = both(d)
    vocab = zinc_vocab(40)
    ids, ln = gtok.ops.ibtt_zinc(batch.to(DEV), gtok.ops.zinc_lut(vocab, 40), 1024, 2)
    rng = np.random.default_rng(0)
    index = rng.permutation(300)[:128]
    lmax = int(ln.cpu().numpy()[index].max())
    for out_ld in (lmax, lmax + 5):
        X, A = gtok.ops.collate(ids, ln, torch.from_numpy(index), 2, out_ld)
        rX, rA, m = orc.collate(ids.cpu().numpy(), ln.cpu().numpy(), index, 2, out_ld)
        assert m == lmax and X.dtype == torch.int64 and A.dtype == torch.bool
        assert np.array_equal(X.cpu().numpy(), rX) and np.array_equal(A.cpu().numpy(), rA)


def _pin_sent(monkeypatch, pin):
    """GTOK_SENT_KERNEL pin; "lane-int32" = the lane kernel staging from the int32 CSR (no byte-packed mirror),
    "lane-unsorted" = the lane kernel on the batch as stored (64 neighbouring graphs per wave) instead of its
    copy reordered by walk length (ops.lane_sorted, the default)."""
    monkeypatch.setenv("GTOK_SENT_KERNEL", pin.split("-")[0])
    if pin == "lane-int32":
        monkeypatch.setenv("GTOK_NO_PACK8", "1")
    else:
        monkeypatch.delenv("GTOK_NO_PACK8", raising=False)
    if pin == "lane-unsorted":
        monkeypatch.setenv("GTOK_NO_LANE_SORT", "1")
    else:
        monkeypatch.delenv("GTOK_NO_LANE_SORT", raising=False)


@pytest.mark.parametrize("pin", ["lane", "lane-int32", "lane-unsorted", "reg", "lds"])
def test_sent_every_kernel_same_tokens(pin, monkeypatch):
    """Three kernels implement the one SENT spec (lane-per-graph - staged from the byte-packed mirror or from the
    int32 CSR -, register-resident wave-per-graph, LDS bit matrix); GTOK_SENT_KERNEL pins one per call.  Each must
    reproduce the oracle on graphs they all accept."""
    _pin_sent(monkeypatch, pin)
    cases = [(gtok.synth.zinc_like(3000, seed=61), True, 37, dict(remap_zinc=True, num_node_types=9, num_edge_types=4)),
             (gtok.synth.zinc_like(800, seed=62, coalesced=False), True, 37, dict(num_node_types=28, num_edge_types=5)),
             (gtok.synth.zinc_like(1500, seed=63), False, 40, {}),
             (gtok.synth.graph_token_like(400, seed=64, with_text=False, algorithms=("er", "ba", "sbm", "path", "star")), False, 49, {}),
             (edge_case_graphs(), True, 8, dict(num_node_types=28, num_edge_types=6)),
             (edge_case_graphs(), False, 8, {})]
    for d, labeled, nn, kw in cases:
        batch, coo = both(d, labeled)
        for max_len in (1024, 40):
            ids, ln = gtok.ops.sent(batch.to(DEV), nn, max_len, 17, 5, labeled=labeled, graph_base=123, **kw)
            ref, rln = orc.sent(coo, nn, max_len, 17, 5, labeled=labeled, graph_base=123, ld=ids.shape[1], **kw)
            _cmp(ids, ln, ref, rln, f"sent[{pin}] labeled={labeled} max_len={max_len}")
    # with a query tail
    g = gtok.synth.graph_token_like(300, seed=65, task="shortest_path", with_text=False, algorithms=("er", "ba", "path"))
    b2, c2 = both(g, False)
    q = np.array([qq if qq is not None else (0, 0) for qq in g["queries"]], np.int32)
    ids, ln = gtok.ops.sent(b2.to(DEV), 49, 600, 5, 2, query=torch.from_numpy(q))
    ref, rln = orc.sent(c2, 49, 600, 5, 2, query=q, ld=ids.shape[1])
    _cmp(ids, ln, ref, rln, f"sent[{pin}]+query")


def test_torch_custom_ops():
    """torch.ops.gtok.* are the same kernels; CUDA only (a CPU tensor must be refused, not tokenized on the host)."""
    d = gtok.synth.zinc_like(500, seed=70)
    batch, coo = both(d)
    b = batch.to(DEV)
    ids, ln = torch.ops.gtok.sent(b.node_ptr, b.edge_ptr, b.rowptr, b.col, b.nattr, b.eattr, None, b.max_nodes,
                                  b.max_edges, 37, 1024, 192, 3, 1, True, 9, 4, True, 5, 0)
    ref, rln = orc.sent(coo, 37, 1024, 3, 1, labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True, ld=192)
    _cmp(ids, ln, ref, rln, "torch.ops.gtok.sent")
    vocab = zinc_vocab(40)
    lut = gtok.ops.zinc_lut(vocab, 40).to(DEV)
    ids2, ln2 = torch.ops.gtok.ibtt_zinc(b.node_ptr, b.edge_ptr, b.rowptr, b.col, b.eorder, b.nattr, b.eattr, lut,
                                         b.max_nodes, b.max_edges, 1024, 2, 240)
    ref2, rln2 = orc.ibtt_zinc(coo, lut.cpu().numpy(), 1024, 2, 240)
    _cmp(ids2, ln2, ref2, rln2, "torch.ops.gtok.ibtt_zinc")
    # the other five entry points, against the GraphBatch-level wrappers (themselves checked against the oracle elsewhere)
    texts = ["<bos> 0 1 <e> 1 2 <e> <n> 0 1 2 <q> has_cycle <p> no <eos>", "<bos> 3 4 <e> <n> 3 4 <q> shortest_distance 3 4 <p> len1 <eos>"]
    tb, tp = gtok.ops.pack_texts(texts)
    tb, tp = tb.to(DEV), tp.to(DEV)
    sv = {t: i for i, t in enumerate(["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no", "0", "1", "2", "3", "4", "has_cycle"])}
    tab = gtok.ops.VocabTable(sv, DEV)
    a1, l1 = torch.ops.gtok.text_to_ids(tb, tp, tab.key_off, tab.key_len, tab.ids, tab.key_bytes, tab.pad_id, 64, True, 64)
    a0, l0 = gtok.ops.text_to_ids(tb, tp, tab, 64, True, ld=64)
    assert torch.equal(a1, a0) and torch.equal(l1, l0)
    r = gtok.ops.parse_graph_texts(tb, tp)
    t8 = torch.ops.gtok.parse_graph_text(tb, tp)
    for got, key in zip(t8, ("num_edges", "num_nodes", "query", "label", "status", "edge_ptr", "src", "dst")):
        assert torch.equal(got, r[key]), key
    xb = torch.tensor([[1, 9, 4, 4], [7, 7, 7, 7]], dtype=torch.int64, device=DEV)
    assert torch.ops.gtok.find_token(xb, 4).tolist() == [2, -1]
    raw, rl = gtok.ops.sent(b, 37, 1024, 3, 1, labeled=True, num_node_types=9, num_edge_types=4)
    dd = gtok.ops.sent_decode(raw, rl, 37, True, 9, 64, 37)
    td = torch.ops.gtok.sent_decode(raw, rl, 37, True, 9, 64, 37)
    td = dict(zip(("num_nodes", "num_edges", "status", "edge_a", "edge_b", "edge_type", "node_type"), td))
    for key in ("num_nodes", "num_edges", "status"):
        assert torch.equal(td[key], dd[key]), key
    em = torch.arange(64, device=DEV)[None, :] < dd["num_edges"][:, None]       # slots past the counts are not written
    nm = torch.arange(37, device=DEV)[None, :] < dd["num_nodes"][:, None]
    for key in ("edge_a", "edge_b", "edge_type"):
        assert torch.equal(td[key][em], dd[key][em]), key
    assert torch.equal(td["node_type"][nm], dd["node_type"][nm])
    s = gtok.synth.graph_token_like(200, seed=74, with_text=False)
    sb, _ = both(s, False)
    sb = sb.to(DEV)
    c0, f0 = gtok.ops.vocab_stats_synth(sb, 64)
    c1, f1 = torch.ops.gtok.vocab_stats_synth(sb.node_ptr, sb.edge_ptr, sb.rowptr, sb.col, sb.eorder, None, sb.max_nodes, sb.max_edges, 64, 0)
    assert torch.equal(c0, c1) and torch.equal(f0, f1)
    X, A = torch.ops.gtok.collate(ids2, ln2, torch.arange(8, device=DEV), 2, int(ln2[:8].max()))
    assert X.dtype == torch.int64 and A.dtype == torch.bool and X.shape == A.shape
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.gtok.remap_zinc(ids.cpu(), ln.cpu(), 6, 43, 52)


def test_c_abi_error_codes():
    """Bad arguments come back as negative GTOK_E_* codes (nothing throws across the ABI, nothing is launched)."""
    import ctypes
    L = gtok.lib()
    lib_mod = gtok._lib
    d = gtok.synth.zinc_like(8, seed=80)
    b, _ = both(d)
    b = b.to(DEV)
    cs = b.c_struct()
    ids = torch.empty((8, 64), dtype=torch.int32, device=DEV); ln = torch.empty(8, dtype=torch.int32, device=DEV)
    p = lib_mod.GtokSentParams(37, 0, 0, 0, 64, 0, 5, 0, 1, 0, 0, None)
    assert L.gtok_sent(ctypes.byref(cs), ctypes.byref(p), ids.data_ptr(), 64, ln.data_ptr(), None) == 0
    assert L.gtok_sent(ctypes.byref(cs), ctypes.byref(p), None, 64, ln.data_ptr(), None) == -1          # GTOK_E_INVAL
    assert L.gtok_sent(ctypes.byref(cs), ctypes.byref(p), ids.data_ptr(), 0, ln.data_ptr(), None) == -1
    lab = lib_mod.GtokSentParams(37, 1, 9, 4, 64, 0, 5, 0, 1, 0, 0, None)
    nolab = gtok.GraphBatch(b.num_graphs, b.max_nodes, b.max_edges, b.node_ptr, b.edge_ptr, b.rowptr, b.col, None, None, None)
    cs2 = nolab.c_struct()
    assert L.gtok_sent(ctypes.byref(cs2), ctypes.byref(lab), ids.data_ptr(), 64, ln.data_ptr(), None) == -1   # labelled, no attrs
    big = gtok.GraphBatch(b.num_graphs, 600, b.max_edges, b.node_ptr, b.edge_ptr, b.rowptr, b.col, None, b.nattr, b.eattr)
    cs3 = big.c_struct()
    assert L.gtok_sent(ctypes.byref(cs3), ctypes.byref(p), ids.data_ptr(), 64, ln.data_ptr(), None) == -2    # GTOK_E_TOO_LARGE
    assert L.gtok_ibtt_zinc(ctypes.byref(cs), None, 64, 64, 2, ids.data_ptr(), 64, ln.data_ptr(), None) == -1
    vt = lib_mod.GtokVocabTable(12, 0, None, None, None, None)                                               # not a power of two
    assert L.gtok_text_to_ids(None, ln.data_ptr(), 0, ctypes.byref(vt), 1, 8, ids.data_ptr(), 64, ln.data_ptr(), None) == -1
    # a batch reordered for the lane-per-graph SENT kernel is for that kernel alone
    lab9 = lib_mod.GtokSentParams(37, 1, 9, 4, 64, 0, 5, 0, 1, 0, 0, None)
    gtok.ops.sent(b, 37, 64, 1, labeled=True, num_node_types=9, num_edge_types=4)                            # (makes nothing: too small a batch ...)
    sb = gtok.ops.lane_sorted(b)                                                                             # ... so ask for the copy
    assert sb is not None and sb.graph_ids is not None
    css = sb.c_struct()
    assert L.gtok_sent(ctypes.byref(css), ctypes.byref(lab9), ids.data_ptr(), 64, ln.data_ptr(), None) == 0
    lut = gtok.ops.zinc_lut(zinc_vocab(40), 40).to(DEV)
    assert L.gtok_ibtt_zinc(ctypes.byref(css), lut.data_ptr(), lut.numel(), 64, 2, ids.data_ptr(), 64, ln.data_ptr(), None) == -1
    info = torch.zeros(1, dtype=torch.int32, device=DEV)
    rows = torch.empty((sb.num_nodes_total, 1), dtype=torch.int64, device=DEV); planes = torch.empty((8, 8, 1), dtype=torch.int64, device=DEV)
    assert L.gtok_csr_adjbits(ctypes.byref(css), 1, rows.data_ptr(), planes.data_ptr(), info.data_ptr(), None) == -1
    assert L.gtok_csr_adjbits(ctypes.byref(cs), 3, rows.data_ptr(), planes.data_ptr(), info.data_ptr(), None) == -1      # words: 1, 2 or 4
    assert L.gtok_csr_adjbits(ctypes.byref(cs), 1, rows.data_ptr(), planes.data_ptr(), info.data_ptr(), None) == 0
    half = gtok.GraphBatch(sb.num_graphs, sb.max_nodes, sb.max_edges, sb.node_ptr, sb.edge_ptr, sb.rowptr, sb.col, None, sb.nattr, sb.eattr,
                           sb.flags, sb.chunk_nodes, sb.chunk_edges, sb.max_degree)
    half.graph_ids = sb.graph_ids                                                                            # ids without a unit table
    csh = half.c_struct()
    assert L.gtok_sent(ctypes.byref(csh), ctypes.byref(lab9), ids.data_ptr(), 64, ln.data_ptr(), None) == -1
    # ABI v4 fields: a negative epoch_count, a nonzero `reserved`, an unknown flag bit, a pad id that does not fit 16-bit rows
    P = lib_mod.GtokSentParams
    for bad in (P(37, 0, 0, 0, 64, 0, 5, 0, 1, 0, 0, None, -1, 0), P(37, 0, 0, 0, 64, 0, 5, 0, 1, 0, 0, None, 1, 7),
                P(37, 0, 0, 0, 64, 0, 5, 4, 1, 0, 0, None, 1, 0), P(37, 0, 0, 0, 64, 0, 70000, lib_mod.SENT_U16, 1, 0, 0, None, 1, 0)):
        assert L.gtok_sent(ctypes.byref(cs), ctypes.byref(bad), ids.data_ptr(), 64, ln.data_ptr(), None) == -1
    ok2 = P(37, 0, 0, 0, 64, 0, 5, lib_mod.SENT_U16 | lib_mod.SENT_NO_PAD, 1, 0, 0, None, 2, 0)                  # 2 epochs of 16-bit rows
    ids16 = torch.empty((16, 64), dtype=torch.int16, device=DEV); ln2 = torch.empty(16, dtype=torch.int32, device=DEV)
    assert L.gtok_sent(ctypes.byref(cs), ctypes.byref(ok2), ids16.data_ptr(), 64, ln2.data_ptr(), None) == 0
    # a per-unit record table without the unit table it describes
    orphan = gtok.GraphBatch(b.num_graphs, b.max_nodes, b.max_edges, b.node_ptr, b.edge_ptr, b.rowptr, b.col, None, b.nattr, b.eattr)
    orphan.unit_info = sb.unit_info
    cso = orphan.c_struct()
    assert L.gtok_sent(ctypes.byref(cso), ctypes.byref(p), ids.data_ptr(), 64, ln.data_ptr(), None) == -1
    # the strided readers refuse segments (a slab has none); the checked unpack takes a NULL status
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    assert L.gtok_unpack_rows_checked(ids16.data_ptr(), 2, None, ln2.data_ptr(), 16, 4, 64, 0, 5, ids.data_ptr(), 64, st.data_ptr(), None) == -1
    wide = torch.empty((16, 64), dtype=torch.int32, device=DEV)
    assert L.gtok_unpack_rows_checked(ids16.data_ptr(), 2, None, ln2.data_ptr(), 16, 0, 0, 16 * 64, 5, wide.data_ptr(), 64, None, None) == 0
    assert L.gtok_pack_rows_u16(ids16.data_ptr(), 64, ln2.data_ptr(), 16, None, 2, wide.data_ptr(), 16 * 64, st.data_ptr(), None) == -1   # no row_ptr
    torch.cuda.synchronize()
    with pytest.raises(gtok.GtokError):
        gtok.ops.sent(big, 37, 64, 0)
    # an empty batch is a no-op, not an error
    empty = gtok.GraphBatch.from_coo([], [], [], []).to(DEV)
    e_ids, e_ln = gtok.ops.sent(empty, 37, 64, 0, ld=8)
    assert e_ids.shape == (0, 8) and e_ln.numel() == 0


@pytest.mark.parametrize("pin", ["lane", "quad", "wave"])
@pytest.mark.parametrize("max_len,ld", [(1024, None), (64, 64), (10, 12), (1, 4), (0, 4), (1024, 40), (57, 60), (1024, 1500), (64, 67), (200, 241)])
def test_ibtt_zinc_both_kernels(pin, max_len, ld, monkeypatch):
    """The lane-per-graph and 16-lanes-per-graph IBTT kernels (simple symmetric batches in list order) and the
    wave-per-graph one give the oracle's ids, including every truncation corner."""
    monkeypatch.setenv("GTOK_IBTT_KERNEL", pin)
    d = gtok.synth.zinc_like(3000, seed=91)
    batch, coo = both(d)
    assert batch.flags & 1 and batch.eorder is None
    vocab = zinc_vocab(30, with_fallbacks=(max_len != 57))     # some node ids (and X/unknown) fall back to <pad>
    lut = gtok.ops.zinc_lut(vocab, 40)
    ids, ln = gtok.ops.ibtt_zinc(batch.to(DEV), lut, max_len, vocab["<pad>"], ld=ld)
    ref, rln = orc.ibtt_zinc(coo, lut.numpy(), max_len, vocab["<pad>"], ids.shape[1])
    _cmp(ids, ln, ref, rln, f"ibtt_zinc[{pin}]")


@pytest.mark.parametrize("pin", ["lane", "lds"])
def test_ticket_queues_survive_many_launches(pin, monkeypatch):
    """The dynamically scheduled kernels take a slot of a 256-entry ring of device counters per launch and re-arm it
    when their last wave retires: 600 launches on two streams must keep giving the oracle's tokens."""
    monkeypatch.setenv("GTOK_SENT_KERNEL", pin)
    d = gtok.synth.zinc_like(700, seed=71)
    batch, coo = both(d, False)
    b = batch.to(DEV)
    ld = gtok.ops.sent_safe_ld(batch, False, 1024)
    ref = {k: orc.sent(coo, 40, 1024, 3, k, ld=ld) for k in (0, 1)}
    side = torch.cuda.Stream(device=DEV)
    outs = []
    for i in range(300):
        outs.append((0, gtok.ops.sent(b, 40, 1024, 3, 0, ld=ld)))
        with torch.cuda.stream(side):
            outs.append((1, gtok.ops.sent(b, 40, 1024, 3, 1, ld=ld)))
        if len(outs) >= 40:
            torch.cuda.synchronize()
            for k, (ids, ln) in outs:
                _cmp(ids, ln, ref[k][0], ref[k][1], f"launch {i} epoch {k} [{pin}]")
            outs = []
    torch.cuda.synchronize()


@pytest.mark.parametrize("pin", ["reg", "lane"])
def test_hip_graph_capture_and_replay(pin, monkeypatch):
    """The launchers only enqueue on the caller's stream, so a tokenisation pass (SENT + IBTT into preallocated
    slabs) can be captured once in a HIP graph and replayed; replays must keep matching the oracle (the lane
    kernel's ticket-counter slot is baked into the captured launch and re-arms itself after every replay)."""
    monkeypatch.setenv("GTOK_SENT_KERNEL", pin)
    d = gtok.synth.zinc_like(2000, seed=81)
    batch, coo = both(d)
    b = batch.to(DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ld = gtok.ops.sent_safe_ld(batch, True, 1024)
    vocab = zinc_vocab(40)
    lut = gtok.ops.zinc_lut(vocab, 40).to(DEV)
    s_out = (torch.empty((2000, ld), dtype=torch.int32, device=DEV), torch.empty(2000, dtype=torch.int32, device=DEV))
    i_out = (torch.empty((2000, 256), dtype=torch.int32, device=DEV), torch.empty(2000, dtype=torch.int32, device=DEV))
    gtok.ops.sent(b, 37, 1024, 9, 4, ld=ld, out=s_out, **kw)             # warm-up: first launch allocates the counters
    gtok.ops.ibtt_zinc(b, lut, 1024, vocab["<pad>"], ld=256, out=i_out)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gtok.ops.sent(b, 37, 1024, 9, 4, ld=ld, out=s_out, **kw)
        gtok.ops.ibtt_zinc(b, lut, 1024, vocab["<pad>"], ld=256, out=i_out)
    ref, rln = orc.sent(coo, 37, 1024, 9, 4, ld=ld, **kw)
    iref, irln = orc.ibtt_zinc(coo, lut.cpu().numpy(), 1024, vocab["<pad>"], 256)
    for rep in range(3):
        s_out[0].fill_(-1); i_out[0].fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        _cmp(s_out[0], s_out[1], ref, rln, f"graph replay {rep}: sent")
        _cmp(i_out[0], i_out[1], iref, irln, f"graph replay {rep}: ibtt")


def _symmetric_er(num_graphs, seed, min_nodes, max_nodes, p_lo, p_hi):
    """ER graphs listed in both directions, row-sorted (PyG-coalesced style), with atom / bond types."""
    d = gtok.synth.er_batch(num_graphs, seed=seed, min_nodes=min_nodes, max_nodes=max_nodes, min_sparsity=p_lo, max_sparsity=p_hi)
    rng = np.random.default_rng(seed + 1)
    nc, ec = d["node_counts"], d["edge_counts"]
    gid = np.repeat(np.arange(num_graphs), ec)
    bond = rng.integers(0, 6, gid.size)
    src = np.concatenate([d["src"], d["dst"]]); dst = np.concatenate([d["dst"], d["src"]])
    gg = np.concatenate([gid, gid]); ea = np.concatenate([bond, bond])
    order = np.lexsort((dst, src, gg))
    return dict(node_counts=nc, edge_counts=2 * ec, src=src[order], dst=dst[order], edge_attr=ea[order],
                x=rng.integers(0, 12, int(nc.sum())))


@pytest.mark.parametrize("pin", ["quad", "wave"])
def test_ibtt_zinc_large_symmetric_graphs(pin, monkeypatch):
    """The 16-lanes-per-molecule kernel on graphs far beyond molecule size: more than 64 nodes and more than 128
    entries per graph take its plain-loop tails (the first 64 / 128 travel through the register pipeline)."""
    monkeypatch.setenv("GTOK_IBTT_KERNEL", pin)
    d = _symmetric_er(1100, 5, 2, 150, 0.03, 0.12)
    batch, coo = both(d)
    assert batch.flags & 1 and batch.eorder is None and batch.max_nodes > 64 and batch.max_edges > 128
    vocab = zinc_vocab(160)
    lut = gtok.ops.zinc_lut(vocab, 160)
    for max_len, ld in ((8192, None), (300, 300), (1024, 5000)):
        ids, ln = gtok.ops.ibtt_zinc(batch.to(DEV), lut, max_len, vocab["<pad>"], ld=ld)
        ref, rln = orc.ibtt_zinc(coo, lut.numpy(), max_len, vocab["<pad>"], ids.shape[1])
        _cmp(ids, ln, ref, rln, f"ibtt_zinc large [{pin}] max_len={max_len}")


@pytest.mark.parametrize("task", ["cycle_check", "shortest_path"])
def test_vocab_stats_synth_and_vocab_from_graphs(task):
    """§8f-1: node-id token statistics on the device == the oracle's restatement (sorted and unsorted edge lists,
    query arguments, shards accumulated into one table), and the vocab built from them == build_vocab_from_texts."""
    import importlib
    dl = importlib.import_module("glearning-benchmark_amd.graph_data_loader.data_loader")
    d = gtok.synth.graph_token_like(900, seed=12, task=task, min_nodes=2, max_nodes=70)
    q = None
    if task == "shortest_path":
        q = np.array([qq if qq is not None else (-1, -1) for qq in d["queries"]], np.int32)
    tq = None if q is None else torch.from_numpy(q)
    for shuffle in (False, True):
        dd = dict(d)
        if shuffle:   # edge lists in arbitrary order inside each graph: the CSR keeps the list position in eorder
            rng = np.random.default_rng(3)
            gid = np.repeat(np.arange(900), d["edge_counts"])
            order = np.lexsort((rng.random(gid.size), gid))
            dd["src"], dd["dst"] = d["src"][order], d["dst"][order]
        batch, coo = both(dd, False)
        assert (batch.eorder is not None) == shuffle
        want = orc.vocab_stats_synth(coo, 80, q, graph_base=1000)
        count, first = gtok.ops.vocab_stats_synth(batch.to(DEV), 80, tq, graph_base=1000)
        assert np.array_equal(count.cpu().numpy(), want[0]) and np.array_equal(first.cpu().numpy(), want[1]), shuffle
        # two shards accumulated into the same tables
        acc = None
        for lo, hi in ((0, 400), (400, 900)):
            acc = gtok.ops.vocab_stats_synth(batch.shard(lo, hi).to(DEV), 80, None if tq is None else tq[lo:hi],
                                             graph_base=1000 + lo, out=acc)
        assert np.array_equal(acc[0].cpu().numpy(), want[0]) and np.array_equal(acc[1].cpu().numpy(), want[1])
    batch, _ = both(d, False)
    tails = [t.split("<p>")[1].split()[0] for t in d["texts"]]
    tname = "has_cycle" if task == "cycle_check" else "shortest_distance"
    for min_freq, max_tokens in ((1, None), (1, 50), (4, 600)):
        want, _ = dl.build_vocab_from_texts(d["texts"], min_freq, max_tokens)
        got, _ = dl.build_vocab_from_graphs(batch.to(DEV), 80, tname, tails, q, min_freq, max_tokens)
        assert got == want


def _rings_with_chords(sizes, chords, seed):
    """Simple symmetric graphs (ring + random chords), both directions listed, row-sorted, typed."""
    rng = np.random.default_rng(seed)
    ncs, ecs, srcs, dsts = [], [], [], []
    for n in sizes:
        pairs = {(i, (i + 1) % n) if i < (i + 1) % n else ((i + 1) % n, i) for i in range(n)} if n > 2 else set()
        while len(pairs) < min((n if n > 2 else 0) + chords, n * (n - 1) // 2) and n > 3:
            a, b = sorted(rng.integers(0, n, 2).tolist())
            if a != b:
                pairs.add((a, b))
        e = np.array(sorted(pairs), np.int64).reshape(-1, 2)
        s = np.concatenate([e[:, 0], e[:, 1]]); t = np.concatenate([e[:, 1], e[:, 0]])
        o = np.lexsort((t, s))
        ncs.append(n); ecs.append(s.size); srcs.append(s[o]); dsts.append(t[o])
    src, dst = np.concatenate(srcs), np.concatenate(dsts)
    N = int(np.sum(ncs))
    return dict(node_counts=np.array(ncs), edge_counts=np.array(ecs), src=src, dst=dst,
                x=rng.integers(0, 30, N), edge_attr=np.minimum(src, dst) % 7)   # symmetric edge types


@pytest.mark.parametrize("top", [64, 65, 128, 129, 256, 257, 512])
def test_sent_at_the_size_boundaries_of_the_kernels(top, monkeypatch):
    """Batches whose largest graph sits exactly at / just past 64, 128, 256 nodes (lane / register kernels stop at
    64; the LDS kernel switches its word count W at each power of two) and at the 512-node limit; 254 entries is
    the most a lane-kernel graph may list (u8 row pointers)."""
    sizes = [top, top - 1, top - 2, 3, 1, 0] * 3
    chords = 63 if top == 64 else 40          # 64-node ring + 63 chords = 127 edges = 254 entries
    d = _rings_with_chords(sizes, chords, seed=top)
    batch, coo = both(d)
    assert batch.flags & 1
    pins = ["lane", "lane-int32", "lane-unsorted", "reg", "lds"] if top <= 64 else ["lds"]
    for pin in pins:
        _pin_sent(monkeypatch, pin)
        for labeled in (True, False):
            b2, c2 = both(d, labeled)
            kw = dict(labeled=labeled, num_node_types=28 if labeled else 0, num_edge_types=5 if labeled else 0)
            for max_len in (4096, 100):
                ids, ln = gtok.ops.sent(b2.to(DEV), top, max_len, 7, 1, **kw)
                ref, rln = orc.sent(c2, top, max_len, 7, 1, ld=ids.shape[1], **kw)
                _cmp(ids, ln, ref, rln, f"sent boundary top={top} [{pin}] labeled={labeled} max_len={max_len}")
    if top == 512:
        big = _rings_with_chords([513, 4], 3, seed=1)
        bb, _ = both(big, False)
        with pytest.raises(gtok.GtokError):
            gtok.ops.sent(bb.to(DEV), 513, 64, 0)


def test_text_to_ids_long_tokens_chunk_borders_and_big_vocab():
    """The text kernel splits 1 KB chunks out of a 2 KB LDS ring: tokens that straddle chunk borders, tokens longer
    than the ring (bytes past it come from global memory), keys longer than the 19 bytes an LDS vocab slot holds, and
    a vocab too large for LDS (> 1024 slots: table probed in global memory) must all give the oracle's ids."""
    rng = np.random.default_rng(9)
    words = ["<bos>", "<e>", "<n>", "<q>", "<p>", "yes", "shortest_distance", "a_token_of_exactly_24_by", "k" * 19, "k" * 20,
             "L" * 1500, "M" * 3000, "q"] + [str(i) for i in range(300)]
    texts = []
    for t in range(40):
        toks = [words[int(j)] for j in rng.integers(0, len(words), int(rng.integers(1, 900)))]
        seps = [" " * int(rng.integers(1, 4)) if rng.random() < 0.9 else "\n\t " for _ in toks]
        texts.append("".join(a + b for a, b in zip(toks, seps)))
    texts += ["w" * 1023 + " x", "w" * 1024 + " x", "w" * 1025 + " x y", " " * 2047 + "q", "q" * 2049, ""]
    small = {"<pad>": 0}
    for w in words:
        small.setdefault(w, len(small))
    big = dict(small)
    for i in range(3000):
        big.setdefault(f"val_{i}", len(big))
    tb, tp = gtok.ops.pack_texts(texts)
    for vocab in (small, big):
        table = gtok.ops.VocabTable(vocab, DEV)
        assert (table.capacity <= 1024) == (vocab is small)
        for strip in (True, False):
            for max_len in (2048, 37):
                ids, ln = gtok.ops.text_to_ids(tb.to(DEV), tp, table, max_len, strip_label=strip)
                ref, rln = orc.text_to_ids(texts, vocab, max_len, ids.shape[1], strip_label=strip)
                _cmp(ids, ln, ref, rln, f"text vocab={len(vocab)} strip={strip} max_len={max_len}")


def test_find_token_on_collated_batches():
    """§8f-3: the model-side `<q>` search (first occurrence per row, -1 when absent) on the int64 batch gtok_collate
    hands to the model."""
    rng = np.random.default_rng(4)
    for B, L in ((128, 600), (1, 1), (5, 64), (7, 65), (300, 1023), (4, 0)):
        x = rng.integers(0, 50, (B, L)).astype(np.int64)
        if L:
            x[::3] = np.where(x[::3] == 4, 5, x[::3])      # every third row has no <q> (token 4) at all
            if B > 2 and L > 2:
                x[1, -1] = 4; x[2, 0] = 4
        got = gtok.ops.find_token(torch.from_numpy(x).to(DEV), 4)
        assert np.array_equal(got.cpu().numpy(), orc.find_token(x, 4)), (B, L)
    d = gtok.synth.graph_token_like(200, seed=21, task="shortest_path")
    vocab = {t: i for i, t in enumerate(["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no", "shortest_distance"]
                                        + [str(i) for i in range(64)])}
    tb, tp = gtok.ops.pack_texts(d["texts"])
    ids, ln = gtok.ops.text_to_ids(tb.to(DEV), tp, gtok.ops.VocabTable(vocab, DEV), 4096)   # no text is cut before <q>
    X, A = gtok.ops.collate(ids, ln, torch.arange(200), vocab["<pad>"], int(ln.max()))
    qpos = gtok.ops.find_token(X, vocab["<q>"]).cpu().numpy()
    Xh = X.cpu().numpy()
    assert (qpos >= 0).all()
    for g, (u, v) in enumerate(d["queries"]):               # "<q> shortest_distance u v <p>": u at +2, v at +3
        assert Xh[g, qpos[g] + 2] == vocab[str(u)] and Xh[g, qpos[g] + 3] == vocab[str(v)]


def test_csr_builder_on_the_device_equals_the_host_builder():
    """GraphBatch.from_coo_device (torch sort / bincount / cumsum on the GPU) gives the host builder's arrays bit for
    bit: sorted and unsorted edge lists, typed and untyped, the hand-made edge cases, flags and chunk maxima included."""
    cases = [(gtok.synth.zinc_like(5000, seed=41), True), (gtok.synth.zinc_like(1500, seed=42, coalesced=False), True),
             (gtok.synth.graph_token_like(400, seed=43, with_text=False), False),
             (gtok.synth.er_batch(50, seed=44, min_nodes=10, max_nodes=200), False), (edge_case_graphs(), True),
             (_rings_with_chords([64, 63, 5, 1, 0], 20, seed=3), True)]
    for d, labeled in cases:
        x, ea = (d.get("x"), d.get("edge_attr")) if labeled else (None, None)
        h = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], x, ea)
        g = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], x, ea, device=DEV)
        meta = lambda b: (b.num_graphs, b.max_nodes, b.max_edges, b.flags, b.chunk_nodes, b.chunk_edges)
        assert meta(g) == meta(h)
        for name in ("node_ptr", "edge_ptr", "rowptr", "col", "eorder", "nattr", "eattr"):
            a, b = getattr(h, name), getattr(g, name)
            assert (a is None) == (b is None), name
            if a is not None:
                assert a.dtype == b.dtype and torch.equal(a, b.cpu()), name
    empty = gtok.GraphBatch.from_coo_device([], [], [], [], device=DEV)
    assert empty.num_graphs == 0 and empty.flags == 0


def test_graph_token_text_parser_on_the_device():
    """§8f-2 / a13: gtok_parse_graph_text == the oracle's restatement of the reference parsers on canonical texts
    (both tasks, INF labels, empty graphs); anything else is flagged (status 1) and parse_texts_on_device hands it to
    the host parser, so the mirror's result equals the host path for every text."""
    import importlib
    gm = importlib.import_module("glearning-benchmark_amd.graph_data_loader.graph_token_dataset_autograph")
    texts = []
    for task in ("cycle_check", "shortest_path"):
        texts += gtok.synth.graph_token_like(300, seed=51, task=task, min_nodes=2, max_nodes=60)["texts"]
    texts += ["<bos> <n> 0 1 2 <q> has_cycle <p> no <eos>",                      # no edges: nodes from the list
              "<bos> 0 1 <e> 1 2 <e> <n> <q> has_cycle <p> YES <eos>",          # empty node list: max endpoint + 1
              "<bos> 3 4 <e> <n> 0 1 2 3 4 <q> shortest_distance 3 4 <p> len1 <eos>",
              "<bos> 3 4 <e> <n> 0 1 2 3 4 <q> shortest_distance 0 3 <p> INF <eos>",
              "<bos> 0 1 <e> <n> 0 1 <p> Len12 <eos>", "<bos> <n> <eos>", "<bos> 007 8 <e> <n> 0 8 <p> no",
              "<bos> 0 1 <e> 1 2 <e> <n> 0 1 2 <q> <e> <p> yes <eos>",          # a stray <e> where any word may stand: not an edge
              "<bos> 5 6 <e> <n> 5 6 <q> has_cycle <p> <e>",
              "<bos> 0 1 <e> <n> 0 1 <q> x<e>y <p> yes <eos>",                  # the sizing pass counts the bytes `<e>`: a slot too many, squeezed
              "<bos>\t0 1\t<e>\n<n> 0 1 <p> yes"]                               # any Python whitespace separates tokens
    canonical = len(texts)
    texts += ["0 1 <e> <n> 0 1 <q> has_cycle <p> yes <eos>",                     # no <bos>
              "<bos> 0 1 <e> 2 <e> <n> 0 1 2 <p> yes", "<bos> 0 1 2 <e> <n> 0 1 2 <p> no",   # tokens out of place
              "<bos> 0 1 <e> <n> 0 x 1 <p> yes", "<bos> 0 1 <e> <p> yes",       # junk in the node list, no <n>
              "<bos> 0 +1 <e> <n> 0 1 <p> yes", "<bos> 0 12345678901 <e> <n> 0 1 <p> yes",
              "<bos> 0 1 <e> <n> 0 1 <p> lenx <p> yes", "<bos> 0 1 <E> <n> 0 1 <p> yes", "", "   "]
    tb, tp = gtok.ops.pack_texts(texts)
    r = gtok.ops.parse_graph_texts(tb.to(DEV), tp)
    st = r["status"].cpu().numpy()
    assert not st[:canonical].any(), np.nonzero(st[:canonical])[0]
    assert st[canonical:].all(), st[canonical:]
    ep = r["edge_ptr"].cpu().numpy(); src = r["src"].cpu().numpy(); dst = r["dst"].cpu().numpy()
    nn = r["num_nodes"].cpu().numpy(); lab = r["label"].cpu().numpy(); q = r["query"].cpu().numpy()
    for g in range(canonical):
        edges, n, query, label = orc.parse_graph_text(texts[g])
        assert list(zip(src[ep[g]:ep[g + 1]].tolist(), dst[ep[g]:ep[g + 1]].tolist())) == edges, g
        assert nn[g] == n and (None if lab[g] == gtok.ops.NO_LABEL else lab[g]) == label, (g, nn[g], n, lab[g], label)
        assert (None if q[g, 0] < 0 else (q[g, 0], q[g, 1])) == query, g
    # the mirror: device parse + host parse of the flagged texts == host parse of everything
    got = gm.parse_texts_on_device(texts, DEV)
    for g, text in enumerate(texts):
        edges, n, label = gm.parse_graph_from_json({"text": text})
        assert got[g] == (edges, n, label, gm.parse_query_nodes_from_text(text)), (g, text[:60])
        oe, on, oq, ol = orc.parse_graph_text(text)
        assert (oe, on, ol, oq) == (edges, n, label, gm.parse_query_nodes_from_text(text)), g   # oracle == mirror too


def test_graph_token_text_parser_with_more_tag_bytes_than_edges_fit():
    """ops.parse_graph_texts sizes its edge arrays by the bytes (an accepted edge is >= 8 bytes) instead of reading the
    count back; a corpus whose junk holds more `<e>` than that is parsed a second time with exact arrays: the canonical
    texts behind the junk keep every edge."""
    good = gtok.synth.graph_token_like(20, seed=5, task="cycle_check", min_nodes=5, max_nodes=40)["texts"]
    texts = ["<e>" * 4000, "<bos> " + "<e> " * 500 + "<n> 0 <p> yes"] + good
    tb, tp = gtok.ops.pack_texts(texts)
    assert 4000 + 500 > tb.numel() // 8 + len(texts) + 1          # the case is the one meant
    r = gtok.ops.parse_graph_texts(tb.to(DEV), tp)
    st = r["status"].cpu().numpy()
    assert st[0] == 1 and st[1] == 1 and not st[2:].any()
    ep = r["edge_ptr"].cpu().numpy(); src = r["src"].cpu().numpy(); dst = r["dst"].cpu().numpy()
    assert ep[1] == ep[0] and ep[2] == ep[1] and ep[-1] == src.size == dst.size
    for g in range(2, len(texts)):
        edges, n, query, label = orc.parse_graph_text(texts[g])
        assert list(zip(src[ep[g]:ep[g + 1]].tolist(), dst[ep[g]:ep[g + 1]].tolist())) == edges, g
        assert int(r["num_nodes"][g]) == n


def test_sent_decode_kernel_equals_the_oracle_decoder():
    """§8f-4: gtok_sent_decode == oracle_sent_decode on the GPU's own SENT rows (labelled, unlabelled, cut at max_len,
    large graphs), on damaged rows (same status, same partial output) and when the output capacities are too small; and
    the decoded graphs have the node / edge counts of their inputs."""
    cases = [(gtok.synth.zinc_like(3000, seed=61), True, 37, 28), (gtok.synth.zinc_like(1000, seed=62), False, 37, 0),
             (gtok.synth.graph_token_like(300, seed=63, with_text=False), False, 49, 0),
             (gtok.synth.er_batch(40, seed=64, min_nodes=100, max_nodes=200), False, 200, 0), (edge_case_graphs(), True, 8, 28)]
    for d, labeled, nn, ntypes in cases:
        batch, coo = both(d, labeled)
        kw = dict(labeled=labeled, num_node_types=ntypes, num_edge_types=5 if labeled else 0)
        for max_len in (4096, 50):
            ids, ln = gtok.ops.sent(batch.to(DEV), nn, max_len, 9, 2, **kw)
            for ecap, ncap in ((None, None), (7, 3)):
                got = gtok.ops.sent_decode(ids, ln, nn, labeled, ntypes, ecap, ncap)
                want = orc.sent_decode_rows(ids.cpu().numpy(), ln.cpu().numpy(), nn, labeled, ntypes, ecap, ncap)
                for k in ("num_nodes", "num_edges", "status"):
                    assert np.array_equal(got[k].cpu().numpy(), want[k]), (k, labeled, max_len, ecap)
                for k, cnt in (("edge_a", "num_edges"), ("edge_b", "num_edges"), ("edge_type", "num_edges"), ("node_type", "num_nodes")):
                    g_, w_ = got[k].cpu().numpy(), want[k]
                    valid = np.arange(w_.shape[1])[None, :] < np.minimum(want[cnt], w_.shape[1])[:, None]
                    assert np.array_equal(g_[valid], w_[valid]), (k, labeled, max_len, ecap)
            if max_len == 4096 and int(ln.max()) < 4096:
                got = gtok.ops.sent_decode(ids, ln, nn, labeled, ntypes)
                assert not got["status"].any()
                und = np.array([len({(min(a, b), max(a, b)) for a, b in zip(coo.src[coo.edge_ptr[g]:coo.edge_ptr[g + 1]],
                                                                       coo.dst[coo.edge_ptr[g]:coo.edge_ptr[g + 1]])})
                                for g in range(coo.G)])
                assert np.array_equal(got["num_nodes"].cpu().numpy(), coo.node_counts) and np.array_equal(got["num_edges"].cpu().numpy(), und)
    # damaged rows: identical verdicts
    d = gtok.synth.zinc_like(600, seed=65)
    batch, coo = both(d, True)
    ids, ln = gtok.ops.sent(batch.to(DEV), 37, 4096, 1, 0, labeled=True, num_node_types=28, num_edge_types=5)
    bad = ids.cpu().numpy().copy(); rng = np.random.default_rng(1)
    for g in range(coo.G):
        L = int(ln[g]); p = int(rng.integers(0, L))
        bad[g, p] = int(rng.integers(0, 120))
    got = gtok.ops.sent_decode(torch.from_numpy(bad).to(DEV), ln, 37, True, 28)
    want = orc.sent_decode_rows(bad, ln.cpu().numpy(), 37, True, 28)
    for k in ("num_nodes", "num_edges", "status"):
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    assert (want["status"] != 0).mean() > 0.3


def test_shard_on_device_then_gather_over_rccl_equals_unsharded():
    """BASELINE config 4 on one GPU: a device-resident corpus is block-sharded ON the device (GraphBatch.shard),
    each block tokenized with graph_base = its first global index, and the blocks are reassembled by
    dist.gather_tokens - run through a real one-rank `nccl` (RCCL) group with force=True, so the collective the
    multi-GPU runs rely on has executed on hardware.  Result == the unsharded slab == the oracle."""
    import socket
    import torch.distributed as tdist
    d = gtok.synth.zinc_like(30000, seed=91)          # above the lane kernel's threshold: both kernels get exercised
    batch, coo = both(d)
    dev_batch = batch.to(DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ld = gtok.ops.sent_safe_ld(batch, True, 1024)
    whole_ids, whole_ln = gtok.ops.sent(dev_batch, 37, 1024, 21, 3, ld=ld, **kw)
    ref, rln = orc.sent(coo, 37, 1024, 21, 3, ld=ld, nthreads=8, **kw)
    _cmp(whole_ids, whole_ln, ref, rln, "unsharded")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        # one rank: its block is the whole corpus, the gather is a real (single-rank) RCCL all_gather_into_tensor
        mine, lo, hi = gtok.dist.shard_of(dev_batch)
        assert (lo, hi) == (0, 30000) and mine.col.is_cuda
        ids, ln = gtok.ops.sent(mine, 37, 1024, 21, 3, ld=ld, graph_base=lo, **kw)
        g_ids, g_ln = gtok.dist.gather_tokens(ids, ln, 30000, 5, force=True)
        assert g_ids.data_ptr() != ids.data_ptr()           # went through the collective, not the early return
        assert torch.equal(g_ids, whole_ids) and torch.equal(g_ln, whole_ln)
        # the 4-rank layout rehearsed on one device: every rank's block tokenized on its own, blocks laid out as the
        # gather lays them out (padded to ceil(G/P) rows)
        world, per = 4, -(-30000 // 4)
        parts_i, parts_l = [], []
        for r in range(world):
            blk, lo, hi = gtok.dist.shard_of(dev_batch, r, world)
            assert blk.col.is_cuda and blk.num_graphs == hi - lo
            i_r, l_r = gtok.ops.sent(blk, 37, 1024, 21, 3, ld=ld, graph_base=lo, **kw)
            parts_i.append(i_r); parts_l.append(l_r)
        assert torch.equal(torch.cat(parts_i), whole_ids) and torch.equal(torch.cat(parts_l), whole_ln)
    finally:
        tdist.destroy_process_group()


def test_ticket_ring_never_shares_a_slot_between_live_launches(monkeypatch):
    """The ring of ticket-counter slots has 256 entries.  700 launches of a ticket-scheduled kernel are queued on
    two streams WITHOUT any synchronisation in between (each into its own slab), so the ring wraps while early
    launches are still in flight: the launcher must wait for a slot's previous launch instead of handing the same
    counters to two live launches (skipped units = rows left unwritten).  Every slab is checked afterwards."""
    monkeypatch.setenv("GTOK_SENT_KERNEL", "lds")
    d = gtok.synth.zinc_like(1500, seed=72)
    batch, coo = both(d, False)
    b = batch.to(DEV)
    ld = gtok.ops.sent_safe_ld(batch, False, 1024)
    ref = {k: orc.sent(coo, 40, 1024, 3, k, ld=ld) for k in (0, 1)}
    side = torch.cuda.Stream(device=DEV)
    n = 350
    outs = [[(torch.full((1500, ld), -7, dtype=torch.int32, device=DEV), torch.full((1500,), -7, dtype=torch.int32, device=DEV))
             for _ in range(n)] for _ in range(2)]
    torch.cuda.synchronize()
    for i in range(n):
        gtok.ops.sent(b, 40, 1024, 3, 0, ld=ld, out=outs[0][i])
        with torch.cuda.stream(side):
            gtok.ops.sent(b, 40, 1024, 3, 1, ld=ld, out=outs[1][i])
    torch.cuda.synchronize()
    for k in (0, 1):
        want_ids = torch.from_numpy(ref[k][0]).to(DEV); want_ln = torch.from_numpy(ref[k][1]).to(DEV)
        for i in range(n):
            assert torch.equal(outs[k][i][1], want_ln) and torch.equal(outs[k][i][0], want_ids), (k, i)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_batch_on_another_device_than_the_current_one():
    """A batch on cuda:1 tokenized while cuda:0 is the current device: the launch takes its device (occupancy, ticket
    counters) from the stream it is given, not from the calling thread."""
    torch.cuda.set_device(0)
    d = gtok.synth.zinc_like(3000, seed=73)
    batch, coo = both(d)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    for pin in ("lane", "reg", "lds"):
        os.environ["GTOK_SENT_KERNEL"] = pin
        try:
            ids, ln = gtok.ops.sent(batch.to("cuda:1"), 37, 1024, 5, 1, **kw)
        finally:
            del os.environ["GTOK_SENT_KERNEL"]
        assert ids.device == torch.device("cuda:1") and torch.cuda.current_device() == 0
        ref, rln = orc.sent(coo, 37, 1024, 5, 1, ld=ids.shape[1], **kw)
        _cmp(ids, ln, ref, rln, f"cuda:1 [{pin}]")
    vocab = zinc_vocab(40)
    ids, ln = gtok.ops.ibtt_zinc(batch.to("cuda:1"), gtok.ops.zinc_lut(vocab, 40), 1024, vocab["<pad>"])
    ref, rln = orc.ibtt_zinc(coo, gtok.ops.zinc_lut(vocab, 40).numpy(), 1024, vocab["<pad>"], ids.shape[1])
    _cmp(ids, ln, ref, rln, "cuda:1 ibtt")


def test_config5_family_mix_up_to_256_nodes():
    """BASELINE config 5's families (er / ba / sbm / sfn / path / star / complete) at 10-256 nodes through the
    large-graph kernels, plus the extremes the ER-only runs never reach: a 256-node star (one row of 255 entries), the
    complete graph K256 (32,640 entries), a 256-node path.  SENT (LDS bit-matrix kernel), the IBTT graph-token grammar
    and the vocab statistics, all bit-exact against the oracle; max_len 600 cuts the long rows (the hot truncation path)."""
    mix = gtok.synth.mix_batch_device(140, DEV, seed=5)
    assert sorted(set(mix["family"].tolist())) == list(range(7)) and int(mix["node_counts"].max()) > 200
    n = 256
    iu, iv = np.triu_indices(n, 1)
    extremes = dict(node_counts=np.array([n, n, n, 3]), edge_counts=np.array([n - 1, iu.size, n - 1, 2]),
                    src=np.concatenate([np.zeros(n - 1, np.int64), iu, np.arange(n - 1), [0, 1]]),
                    dst=np.concatenate([np.arange(1, n), iv, np.arange(1, n), [1, 2]]))
    for name, d in (("mix", mix), ("extremes", extremes)):
        batch, coo = both(d, False)
        assert 128 < batch.max_nodes <= 256
        b = batch.to(DEV)
        for max_len in (600, 100000 if name == "mix" else 70000):
            ids, ln = gtok.ops.sent(b, 256, max_len, 9, 2)
            assert gtok.ops.sent_kernel_name(b, 256, max_len).startswith("sent_lds_kernel<W=4>")
            ref, rln = orc.sent(coo, 256, max_len, 9, 2, ld=ids.shape[1], nthreads=8)
            _cmp(ids, ln, ref, rln, f"sent {name} max_len={max_len}")
        vocab = {t: i for i, t in enumerate(["<pad>", "<bos>", "<e>", "<n>", "<q>", "<p>", "<eos>", "yes", "no", "has_cycle"]
                                            + [str(i) for i in range(256)])}
        lut = gtok.ops.synth_lut(vocab, 256)
        q = np.zeros((batch.num_graphs, 4), np.int32); q[:, 0] = 1; q[:, 1] = vocab["has_cycle"]
        ids, ln = gtok.ops.ibtt_synth(b, lut, torch.from_numpy(q), 600, 0)
        ref, rln = orc.ibtt_synth(coo, lut.numpy(), q, 600, 0, ids.shape[1])
        _cmp(ids, ln, ref, rln, f"ibtt_synth {name}")
        c, f = gtok.ops.vocab_stats_synth(b, 256)
        rc, rf = orc.vocab_stats_synth(coo, 256)
        assert np.array_equal(c.cpu().numpy(), rc) and np.array_equal(f.cpu().numpy(), rf), name


def test_vocab_from_texts_on_the_device():
    """the corpus pass of build_vocab_from_texts (data_loader.py:451-463) and of the ZINC dynamic-token
    scan (trainer/train_ibtt.py:361-372) over arbitrary texts, on the device.  Table entries == the oracle's
    Counter-based restatement (token, count, first offset) in most_common order; vocab == the reference's (golden) for
    every min_freq / max_tokens cut; ZINC: same token set as the reference's hash-seed-0 vocab, dynamic ids by first
    appearance (the deterministic stand-in for the reference's `set` order)."""
    from _util import config1_examples, golden, golden2
    gdl = gtok.graph_data_loader
    arr, meta = golden()
    for c in meta["vocab_cases"]:                                  # ties, min_freq, max_tokens corner cases
        v, itos = gdl.build_vocab_from_texts_on_device(c["texts"], min_freq=c["min_freq"], max_tokens=c["max_tokens"], device=DEV)
        assert list(v.items()) == [tuple(p) for p in c["vocab"]] and itos == {i: t for t, i in v.items()}
    for task in ("cycle_check", "shortest_path"):
        texts = [e["text"] for e in config1_examples(task)]        # 1,002 graph-token records
        blob, ptr = gtok.ops.pack_texts(texts)
        table = gtok.ops.vocab_stats_text(blob.to(DEV), ptr, 1 << 12)
        assert gtok.ops.text_stats_entries(table, blob.to(DEV)) == orc.vocab_stats_text(texts)
        # accumulated over two shards (base_offset = the shard's byte offset) == one pass
        half = len(texts) // 2
        b1, p1 = gtok.ops.pack_texts(texts[:half]); b2, p2 = gtok.ops.pack_texts(texts[half:])
        t2 = gtok.ops.vocab_stats_text(b1.to(DEV), p1, 1 << 12)
        t2 = gtok.ops.vocab_stats_text(b2.to(DEV), p2, 1 << 12, base_offset=int(p1[-1]), out=t2)
        assert gtok.ops.text_stats_entries(t2, blob.to(DEV)) == orc.vocab_stats_text(texts)
        _, m2 = golden2()
        for mf, mt in ((1, 600), (1, 40), (3, None), (50, 600)):
            v, _ = gdl.build_vocab_from_texts_on_device(texts, min_freq=mf, max_tokens=mt, device=DEV)
            want, _ = gdl.build_vocab_from_texts(texts, min_freq=mf, max_tokens=mt)
            assert list(v.items()) == list(want.items())
            if (mf, mt) == (1, 600):
                assert list(v.items()) == [tuple(p) for p in m2[f"config1_{task}_vocab"]]   # the reference's own vocab
    # a table that is too small says so
    texts = [e["text"] for e in config1_examples("cycle_check")]
    blob, ptr = gtok.ops.pack_texts(texts)
    with pytest.raises(gtok.GtokError):
        gtok.ops.text_stats_entries(gtok.ops.vocab_stats_text(blob.to(DEV), ptr, 16), blob.to(DEV))
    # ZINC: fixed table + unseen tokens
    ztexts = meta["zinc_L1024_texts"]
    ref_vocab = dict(meta["zinc_L1024_vocab"])                     # the reference's, PYTHONHASHSEED=0 set order
    v = gdl.build_zinc_vocab_on_device(ztexts, device=DEV)
    fixed, _ = gdl.build_fixed_zinc_vocab()
    assert set(v) == set(ref_vocab) and all(v[t] == i for t, i in fixed.items())
    dyn = [t for t in v if t not in fixed]
    assert [v[t] for t in dyn] == list(range(22, 22 + len(dyn)))
    seen = []
    for t in ztexts:
        for tok in t.split():
            if tok not in fixed and tok not in seen:
                seen.append(tok)
    assert dyn == seen                                              # first appearance order
    # and the vocab it gives tokenizes the corpus to the reference's ids up to that permutation of the dynamic ids
    ex = [{"text": t, "label": 0} for t in ztexts]
    a = gdl.TokenDataset(ex, v, 1024); b = gdl.TokenDataset(ex, ref_vocab, 1024)
    perm = {ref_vocab[t]: v[t] for t in v}
    assert all([perm[int(x)] for x in sb] == sa.tolist() for sa, sb in zip(a.seqs, b.seqs))


def test_sent_without_padding_writes_the_same_rows():
    """GTOK_SENT_NO_PAD: every row equals the padded run up to its length; what lies beyond is left alone (checked
    against a sentinel for the rows' far ends), lengths are identical.  TokenizedGraphDataset tokenizes this way."""
    d = gtok.synth.zinc_like(40000, seed=93)
    batch, coo = both(d)
    b = batch.to(DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ld = gtok.ops.sent_safe_ld(batch, True, 1024)
    ids, ln = gtok.ops.sent(b, 37, 1024, 4, 6, ld=ld, **kw)
    assert gtok.ops.sent_kernel_name(b, 37, 1024, **kw) == "sent_lane_kernel"
    raw = torch.full((40000, ld), -9, dtype=torch.int32, device=DEV)
    ln2 = torch.empty(40000, dtype=torch.int32, device=DEV)
    gtok.ops.sent(b, 37, 1024, 4, 6, ld=ld, out=(raw, ln2), pad=False, **kw)
    assert torch.equal(ln, ln2)
    col = torch.arange(ld, device=DEV)[None, :]
    inside = col < ln[:, None]
    assert torch.equal(torch.where(inside, raw, 0), torch.where(inside, ids, 0))
    far = col >= ((ln[:, None] + 15) // 16) * 16
    assert bool((raw[far] == -9).all())
    X, A = gtok.ops.collate(raw, ln2, torch.arange(128, device=DEV), 5, int(ln2[:128].max()))
    X0, A0 = gtok.ops.collate(ids, ln, torch.arange(128, device=DEV), 5, int(ln[:128].max()))
    assert torch.equal(X, X0) and torch.equal(A, A0)


@pytest.mark.parametrize("pin", ["lane", "lane-int32"])
def test_lane_kernel_many_units_per_wave_and_chunks_beyond_the_staging_registers(pin, monkeypatch):
    """Two paths of sent_lane_kernel the big corpora never take (ZINC-full gives every wave exactly one unit whose chunk
    fits the staging registers): (1) a grid smaller than the number of 64-graph units (GTOK_LANE_BLOCKS_PER_CU=1: 256
    waves for 625 units) - waves loop over units, padding the previous unit's rows behind the next unit's loads;
    (2) units of dense 60-64-node graphs with ~250 entries each (16 k entries per unit) - the staging tail loops beyond
    the register-held vectors, counter planes P = 6 (degree up to 63), rows longer than four entries everywhere."""
    _pin_sent(monkeypatch, pin)
    monkeypatch.setenv("GTOK_LANE_BLOCKS_PER_CU", "1")
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    d = gtok.synth.zinc_like(40000, seed=95)
    batch, coo = both(d)
    for max_len in (1024, 57):
        ids, ln = gtok.ops.sent(batch.to(DEV), 37, max_len, 8, 3, **kw)
        ref, rln = orc.sent(coo, 37, max_len, 8, 3, ld=ids.shape[1], nthreads=8, **kw)
        _cmp(ids, ln, ref, rln, f"many units per wave [{pin}] max_len={max_len}")
    monkeypatch.delenv("GTOK_LANE_BLOCKS_PER_CU")
    dense = _rings_with_chords([64, 63, 62, 61, 60, 64, 64, 33] * 25, 60, seed=5)      # 200 graphs, ~3 units, up to 248 entries each
    # one hub per graph (node 3 tied to every third node): degrees past 15 -> six counter planes, long rows
    nptr = np.concatenate([[0], np.cumsum(dense["node_counts"])]); eptr = np.concatenate([[0], np.cumsum(dense["edge_counts"])])
    srcs, dsts, ecs = [], [], []
    for g, n in enumerate(dense["node_counts"]):
        pairs = set(zip(dense["src"][eptr[g]:eptr[g + 1]].tolist(), dense["dst"][eptr[g]:eptr[g + 1]].tolist()))
        pairs = {(a, b) for a, b in pairs if a < b and len(pairs) and (a, b) in pairs}
        keep = sorted(pairs)[:90]                                   # room for the hub inside the 255-entry limit
        hub = {(min(3, j), max(3, j)) for j in range(0, int(n), 3) if j != 3}
        e = np.array(sorted(set(keep) | hub), np.int64).reshape(-1, 2)
        s_ = np.concatenate([e[:, 0], e[:, 1]]); t_ = np.concatenate([e[:, 1], e[:, 0]])
        o = np.lexsort((t_, s_))
        srcs.append(s_[o]); dsts.append(t_[o]); ecs.append(s_.size)
    dense = dict(dense, edge_counts=np.array(ecs), src=np.concatenate(srcs), dst=np.concatenate(dsts))
    dense["edge_attr"] = np.minimum(dense["src"], dense["dst"]) % 7
    assert int(dense["edge_counts"].max()) <= 255
    for labeled in (True, False):
        b2, c2 = both(dense, labeled)
        assert b2.max_degree > 15 and b2.flags & 1
        kw2 = dict(labeled=labeled, num_node_types=28 if labeled else 0, num_edge_types=5 if labeled else 0)
        for max_len in (4096, 300):
            ids, ln = gtok.ops.sent(b2.to(DEV), 64, max_len, 2, 9, **kw2)
            ref, rln = orc.sent(c2, 64, max_len, 2, 9, ld=ids.shape[1], **kw2)
            _cmp(ids, ln, ref, rln, f"dense units [{pin}] labeled={labeled} max_len={max_len}")


@pytest.mark.parametrize("pin", ["lane", "lane-int32", "lane-unsorted", "reg"])
def test_lane_kernel_row_ends_query_tail_cuts_and_odd_slabs(pin, monkeypatch):
    """Everything sent_lane_kernel does at the END of a row, on graphs it accepts (symmetric molecules): the query tail
    (labelled and unlabelled, remapped or not), rows cut by max_len at every phase of the 4-token window and of the
    store burst, slabs narrower than the rows (len > ld), slab widths that are not multiples of 4 or 16 (element-wise
    stores instead of the 16-byte ones), and GTOK_SENT_NO_PAD on top.  The reg kernel runs the same matrix."""
    _pin_sent(monkeypatch, pin)
    d = gtok.synth.zinc_like(700, seed=97)
    rng = np.random.default_rng(3)
    q = np.stack([rng.integers(0, d["node_counts"]), rng.integers(0, d["node_counts"])], 1).astype(np.int32)
    for labeled, remap in ((True, True), (True, False), (False, False)):
        batch, coo = both(d, labeled)
        b = batch.to(DEV)
        kw = dict(labeled=labeled, num_node_types=9 if labeled else 0, num_edge_types=4 if labeled else 0, remap_zinc=remap)
        for max_len, ld in ((1024, None), (1024, 64), (1024, 61), (1024, 203), (33, None), (34, 48), (35, 41), (36, 36), (37, 20),
                            (5, 8), (2, 4), (1, 7), (0, 4), (90, 96), (91, 91)):
            for query in (None, q):
                kwq = dict(kw, query=None if query is None else torch.from_numpy(query))
                ids, ln = gtok.ops.sent(b, 37, max_len, 13, 4, ld=ld, **kwq)
                ref, rln = orc.sent(coo, 37, max_len, 13, 4, ld=ids.shape[1], query=query, **kw)
                _cmp(ids, ln, ref, rln, f"[{pin}] labeled={labeled} remap={remap} max_len={max_len} ld={ld} query={query is not None}")
        # rows without padding: equal inside their lengths
        raw = torch.full((700, 208), -3, dtype=torch.int32, device=DEV); l2 = torch.empty(700, dtype=torch.int32, device=DEV)
        gtok.ops.sent(b, 37, 1024, 13, 4, ld=208, out=(raw, l2), pad=False, query=torch.from_numpy(q), **kw)
        ref, rln = orc.sent(coo, 37, 1024, 13, 4, ld=208, query=q, **kw)
        inside = np.arange(208)[None, :] < rln[:, None]
        assert np.array_equal(l2.cpu().numpy(), rln) and np.array_equal(np.where(inside, raw.cpu().numpy(), 0), np.where(inside, ref, 0))


@pytest.mark.parametrize("order", ["1", "0"])
def test_sent_bit_matrix_lane_kernel(order, monkeypatch):
    """sent_blane_kernel (lane per graph over the adjacency bit-matrix mirror; unlabelled, <= 256 nodes, ANY edge list)
    against the oracle: molecules, the graph-token families, the hand-made edge cases (self loops, duplicates, one
    direction only, isolated nodes, empty graphs), every word count W (batches topping out at 64 / 65 / 128 / 129 / 256
    nodes), 4 and 8 counter planes, the 256-node star / complete graph / path, the query tail, rows cut by max_len at
    every phase of the token window, odd slab widths, GTOK_SENT_NO_PAD, a non-zero graph_base, and the graphs dealt
    to lanes in dataset order (order = "0") or longest walk first."""
    monkeypatch.setenv("GTOK_SENT_KERNEL", "blane")
    monkeypatch.setenv("GTOK_BLANE_ORDER", order)
    cases = [("zinc", gtok.synth.zinc_like(1500, seed=63), 40),
             ("families", gtok.synth.graph_token_like(400, seed=64, with_text=False, algorithms=("er", "ba", "sbm", "path", "star")), 49),
             ("edge cases", edge_case_graphs(), 8)]
    for top in (64, 65, 128, 129, 256):
        cases.append((f"top={top}", _rings_with_chords([top, top - 1, top - 2, 3, 1, 0] * 3 + [top] * 70, 3, seed=top), top))
    n = 256
    iu, iv = np.triu_indices(n, 1)
    cases.append(("extremes", dict(node_counts=np.array([n, n, n, 3]), edge_counts=np.array([n - 1, iu.size, n - 1, 2]),
                                   src=np.concatenate([np.zeros(n - 1, np.int64), iu, np.arange(n - 1), [0, 1]]),
                                   dst=np.concatenate([np.arange(1, n), iv, np.arange(1, n), [1, 2]])), 256))
    for name, d, nn in cases:
        batch, coo = both(d, False)
        b = batch.to(DEV)
        for max_len in (100000, 40):
            ids, ln = gtok.ops.sent(b, nn, max_len, 17, 5, graph_base=123)
            assert b.adj_rows is not None and (b.lane_order is not None) == (order == "1")
            assert gtok.ops.sent_kernel_name(b, nn, max_len).startswith("sent_blane_kernel<W=%d>" % b.adj_words), name
            ref, rln = orc.sent(coo, nn, max_len, 17, 5, graph_base=123, ld=ids.shape[1], nthreads=8)
            _cmp(ids, ln, ref, rln, f"blane {name} max_len={max_len}")
    # row ends: query tail x cuts x slab widths (the matrix of test_lane_kernel_row_ends_query_tail_cuts_and_odd_slabs)
    d = gtok.synth.zinc_like(700, seed=97)
    rng = np.random.default_rng(3)
    q = np.stack([rng.integers(0, d["node_counts"]), rng.integers(0, d["node_counts"])], 1).astype(np.int32)
    batch, coo = both(d, False)
    b = batch.to(DEV)
    for max_len, ld in ((1024, None), (1024, 64), (1024, 61), (1024, 203), (33, None), (34, 48), (35, 41), (36, 36), (37, 20),
                        (5, 8), (2, 4), (1, 7), (0, 4), (90, 96), (91, 91)):
        for query in (None, q):
            ids, ln = gtok.ops.sent(b, 37, max_len, 13, 4, ld=ld, query=None if query is None else torch.from_numpy(query))
            ref, rln = orc.sent(coo, 37, max_len, 13, 4, ld=ids.shape[1], query=query)
            _cmp(ids, ln, ref, rln, f"blane max_len={max_len} ld={ld} query={query is not None}")
    raw = torch.full((700, 208), -3, dtype=torch.int32, device=DEV); l2 = torch.empty(700, dtype=torch.int32, device=DEV)
    gtok.ops.sent(b, 37, 1024, 13, 4, ld=208, out=(raw, l2), pad=False, query=torch.from_numpy(q))
    ref, rln = orc.sent(coo, 37, 1024, 13, 4, ld=208, query=q)
    inside = np.arange(208)[None, :] < rln[:, None]
    assert np.array_equal(l2.cpu().numpy(), rln) and np.array_equal(np.where(inside, raw.cpu().numpy(), 0), np.where(inside, ref, 0))
    # K256 plus a self loop: a degree of 256 does not fit 8 counter planes -> no mirror, the LDS kernel takes the batch
    full = dict(node_counts=np.array([n, 5]), edge_counts=np.array([iu.size + 1, 1]),
                src=np.concatenate([iu, [7], [0]]), dst=np.concatenate([iv, [7], [1]]))
    batch, coo = both(full, False)
    b = batch.to(DEV)
    ids, ln = gtok.ops.sent(b, 256, 100000, 3, 1)
    assert b.adj_rows is None and gtok.ops.sent_kernel_name(b, 256, 100000).startswith("sent_lds_kernel")
    ref, rln = orc.sent(coo, 256, 100000, 3, 1, ld=ids.shape[1])
    _cmp(ids, ln, ref, rln, "K256 + self loop")


def test_config5_sized_batches_take_the_bit_matrix_lane_kernel():
    """A config-5 sized batch (>= 20 k unlabelled graphs of 10-256 nodes) is tokenized by sent_blane_kernel by default;
    bit-exact against the oracle, ER and the seven-family mix."""
    for name, d in (("er", gtok.synth.er_batch_device(20480, DEV, seed=3)), ("mix", gtok.synth.mix_batch_device(20480, DEV, seed=4))):
        batch, coo = both(d, False)
        b = batch.to(DEV)
        for max_len in (100000, 600):
            ids, ln = gtok.ops.sent(b, 256, max_len, 9, 2)
            assert gtok.ops.sent_kernel_name(b, 256, max_len).startswith("sent_blane_kernel<W=4>")
            ref, rln = orc.sent(coo, 256, max_len, 9, 2, ld=ids.shape[1], nthreads=8)
            _cmp(ids, ln, ref, rln, f"blane default {name} max_len={max_len}")


def test_lane_kernel_on_the_reordered_batch(monkeypatch):
    """ops.lane_sorted: the copy of a batch that sent_lane_kernel walks by default - graphs stored by descending
    expected walk length, units of <= 64 neighbours cut to the 10 KB LDS budget, graph_ids carrying every slot's
    dataset index.  Sizes chosen against the packing: 64-node graphs with ~250 entries (about 19 to a unit), single
    nodes and triangles (64 to a unit), molecule-sized rings in between, a batch smaller than one unit.  Tokens, lengths,
    query tails and RNG identity (graph_base) must be those of the batch in dataset order = the oracle's."""
    _pin_sent(monkeypatch, "lane")
    rng = np.random.default_rng(11)
    sizes = [64] * 150 + [3] * 700 + [1] * 130 + [37] * 400 + [20] * 300 + [2] * 64
    rng.shuffle(sizes)
    big = _rings_with_chords(sizes, 58, seed=21)
    small = _rings_with_chords([9, 5, 33, 1, 64, 12], 3, seed=22)
    for name, d in (("mixed", big), ("one unit", small)):
        q = np.stack([rng.integers(0, np.maximum(d["node_counts"], 1)), rng.integers(0, np.maximum(d["node_counts"], 1))], 1).astype(np.int32)
        for labeled in (True, False):
            batch, coo = both(d, labeled)
            assert batch.flags & 1 and batch.max_edges <= 255
            b = batch.to(DEV)
            kw = dict(labeled=labeled, num_node_types=30 if labeled else 0, num_edge_types=7 if labeled else 0)
            for max_len, query in ((4096, None), (90, None), (4096, q)):
                ids, ln = gtok.ops.sent(b, 64, max_len, 5, 3, graph_base=1000, query=None if query is None else torch.from_numpy(query), **kw)
                ref, rln = orc.sent(coo, 64, max_len, 5, 3, graph_base=1000, query=query, ld=ids.shape[1], nthreads=8, **kw)
                _cmp(ids, ln, ref, rln, f"reordered {name} labeled={labeled} max_len={max_len} query={query is not None}")
            assert gtok.ops.sent_kernel_name(b, 64, 4096, **kw) == "sent_lane_kernel"
            sb = b.lane_sorted
            assert sb is not None and sb.graph_ids is not None and sb.rowptr8 is not None
            up = sb.unit_ptr.cpu().numpy(); gi = sb.graph_ids.cpu().numpy()
            assert up[0] == 0 and up[-1] == b.num_graphs and sb.num_units == up.size - 1
            assert np.array_equal(np.sort(gi), np.arange(b.num_graphs)) and (np.diff(up) >= 1).all() and (np.diff(up) <= 64).all()
            nc = (b.node_ptr[1:] - b.node_ptr[:-1]).cpu().numpy()[gi]; ec = (b.edge_ptr[1:] - b.edge_ptr[:-1]).cpu().numpy()[gi]
            cn, ce = np.concatenate([[0], np.cumsum(nc)]), np.concatenate([[0], np.cumsum(ec)])
            un, ue = cn[up[1:]] - cn[up[:-1]], ce[up[1:]] - ce[up[:-1]]
            a16 = lambda v: (v + 15) // 16 * 16
            assert sb.chunk_nodes == un.max() and sb.chunk_edges == ue.max()
            assert a16(un.max() + 80) + 2 * a16(ue.max() + 8) + a16(un.max() + 8) <= gtok.ops.LANE_UNIT_LDS
            if name == "mixed":
                assert sb.num_units > (b.num_graphs + 63) // 64          # the 64-node graphs do not fit 64 to a unit


def test_hip_graph_capture_of_the_one_workgroup_per_cu_launches(monkeypatch):
    """The lane-per-graph kernels launch one workgroup per CU with more than 64 KB of dynamic LDS (an attribute the
    launcher sets on every call: it must be legal while a stream is capturing).  70 k molecules (reordered batch, 16-wave
    workgroups) and 6 k unlabelled graphs of up to 256 nodes (bit-matrix kernel, 8-wave workgroups): captured once,
    replayed, equal to the oracle every time."""
    d = gtok.synth.zinc_like(70000, seed=83)
    batch, coo = both(d)
    b = batch.to(DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ld = gtok.ops.sent_safe_ld(batch, True, 1024)
    e = gtok.synth.er_batch_device(6000, DEV, seed=84)
    eb, ecoo = both(e, False)
    ebd = eb.to(DEV)
    eld = 608
    s_out = (torch.empty((70000, ld), dtype=torch.int32, device=DEV), torch.empty(70000, dtype=torch.int32, device=DEV))
    e_out = (torch.empty((6000, eld), dtype=torch.int32, device=DEV), torch.empty(6000, dtype=torch.int32, device=DEV))
    gtok.ops.sent(b, 37, 1024, 9, 4, ld=ld, out=s_out, **kw)               # warm-up: the resident layouts are made here
    monkeypatch.setenv("GTOK_SENT_KERNEL", "blane")                         # (6 k graphs: below the size where it is the default)
    gtok.ops.sent(ebd, 256, 600, 9, 4, ld=eld, out=e_out)
    assert gtok.ops.sent_kernel_name(b, 37, 1024, **kw) == "sent_lane_kernel" and b.lane_sorted is not None and b.lane_sorted.num_units >= 1024
    assert gtok.ops.sent_kernel_name(ebd, 256, 600).startswith("sent_blane_kernel")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        monkeypatch.delenv("GTOK_SENT_KERNEL")
        gtok.ops.sent(b, 37, 1024, 9, 4, ld=ld, out=s_out, **kw)
        monkeypatch.setenv("GTOK_SENT_KERNEL", "blane")
        gtok.ops.sent(ebd, 256, 600, 9, 4, ld=eld, out=e_out)
    ref, rln = orc.sent(coo, 37, 1024, 9, 4, ld=ld, nthreads=8, **kw)
    eref, erln = orc.sent(ecoo, 256, 600, 9, 4, ld=eld, nthreads=8)
    for rep in range(2):
        s_out[0].fill_(-1); e_out[0].fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        _cmp(s_out[0], s_out[1], ref, rln, f"graph replay {rep}: sent_lane_kernel, one workgroup per CU")
        _cmp(e_out[0], e_out[1], eref, erln, f"graph replay {rep}: sent_blane_kernel")


def test_graph_token_text_parser_streamed_edge_zone_against_the_host_parser():
    """parse_graph_text_kernel streams the edge zone 1 KB at a time (16 bytes per lane) and hands the first window that
    holds anything but `INT INT <e>` triples - and everything behind it - to its general loop.  Texts of up to ~20 KB
    (a dozen windows), then the same texts damaged at random places: runs of spaces, tabs and newlines (whitespace the
    stream does not take), junk tokens, long numbers, tags out of place, cuts in the middle of a token or of a triple,
    leading whitespace.  Whatever the device accepts (status 0) must be what the host parser (the reference's scan) says,
    and the mirror's device + host-fallback result must equal the host parser on every text."""
    import importlib
    gm = importlib.import_module("glearning-benchmark_amd.graph_data_loader.graph_token_dataset_autograph")
    rng = np.random.default_rng(77)
    base = []
    for task in ("cycle_check", "shortest_path"):
        base += gtok.synth.graph_token_like(60, seed=91, task=task, min_nodes=40, max_nodes=230)["texts"]
    assert max(len(t) for t in base) > 12000
    # ids of one to five digits (the stream reads numbers of up to four digits, at most seven per edge; longer ones go to the
    # general loop), every tag position modulo 16 and every window border
    for k in range(40):
        m = int(rng.integers(3, 1500))
        hi = [10, 100, 1000, 10000, 100000][k % 5]
        e = rng.integers(0, hi, (m, 2))
        nn = int(e.max()) + 1
        base.append("<bos> " + " ".join(f"{a} {b} <e>" for a, b in e.tolist()) + " <n> " + " ".join(map(str, range(min(nn, 300)))) + " <q> has_cycle <p> yes <eos>")
    texts = list(base)
    junk = ["x", "12345678901", "1234567890", "007", "<e>", "<n>", "<q>", "+1", "-2", "3.5", "<bos>", "<E>", "9" * 9, "\x01", "<e", "e>"]
    for t in base:
        for _ in range(6):
            u = t
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(0, len(u)))
                op = int(rng.integers(0, 6))
                if op == 0:                                          # stretch a separator
                    sp = u.find(" ", k)
                    if sp >= 0:
                        u = u[:sp] + [" " * int(rng.integers(2, 40)), "\t", "\n", " \r\n ", "\x0b"][int(rng.integers(0, 5))] + u[sp + 1:]
                elif op == 1:                                        # a token that does not belong
                    sp = u.find(" ", k)
                    if sp >= 0:
                        u = u[:sp] + " " + junk[int(rng.integers(0, len(junk)))] + u[sp:]
                elif op == 2:                                        # cut
                    u = u[:k]
                elif op == 3:                                        # drop a byte (glues or shortens tokens)
                    u = u[:k] + u[k + 1:]
                elif op == 4:
                    u = " " * int(rng.integers(1, 70)) + u
                else:                                                # overwrite a byte
                    u = u[:k] + "0a< >e\t9"[int(rng.integers(0, 8))] + u[k + 1:]
            texts.append(u)
    got = gm.parse_texts_on_device(texts, DEV)
    tb, tp = gtok.ops.pack_texts(texts)
    st = gtok.ops.parse_graph_texts(tb.to(DEV), tp)["status"].cpu().numpy()
    assert not st[:len(base)].any() and st[len(base):].any() and not st[len(base):].all()
    for g, text in enumerate(texts):
        edges, n, label = gm.parse_graph_from_json({"text": text})
        assert got[g] == (edges, n, label, gm.parse_query_nodes_from_text(text)), (g, int(st[g]), text[:80])
