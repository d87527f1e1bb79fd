"""Device-side layout steps of the C ABI (ABI v5): gtok_csr_check, gtok_csr_lane_sort, and the routes that reach the
benchmarked sent_lane_kernel through them - torch.ops.gtok.* and bare ctypes (no ops.py)."""
import ctypes

import numpy as np
import pytest
import torch

from _util import both, edge_case_graphs, gtok, lane_sort_reference, orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ZKW = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)


def _dev_eq(t, a, what):
    assert np.array_equal(t.cpu().numpy(), a), what


@pytest.mark.parametrize("G,seed", [(1, 3), (63, 4), (64, 5), (65, 6), (2049, 7), (40000, 8)])
def test_lane_sort_on_the_device_equals_the_restated_rule(G, seed):
    d = gtok.synth.zinc_like(G, seed=seed)
    host, _ = both(d)
    assert host.flags & gtok._lib.CSR_SIMPLE_SYMMETRIC
    sb = gtok.ops.lane_sorted(host.to(DEV))
    ref = lane_sort_reference(host)
    for k in ("graph_ids", "node_ptr", "edge_ptr", "rowptr", "col", "nattr", "eattr", "unit_ptr"):
        _dev_eq(getattr(sb, k), ref[k], f"{k} (G={G})")
    _dev_eq(sb.unit_info, ref["unit_info"], "unit_info")
    assert (sb.num_units, sb.chunk_nodes, sb.chunk_edges) == (ref["unit_ptr"].size - 1, ref["chunk_nodes"], ref["chunk_edges"])
    nr, ne = ref["rowptr"].size, ref["col"].size
    _dev_eq(sb.rowptr8[:nr], ref["rowptr"].astype(np.uint8), "rowptr8")
    _dev_eq(sb.col8[:ne], ref["col"].astype(np.uint8), "col8")


def test_lane_sort_with_empty_graphs_unlabelled_and_a_small_budget():
    # graphs of 0 and 1 nodes between molecules (slots without rows or entries), no attributes, units cut by the caps
    d = gtok.synth.zinc_like(3000, seed=11)
    nc, ec = d["node_counts"].copy(), d["edge_counts"].copy()
    eptr = np.concatenate([[0], np.cumsum(ec)])
    keep = np.ones(ec.sum(), bool)
    for g in range(0, 3000, 7):                       # every 7th graph loses its edges; every 21st its nodes too
        keep[eptr[g]:eptr[g + 1]] = False
        ec[g] = 0
        if g % 21 == 0:
            nc[g] = 0 if g % 42 == 0 else 1
    d2 = dict(node_counts=nc, edge_counts=ec, src=d["src"][keep], dst=d["dst"][keep])
    host, _ = both(d2, labeled=False)
    assert host.flags & gtok._lib.CSR_SIMPLE_SYMMETRIC
    db = host.to(DEV)
    for budget in (10224, 2048):
        db.lane_sorted = None
        old = gtok.ops.LANE_UNIT_LDS
        gtok.ops.LANE_UNIT_LDS = budget
        try:
            sb = gtok.ops.lane_sorted(db)
        finally:
            gtok.ops.LANE_UNIT_LDS = old
        ref = lane_sort_reference(host, budget)
        for k in ("graph_ids", "node_ptr", "edge_ptr", "rowptr", "col", "unit_ptr"):
            _dev_eq(getattr(sb, k), ref[k], f"{k} (budget {budget})")
        _dev_eq(sb.unit_info, ref["unit_info"], "unit_info")
        assert sb.nattr is None and sb.eattr is None


def _damage(d, kind, rng):
    """One defect in one graph of a simple symmetric COO corpus."""
    src, dst, ec = d["src"].copy(), d["dst"].copy(), d["edge_counts"].copy()
    eptr = np.concatenate([[0], np.cumsum(ec)])
    g = int(rng.integers(0, ec.size))
    while ec[g] < 4:
        g = (g + 1) % ec.size
    k = int(eptr[g] + rng.integers(0, ec[g]))
    if kind == "self_loop":
        dst[k] = src[k]
    elif kind == "missing_reverse":                   # redirect one entry: (u, v) stays without (v, u) ... unless it lands on a neighbour
        u = src[k]
        nb = set(dst[eptr[g]:eptr[g + 1]][src[eptr[g]:eptr[g + 1]] == u].tolist())
        cand = [v for v in range(int(d["node_counts"][g])) if v != u and v not in nb]
        if not cand:
            return None
        dst[k] = cand[0]
    elif kind == "duplicate":                         # an entry listed twice (its reverse is listed once)
        src = np.insert(src, k, src[k]); dst = np.insert(dst, k, dst[k]); ec[g] += 1
        return dict(d, src=src, dst=dst, edge_counts=ec, edge_attr=np.insert(d["edge_attr"], k, d["edge_attr"][k]))
    return dict(d, src=src, dst=dst, edge_counts=ec)


@pytest.mark.parametrize("kind", ["self_loop", "missing_reverse", "duplicate"])
def test_csr_check_finds_what_the_host_check_finds(kind):
    rng = np.random.default_rng(5)
    d = gtok.synth.zinc_like(700, seed=31)
    clean, _ = both(d)
    r = gtok.ops.csr_check(clean.to(DEV))
    assert r == dict(violations=0, max_degree=clean.max_degree, max_nodes=clean.max_nodes, max_edges=clean.max_edges)
    for _ in range(6):
        dd = _damage(d, kind, rng)
        if dd is None:
            continue
        bad, _ = both(dd)
        assert not (bad.flags & gtok._lib.CSR_SIMPLE_SYMMETRIC)            # the host's numpy check
        assert gtok.ops.csr_check(bad.to(DEV))["violations"] > 0, kind


def test_csr_check_on_unsorted_rows_edge_cases_and_large_graphs():
    # rows in arbitrary order (from_coo only sorts by row): still simple and symmetric
    rng = np.random.default_rng(9)
    d = gtok.synth.zinc_like(400, seed=32)
    eptr = np.concatenate([[0], np.cumsum(d["edge_counts"])])
    src, dst, ea = d["src"].copy(), d["dst"].copy(), d["edge_attr"].copy()
    for g in range(400):
        p = rng.permutation(int(d["edge_counts"][g])) + eptr[g]
        src[eptr[g]:eptr[g + 1]], dst[eptr[g]:eptr[g + 1]], ea[eptr[g]:eptr[g + 1]] = src[p], dst[p], ea[p]
    b, _ = both(dict(d, src=src, dst=dst, edge_attr=ea))
    assert b.flags & gtok._lib.CSR_SIMPLE_SYMMETRIC
    assert gtok.ops.csr_check(b.to(DEV))["violations"] == 0
    # the hand-made edge cases hold self loops and duplicates; one direction only: never symmetric
    e, _ = both(edge_case_graphs())
    assert not e.flags and gtok.ops.csr_check(e.to(DEV))["violations"] > 0
    # graphs of up to 256 nodes (both directions stored)
    er = gtok.synth.er_batch(300, seed=2)
    sym = dict(node_counts=er["node_counts"], edge_counts=None)
    eptr = np.concatenate([[0], np.cumsum(er["edge_counts"])])
    s2, d2, c2 = [], [], []
    for g in range(300):
        u, v = er["src"][eptr[g]:eptr[g + 1]], er["dst"][eptr[g]:eptr[g + 1]]
        s2 += [u, v]; d2 += [v, u]; c2.append(2 * u.size)
    big, _ = both(dict(node_counts=er["node_counts"], edge_counts=np.array(c2), src=np.concatenate(s2), dst=np.concatenate(d2)), labeled=False)
    r = gtok.ops.csr_check(big.to(DEV))
    assert bool(big.flags & gtok._lib.CSR_SIMPLE_SYMMETRIC) == (r["violations"] == 0)
    assert (r["max_degree"], r["max_nodes"], r["max_edges"]) == (big.max_degree, big.max_nodes, big.max_edges)
    one, _ = both(er, labeled=False)                   # one direction per edge: not symmetric
    assert gtok.ops.csr_check(one.to(DEV))["violations"] > 0 and not one.flags
    # damaged structure: nothing is read through an unchecked offset
    bad = clean_copy = both(d)[0].to(DEV)
    rp = clean_copy.rowptr.clone(); rp[5] = 10 ** 6; rp[40] = -3
    bad = gtok.GraphBatch(bad.num_graphs, bad.max_nodes, bad.max_edges, bad.node_ptr, bad.edge_ptr, rp, bad.col, None, bad.nattr, bad.eattr)
    assert gtok.ops.csr_check(bad)["violations"] > 0
    c = clean_copy.col.clone(); c[17] = 99999; c[18] = -1
    bad = gtok.GraphBatch(bad.num_graphs, bad.max_nodes, bad.max_edges, bad.node_ptr, bad.edge_ptr, clean_copy.rowptr, c, None, bad.nattr, bad.eattr)
    assert gtok.ops.csr_check(bad)["violations"] > 0


def _tensors(host):
    t = lambda a: None if a is None else a.to(DEV)
    return dict(node_ptr=t(host.node_ptr), edge_ptr=t(host.edge_ptr), rowptr=t(host.rowptr), col=t(host.col), nattr=t(host.nattr), eattr=t(host.eattr))


def test_torch_op_reaches_the_benchmarked_kernel_on_a_zinc_full_shaped_corpus():
    """torch.ops.gtok.sent_epochs on ZINC-full-shaped input must run sent_lane_kernel - with the batch
    prepared explicitly (csr_prepare) and with raw tensors alone (prepared behind the op, once) - and equal oracle_sent."""
    G, K = 249456, 2
    d = gtok.synth.zinc_like(G, seed=1000)
    host, coo = both(d)
    T = _tensors(host)
    common = dict(query=None, max_num_nodes=host.max_nodes, max_len=1024, ld=176, seed=3, epoch=5, epochs=K, labeled=True, num_node_types=9,
                  num_edge_types=4, remap_zinc=True, pad_id=5, graph_base=0, pad=False, u16=True)
    # (a) raw tensors, flags unknown to the caller: the op verifies and prepares on the first call, finds it again on the second
    a1, l1 = torch.ops.gtok.sent_epochs(**T, max_nodes=host.max_nodes, max_edges=host.max_edges, **common)
    assert gtok.ops.last_sent_kernel() == "sent_lane_kernel"
    cached = len(gtok.torch_ops._BATCHES)
    a2, l2 = torch.ops.gtok.sent_epochs(**T, max_nodes=host.max_nodes, max_edges=host.max_edges, **common)
    assert len(gtok.torch_ops._BATCHES) == cached and gtok.ops.last_sent_kernel() == "sent_lane_kernel"
    # (b) prepared explicitly
    P = gtok.torch_ops.prepared_args(**T, max_nodes=host.max_nodes, max_edges=host.max_edges)
    assert P["layout"][2] > 0 and P["graph_ids"] is not None
    b1, m1 = torch.ops.gtok.sent_epochs(**P, **common)
    assert gtok.ops.last_sent_kernel() == "sent_lane_kernel"
    assert torch.equal(l1, l2) and torch.equal(l1, m1)
    S = 30000                                          # oracle on a slice of both epochs (the full-size oracle test is test_gpu_fullsize)
    sub = coo.slice(0, S)
    for e in range(K):
        ref, rln = orc.sent(sub, host.max_nodes, 1024, 3, 5 + e, ld=176, **ZKW)
        for ids, ln in ((a1, l1), (a2, l2), (b1, m1)):
            got_l = ln.view(K, G)[e, :S].cpu().numpy()
            assert np.array_equal(got_l, rln)
            got = (ids.view(K, G, 176)[e, :S].cpu().numpy().astype(np.int32) & 0xFFFF)
            inside = np.arange(176)[None, :] < rln[:, None]
            assert np.array_equal(np.where(inside, got, 0), np.where(inside, ref, 0))
    # whole-corpus agreement between the three routes, inside the row lengths
    inside = torch.arange(176, device=DEV)[None, :] < l1[:, None]
    assert torch.equal(torch.where(inside, a1, 0), torch.where(inside, b1, 0)) and torch.equal(torch.where(inside, a1, 0), torch.where(inside, a2, 0))
    # a tensor written in place is a different batch: the cache must not serve the old one
    T["col"].add_(0)
    torch.ops.gtok.sent_epochs(**T, max_nodes=host.max_nodes, max_edges=host.max_edges, **common)
    assert gtok.ops.last_sent_kernel() == "sent_lane_kernel"


def test_torch_op_sent_int32_padded_prepared_and_batches_the_lane_kernel_does_not_take():
    d = gtok.synth.zinc_like(30000, seed=77)
    host, coo = both(d)
    T = _tensors(host)
    P = gtok.torch_ops.prepared_args(**T, max_nodes=host.max_nodes, max_edges=host.max_edges)
    args = dict(query=None, max_num_nodes=host.max_nodes, max_len=1024, ld=176, seed=1, epoch=2, labeled=True, num_node_types=9,
                num_edge_types=4, remap_zinc=True, pad_id=5, graph_base=7)
    ids, ln = torch.ops.gtok.sent(**P, **args)
    assert gtok.ops.last_sent_kernel() == "sent_lane_kernel"
    ref, rln = orc.sent(coo, host.max_nodes, 1024, 1, 2, graph_base=7, ld=176, **ZKW)
    assert np.array_equal(ids.cpu().numpy(), ref) and np.array_equal(ln.cpu().numpy(), rln)
    # not simple / symmetric: csr_prepare hands back layout[2] == 0 and flags 0; the op still tokenizes (another kernel)
    e, ecoo = both(edge_case_graphs())
    TE = _tensors(e)
    PE = gtok.torch_ops.prepared_args(**TE, max_nodes=e.max_nodes, max_edges=e.max_edges)
    assert PE["layout"][0] == 0 and PE["layout"][2] == 0 and "graph_ids" not in PE
    kw = dict(query=None, max_num_nodes=8, max_len=1024, ld=128, seed=1, epoch=0, labeled=True, num_node_types=28, num_edge_types=5,
              remap_zinc=False, pad_id=5, graph_base=0)
    ids, ln = torch.ops.gtok.sent(**PE, **kw)
    ref, rln = orc.sent(ecoo, 8, 1024, 1, 0, labeled=True, num_node_types=28, num_edge_types=5, ld=128)
    assert np.array_equal(ids.cpu().numpy(), ref) and np.array_equal(ln.cpu().numpy(), rln)


def test_lane_kernel_through_bare_ctypes_without_ops_py():
    """A C caller's route (no GraphBatch, no ops.py): int32 CSR in HBM -> gtok_csr_lane_sort (check included) -> one
    32-byte read-back -> gtok_sent on the reordered batch; the rows are those of the batch in dataset order."""
    L, lb = gtok.lib(), gtok._lib
    G = 60000
    d = gtok.synth.zinc_like(G, seed=123)
    host, coo = both(d)
    dev = torch.device(DEV)
    T = _tensors(host)
    N, E = host.num_nodes_total, host.num_edges_total
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    z = lambda n, dt: torch.empty(n, dtype=dt, device=dev)
    o = dict(graph_ids=z(G, torch.int32), node_ptr=z(G + 1, torch.int32), edge_ptr=z(G + 1, torch.int64), rowptr=z(N + G, torch.int32), col=z(E, torch.int32),
             nattr=z(N, torch.uint8), eattr=z(E, torch.uint8), rowptr8=z(N + G + 16, torch.uint8), col8=z(E + 16, torch.uint8),
             unit_ptr=z(G + 1, torch.int32), unit_info=z(8 * G, torch.int32), info=z(8, torch.int32))
    src = lb.GtokCsr(G, host.max_nodes, host.max_edges, 0, T["node_ptr"].data_ptr(), T["edge_ptr"].data_ptr(), T["rowptr"].data_ptr(), T["col"].data_ptr(),
                     None, T["nattr"].data_ptr(), T["eattr"].data_ptr(), 0, 0, 0, 0, None, None, None, None, None, 0, 0, None, None, 0, 0, None)
    ws = z(int(L.gtok_csr_lane_sort_workspace(G)), torch.uint8)
    outs = lb.GtokCsrSorted(*[o[n].data_ptr() for n, _ in lb.GtokCsrSorted._fields_])
    assert L.gtok_csr_lane_sort(ctypes.byref(src), 0, 1, ctypes.byref(outs), ws.data_ptr(), ws.numel(), stream) == 0
    viol, maxdeg, maxn, maxe, units, chunk_n, chunk_e, _ = o["info"].tolist()
    assert viol == 0 and (maxdeg, maxn, maxe) == (host.max_degree, host.max_nodes, host.max_edges) and units > 0
    srt = lb.GtokCsr(G, host.max_nodes, host.max_edges, lb.CSR_SIMPLE_SYMMETRIC, o["node_ptr"].data_ptr(), o["edge_ptr"].data_ptr(), o["rowptr"].data_ptr(),
                     o["col"].data_ptr(), None, o["nattr"].data_ptr(), o["eattr"].data_ptr(), chunk_n, chunk_e, maxdeg, 0, o["rowptr8"].data_ptr(),
                     o["col8"].data_ptr(), None, None, None, 0, 0, o["graph_ids"].data_ptr(), o["unit_ptr"].data_ptr(), units, 0, o["unit_info"].data_ptr())
    p = lb.GtokSentParams(host.max_nodes, 1, 9, 4, 1024, 1, 5, 0, 42, 9, 1000, None, 1, 0)
    assert L.gtok_sent_kernel_name(ctypes.byref(srt), ctypes.byref(p)) == b"sent_lane_kernel"
    ids, ln = z((G, 176), torch.int32), z(G, torch.int32)
    assert L.gtok_sent(ctypes.byref(srt), ctypes.byref(p), ids.data_ptr(), 176, ln.data_ptr(), stream) == 0
    ref, rln = orc.sent(coo, host.max_nodes, 1024, 42, 9, graph_base=1000, ld=176, **ZKW)
    assert np.array_equal(ln.cpu().numpy(), rln) and np.array_equal(ids.cpu().numpy(), ref)
    # error codes of the new entry points
    assert L.gtok_csr_lane_sort(ctypes.byref(src), 0, 1, ctypes.byref(outs), ws.data_ptr(), 16, stream) == -1          # workspace too small
    big = lb.GtokCsr(G, 65, host.max_edges, 0, T["node_ptr"].data_ptr(), T["edge_ptr"].data_ptr(), T["rowptr"].data_ptr(), T["col"].data_ptr(),
                     None, None, None, 0, 0, 0, 0, None, None, None, None, None, 0, 0, None, None, 0, 0, None)
    assert L.gtok_csr_lane_sort(ctypes.byref(big), 0, 0, ctypes.byref(outs), ws.data_ptr(), ws.numel(), stream) == -2  # GTOK_E_TOO_LARGE
    assert L.gtok_csr_check(ctypes.byref(srt), o["info"].data_ptr(), stream) == -1                                      # a reordered batch
    assert L.gtok_csr_check(None, o["info"].data_ptr(), stream) == -1
