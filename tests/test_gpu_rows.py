"""GPU: the packed-row format (gtok_row_offsets / gtok_pack_rows / gtok_unpack_rows / gtok_collate_packed) against the
oracle's numpy statement of it, and the compact all-gather built on it (real one-rank RCCL group + the segmented layout
of a 4-rank gather rehearsed on one device)."""
import os

import numpy as np
import pytest
import torch

from _util import both, gtok, orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _slab(rows, ld, seed, max_id=300, long_rows=False):
    rng = np.random.default_rng(seed)
    ln = rng.integers(0, ld + 1, rows).astype(np.int32)
    if long_rows:                       # lengths beyond the slab width: the packed row holds the first ld ids
        ln[rng.integers(0, rows, max(1, rows // 7))] += ld
    ln[rng.integers(0, rows, max(1, rows // 9))] = 0
    ids = rng.integers(0, max_id, (rows, ld)).astype(np.int32)
    return ids, ln


@pytest.mark.parametrize("rows,ld", [(1, 16), (7, 4), (4097, 48), (10000, 208), (513, 1024), (300, 13)])
@pytest.mark.parametrize("align", [8, 1])
def test_row_offsets_pack_unpack_roundtrip(rows, ld, align):
    ids, ln = _slab(rows, ld, seed=rows + ld, long_rows=True)
    d_ids, d_ln = torch.from_numpy(ids).to(DEV), torch.from_numpy(ln).to(DEV)
    ptr = gtok.ops.row_offsets(d_ln, ld, align)
    assert np.array_equal(ptr.cpu().numpy(), orc.row_offsets(ln, ld, align))
    for eb in (2, 4):
        packed, p2 = gtok.ops.pack_rows(d_ids, d_ln, ptr, elem_bytes=eb)
        ref, rptr = orc.pack_rows(ids, ln, ld, eb, align)
        got = packed.cpu().numpy()
        got = got.view(np.uint16) if eb == 2 else got
        n = np.clip(ln, 0, ld)
        for r in range(rows):           # the slots between rows are not written: compare row by row
            assert np.array_equal(got[rptr[r]:rptr[r] + n[r]], ref[rptr[r]:rptr[r] + n[r]]), (r, eb)
        back = gtok.ops.unpack_rows(packed, ptr, d_ln, ld, pad_id=5)
        want = orc.unpack_rows(ref, rptr, ln, ld, 5)
        assert np.array_equal(back.cpu().numpy(), want)
        keep = np.arange(ld)[None, :] < n[:, None]
        assert np.array_equal(np.where(keep, ids, 5), want)          # = the slab with its tails re-padded


def test_pack_rows_flags_wide_ids_and_small_buffers():
    ids, ln = _slab(2000, 64, seed=3)
    ids[1234, 0] = 70000
    ln[1234] = max(ln[1234], 1)
    d_ids, d_ln = torch.from_numpy(ids).to(DEV), torch.from_numpy(ln).to(DEV)
    with pytest.raises(gtok.GtokError, match="16 bits"):
        gtok.ops.pack_rows(d_ids, d_ln, elem_bytes=2)
    packed, ptr = gtok.ops.pack_rows(d_ids, d_ln, elem_bytes=4)
    assert np.array_equal(gtok.ops.unpack_rows(packed, ptr, d_ln, 64, 0).cpu().numpy(),
                          orc.unpack_rows(*orc.pack_rows(ids, ln, 64, 4), ln, 64, 0))
    with pytest.raises(gtok.GtokError, match="capacity"):
        gtok.ops.pack_rows(d_ids, d_ln, elem_bytes=4, capacity=int(ptr[-1]) - 8)
    # empty inputs are no-ops
    e_ids, e_ln = torch.empty((0, 16), dtype=torch.int32, device=DEV), torch.empty(0, dtype=torch.int32, device=DEV)
    p, q = gtok.ops.pack_rows(e_ids, e_ln)
    assert q.tolist() == [0] and gtok.ops.unpack_rows(p, q, e_ln, 16, 5).shape == (0, 16)


@pytest.mark.parametrize("rows,ld", [(1, 16), (255, 24), (256, 24), (257, 24), (4097, 48), (100003, 176), (513, 1024), (300, 13), (70000, 8)])
@pytest.mark.parametrize("align", [8, 1])
def test_pack_rows_scan_is_row_offsets_plus_pack_in_one_pass(rows, ld, align):
    """gtok_pack_rows_scan (ABI v5): same row_ptr and same packed rows as gtok_row_offsets + gtok_pack_rows(_u16), from the
    int32 slab and from the 16-bit one, at every packed width; tile boundaries (256 rows), lengths outside [0, ld]."""
    ids, ln = _slab(rows, ld, seed=rows * 3 + ld, long_rows=True)
    ln[::11] = -3                                                   # negative lengths count as empty rows
    d_ids, d_ln = torch.from_numpy(ids).to(DEV), torch.from_numpy(ln).to(DEV)
    d_ids16 = d_ids.to(torch.int16)
    ptr = gtok.ops.row_offsets(d_ln, ld, align)
    want_ptr = orc.row_offsets(ln, ld, align)
    assert np.array_equal(ptr.cpu().numpy(), want_ptr)
    total, n = int(want_ptr[-1]), np.clip(ln, 0, ld)
    for src, ebs in ((d_ids, (2, 4)), (d_ids16, (2, 4, 8))):
        for eb in ebs:
            cap = total + 64
            packed, p2, st = gtok.ops.pack_rows_scan(src, d_ln, eb, cap, align)
            assert int(st.item()) == 0 and np.array_equal(p2.cpu().numpy(), want_ptr), (src.dtype, eb)
            two = (gtok.ops.pack_rows_u16 if src.dtype == torch.int16 else gtok.ops.pack_rows)(src, d_ln, ptr, elem_bytes=eb, capacity=cap, check_status=False)[0]
            a, b = packed.cpu().numpy(), two.cpu().numpy()
            a, b = (a.view(np.uint16), b.view(np.uint16)) if eb == 2 else (a, b)
            idx = np.concatenate([np.arange(want_ptr[r], want_ptr[r] + n[r]) for r in range(0, rows, max(1, rows // 4000))] + [np.zeros(0, np.int64)]).astype(np.int64)
            assert np.array_equal(a[idx], b[idx]), (src.dtype, eb)
            if rows <= 5000:                                        # every row, against the slab itself
                for r in range(rows):
                    assert np.array_equal(a[want_ptr[r]:want_ptr[r] + n[r]].astype(np.int64), ids[r, :n[r]]), (r, eb)
    # through the public wrappers: row_ptr=None + a capacity = the one-pass route
    packed, p3 = gtok.ops.pack_rows_u16(d_ids16, d_ln, None, elem_bytes=2, capacity=total + 8, align=align)
    assert np.array_equal(p3.cpu().numpy(), want_ptr)
    back = gtok.ops.unpack_rows(packed, p3, d_ln, ld, pad_id=5)
    assert np.array_equal(back.cpu().numpy(), np.where(np.arange(ld)[None, :] < n[:, None], ids, 5))


def test_pack_rows_scan_status_bits_and_repeated_launches():
    ids, ln = _slab(3000, 64, seed=3)
    ids[1234, 0] = 70000
    ln[1234] = max(ln[1234], 1)
    d_ids, d_ln = torch.from_numpy(ids).to(DEV), torch.from_numpy(ln).to(DEV)
    total = int(orc.row_offsets(ln, 64, 8)[-1])
    assert int(gtok.ops.pack_rows_scan(d_ids, d_ln, 2, total, 8)[2].item()) == 1          # an id beyond 16 bits
    assert int(gtok.ops.pack_rows_scan(d_ids, d_ln, 4, total, 8)[2].item()) == 0
    packed, ptr, st = gtok.ops.pack_rows_scan(d_ids, d_ln, 4, total - 8, 8)                # the last rows do not fit: skipped, flagged
    assert int(st.item()) == 2 and np.array_equal(ptr.cpu().numpy(), orc.row_offsets(ln, 64, 8))
    with pytest.raises(gtok.GtokError, match="capacity"):
        gtok.ops.pack_rows(d_ids, d_ln, None, elem_bytes=4, capacity=total - 8)
    # the tile ticket is re-armed by every launch: 300 launches back to back (beyond the ring of 256 counter blocks), two streams
    want = orc.row_offsets(ln, 64, 8)
    side = torch.cuda.Stream(device=DEV)
    outs = []
    for k in range(300):
        if k % 3 == 0:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                outs.append(gtok.ops.pack_rows_scan(d_ids, d_ln, 4, total, 8))
        else:
            outs.append(gtok.ops.pack_rows_scan(d_ids, d_ln, 4, total, 8))
    torch.cuda.synchronize()
    ref = outs[0][0].cpu().numpy()
    for packed, ptr, st in outs[::17]:
        assert np.array_equal(ptr.cpu().numpy(), want) and int(st.item()) == 0
    e_ids, e_ln = torch.empty((0, 16), dtype=torch.int32, device=DEV), torch.empty(0, dtype=torch.int32, device=DEV)
    p, q, st = gtok.ops.pack_rows_scan(e_ids, e_ln, 2, 0, 8)
    assert q.tolist() == [0] and int(st.item()) == 0


def test_collate_packed_equals_collate_on_the_slab():
    d = gtok.synth.zinc_like(3000, seed=12)
    batch, coo = both(d)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ids, ln = gtok.ops.sent(batch.to(DEV), 37, 1024, 4, 1, **kw)
    packed, ptr = gtok.ops.pack_rows(ids, ln)
    idx = torch.randperm(3000, generator=torch.Generator().manual_seed(1))[:128]
    out_ld = int(ln.cpu()[idx].max())
    X, A = gtok.ops.collate(ids, ln, idx, 5, out_ld)
    Xp, Ap = gtok.ops.collate_packed(packed, ptr, ln, ids.shape[1], idx, 5, out_ld)
    assert torch.equal(X, Xp) and torch.equal(A, Ap)
    ref, rln = orc.sent(coo, 37, 1024, 4, 1, ld=ids.shape[1], **kw)
    RX, RA, _ = orc.collate(ref, rln, idx.numpy(), 5, out_ld)
    assert np.array_equal(Xp.cpu().numpy(), RX) and np.array_equal(Ap.cpu().numpy(), RA)


def test_segmented_unpack_is_the_layout_of_a_four_rank_gather():
    """What dist.gather_tokens(compact=True) does on 4 ranks, on one device: every block packed into its own
    capacity-sized segment, segments concatenated, one row_offsets over all lengths, one segmented unpack."""
    ids, ln = _slab(1001, 96, seed=8)
    world, per = 4, -(-1001 // 4)
    d_ids, d_ln = torch.from_numpy(ids).to(DEV), torch.from_numpy(ln).to(DEV)
    parts, lens, totals = [], [], []
    for r in range(world):
        lo, hi = gtok.dist.block_bounds(1001, world)[r]
        bi, bl = d_ids[lo:hi].contiguous(), d_ln[lo:hi].contiguous()
        if hi - lo < per:
            bi = torch.cat([bi, torch.full((per - (hi - lo), 96), 5, dtype=torch.int32, device=DEV)])
            bl = torch.cat([bl, torch.zeros(per - (hi - lo), dtype=torch.int32, device=DEV)])
        parts.append((bi, bl)); lens.append(bl)
        totals.append(int(gtok.ops.row_offsets(bl, 96)[-1]))
    cap = -(-max(totals) // 8) * 8
    segs = [gtok.ops.pack_rows(bi, bl, capacity=cap)[0] for bi, bl in parts]
    all_packed, all_ln = torch.cat(segs), torch.cat(lens)
    out = gtok.ops.unpack_rows(all_packed, gtok.ops.row_offsets(all_ln, 96), all_ln, 96, 5, segment_rows=per, segment_stride=cap)
    keep = np.arange(96)[None, :] < ln[:, None]
    assert np.array_equal(out[:1001].cpu().numpy(), np.where(keep, ids, 5))


def test_compact_gather_over_rccl_equals_padded_gather():
    """dist.gather_tokens(compact=True) through a real one-rank `nccl` (RCCL) group: packed 16-bit rows + lengths over
    the collective, re-padded locally; equal to the padded exchange and to the oracle, at a fraction of the bytes."""
    import socket
    import torch.distributed as tdist
    d = gtok.synth.zinc_like(30001, seed=92)
    batch, coo = both(d)
    kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
    ld = gtok.ops.sent_safe_ld(batch, True, 1024)
    ids, ln = gtok.ops.sent(batch.to(DEV), 37, 1024, 21, 3, ld=ld, **kw)
    ref, rln = orc.sent(coo, 37, 1024, 21, 3, ld=ld, nthreads=8, **kw)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        sp, sc = {}, {}
        p_ids, p_ln = gtok.dist.gather_tokens(ids, ln, 30001, 5, force=True, stats=sp)
        c_ids, c_ln = gtok.dist.gather_tokens(ids, ln, 30001, 5, force=True, compact=True, stats=sc)
        assert c_ids.data_ptr() != ids.data_ptr()
        assert torch.equal(c_ids, p_ids) and torch.equal(c_ln, p_ln)
        assert np.array_equal(c_ids.cpu().numpy(), ref) and np.array_equal(c_ln.cpu().numpy(), rln)
        assert sc["bytes_sent_per_rank"] < 0.3 * sp["bytes_sent_per_rank"]
        # a caller-given capacity skips the size exchange
        k_ids, _ = gtok.dist.gather_tokens(ids, ln, 30001, 5, force=True, compact=True, capacity=sc["capacity"] + 64)
        assert torch.equal(k_ids, p_ids)
        # 16-bit rows in (two epochs in one launch), 16-bit slab out of both exchanges: packed in one pass, re-padded at 16 bits
        i16, l16 = gtok.ops.sent(batch.to(DEV), 37, 1024, 21, 3, ld=ld, u16=True, epochs=2, **kw)
        i16, l16 = i16.reshape(-1, ld), l16.reshape(-1)
        u_ids, u_ln = gtok.dist.gather_tokens(i16, l16, 2 * 30001, 5, force=True, compact=True, capacity=int(2.05 * sc["capacity"]))
        q_ids, q_ln = gtok.dist.gather_tokens(i16, l16, 2 * 30001, 5, force=True)
        assert u_ids.dtype == torch.int16 and torch.equal(u_ids, q_ids) and torch.equal(u_ln, q_ln) and u_ids.data_ptr() != i16.data_ptr()
        assert np.array_equal(u_ids[:30001].cpu().numpy().view(np.uint16).astype(np.int32), ref)
    finally:
        tdist.destroy_process_group()


def test_packed_rows_as_torch_custom_ops():
    ids, ln = _slab(500, 48, seed=21)
    d_ids, d_ln = torch.from_numpy(ids).to(DEV), torch.from_numpy(ln).to(DEV)
    ptr = torch.ops.gtok.row_offsets(d_ln, 48, 8)
    cap = int(ptr[-1]) + 16
    packed, st = torch.ops.gtok.pack_rows(d_ids, d_ln, ptr, 2, cap)
    assert int(st) == 0 and packed.dtype == torch.int16 and packed.numel() == cap
    back = torch.ops.gtok.unpack_rows(packed, ptr, d_ln, 48, 7, 0, 0)
    keep = np.arange(48)[None, :] < ln[:, None]
    assert np.array_equal(back.cpu().numpy(), np.where(keep, ids, 7))
    idx = torch.arange(0, 500, 7, device=DEV)
    X, A = torch.ops.gtok.collate_packed(packed, ptr, d_ln, 48, idx, 7, 48)
    assert torch.equal(X.to(torch.int32), back[idx]) and torch.equal(A, torch.from_numpy(keep).to(DEV)[idx])


def test_ids_to_text_against_the_python_statement_of_the_format():
    """gtok_ids_to_text (two passes: lengths, then bytes) on its own: empty rows, take of 0 / 1 / the whole row / beyond it,
    ids outside the string table (empty strings, their separators stay), empty strings inside the table, with and without
    suffixes (also empty ones), one row and thousands.  The ZINC strings as a whole are checked against the reference's golden
    texts in test_gpu_boundary.py; this pins the format."""
    rng = np.random.default_rng(17)
    strings = ["<bos>", "", "C", "Cl", "aromatic", "x" * 37, "7"] + [str(i) for i in range(40)]
    for rows, ld in ((1, 1), (1, 9), (5, 16), (3001, 70), (64, 257)):
        ids = rng.integers(-2, len(strings) + 3, (rows, ld)).astype(np.int32)
        take = rng.integers(0, ld + 1, rows).astype(np.int32)
        take[rng.integers(0, rows, max(1, rows // 5))] = 0
        take[rng.integers(0, rows, max(1, rows // 7))] = ld
        if rows > 2:
            take[1] = ld + 5                                   # clamped to the row
        suf = [b"" if r % 3 == 0 else (b" val_%d_%02d <eos>" % (r, r % 100)) for r in range(rows)]
        d_ids, d_take = torch.from_numpy(ids).to(DEV), torch.from_numpy(take).to(DEV)
        for sfx in (suf, None):
            blob, ptr = gtok.ops.ids_to_text(d_ids, d_take, strings, sfx)
            want = orc.ids_to_text(ids, take, strings, sfx)
            p, b = ptr.cpu().numpy(), blob.cpu().numpy().tobytes()
            assert p[0] == 0 and p[-1] == len(b) == sum(map(len, want))
            got = [b[p[r]:p[r + 1]] for r in range(rows)]
            assert got == want, next((r, got[r], want[r]) for r in range(rows) if got[r] != want[r])
    e_ids = torch.empty((0, 8), dtype=torch.int32, device=DEV)
    blob, ptr = gtok.ops.ids_to_text(e_ids, torch.empty(0, dtype=torch.int32, device=DEV), strings, [])
    assert blob.numel() == 0 and ptr.tolist() == [0]


def test_zinc_text_tails_format_labels_as_python_does():
    """gtok_zinc_text_tails against the reference's own expression (zinc_dataset_indexbase.py:192, f"val_{label:.2f}" with
    '.' -> '_' and '-' -> 'neg', restated in oracle.zinc_text_tails): "%.2f" of the float's exact value with ties to even.
    Random bit patterns of every exponent (denormals, huge values printed in full, nan / inf), exact ties (k / 8: 0.125 ->
    0.12, 0.375 -> 0.38), values next to a hundredth, signed zeros and tiny negatives (val_neg0_00), ZINC-like labels; the cut
    rule (:217-221) at max_len 1 / 2 / around the row lengths; then ids_to_text with the device tails == with host bytes."""
    rng = np.random.default_rng(23)
    bits = rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    ties = (rng.integers(-40000, 40000, 50000) / 8.0).astype(np.float32)
    near = (rng.integers(-10 ** 6, 10 ** 6, 50000) / 100.0).astype(np.float32)
    near = np.concatenate([near, np.nextafter(near, np.float32(np.inf)), np.nextafter(near, np.float32(-np.inf))])
    halves = ((rng.integers(-10 ** 5, 10 ** 5, 50000) * 2 + 1) / 200.0).astype(np.float32)       # x.xx5 as float32 sees it
    special = np.array([0.0, -0.0, -0.001, 0.004999, 0.005, 0.015, 0.125, 0.375, 2.675, -2.675, 1e-45, -1e-45, 1e10, -1e20, 3.4028235e38,
                        -3.4028235e38, np.inf, -np.inf, np.nan, 8388608.0, 16777216.0, 16777215.0, 8388607.5, 4194303.75, 99.995,
                        999999.995, 0.994999, 0.995, 1.005, 4.23, -2.1], np.float32)
    zinc = rng.normal(0.0, 2.0, 100000).astype(np.float32)
    y = np.concatenate([bits, ties, near, halves, special, zinc])
    G = y.size
    ln = rng.integers(1, 120, G).astype(np.int32)
    d_y, d_ln = torch.from_numpy(y).to(DEV), torch.from_numpy(ln).to(DEV)
    for max_len in (1 << 30, 60, 2, 1):
        take, sb, sp = gtok.ops.zinc_text_tails(d_y, d_ln, max_len)
        want_take, want = orc.zinc_text_tails(y, ln, max_len)
        assert np.array_equal(take.cpu().numpy(), want_take)
        p, b = sp.cpu().numpy(), sb.cpu().numpy().tobytes()
        assert p[0] == 0 and p[-1] == sum(map(len, want))
        got = [b[p[r]:p[r + 1]] for r in range(G)]
        assert got == want, next((r, y[r], got[r], want[r]) for r in range(G) if got[r] != want[r])
    strings = ["<bos>", "C", "single", "<p>"] + [str(i) for i in range(40)]
    ids = rng.integers(0, len(strings), (4000, 120)).astype(np.int32)
    d_ids = torch.from_numpy(ids).to(DEV)
    for max_len in (1 << 30, 60):
        take, sb, sp = gtok.ops.zinc_text_tails(d_y[:4000], d_ln[:4000], max_len)
        blob, ptr = gtok.ops.ids_to_text(d_ids, take, strings, (sb, sp))
        want_take, tails = orc.zinc_text_tails(y[:4000], ln[:4000], max_len)
        blob2, ptr2 = gtok.ops.ids_to_text(d_ids, take, strings, tails)
        assert torch.equal(blob, blob2) and torch.equal(ptr, ptr2)
        want = orc.ids_to_text(ids, want_take, strings, tails)
        b, p = blob.cpu().numpy().tobytes(), ptr.cpu().numpy()
        assert [b[p[r]:p[r + 1]] for r in range(4000)] == want
    with pytest.raises(gtok.GtokError):
        gtok.ops.zinc_text_tails(d_y, d_ln, 0)


def test_c_abi_error_codes_of_the_round_5_row_entry_points():
    """gtok_pack_rows_scan / gtok_unpack_rows_u16 / gtok_collate_batch / gtok_collate_epoch*: bad arguments come back as GTOK_E_*
    codes before anything is launched; a host index outside the slab never reaches the kernel."""
    import ctypes
    L = gtok.lib()
    ids, ln = _slab(300, 32, seed=5)
    d_ids, d_ln = torch.from_numpy(ids).to(DEV), torch.from_numpy(ln).to(DEV)
    d16 = d_ids.to(torch.int16)
    z = lambda n, dt: torch.empty(n, dtype=dt, device=DEV)
    packed, ptr, st = z(300 * 32, torch.int16), z(301, torch.int64), z(1, torch.int32)
    ok = lambda *a: L.gtok_pack_rows_scan(*a, None)
    assert ok(d16.data_ptr(), 2, 32, d_ln.data_ptr(), 300, 8, 2, packed.data_ptr(), 300 * 32, ptr.data_ptr(), st.data_ptr()) == 0
    assert ok(d16.data_ptr(), 3, 32, d_ln.data_ptr(), 300, 8, 2, packed.data_ptr(), 300 * 32, ptr.data_ptr(), st.data_ptr()) == -1    # source width
    assert ok(d16.data_ptr(), 2, 32, d_ln.data_ptr(), 300, 6, 2, packed.data_ptr(), 300 * 32, ptr.data_ptr(), st.data_ptr()) == -1    # align not a power of two
    assert ok(d_ids.data_ptr(), 4, 32, d_ln.data_ptr(), 300, 8, 8, packed.data_ptr(), 300 * 32, ptr.data_ptr(), st.data_ptr()) == -1  # int64 only from 16-bit rows
    assert ok(d16.data_ptr(), 2, 32, d_ln.data_ptr(), 300, 8, 2, packed.data_ptr(), 300 * 32, None, st.data_ptr()) == -1
    assert ok(None, 2, 32, d_ln.data_ptr(), 300, 8, 2, packed.data_ptr(), 300 * 32, ptr.data_ptr(), st.data_ptr()) == -1
    out16 = z((300, 32), torch.int16)
    un = lambda pad: L.gtok_unpack_rows_u16(packed.data_ptr(), 2, ptr.data_ptr(), d_ln.data_ptr(), 300, 0, 0, 300 * 32, pad, out16.data_ptr(), 32, None, None)
    assert un(5) == 0 and un(70000) == -1 and un(-1) == -1
    assert np.array_equal(out16.cpu().numpy().view(np.uint16).astype(np.int32), np.where(np.arange(32)[None, :] < np.clip(ln, 0, 32)[:, None], ids, 5))
    X, A = z((4, 32), torch.int64), z((4, 32), torch.bool)
    idx = (ctypes.c_int64 * 4)(0, 299, 7, 7)
    cb = lambda ix, rows: L.gtok_collate_batch(d16.data_ptr(), 2, None, d_ln.data_ptr(), 32, ix, 4, rows, 5, X.data_ptr(), A.data_ptr(), 32, None, 0, None, None)
    assert cb(idx, 300) == 0
    bad = (ctypes.c_int64 * 4)(0, 300, 7, 7)
    assert cb(bad, 300) == -1 and cb((ctypes.c_int64 * 4)(0, -1, 7, 7), 300) == -1 and cb(None, 300) == -1
    lmax, off = z(3, torch.int32), z(4, torch.int64)
    order = torch.arange(300, device=DEV)
    assert L.gtok_collate_epoch_plan(d_ln.data_ptr(), 32, order.data_ptr(), 300, 128, lmax.data_ptr(), off.data_ptr(), None) == 0
    assert L.gtok_collate_epoch_plan(d_ln.data_ptr(), 32, order.data_ptr(), 300, 0, lmax.data_ptr(), off.data_ptr(), None) == -1
    assert L.gtok_collate_epoch_plan(d_ln.data_ptr(), 32, order.data_ptr(), 300, 128, lmax.data_ptr(), None, None) == -1
    total = int(off[-1])
    Xa, Aa = z(total, torch.int64), z(total, torch.bool)
    ce = lambda eb, arena: L.gtok_collate_epoch(d16.data_ptr(), eb, None, d_ln.data_ptr(), 32, order.data_ptr(), 300, 128, 5, lmax.data_ptr(), off.data_ptr(),
                                                Xa.data_ptr(), Aa.data_ptr(), arena, None)
    assert ce(2, total) == 0 and ce(3, total) == -1 and ce(2, -1) == -1
    # an arena smaller than the plan says: rows that would end beyond it are skipped, nothing is written past it
    Xa.fill_(-7)
    assert ce(2, total // 2) == 0
    torch.cuda.synchronize()
    assert bool((Xa[total // 2:] == -7).all())


def test_pack_rows_scan_with_several_tiles_per_ticket_and_large_host_batches(monkeypatch):
    """The chunked-ticket path of gtok_pack_rows_scan (several tiles per workgroup: what slabs beyond ~500 k rows get) forced on a
    small slab, and gtok_collate_batch with more indices than one launch's arguments hold (512)."""
    ids, ln = _slab(5000, 40, seed=77, long_rows=True)
    d_ln = torch.from_numpy(ln).to(DEV)
    d16 = torch.from_numpy(ids).to(DEV).to(torch.int16)
    want_ptr = orc.row_offsets(ln, 40, 8)
    n = np.clip(ln, 0, 40)
    for chunk in ("3", "8"):
        monkeypatch.setenv("GTOK_PACK_CHUNK", chunk)
        for tile in ("64", "256"):
            monkeypatch.setenv("GTOK_PACK_TILE", tile)
            packed, ptr, st = gtok.ops.pack_rows_scan(d16, d_ln, 2, int(want_ptr[-1]) + 8)
            assert int(st.item()) == 0 and np.array_equal(ptr.cpu().numpy(), want_ptr), (chunk, tile)
            got = packed.cpu().numpy().view(np.uint16)
            for r in range(0, 5000, 7):
                assert np.array_equal(got[want_ptr[r]:want_ptr[r] + n[r]].astype(np.int64), ids[r, :n[r]]), (chunk, tile, r)
    monkeypatch.delenv("GTOK_PACK_CHUNK"); monkeypatch.delenv("GTOK_PACK_TILE")
    idx = np.random.default_rng(1).integers(0, 5000, 1300).astype(np.int64)
    y = torch.arange(5000, dtype=torch.float32, device=DEV) * 0.5
    X, A, Y = gtok.ops.collate_batch(d16, None, d_ln, 40, idx, 5, 40, y)
    Xr, Ar = gtok.ops.collate_packed(d16, None, d_ln, 40, torch.from_numpy(idx).to(DEV), 5, 40)
    assert torch.equal(X, Xr) and torch.equal(A, Ar) and torch.equal(Y, y[torch.from_numpy(idx).to(DEV)])
    yl = torch.arange(5000, dtype=torch.int64, device=DEV) * 3
    assert torch.equal(gtok.ops.collate_batch(d16, None, d_ln, 40, idx[:5], 5, 16, yl)[2], yl[torch.from_numpy(idx[:5]).to(DEV)])
