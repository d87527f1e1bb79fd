"""CPU, world_size 2 over gloo: block sharding + the all-gather that reassembles the padded slab, and the
shard-invariant RNG keying (global graph index).  The GPU box runs the same code over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import both, gtok, orc


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class _OracleRows:
    """rows_impl for dist.gather_tokens on CPU tensors: the oracle's numpy statement of the packed format (the product's
    implementation is the HIP kernels; the collective logic around them is what this test covers)."""

    @staticmethod
    def row_offsets(ln, ld, align=8):
        return torch.from_numpy(orc.row_offsets(ln.numpy(), ld, align))

    @staticmethod
    def pack_rows(ids, ln, row_ptr, elem_bytes=2, capacity=None, check_status=True):
        packed, ptr, st = orc.pack_rows(ids.numpy(), ln.numpy(), ids.shape[1], elem_bytes, capacity=capacity, fill=0x7ABC, with_status=True)
        if row_ptr is None:               # a caller-given capacity: offsets and packing are one pass (gtok_pack_rows_scan)
            row_ptr = torch.from_numpy(ptr)
        assert np.array_equal(ptr, row_ptr.numpy())
        if check_status and st:
            raise ValueError(f"pack status {st}")
        out = (torch.from_numpy(packed.view(np.int16) if elem_bytes == 2 else packed), row_ptr)
        return out if check_status else out + (torch.tensor([st], dtype=torch.int32),)

    pack_rows_u16 = pack_rows           # (the oracle's statement takes either slab)

    @staticmethod
    def unpack_rows(packed, row_ptr, ln, ld, pad_id, segment_rows=0, segment_stride=0, status=None, u16=False):
        p = packed.numpy()
        p = p.view(np.uint16) if p.dtype == np.int16 else p
        out, st = orc.unpack_rows(p, None if row_ptr is None else row_ptr.numpy(), ln.numpy(), ld, pad_id, segment_rows, segment_stride, with_status=True)
        if status is not None:
            status |= st
        return torch.from_numpy(out.astype(np.uint16).view(np.int16)) if u16 else torch.from_numpy(out)     # (gtok_unpack_rows_u16: a 16-bit slab)


    @staticmethod
    def unpack_rows_at(packed, row_start, ln, ld, pad_id, segment_rows=0, segment_stride=0, status=None, u16=False):
        """rows with explicit starts, relative to their rank's segment (gtok_unpack_rows_at): a plain numpy loop"""
        p = packed.numpy()
        p = p.view(np.uint16) if p.dtype == np.int16 else p
        st, n = row_start.numpy(), np.clip(ln.numpy(), 0, ld)
        out = np.full((n.size, ld), pad_id, dtype=np.int32)
        for r in range(n.size):
            seg = r // segment_rows if segment_rows else 0
            if st[r] < 0 or (segment_rows and st[r] + n[r] > segment_stride):
                if n[r] and status is not None:
                    status |= 2
                continue
            at = seg * segment_stride + st[r]
            out[r, :n[r]] = p[at:at + n[r]]
        return torch.from_numpy(out.astype(np.uint16).view(np.int16)) if u16 else torch.from_numpy(out)


class _Prepacked:
    """what ops.sent(..., packed=) leaves (ops.PackedRows), built by hand: rows in a scrambled order, each from an 8-id boundary,
    in two regions of the buffer"""

    def __init__(self, ids16, ln, capacity, seed, drop=()):
        rows, ld = ids16.shape
        self.capacity = capacity
        buf = np.full(capacity, 0x7ABC, dtype=np.uint16)
        self.row_start = torch.full((rows,), -1, dtype=torch.int64)
        fill = [0, capacity // 2]
        st = 0
        for i, r in enumerate(np.random.default_rng(seed).permutation(rows)):
            n = int(min(max(ln[r], 0), ld))
            reg = i & 1
            if r in drop or fill[reg] + n > (reg + 1) * (capacity // 2):
                st |= 2
                continue
            buf[fill[reg]:fill[reg] + n] = ids16[r, :n]
            self.row_start[r] = fill[reg]
            fill[reg] += (n + 7) // 8 * 8
        self.buf = torch.from_numpy(buf.view(np.int16))
        self._st = st

    def status(self):
        return torch.tensor([self._st], dtype=torch.int32)


def _worker(rank, world, port, G, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = gtok.synth.zinc_like(G, seed=77)
        _, coo = both(d)
        lo, hi = gtok.dist.block_bounds(G, world)[rank]
        kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
        # each rank tokenizes only its block (here with the oracle standing in for the kernel: no GPU in CI)
        max_nodes = gtok.dist.all_reduce_max_int(int(coo.node_counts[lo:hi].max()), "cpu")
        ids, ln = orc.sent(coo.slice(lo, hi), max_nodes, 1024, 11, 2, graph_base=lo, ld=160, **kw)
        full_ids, full_ln = gtok.dist.gather_tokens(torch.from_numpy(ids), torch.from_numpy(ln), G, 5)
        ref_ids, ref_ln = orc.sent(coo, int(coo.node_counts.max()), 1024, 11, 2, ld=160, **kw)
        ok = (max_nodes == int(coo.node_counts.max()) and tuple(full_ids.shape) == (G, 160)
              and np.array_equal(full_ids.numpy(), ref_ids) and np.array_equal(full_ln.numpy(), ref_ln))
        # the compact exchange (packed 16-bit rows + lengths, re-padded locally) gives the same slab, at both widths
        for eb in (2, 4):
            st = {}
            c_ids, c_ln = gtok.dist.gather_tokens(torch.from_numpy(ids), torch.from_numpy(ln), G, 5, compact=True,
                                                  elem_bytes=eb, rows_impl=_OracleRows, stats=st)
            ok = ok and torch.equal(c_ids, full_ids) and torch.equal(c_ln, full_ln) and st["compact"] \
                and st["bytes_sent_per_rank"] < ids.size * 4 * (0.4 if eb == 2 else 0.7)
            k_ids, k_ln = gtok.dist.gather_tokens(torch.from_numpy(ids), torch.from_numpy(ln), G, 5, compact=True, elem_bytes=eb,
                                                  capacity=st["capacity"] + 40, rows_impl=_OracleRows, stats=st)
            ok = ok and torch.equal(k_ids, full_ids) and int(st["status"]) == 0
            # a 16-bit slab (ops.sent(..., u16=True)) goes through both exchanges as it is
            i16 = torch.from_numpy(ids.astype(np.uint16).view(np.int16))
            u_ids, u_ln = gtok.dist.gather_tokens(i16, torch.from_numpy(ln), G, 5, compact=True, elem_bytes=eb, rows_impl=_OracleRows)
            p_ids, p_ln = gtok.dist.gather_tokens(i16, torch.from_numpy(ln), G, 5)
            ok = ok and u_ids.dtype == torch.int16 and p_ids.dtype == torch.int16 and torch.equal(u_ids, p_ids) and torch.equal(u_ln, full_ln) \
                and np.array_equal(p_ids.numpy().view(np.uint16).astype(np.int32), full_ids.numpy()) and torch.equal(p_ln, full_ln)
        # rows the walk has packed itself (ops.sent(..., packed=): completion order, explicit row starts) travel as they are
        i16n = ids.astype(np.uint16)
        cap = (int(orc.row_offsets(ref_ln[:-(-G // world)], 160)[-1]) + 64 * 8) // 16 * 16
        st = {}
        pp = _Prepacked(i16n, ln, cap, seed=rank)
        f_ids, f_ln = gtok.dist.gather_tokens(torch.from_numpy(i16n.view(np.int16)), torch.from_numpy(ln), G, 5, compact=True, packed=pp,
                                              rows_impl=_OracleRows, stats=st)
        ok = ok and f_ids.dtype == torch.int16 and torch.equal(f_ln, full_ln) and int(st["status"]) == 0 and st["prepacked"] \
            and np.array_equal(f_ids.numpy().view(np.uint16).astype(np.int32), full_ids.numpy())
        # ... or stay packed on the receiving side too (as_packed=True): absolute row starts into the gathered buffer
        (g_buf, g_start), g_ln = gtok.dist.gather_tokens(None, torch.from_numpy(ln), G, 5, compact=True, packed=pp, ld=160, as_packed=True,
                                                         rows_impl=_OracleRows, stats=st)
        back = _OracleRows.unpack_rows_at(g_buf, g_start, g_ln, 160, 5, u16=True)
        ok = ok and st["as_packed"] and torch.equal(g_ln, full_ln) and np.array_equal(back.numpy().view(np.uint16).astype(np.int32), full_ids.numpy())
        pp = _Prepacked(i16n, ln, cap, seed=rank, drop=(3,) if rank == 1 else ())      # one rank skipped a row: every rank sees the verdict
        d_ids, d_ln = gtok.dist.gather_tokens(torch.from_numpy(i16n.view(np.int16)), torch.from_numpy(ln), G, 5, compact=True, packed=pp,
                                              rows_impl=_OracleRows, stats=st)
        gone = gtok.dist.block_bounds(G, world)[1][0] + 3
        keep = np.arange(G) != gone
        ok = ok and int(st["status"]) == 2 and bool((d_ids[gone] == 5).all()) \
            and np.array_equal(d_ids.numpy().view(np.uint16).astype(np.int32)[keep], full_ids.numpy()[keep])
        (g_buf, g_start), g_ln = gtok.dist.gather_tokens(None, torch.from_numpy(ln), G, 5, compact=True, packed=pp, ld=160, as_packed=True,
                                                         rows_impl=_OracleRows, stats=st)
        ok = ok and int(st["status"]) == 2 and int(g_ln[gone]) == 0 and int(g_start[gone]) < 0 and torch.equal(g_ln[torch.from_numpy(keep)], full_ln[torch.from_numpy(keep)])
        # a caller-given capacity that turns out too small: no rank reads beyond a segment, every rank sees the same verdict,
        # the rows that did not fit come out as pad, the others are right
        st = {}
        small = int(orc.row_offsets(ref_ln[:-(-G // world)], 160)[-1]) // 2 // 8 * 8      # the same bound on every rank
        s_ids, s_ln = gtok.dist.gather_tokens(torch.from_numpy(ids), torch.from_numpy(ln), G, 5, compact=True, capacity=small,
                                              rows_impl=_OracleRows, stats=st)
        ok = ok and int(st["status"]) & 2 and torch.equal(s_ln, full_ln)
        same = (s_ids == full_ids).all(1) | (s_ids == 5).all(1)
        ok = ok and bool(same.all()) and bool((s_ids == full_ids).all(1).any()) and bool((s_ids != full_ids).any())
        # an id beyond 16 bits with the default elem_bytes=2 and no caller capacity: EVERY rank raises, after the collectives
        wide = torch.from_numpy(ids).clone()
        if rank == 1:
            wide[0, 0] = 70000
        try:
            gtok.dist.gather_tokens(wide, torch.from_numpy(ln), G, 5, compact=True, rows_impl=_OracleRows)
            ok = False
        except gtok.GtokError:
            pass
        # corpus-wide vocab statistics from per-rank tables (SUM / MIN all-reduce)
        s = gtok.synth.graph_token_like(G, seed=78, with_text=False)
        sc = orc.Coo(s["node_counts"], s["edge_counts"], s["src"], s["dst"])
        c, f = orc.vocab_stats_synth(sc.slice(lo, hi), 64, graph_base=lo)
        c, f = gtok.dist.reduce_vocab_stats(torch.from_numpy(c), torch.from_numpy(f))
        wc, wf = orc.vocab_stats_synth(sc, 64)
        ok = ok and np.array_equal(c.numpy(), wc) and np.array_equal(f.numpy(), wf)
        q.put((rank, bool(ok)))
    except Exception:                       # report instead of leaving the parent to time out
        import traceback
        traceback.print_exc()
        q.put((rank, False))
    finally:
        dist.destroy_process_group()


def test_block_bounds():
    assert gtok.dist.block_bounds(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert gtok.dist.block_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert gtok.dist.block_bounds(8, 1) == [(0, 8)]


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    G = 101                                       # odd: the last block is short and gets padded for the gather
    procs = [ctx.Process(target=_worker, args=(r, 2, port, G, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
