"""numpy/ctypes front end of the CPU oracle (oracle/gtok_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package never imports this module.
Inputs are PyG-style batched COO (what the reference's Python iterates over).
SENT parity vs upstream AutoGraph is UNPINNED (see the C file's header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgtok_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gtok_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libgtok_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.int32)


def _i64(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.int64)


class Coo:
    """Batched COO: per-graph node/edge counts + concatenated local-id edge lists and raw int attrs."""

    def __init__(self, node_counts, edge_counts, src, dst, x=None, edge_attr=None):
        self.node_counts = _i64(node_counts)
        self.edge_counts = _i64(edge_counts)
        self.G = int(self.node_counts.size)
        self.node_ptr = np.zeros(self.G + 1, np.int32); self.node_ptr[1:] = np.cumsum(self.node_counts)
        self.edge_ptr = np.zeros(self.G + 1, np.int64); self.edge_ptr[1:] = np.cumsum(self.edge_counts)
        self.src, self.dst = _i32(src), _i32(dst)
        self.x, self.edge_attr = _i32(x), _i32(edge_attr)

    def slice(self, lo, hi):
        n0, n1, e0, e1 = self.node_ptr[lo], self.node_ptr[hi], self.edge_ptr[lo], self.edge_ptr[hi]
        s = lambda a, i, j: None if a is None else a[i:j]
        return Coo(self.node_counts[lo:hi], self.edge_counts[lo:hi], self.src[e0:e1], self.dst[e0:e1],
                   s(self.x, n0, n1), s(self.edge_attr, e0, e1))


def ibtt_zinc(coo: Coo, lut, max_len, pad_id, ld, nthreads=1):
    lut = _i32(lut)
    out = np.empty((coo.G, ld), np.int32); ln = np.empty(coo.G, np.int32)
    x = coo.x if coo.x is not None else np.full(int(coo.node_ptr[-1]), 255, np.int32)
    lib().oracle_ibtt_zinc(ctypes.c_int32(coo.G), _p(coo.node_ptr), _p(coo.edge_ptr), _p(x), _p(coo.src), _p(coo.dst),
                           _p(coo.edge_attr), _p(lut), ctypes.c_int32(lut.size), ctypes.c_int32(max_len),
                           ctypes.c_int32(pad_id), _p(out), ctypes.c_int32(ld), _p(ln), ctypes.c_int32(nthreads))
    return out, ln


def ibtt_synth(coo: Coo, lut, query, max_len, pad_id, ld, nthreads=1):
    lut = _i32(lut)
    q = None if query is None else np.ascontiguousarray(query, dtype=np.int32)
    out = np.empty((coo.G, ld), np.int32); ln = np.empty(coo.G, np.int32)
    lib().oracle_ibtt_synth(ctypes.c_int32(coo.G), _p(coo.node_ptr), _p(coo.edge_ptr), _p(coo.src), _p(coo.dst),
                            _p(lut), ctypes.c_int32(lut.size), _p(q), ctypes.c_int32(max_len), ctypes.c_int32(pad_id),
                            _p(out), ctypes.c_int32(ld), _p(ln), ctypes.c_int32(nthreads))
    return out, ln


def text_to_ids(texts, vocab, max_len, ld, strip_label=True, nthreads=1):
    enc = [t.encode("utf-8") for t in texts]
    ptr = np.zeros(len(enc) + 1, np.int64); ptr[1:] = np.cumsum([len(b) for b in enc])
    blob = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy()
    keys = [k.encode("utf-8") for k in vocab]
    voff = np.zeros(len(keys) + 1, np.int32); voff[1:] = np.cumsum([len(k) for k in keys])
    vbytes = np.frombuffer(b"".join(keys) + b"\0", dtype=np.uint8).copy()
    vid = np.asarray(list(vocab.values()), np.int32)
    out = np.empty((len(enc), ld), np.int32); ln = np.empty(len(enc), np.int32)
    lib().oracle_text_to_ids(_p(blob), _p(ptr), ctypes.c_int32(len(enc)), _p(vbytes), _p(voff), _p(vid),
                             ctypes.c_int32(len(keys)), ctypes.c_int32(vocab["<pad>"]), ctypes.c_int32(int(strip_label)),
                             ctypes.c_int32(max_len), _p(out), ctypes.c_int32(ld), _p(ln), ctypes.c_int32(nthreads))
    return out, ln


def collate(ids, ln, index, pad_id, out_ld):
    ids = np.ascontiguousarray(ids, np.int32); ln = _i32(ln); index = _i64(index)
    B = index.size
    X = np.empty((B, out_ld), np.int64); A = np.empty((B, out_ld), np.uint8); m = np.zeros(1, np.int32)
    lib().oracle_collate(_p(ids), ctypes.c_int32(ids.shape[1]), _p(ln), _p(index), ctypes.c_int32(B),
                         ctypes.c_int32(pad_id), _p(X), _p(A), ctypes.c_int32(out_ld), _p(m))
    return X, A.astype(bool), int(m[0])


def parse_graph_text(text, task_query=True):
    """One graph-token text -> (edges, num_nodes, query or None, label or None), restating
    graph_data_loader/graph_token_dataset_autograph.py: parse_graph_from_text :14-55 (greedy `int int <e>` scan up to the
    first <n>, then the integer list), parse_query_nodes_from_text :58-78, parse_label_from_text :80-113 and the
    num_nodes rule of parse_graph_from_json :116-158.  Pure Python: small cases only."""
    toks = text.split()
    edges, nodes = [], []
    i = 0
    while i < len(toks):
        if i + 2 < len(toks) and toks[i + 2] == "<e>":
            try:
                edges.append((int(toks[i]), int(toks[i + 1])))
                i += 3
            except ValueError:
                i += 1
        elif toks[i] == "<n>" and i + 1 < len(toks):
            i += 1
            while i < len(toks) and toks[i] not in ("<q>", "<p>", "<eos>"):
                try:
                    nodes.append(int(toks[i]))
                    i += 1
                except ValueError:
                    break
            break
        else:
            i += 1
    query = None
    for i, tok in enumerate(toks):
        if tok == "<q>" and i + 3 < len(toks) and toks[i + 1] == "shortest_distance":
            try:
                query = (int(toks[i + 2]), int(toks[i + 3]))
                break
            except ValueError:
                pass
    label = None
    for i in range(len(toks) - 1):
        if toks[i] != "<p>":
            continue
        lab = toks[i + 1].upper()
        if lab in ("YES", "NO"):
            label = int(lab == "YES")
            break
        if lab.startswith("LEN"):
            try:
                label = int(lab[3:]) - 1
                break
            except ValueError:
                pass
        if lab in ("INF", "INFINITY"):
            break
    num_nodes = max(nodes) + 1 if nodes else (max(max(a, b) for a, b in edges) + 1 if edges else 0)
    return edges, num_nodes, query, label


def find_token(x, token):
    """trainer/train_ibtt.py:88-97: `(x[b] == q).nonzero()[0]` per sample, -1 when the token is absent."""
    x = np.asarray(x)
    out = np.full(x.shape[0], -1, np.int32)
    for b in range(x.shape[0]):
        hits = np.nonzero(x[b] == token)[0]
        if hits.size:
            out[b] = hits[0]
    return out


def vocab_stats_synth(coo: Coo, num_ids, query_nodes=None, graph_base=0):
    """(count, first) int64 [num_ids]: what build_vocab_from_texts would see for the tokens str(0..num_ids-1)."""
    count = np.zeros(num_ids, np.int64); first = np.full(num_ids, np.iinfo(np.int64).max, np.int64)
    q = None if query_nodes is None else np.ascontiguousarray(query_nodes, dtype=np.int32)
    lib().oracle_vocab_stats_synth(ctypes.c_int32(coo.G), _p(coo.node_ptr), _p(coo.edge_ptr), _p(coo.src), _p(coo.dst),
                                   _p(q), ctypes.c_int64(graph_base), ctypes.c_int32(num_ids), _p(count), _p(first))
    return count, first


def vocab_stats_text(texts):
    """[(token, count, first byte offset in the concatenated texts)] in Counter.most_common order - the corpus pass
    of build_vocab_from_texts (data_loader.py:451-463: Counter.update(t.split()) text by text; most_common = count
    descending, ties in first-insertion order).  Pure Python: small cases only."""
    import re
    from collections import Counter
    counts, first, base = Counter(), {}, 0
    for t in texts:
        toks = t.split()
        counts.update(toks)
        # byte offsets of the tokens (ASCII texts: str.split() separators = the bytes py_isspace accepts)
        pos = [m.start() for m in re.finditer(r"[^ \t\n\r\x0b\x0c\x1c\x1d\x1e\x1f]+", t)]
        assert len(pos) == len(toks)
        for tok, p in zip(toks, pos):
            first.setdefault(tok, base + p)
        base += len(t.encode())
    return [(tok, c, first[tok]) for tok, c in counts.most_common()]


def remap_zinc(ids, ln, idx_off, node_off, edge_off):
    ids = np.ascontiguousarray(ids, np.int32); ln = _i32(ln)
    out = np.empty_like(ids)
    lib().oracle_remap_zinc(_p(ids), _p(out), ctypes.c_int32(ids.shape[1]), _p(ln), ctypes.c_int32(ids.shape[0]),
                            ctypes.c_int32(idx_off), ctypes.c_int32(node_off), ctypes.c_int32(edge_off))
    return out


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, np.uint32); k = np.asarray(key, np.uint32); o = np.zeros(4, np.uint32)
    lib().oracle_philox4x32_10(_p(c), _p(k), _p(o))
    return o


def sent(coo: Coo, max_num_nodes, max_len, seed, epoch=0, labeled=False, num_node_types=0, num_edge_types=0,
         remap_zinc=False, pad_id=5, graph_base=0, query=None, ld=None, nthreads=1):
    if ld is None:
        ld = max_len + (3 if query is not None else 0)
    q = None if query is None else np.ascontiguousarray(query, dtype=np.int32)
    out = np.empty((coo.G, ld), np.int32); ln = np.empty(coo.G, np.int32)
    lib().oracle_sent(ctypes.c_int32(coo.G), _p(coo.node_ptr), _p(coo.edge_ptr), _p(coo.x), _p(coo.src), _p(coo.dst),
                      _p(coo.edge_attr), ctypes.c_int32(int(labeled)), ctypes.c_int32(max_num_nodes),
                      ctypes.c_int32(num_node_types), ctypes.c_int32(num_edge_types), ctypes.c_int32(max_len),
                      ctypes.c_int32(int(remap_zinc)), ctypes.c_int32(pad_id), ctypes.c_uint64(seed & (2 ** 64 - 1)),
                      ctypes.c_uint64(epoch & (2 ** 64 - 1)), ctypes.c_int64(graph_base), _p(q), _p(out),
                      ctypes.c_int32(ld), _p(ln), ctypes.c_int32(nthreads))
    return out, ln


def sent_decode_rows(ids, ln, max_num_nodes, labeled=False, num_node_types=0, edge_cap=None, node_cap=None):
    """C decoder with gtok_sent_decode's outputs (dict of numpy arrays); untouched slots of the [G, cap] arrays are -9."""
    ids = np.ascontiguousarray(ids, np.int32); ln = _i32(ln)
    G, ld = ids.shape
    ecap = ld if edge_cap is None else int(edge_cap)
    ncap = max(1, max_num_nodes) if node_cap is None else int(node_cap)
    out = dict(num_nodes=np.empty(G, np.int32), num_edges=np.empty(G, np.int32), status=np.empty(G, np.int32),
               edge_a=np.full((G, ecap), -9, np.int32), edge_b=np.full((G, ecap), -9, np.int32),
               edge_type=np.full((G, ecap), -9, np.int32), node_type=np.full((G, ncap), -9, np.int32))
    lib().oracle_sent_decode(_p(ids), ctypes.c_int32(ld), _p(ln), ctypes.c_int32(G), ctypes.c_int32(max_num_nodes),
                             ctypes.c_int32(int(labeled)), ctypes.c_int32(num_node_types), _p(out["num_nodes"]),
                             _p(out["num_edges"]), _p(out["edge_a"]), _p(out["edge_b"]), _p(out["edge_type"]),
                             ctypes.c_int32(ecap), _p(out["node_type"]), ctypes.c_int32(ncap), _p(out["status"]))
    return out


def sent_roundtrip(coo: Coo, ids, ln, max_num_nodes, max_len, seed, epoch=0, labeled=False, num_node_types=0,
                   graph_base=0, nthreads=1):
    """status int32 [G]: 0 = the (un-remapped) row decodes exactly to the input graph (a row cut at max_len: to a
    part of it); see oracle_sent_roundtrip in the C file for the other codes."""
    ids = np.ascontiguousarray(ids, np.int32); ln = _i32(ln)
    status = np.empty(coo.G, np.int32)
    lib().oracle_sent_roundtrip(ctypes.c_int32(coo.G), _p(coo.node_ptr), _p(coo.edge_ptr), _p(coo.x), _p(coo.src),
                                _p(coo.dst), _p(coo.edge_attr), ctypes.c_int32(int(labeled)), ctypes.c_int32(max_num_nodes),
                                ctypes.c_int32(num_node_types), ctypes.c_int32(max_len),
                                ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.c_uint64(epoch & (2 ** 64 - 1)),
                                ctypes.c_int64(graph_base), _p(ids), ctypes.c_int32(ids.shape[1]), _p(ln), _p(status),
                                ctypes.c_int32(nthreads))
    return status


# ------------------------------------------------------------------------------------------------
# SENT decoder (pure Python, small cases): token stream -> graph, used for the losslessness and
# reference-visible-invariant property tests.
# ------------------------------------------------------------------------------------------------
def sent_decode(tokens, max_num_nodes, labeled=False, num_node_types=0):
    """Returns (num_nodes, set of frozenset edges, node_types dict by visit idx, edge_types dict)."""
    idx_off, node_off = 6, 6 + max_num_nodes
    edge_off = node_off + num_node_types
    toks = list(tokens)
    assert toks[0] == 0, "must start with SOS"
    i, prev, nseen = 1, None, 0
    edges, ntypes, etypes = set(), {}, {}
    pending_et = None

    def is_pos(t):
        return idx_off <= t < node_off

    while i < len(toks):
        t = toks[i]
        if t == 4:  # EOS
            i += 1
            break
        if t == 1:  # RESET
            prev, pending_et = None, None
            i += 1
            continue
        if t == 2:  # LADJ ... RADJ: edges from the current node to earlier nodes
            i += 1
            while toks[i] != 3:
                et = None
                if labeled:
                    et = toks[i] - edge_off; i += 1
                a = toks[i] - idx_off; i += 1
                e = frozenset((prev, a))
                assert e not in edges, "edge encoded twice"
                edges.add(e)
                if labeled:
                    etypes[e] = et
            i += 1
            continue
        if labeled and not is_pos(t):  # edge type preceding a trail step
            pending_et = t - edge_off
            i += 1
            continue
        assert is_pos(t), f"unexpected token {t}"
        k = t - idx_off
        i += 1
        if k == nseen:  # first visit
            nseen += 1
            if labeled:
                ntypes[k] = toks[i] - node_off; i += 1
        else:
            assert k < nseen
        if prev is not None:
            e = frozenset((prev, k))
            assert e not in edges, "edge encoded twice"
            edges.add(e)
            if labeled:
                etypes[e] = pending_et
        pending_et = None
        prev = k
    return nseen, edges, ntypes, etypes, i


# ---- rows of ids -> texts (include/gtok.h, gtok_ids_to_text): the format itself, in Python ------------------------------
def ids_to_text(ids, take, strings, suffixes=None):
    """Row r = the strings of its first take[r] ids (ids outside the table: empty strings) joined by single spaces, then the
    row's suffix bytes verbatim - what zinc_dataset_indexbase.py:143-227 builds with ' '.join(tokens) once the tokens are ids
    of a string table.  Returns a list of bytes objects."""
    ids = np.asarray(ids, np.int32)
    enc = [t.encode("ascii") if isinstance(t, str) else bytes(t) for t in strings]
    out = []
    for r in range(ids.shape[0]):
        k = min(max(int(take[r]), 0), ids.shape[1])
        words = [enc[t] if 0 <= t < len(enc) else b"" for t in ids[r, :k].tolist()]
        out.append(b" ".join(words) + (bytes(suffixes[r]) if suffixes is not None else b""))
    return out


def zinc_text_tails(y, ln, max_len):
    """What follows the serialiser's ids in a ZINC text (include/gtok.h, gtok_zinc_text_tails): the label token
    f"val_{label:.2f}" with '.' -> '_' and '-' -> 'neg' (zinc_dataset_indexbase.py:192; label = data.y.item(), :205: the
    Python float that holds the stored value exactly) and `<eos>` (:195); a text of more than max_len tokens keeps max_len - 1
    of them and `<eos>` (:217-221).  ln[r] = ids of row r (up to and including `<p>`).  Returns (take int32 [G], [bytes])."""
    ln = np.asarray(ln, np.int64)
    cut = ln + 2 > max_len
    take = np.where(cut, max_len - 1, ln)
    tails = []
    for c, k, v in zip(cut.tolist(), take.tolist(), [float(v) for v in np.asarray(y).tolist()]):
        label = f"val_{v:.2f}".replace(".", "_").replace("-", "neg")
        tails.append(((" " if k > 0 else "") + ("" if c else label + " ") + "<eos>").encode("ascii"))
    return take.astype(np.int32), tails


# ---- packed rows (include/gtok.h, "packed (ragged) rows"): numpy restatement of the format itself ----------------
def row_offsets(ln, ld, align=8):
    n = np.clip(np.asarray(ln, np.int64), 0, ld)
    cost = (n + align - 1) // align * align
    ptr = np.zeros(n.size + 1, np.int64)
    np.cumsum(cost, out=ptr[1:])
    return ptr


def pack_rows(ids, ln, ld=None, elem_bytes=2, align=8, capacity=None, fill=0, with_status=False):
    """(packed, row_ptr): row r's min(len, ld) ids from packed[row_ptr[r]]; slots between rows hold `fill`
    (the product leaves them unwritten: compare through unpack_rows or row by row).  with_status: (packed, row_ptr,
    status) with gtok_pack_rows' bits - 1: an id does not fit 16 bits (stored truncated), 2: a row did not fit `capacity`
    (skipped) - instead of raising."""
    ids = np.asarray(ids)
    if ids.dtype == np.int16:                       # a 16-bit slab (GTOK_SENT_U16)
        ids = ids.view(np.uint16)
    ids = ids.astype(np.int64)
    ld = ids.shape[1] if ld is None else ld
    ptr = row_offsets(ln, ld, align)
    dt = {2: np.uint16, 4: np.int32, 8: np.int64}[elem_bytes]
    status = 0
    packed = np.full(int(ptr[-1]) if capacity is None else capacity, fill, dt)
    for r, l in enumerate(np.clip(np.asarray(ln, np.int64), 0, ld)):
        if ptr[r] + l > packed.size:
            status |= 2
            continue
        if elem_bytes == 2 and (ids[r, :l] >> 16).any():
            status |= 1
        packed[ptr[r]:ptr[r] + l] = (ids[r, :l] & 0xFFFF if elem_bytes == 2 else ids[r, :l]).astype(dt)
    if with_status:
        return packed, ptr, status
    if status & 1:
        raise ValueError("an id does not fit 16 bits")
    if status & 2:
        raise ValueError("capacity too small")
    return packed, ptr


def unpack_rows(packed, row_ptr, ln, ld, pad_id, segment_rows=0, segment_stride=0, with_status=False):
    """row_ptr None: the strided form (row r at r * ld).  A row that would end beyond its segment or the buffer comes out
    as all pad (status bit 2), as gtok_unpack_rows_checked does."""
    ln = np.asarray(ln, np.int64)
    out = np.full((ln.size, ld), pad_id, np.int32)
    status = 0
    for r, l in enumerate(np.clip(ln, 0, ld)):
        if row_ptr is None:
            start = r * ld
        elif segment_rows > 0:
            s = r // segment_rows
            rel = row_ptr[r] - row_ptr[s * segment_rows]
            if rel + l > segment_stride:
                status |= 2 if l > 0 else 0
                continue
            start = s * segment_stride + rel
        else:
            start = row_ptr[r]
        if start + l > packed.size:
            status |= 2 if l > 0 else 0
            continue
        out[r, :l] = packed[start:start + l].astype(np.int64)
    return (out, status) if with_status else out
