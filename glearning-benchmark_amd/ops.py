"""Tensor-level entry points over the C ABI (include/gtok.h).

PyTorch is plumbing here: it owns the device buffers and the HIP stream the
kernels are enqueued on.  Every function raises if libgtok.so is missing or
the tensors are not on a GPU — there is no CPU path in the product.
"""
import ctypes
import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import GtokSentParams, GtokVocabTable, check, lib
from .csr import GraphBatch

SENT_SOS, SENT_RESET, SENT_LADJ, SENT_RADJ, SENT_EOS, SENT_PAD, SENT_IDX_OFFSET = 0, 1, 2, 3, 4, 5, 6

ZINC_ATOM_SYMBOLS = ("C", "N", "O", "F", "P", "S", "Cl", "Br", "I", "X")
ZINC_BOND_NAMES = ("unknown", "single", "double", "triple", "aromatic")


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _need_gpu(t: torch.Tensor, what: str) -> None:
    if t.device.type != "cuda":
        raise _lib.GtokError(f"{what}: tensors must live on the GPU (got {t.device}); the tokenizer has no CPU path")


def _round4(v: int) -> int:
    return max(4, (int(v) + 3) // 4 * 4)


def _alloc_out(G: int, ld: int, device, out=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fresh [G, ld] int32 slab + [G] lengths, or the caller's preallocated pair (`out`)."""
    if out is not None:
        ids, ln = out
        if ids.dtype != torch.int32 or ln.dtype != torch.int32 or tuple(ids.shape) != (G, ld) \
                or ln.numel() != G or not ids.is_contiguous() or ids.device != torch.device(device):
            raise ValueError("out must be (int32 [G, ld] contiguous, int32 [G]) on the batch's device")
        return ids, ln
    return (torch.empty((G, ld), dtype=torch.int32, device=device),
            torch.empty((G,), dtype=torch.int32, device=device))


# ------------------------------------------------------------------------------------------------
# LUT builders (vocab is an INPUT: SURVEY.md F5)
# ------------------------------------------------------------------------------------------------
def zinc_lut(vocab: Dict[str, int], num_node_ids: int) -> torch.Tensor:
    """int32 LUT for gtok_ibtt_zinc; a token absent from `vocab` maps to vocab['<pad>'] exactly as
    TokenDataset does (graph_data_loader/data_loader.py:482)."""
    pad = vocab["<pad>"]
    toks = ["<bos>", "<eos>", "<atom>", "<bond>", "<q>", "regression", "<p>"]
    toks += list(ZINC_ATOM_SYMBOLS) + list(ZINC_BOND_NAMES) + [str(i) for i in range(num_node_ids)]
    return torch.tensor([vocab.get(t, pad) for t in toks], dtype=torch.int32)


def synth_lut(vocab: Dict[str, int], num_node_ids: int) -> torch.Tensor:
    pad = vocab["<pad>"]
    toks = ["<bos>", "<e>", "<n>", "<q>", "<p>"] + [str(i) for i in range(num_node_ids)]
    return torch.tensor([vocab.get(t, pad) for t in toks], dtype=torch.int32)


# ------------------------------------------------------------------------------------------------
# tokenizers
# ------------------------------------------------------------------------------------------------
def ibtt_zinc(batch: GraphBatch, lut: torch.Tensor, max_len: int, pad_id: int,
              ld: Optional[int] = None, out=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """CSR -> IBTT molecular ids.  Returns (ids int32 [G, ld], len int32 [G])."""
    _need_gpu(batch.col, "ibtt_zinc")
    dev = batch.device
    lut = lut.to(dev, dtype=torch.int32).contiguous()
    if ld is None:
        ld = _round4(min(max_len, 4 + 2 * batch.max_nodes + 4 * batch.max_edges))
    ids, ln = _alloc_out(batch.num_graphs, ld, dev, out)
    pack8(batch)
    cs = batch.c_struct()
    check(lib().gtok_ibtt_zinc(ctypes.byref(cs), lut.data_ptr(), lut.numel(), max_len, pad_id,
                               ids.data_ptr(), ld, ln.data_ptr(), _stream(dev)), "gtok_ibtt_zinc")
    return ids, ln


def ibtt_synth(batch: GraphBatch, lut: torch.Tensor, query: Optional[torch.Tensor], max_len: int, pad_id: int,
               ld: Optional[int] = None, out=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """CSR -> graph-token grammar ids.  query: int32 [G,4] = (count, id0, id1, id2) or None."""
    _need_gpu(batch.col, "ibtt_synth")
    dev = batch.device
    lut = lut.to(dev, dtype=torch.int32).contiguous()
    if query is not None:
        query = query.to(dev, dtype=torch.int32).contiguous()
        if tuple(query.shape) != (batch.num_graphs, 4):
            raise ValueError("query must be [G, 4]")
    if ld is None:
        ld = _round4(min(max_len, 3 * batch.max_edges + batch.max_nodes + 7))
    ids, ln = _alloc_out(batch.num_graphs, ld, dev, out)
    cs = batch.c_struct()
    check(lib().gtok_ibtt_synth(ctypes.byref(cs), lut.data_ptr(), lut.numel(),
                                None if query is None else query.data_ptr(), max_len, pad_id,
                                ids.data_ptr(), ld, ln.data_ptr(), _stream(dev)), "gtok_ibtt_synth")
    return ids, ln


def sent_safe_ld(batch: GraphBatch, labeled: bool, max_len: int, with_query: bool = False) -> int:
    """Width no SENT row of this batch can exceed (length bound)."""
    n, e = batch.max_nodes, batch.max_edges
    bound = (2 + 7 * n + 2 * e) if labeled else (2 + 5 * n + e)
    # a multiple of 16 ids: every row then starts on a 64-byte boundary and the kernels' 64-byte store bursts are whole
    return (min(max_len, bound) + (3 if with_query else 0) + 15) // 16 * 16


LANE_MIN_GRAPHS = 28000       # = GTOK_LANE_MIN_GRAPHS of gtok_sent.hip: from here on the molecule lane kernel takes small symmetric graphs
ADJBITS_MIN_GRAPHS = 20000    # = GTOK_BLANE_MIN_GRAPHS of gtok_sent.hip: below it gtok_sent never picks the bit-matrix lane kernel


def pack8(batch: GraphBatch) -> bool:
    """Give a device batch of small graphs its byte-packed rowptr / col mirror (gtok_csr_pack8; once per batch,
    kept on the batch object).  Returns whether the batch has one now.  A layout step like the CSR build: the
    lane-per-graph SENT kernel reads the mirror instead of the int32 arrays."""
    if batch.rowptr8 is not None:
        return True
    if os.environ.get("GTOK_NO_PACK8") == "1":      # tests: the lane kernel's int32 staging path (C-ABI callers without a mirror)
        return False
    if batch.col.device.type != "cuda" or batch.num_graphs == 0 or batch.max_edges > 255 or batch.max_nodes > 64 \
            or not (batch.flags & _lib.CSR_SIMPLE_SYMMETRIC):
        return False
    dev = batch.device
    r8 = torch.empty(batch.rowptr.numel() + 16, dtype=torch.uint8, device=dev)   # + slack: 16-byte vector loads of the
    c8 = torch.empty(batch.col.numel() + 16, dtype=torch.uint8, device=dev)      # last chunk may run past the end
    cs = batch.c_struct()
    check(lib().gtok_csr_pack8(ctypes.byref(cs), batch.rowptr.numel(), batch.col.numel(), r8.data_ptr(), c8.data_ptr(),
                               _stream(dev)), "gtok_csr_pack8")
    batch.rowptr8, batch.col8 = r8, c8
    return True


LANE_UNIT_LDS = 10224      # LDS bytes a unit of the reordered batch may need: 16 waves per CU stay resident (160 KB / 16, less the 16 bytes of the workgroup's ticket counter)


def csr_check(batch: GraphBatch) -> Dict[str, int]:
    """gtok_csr_check: verify on the device what GTOK_CSR_SIMPLE_SYMMETRIC claims and measure the batch maxima.  Returns
    dict(violations, max_degree, max_nodes, max_edges); one 32-byte read-back.  violations == 0 (and at least one entry)
    is what GraphBatch.flags |= CSR_SIMPLE_SYMMETRIC needs."""
    _need_gpu(batch.rowptr, "csr_check")
    dev = batch.device
    info = torch.empty(8, dtype=torch.int32, device=dev)
    cs = batch.c_struct()
    check(lib().gtok_csr_check(ctypes.byref(cs), info.data_ptr(), _stream(dev)), "gtok_csr_check")
    v = info.tolist()
    return dict(violations=v[0], max_degree=v[1], max_nodes=v[2], max_edges=v[3])


def lane_sorted(batch: GraphBatch, verify: bool = False) -> Optional[GraphBatch]:
    """The copy of a device batch of small symmetric graphs that sent_lane_kernel walks fastest (made once per batch,
    kept on the batch object): graphs stored by descending expected walk length - nodes + leaves, a walk restarts once
    per dead end - and dealt to waves in units of <= 64 neighbours whose staged bytes fit LANE_UNIT_LDS, so that the 64
    walks of a wave end together (in dataset order a unit of ZINC runs 41 steps for walks of 28 on average).
    `graph_ids` carries every slot's dataset index: output rows, lengths, query entries and RNG identity follow it, the
    tokens are those of the batch in dataset order.  A layout step like the CSR build, never part of an epoch.
    Built on the device by gtok_csr_lane_sort (counting sort, single-pass scans, the byte mirror in the same pass): a dozen
    small launches and ONE 32-byte read-back (the unit count and the LDS sizes of the launch).  verify=True: the batch's
    flags are not trusted - the same call checks GTOK_CSR_SIMPLE_SYMMETRIC on the device and sets it on `batch`
    (None is returned when the batch turns out not to be simple and symmetric)."""
    if batch.lane_sorted is not None:
        return batch.lane_sorted
    if batch.col.device.type != "cuda" or batch.num_graphs == 0 or batch.max_nodes > 64 or batch.max_edges > 255 \
            or batch.graph_ids is not None or batch.num_edges_total == 0:
        return None
    if not verify and not (batch.flags & _lib.CSR_SIMPLE_SYMMETRIC):
        return None
    dev, G = batch.device, batch.num_graphs
    N, E = batch.num_nodes_total, batch.num_edges_total
    L = lib()
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
    o = dict(graph_ids=i32(G), node_ptr=i32(G + 1), edge_ptr=torch.empty(G + 1, dtype=torch.int64, device=dev), rowptr=i32(N + G), col=i32(E),
             nattr=None if batch.nattr is None else u8(N), eattr=None if batch.eattr is None else u8(E),
             rowptr8=u8(N + G + 16), col8=u8(E + 16),          # + slack: 16-byte vector loads of the last chunk may run past the end
             unit_ptr=i32(G + 1), unit_info=i32(8 * G), info=i32(8))
    if os.environ.get("GTOK_NO_PACK8") == "1":              # tests: the lane kernel's int32 staging path
        o["rowptr8"] = o["col8"] = None
    ws = torch.empty(int(L.gtok_csr_lane_sort_workspace(G)), dtype=torch.uint8, device=dev)
    outs = _lib.GtokCsrSorted(*[None if o[n] is None else o[n].data_ptr() for n, _ in _lib.GtokCsrSorted._fields_])
    cs = batch.c_struct()
    check(L.gtok_csr_lane_sort(ctypes.byref(cs), LANE_UNIT_LDS, int(verify), ctypes.byref(outs), ws.data_ptr(), ws.numel(), _stream(dev)),
          "gtok_csr_lane_sort")
    viol, maxdeg, _, _, units, chunk_n, chunk_e, _ = o["info"].tolist()          # the one host read of the preparation
    if verify:
        if viol:
            return None
        batch.flags |= _lib.CSR_SIMPLE_SYMMETRIC
        batch.max_degree = maxdeg
    out = GraphBatch(G, batch.max_nodes, batch.max_edges, o["node_ptr"], o["edge_ptr"], o["rowptr"], o["col"], None, o["nattr"], o["eattr"],
                     batch.flags, chunk_n, chunk_e, batch.max_degree or maxdeg)
    out.graph_ids, out.num_units = o["graph_ids"], units
    out.unit_ptr = o["unit_ptr"][:units + 1].clone()        # (the arrays are sized for one unit per graph: keep what is used)
    out.unit_info = o["unit_info"][:8 * units].clone().view(units, 8)
    out.rowptr8, out.col8 = o["rowptr8"], o["col8"]
    batch.lane_sorted = out
    return out


_POPCOUNT8 = None


def _lane_order(batch: GraphBatch, typical_len: int) -> torch.Tensor:
    """The order sent_blane_kernel's lanes take the graphs in (64 consecutive entries share a wave, and a wave lasts as
    long as its slowest walk and, at every step, as its largest bracket).  A heuristic that only moves time, never tokens:
    graphs are grouped by the number of trail steps they are expected to take and, within a group, by density, so that
    the brackets of a wave's walks grow at the same pace.  The model: a walk takes n + r steps (r restarts ~ leaves - 1:
    the walk comes back once per dead end) and writes n position tokens, 2 per restart, the e - n + 1 bracket members and
    2 per non-empty bracket; members pile up with the square of the steps taken, the rest linearly - a t + q t^2 tokens
    after t steps - and a row cut at `typical_len` tokens stops where that reaches it (a 256-node star: 1.5 tokens per
    step, 400 steps of its 510; a dense random graph: 3 t + p t^2 / 2)."""
    global _POPCOUNT8
    dev = batch.device
    n = (batch.node_ptr[1:] - batch.node_ptr[:-1]).to(torch.float64)
    e = (batch.edge_ptr[1:] - batch.edge_ptr[:-1]).to(torch.float64)
    if batch.flags & _lib.CSR_SIMPLE_SYMMETRIC:
        e = e / 2
    # leaves = nodes of closure degree 1: bit 0 of the degree set, every higher bit clear (the mirror's degree planes)
    pl = batch.adj_planes
    higher = pl[:, 1]
    for k in range(2, 8):
        higher = higher | pl[:, k]
    deg1 = (pl[:, 0] & ~higher).contiguous().view(torch.uint8)
    if _POPCOUNT8 is None or _POPCOUNT8.device != dev:
        _POPCOUNT8 = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
    leaves = _POPCOUNT8[deg1.long()].reshape(batch.num_graphs, -1).sum(1).to(torch.float64)
    r = (leaves - 1).clamp(min=0)
    members = (e - n + 1).clamp(min=0)
    steps_full = (n + r).clamp(min=1)
    a = (n + 2 * r + 2 * torch.minimum(n, members)) / steps_full
    q = members / (steps_full * steps_full)
    L = float(typical_len)
    cut = torch.where(q > 1e-9, (torch.sqrt(a * a + 4 * q * L) - a) / (2 * q).clamp(min=1e-9), L / a.clamp(min=1e-9))
    steps = torch.minimum(cut, steps_full).round().clamp(0, 1023).to(torch.int64)
    p = (2 * e / (n * n).clamp(min=1)).clamp(min=0.0, max=1.0)
    key = steps * 1024 + (p * 1023).round().to(torch.int64)
    return torch.argsort(key, descending=True, stable=True).to(torch.int32)


def adjbits(batch: GraphBatch) -> bool:
    """Give a device batch of graphs with <= 256 nodes its adjacency bit-matrix mirror (gtok_csr_adjbits; once per
    batch, kept on the batch object) and the order its graphs are dealt to lanes (longest walks first, so that the 64
    walks of a wave have similar lengths).  Returns whether the batch has one now.  A layout step like the CSR build:
    the lane-per-graph SENT kernel for unlabelled graphs reads one row of it per trail step."""
    if batch.adj_rows is not None:
        return True
    if os.environ.get("GTOK_NO_ADJBITS") == "1":
        return False
    if batch.col.device.type != "cuda" or batch.num_graphs == 0 or batch.max_nodes > 256:
        return False
    dev, G = batch.device, batch.num_graphs
    W = 1 if batch.max_nodes <= 64 else 2 if batch.max_nodes <= 128 else 4
    total_nodes = batch.rowptr.numel() - G
    rows = torch.empty((max(total_nodes, 1), W), dtype=torch.int64, device=dev)
    planes = torch.empty((G, 8, W), dtype=torch.int64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    cs = batch.c_struct()
    check(lib().gtok_csr_adjbits(ctypes.byref(cs), W, rows.data_ptr(), planes.data_ptr(), info.data_ptr(), _stream(dev)),
          "gtok_csr_adjbits")
    maxdeg = int(info.item())
    if maxdeg > 255:          # a 256-node graph with a full row and a self loop: 8 counter planes cannot hold it
        return False
    batch.adj_rows, batch.adj_planes, batch.adj_words, batch.adj_max_degree = rows, planes, W, maxdeg
    return True


PACK_STATE_FILL, PACK_STATE_STRIDE, PACK_REGIONS = 16, 16, 64       # include/gtok.h: GTOK_PACK_STATE_*
PACK_STATE_WORDS = PACK_STATE_FILL + PACK_STATE_STRIDE * PACK_REGIONS


class PackedRows:
    """The packed twin of a SENT launch's rows (ops.sent(..., packed=PackedRows(...)): gtok_sent_packed).  buf: `capacity` ids of
    the slab's width, every row from a 16-byte boundary; row_start int64 [rows]: where row (epoch, graph) starts in buf, -1 for a row
    that did not fit (rows lie in the order their 64-graph units finished, inside up to 64 regions of the buffer that fill side by
    side, not in dataset order - readers go through row_start: ops.unpack_rows_at, ops.collate_packed(buf, row_start, ...)); state
    int64 [PACK_STATE_WORDS]: the status word and the regions' fill marks (include/gtok.h).  Size the buffer as the expected total
    + a few per cent.  fused: whether the walk kernel itself wrote it (else gtok_sent + gtok_pack_rows_scan did: dataset order)."""
    __slots__ = ("buf", "row_start", "_ring", "_slot", "capacity", "fused")
    RING = 32          # state blocks (8 KB each) zeroed together: a launch takes the next one instead of paying a memset of its own

    def __init__(self, rows: int, capacity: int, u16: bool, device):
        self.capacity = max(8, -(-int(capacity) // 8) * 8)
        self.buf = torch.empty(self.capacity, dtype=torch.int16 if u16 else torch.int32, device=device)
        self.row_start = torch.empty(max(1, int(rows)), dtype=torch.int64, device=device)
        self._ring = torch.zeros((self.RING, PACK_STATE_WORDS), dtype=torch.int64, device=device)
        self._slot = 0
        self.fused = None

    @property
    def state(self) -> torch.Tensor:
        """int64 [PACK_STATE_WORDS]: the status word and the fill marks of the LAST fill (include/gtok.h)"""
        return self._ring[self._slot]

    def _fresh_state(self) -> torch.Tensor:
        """a zeroed state block for the next launch: the next one of the ring (all of them are zeroed again, in one launch, when the
        ring wraps); under stream capture the block is zeroed inside the capture - replays reuse it"""
        if torch.cuda.is_current_stream_capturing():
            self.state.zero_()
            return self.state
        self._slot = (self._slot + 1) % self.RING
        if self._slot == 0:
            self._ring.zero_()
        return self.state

    def status(self) -> torch.Tensor:
        """int32 [1] on the device: the pack_rows status bits of this fill (bit 1: the buffer was too small for some rows)."""
        return self.state[0:1].to(torch.int32)

    def used(self) -> torch.Tensor:
        """int64 scalar on the device: ids the fill took (the regions' fill marks added up; skipped rows included)."""
        return self.state[PACK_STATE_FILL::PACK_STATE_STRIDE].sum()


def sent(batch: GraphBatch, max_num_nodes: int, max_len: int, seed: int, epoch: int = 0, labeled: bool = False,
         num_node_types: int = 0, num_edge_types: int = 0, remap_zinc: bool = False, pad_id: int = SENT_PAD,
         graph_base: int = 0, query: Optional[torch.Tensor] = None,
         ld: Optional[int] = None, out=None, pad: bool = True, epochs: int = 1, u16: bool = False,
         packed: Optional[PackedRows] = None, slab: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """SENT trail walk.  Returns (ids int32 [G, ld], len int32 [G]); len > ld flags a too-narrow slab.
    pad=False (GTOK_SENT_NO_PAD): rows are only written up to their length - for consumers that go through `len`
    (ops.collate does); the rest of the slab keeps whatever it held.
    epochs=K > 1 (gtok_sent_params.epoch_count): the trails of epochs epoch .. epoch + K - 1 in ONE launch - returns
    (ids [K, G, ld], len [K, G]); slice e is what a call with epoch + e returns.  The trainer re-tokenizes its splits every
    epoch (trainer/train_agtt.py:246-250): K epochs of a small split fill the chip where one cannot.
    u16=True (GTOK_SENT_U16): the slab holds 16-bit ids (torch.int16 storage, to be read as unsigned: every SENT id fits) -
    half the bytes; readers: ops.collate_packed / ops.unpack_rows with row_ptr=None, ops.pack_rows_u16.
    packed=PackedRows(K * G, capacity, u16, device): the rows are ALSO appended to packed.buf (a fresh fill: its state is
    zeroed first) - by the walk kernel itself where the batch runs sent_lane_kernel (gtok_sent_packed: no second pass over the
    rows), else by gtok_pack_rows_scan behind the walk; either way packed.row_start / packed.state describe the result.
    slab=False (with packed=; GTOK_SENT_PACK_ONLY): the caller reads the packed rows alone - no [K, G, ld] slab is written or
    allocated (the walk stages its rows in a per-device scratch of 64 rows per resident wave: 92 MB on an MI355X whatever K is) and
    the first return value is None; out=(staging, len) then names the caller's own staging space (or None) and length tensor."""
    _need_gpu(batch.col, "sent")
    dev = batch.device
    K = max(1, int(epochs))
    if query is not None:
        query = query.to(dev, dtype=torch.int32).contiguous()
        if tuple(query.shape) != (batch.num_graphs, 2):
            raise ValueError("query must be [G, 2] (query_u, query_v)")
    if ld is None:
        ld = sent_safe_ld(batch, labeled, max_len, query is not None)
    G = batch.num_graphs
    staging = None
    if not slab:
        if packed is None:
            raise ValueError("slab=False goes with packed=")
        ids = None
        if out is not None:           # (staging rows of the caller's own - launches on several streams must not share the module's - , lengths)
            staging, ln = out
            if staging is not None and (staging.dtype != (torch.int16 if u16 else torch.int32) or not staging.is_contiguous()
                                        or staging.device != torch.device(dev)):
                raise ValueError("out[0] with slab=False: contiguous staging space of the rows' id width on the batch's device (or None)")
            if ln.dtype != torch.int32 or ln.numel() != K * G or not ln.is_contiguous():
                raise ValueError("out[1]: contiguous int32 [K * G]")
        else:
            ln = torch.empty((K * G,), dtype=torch.int32, device=dev)
    elif out is not None:
        ids, ln = out
        want = torch.int16 if u16 else torch.int32
        if ids.dtype != want or ln.dtype != torch.int32 or ids.numel() != K * G * ld or ln.numel() != K * G \
                or not ids.is_contiguous() or not ln.is_contiguous() or ids.device != torch.device(dev) or (G and ids.shape[-1] != ld):
            raise ValueError(f"out must be ({want} [K * G, ld] contiguous, int32 [K * G]) on the batch's device")
    else:
        ids = torch.empty((K * G, ld), dtype=torch.int16 if u16 else torch.int32, device=dev)
        ln = torch.empty((K * G,), dtype=torch.int32, device=dev)
    if _lib.library_version() < 4 and (K > 1 or u16 or not pad):
        # (GTOK_LIB pointing at an ABI v3 build: it would ignore epoch_count and the row flags and write an int32 [G, ld] slab)
        raise _lib.GtokError(f"the loaded library has ABI version {_lib.library_version()}: epochs > 1, u16 and pad=False need version 4")
    walks = G * K
    if not batch.prepared:          # (a batch prepared by the caller - torch.ops.gtok.csr_prepare - is launched as it stands)
        pack8(batch)
        pin = os.environ.get("GTOK_SENT_KERNEL", "")
        # the bit-matrix mirror is only built where gtok_sent would pick the kernel that reads it: unlabelled batches of
        # >= ADJBITS_MIN_GRAPHS walks that the molecule kernel (<= 64 nodes, simple symmetric, >= LANE_MIN_GRAPHS) does not
        # take - and never again for a batch that turned out unusable (a closure degree above 255)
        lane_takes = batch.max_nodes <= 64 and batch.max_edges <= 255 and bool(batch.flags & _lib.CSR_SIMPLE_SYMMETRIC) \
            and walks >= LANE_MIN_GRAPHS
        wants_blane = pin == "blane" or (not pin and not labeled and not remap_zinc and walks >= ADJBITS_MIN_GRAPHS and not lane_takes)
        if wants_blane and not batch.adj_unusable:
            if adjbits(batch):
                if os.environ.get("GTOK_BLANE_ORDER", "1") != "0" and batch.lane_order_len != max(1, max_len):
                    batch.lane_order, batch.lane_order_len = _lane_order(batch, max(1, max_len)), max(1, max_len)   # once per (batch, max_len)
            elif batch.col.device.type == "cuda" and batch.max_nodes <= 256 and os.environ.get("GTOK_NO_ADJBITS") != "1":
                batch.adj_unusable = True
    flags = (0 if pad else _lib.SENT_NO_PAD) | (_lib.SENT_U16 if u16 else 0)
    p = GtokSentParams(max_num_nodes, int(labeled), num_node_types, num_edge_types, max_len, int(remap_zinc),
                       pad_id, flags, seed & (2 ** 64 - 1), epoch & (2 ** 64 - 1), graph_base,
                       None if query is None else query.data_ptr(), K, 0)
    cs = batch.c_struct()
    if os.environ.get("GTOK_NO_LANE_SORT") != "1" and batch.graph_ids is None and not batch.prepared \
            and lib().gtok_sent_kernel_name(ctypes.byref(cs), ctypes.byref(p)) == b"sent_lane_kernel":
        sb = lane_sorted(batch)                                       # once per resident batch
        if sb is not None:
            cs = sb.c_struct()
    global _LAST_SENT
    _LAST_SENT = (cs, p)
    rc = _lib.E_UNSUPPORTED
    if packed is not None:
        if _lib.library_version() < 6:
            raise _lib.GtokError(f"the loaded library has ABI version {_lib.library_version()}: packed= needs version 6")
        if packed.buf.dtype != (torch.int16 if u16 else torch.int32) or packed.row_start.numel() < K * G or packed.buf.device != ln.device:
            raise ValueError("packed: a PackedRows of this launch's rows, id width and device")
        packed._fresh_state()
        if G and ld % (8 if u16 else 4) == 0:
            dst, pp = ids, p
            if not slab:
                need = int(lib().gtok_sent_pack_scratch_rows(_stream(dev))) * ld
                if staging is not None and staging.numel() < need:
                    raise ValueError(f"out[0] with slab=False: {need} ids of staging space (gtok_sent_pack_scratch_rows() x ld)")
                dst = staging if staging is not None else _pack_scratch(dev, ld, u16)
                pp = GtokSentParams(max_num_nodes, int(labeled), num_node_types, num_edge_types, max_len, int(remap_zinc), pad_id,
                                    flags | _lib.SENT_PACK_ONLY, seed & (2 ** 64 - 1), epoch & (2 ** 64 - 1), graph_base,
                                    None if query is None else query.data_ptr(), K, 0)
            rc = lib().gtok_sent_packed(ctypes.byref(cs), ctypes.byref(pp), dst.data_ptr(), ld, ln.data_ptr(), packed.buf.data_ptr(),
                                        packed.capacity, packed.row_start.data_ptr(), packed.state.data_ptr(), _stream(dev))
            if rc != _lib.E_UNSUPPORTED:
                check(rc, "gtok_sent_packed")
        packed.fused = rc == 0
    if rc == _lib.E_UNSUPPORTED:
        if ids is None:             # another kernel walks: it needs the slab after all (a temporary one)
            ids = torch.empty((K * G, ld), dtype=torch.int16 if u16 else torch.int32, device=dev)
        check(lib().gtok_sent(ctypes.byref(cs), ctypes.byref(p), ids.data_ptr(), ld, ln.data_ptr(), _stream(dev)),
              "gtok_sent")
        if packed is not None and G:           # another kernel walked: the one-pass pack behind it (dataset order)
            ptr = torch.empty(K * G + 1, dtype=torch.int64, device=dev)
            st = torch.empty(1, dtype=torch.int32, device=dev)
            eb = 2 if u16 else 4
            check(lib().gtok_pack_rows_scan(ids.data_ptr(), eb, ld, ln.data_ptr(), K * G, 8, eb, packed.buf.data_ptr(), packed.capacity,
                                            ptr.data_ptr(), st.data_ptr(), _stream(dev)), "gtok_pack_rows_scan")
            packed.row_start[:K * G].copy_(ptr[:-1])
            packed.state[PACK_STATE_FILL:PACK_STATE_FILL + 1].copy_(ptr[-1:])
            packed.state[0:1].copy_(st.to(torch.int64) & 2)
    if not slab:
        return None, (ln.view(K, G) if K > 1 else ln.view(G))
    if K > 1:
        return ids.view(K, G, ld), ln.view(K, G)
    return ids.view(G, ld), ln.view(G)


_PACK_SCRATCH = {}


def _pack_scratch(dev, ld: int, u16: bool) -> torch.Tensor:
    """the staging rows of GTOK_SENT_PACK_ONLY launches: one buffer per (device, width, id size), grown on demand, shared by every
    launch on that device (launches on one stream follow each other; callers with several streams pass their own slab)"""
    key = (torch.device(dev).index or 0, bool(u16))
    rows = int(lib().gtok_sent_pack_scratch_rows(_stream(dev)))
    if rows < 0:
        check(rows, "gtok_sent_pack_scratch_rows")
    t = _PACK_SCRATCH.get(key)
    if t is None or t.numel() < rows * ld:
        t = torch.empty(rows * ld, dtype=torch.int16 if u16 else torch.int32, device=dev)
        _PACK_SCRATCH[key] = t
    return t


_LAST_SENT = None


def last_sent_kernel() -> str:
    """Name of the kernel the most recent ops.sent call of this process launched (tests, bench labels): what the C ABI's
    gtok_sent_kernel_name answers for the very structs that call passed to gtok_sent."""
    if _LAST_SENT is None:
        return ""
    cs, p = _LAST_SENT
    return lib().gtok_sent_kernel_name(ctypes.byref(cs), ctypes.byref(p)).decode()


def sent_kernel_name(batch: GraphBatch, max_num_nodes: int, max_len: int, labeled: bool = False, num_node_types: int = 0,
                     num_edge_types: int = 0, remap_zinc: bool = False, epochs: int = 1) -> str:
    """Name of the kernel gtok_sent() picks for this batch (for profiles and bench labels)."""
    p = GtokSentParams(max_num_nodes, int(labeled), num_node_types, num_edge_types, max_len, int(remap_zinc), SENT_PAD,
                       0, 0, 0, 0, None, max(1, int(epochs)), 0)
    cs = batch.c_struct()
    return lib().gtok_sent_kernel_name(ctypes.byref(cs), ctypes.byref(p)).decode()


def sent_decode(ids: torch.Tensor, ln: torch.Tensor, max_num_nodes: int, labeled: bool = False, num_node_types: int = 0,
                edge_cap: Optional[int] = None, node_cap: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Un-remapped SENT rows -> graphs in visit-index space (node k = the k-th node the trail visited): num_nodes,
    num_edges, status int32 [G]; edge_a, edge_b, edge_type int32 [G, edge_cap]; node_type int32 [G, node_cap].
    status: 0 complete, 1 malformed, 2 capacity exceeded, 3 cut before EOS."""
    _need_gpu(ids, "sent_decode")
    if ids.dtype != torch.int32 or ids.dim() != 2 or not ids.is_contiguous():
        raise ValueError("sent_decode expects a contiguous int32 [G, ld] slab")
    dev, (G, ld) = ids.device, ids.shape
    ecap = int(edge_cap) if edge_cap is not None else ld
    ncap = int(node_cap) if node_cap is not None else max(1, max_num_nodes)
    i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    out = dict(num_nodes=i32(G), num_edges=i32(G), status=i32(G), edge_a=i32(G, ecap), edge_b=i32(G, ecap),
               edge_type=i32(G, ecap), node_type=i32(G, ncap))
    _lib.check(_lib.lib().gtok_sent_decode(ids.data_ptr(), ld, ln.data_ptr(), G, max_num_nodes, int(labeled), num_node_types,
                                           out["num_nodes"].data_ptr(), out["num_edges"].data_ptr(), out["edge_a"].data_ptr(),
                                           out["edge_b"].data_ptr(), out["edge_type"].data_ptr(), ecap,
                                           out["node_type"].data_ptr(), ncap, out["status"].data_ptr(), _stream(dev)),
               "gtok_sent_decode")
    return out


def remap_zinc(ids: torch.Tensor, ln: torch.Tensor, idx_offset: int, node_idx_offset: int,
               edge_idx_offset: int) -> torch.Tensor:
    _need_gpu(ids, "remap_zinc")
    ids = ids.contiguous()
    out = torch.empty_like(ids)
    check(lib().gtok_remap_zinc(ids.data_ptr(), out.data_ptr(), ids.shape[1], ln.data_ptr(), ids.shape[0],
                                idx_offset, node_idx_offset, edge_idx_offset, _stream(ids.device)),
          "gtok_remap_zinc")
    return out


def collate(ids: torch.Tensor, ln: torch.Tensor, index: torch.Tensor, pad_id: int,
            out_ld: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rows `index` of the slab -> (X int64 [B, out_ld], attn bool [B, out_ld])."""
    _need_gpu(ids, "collate")
    dev = ids.device
    index = index.to(dev, dtype=torch.int64).contiguous()
    B = int(index.numel())
    X = torch.empty((B, out_ld), dtype=torch.int64, device=dev)
    A = torch.empty((B, out_ld), dtype=torch.bool, device=dev)
    check(lib().gtok_collate(ids.data_ptr(), ids.shape[1], ln.data_ptr(), index.data_ptr(), B, pad_id,
                             X.data_ptr(), A.data_ptr(), out_ld, None, _stream(dev)), "gtok_collate")
    return X, A


# ------------------------------------------------------------------------------------------------
# packed (ragged) rows: the form token rows take when they leave the GPU (all-gather, D2H)
# ------------------------------------------------------------------------------------------------
ROW_ALIGN = 8      # ids: 16-byte aligned row starts at 2 bytes per id (the vector path of pack / unpack)


def row_offsets(ln: torch.Tensor, ld: int, align: int = ROW_ALIGN) -> torch.Tensor:
    """int64 [rows + 1]: row r of the packed form starts at element row_ptr[r] and holds min(len[r], ld) ids;
    starts are multiples of `align` (gtok_row_offsets)."""
    _need_gpu(ln, "row_offsets")
    if ln.dtype != torch.int32 or not ln.is_contiguous():
        raise ValueError("row_offsets expects a contiguous int32 length vector")
    ptr = torch.empty(ln.numel() + 1, dtype=torch.int64, device=ln.device)
    check(lib().gtok_row_offsets(ln.data_ptr(), ln.numel(), int(ld), int(align), ptr.data_ptr(), _stream(ln.device)),
          "gtok_row_offsets")
    return ptr


def _pack_verdict(packed, row_ptr, status, cap, what):
    st = int(status.item())
    if st & 1:
        raise _lib.GtokError(f"{what}: an id does not fit 16 bits; pack with elem_bytes=4")
    if st & 2:
        raise _lib.GtokError(f"{what}: capacity {cap} is too small for these rows")
    return packed, row_ptr


def pack_rows_scan(ids: torch.Tensor, ln: torch.Tensor, elem_bytes: int, capacity: int, align: int = ROW_ALIGN):
    """row_offsets + pack_rows / pack_rows_u16 in ONE pass (gtok_pack_rows_scan): for callers that know a capacity before
    they know the sizes.  ids: the int32 slab or the 16-bit slab (int16 storage).  Returns (packed, row_ptr, status) - nothing
    waits for the host (status bit 0: an id needs more than 16 bits, bit 1: rows that did not fit `capacity` were skipped)."""
    _need_gpu(ids, "pack_rows_scan")
    if ids.dtype not in (torch.int32, torch.int16) or ids.dim() != 2 or not ids.is_contiguous() or ln.dtype != torch.int32 or not ln.is_contiguous():
        raise ValueError("pack_rows_scan expects a contiguous int32 / int16 [rows, ld] slab and contiguous int32 lengths")
    src_bytes = ids.element_size()
    if elem_bytes not in ((2, 4, 8) if src_bytes == 2 else (2, 4)):
        raise ValueError("elem_bytes must be 2 or 4 (8 also from a 16-bit slab)")
    dev, (rows, ld) = ids.device, ids.shape
    if int(ln.numel()) != rows:
        raise ValueError("pack_rows_scan: one length per row")
    cap = int(capacity)
    packed = torch.empty(max(cap, 1), dtype={2: torch.int16, 4: torch.int32, 8: torch.int64}[elem_bytes], device=dev)
    row_ptr = torch.empty(rows + 1, dtype=torch.int64, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)          # (set by the call)
    check(lib().gtok_pack_rows_scan(ids.data_ptr(), src_bytes, ld, ln.data_ptr(), rows, int(align), elem_bytes, packed.data_ptr(), cap,
                                    row_ptr.data_ptr(), status.data_ptr(), _stream(dev)), "gtok_pack_rows_scan")
    return packed, row_ptr, status


def pack_rows(ids: torch.Tensor, ln: torch.Tensor, row_ptr: Optional[torch.Tensor] = None, elem_bytes: int = 2,
              capacity: Optional[int] = None, align: int = ROW_ALIGN, check_status: bool = True):
    """[rows, ld] int32 slab + lengths -> (packed, row_ptr): row r's ids back to back from packed[row_ptr[r]], int16
    storage (ids 0..65535 kept as their 16 bits) or int32.  capacity: elements to allocate (an all-gather needs the
    same size on every rank); default = exactly row_ptr[-1] (one host read).  Raises when elem_bytes == 2 and an id
    needs more bits, or when `capacity` is too small; check_status=False hands the status tensor back instead
    (packed, row_ptr, status) and leaves the host out of it."""
    _need_gpu(ids, "pack_rows")
    if ids.dtype != torch.int32 or ids.dim() != 2 or not ids.is_contiguous() or ln.dtype != torch.int32:
        raise ValueError("pack_rows expects a contiguous int32 [rows, ld] slab and int32 lengths")
    dev, (rows, ld) = ids.device, ids.shape
    if int(ln.numel()) != rows:
        raise ValueError("pack_rows: one length per row")
    if row_ptr is None and capacity is not None and _lib.library_version() >= 5:       # sizes not needed first: one pass
        packed, row_ptr, status = pack_rows_scan(ids, ln.contiguous(), elem_bytes, capacity, align)
        return (packed, row_ptr, status) if not check_status else _pack_verdict(packed, row_ptr, status, int(capacity), "pack_rows")
    if row_ptr is None:
        row_ptr = row_offsets(ln, ld, align)
    cap = int(row_ptr[-1]) if capacity is None else int(capacity)
    packed = torch.empty(max(cap, 1), dtype=torch.int16 if elem_bytes == 2 else torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().gtok_pack_rows(ids.data_ptr(), ld, ln.data_ptr(), rows, row_ptr.data_ptr(), elem_bytes, packed.data_ptr(),
                               cap, status.data_ptr(), _stream(dev)), "gtok_pack_rows")
    if not check_status:
        return packed, row_ptr, status
    st = int(status.item())
    if st & 1:
        raise _lib.GtokError("pack_rows: an id does not fit 16 bits; pack with elem_bytes=4")
    if st & 2:
        raise _lib.GtokError(f"pack_rows: capacity {cap} is too small for these rows")
    return packed, row_ptr


def pack_rows_u16(ids16: torch.Tensor, ln: torch.Tensor, row_ptr: Optional[torch.Tensor] = None, elem_bytes: int = 2,
                  capacity: Optional[int] = None, align: int = ROW_ALIGN, check_status: bool = True):
    """pack_rows reading a 16-bit slab (ops.sent(..., u16=True)): -> (packed, row_ptr) with int16, int32 or int64 storage
    (elem_bytes 2 / 4 / 8; 8 = the dtype the reference's token tensors have: the packed buffer can leave for the host as
    it is).  check_status=False: (packed, row_ptr, status) and no host read."""
    _need_gpu(ids16, "pack_rows_u16")
    if ids16.dtype != torch.int16 or ids16.dim() != 2 or not ids16.is_contiguous() or ln.dtype != torch.int32:
        raise ValueError("pack_rows_u16 expects a contiguous int16 [rows, ld] slab and int32 lengths")
    if elem_bytes not in (2, 4, 8):
        raise ValueError("elem_bytes must be 2, 4 or 8")
    dev, (rows, ld) = ids16.device, ids16.shape
    if int(ln.numel()) != rows:
        raise ValueError("pack_rows_u16: one length per row")
    if row_ptr is None and capacity is not None and _lib.library_version() >= 5:       # sizes not needed first: one pass
        packed, row_ptr, status = pack_rows_scan(ids16, ln.contiguous(), elem_bytes, capacity, align)
        return (packed, row_ptr, status) if not check_status else _pack_verdict(packed, row_ptr, status, int(capacity), "pack_rows_u16")
    if row_ptr is None:
        row_ptr = row_offsets(ln, ld, align)
    cap = int(row_ptr[-1]) if capacity is None else int(capacity)
    packed = torch.empty(max(cap, 1), dtype={2: torch.int16, 4: torch.int32, 8: torch.int64}[elem_bytes], device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().gtok_pack_rows_u16(ids16.data_ptr(), ld, ln.data_ptr(), rows, row_ptr.data_ptr(), elem_bytes, packed.data_ptr(),
                                   cap, status.data_ptr(), _stream(dev)), "gtok_pack_rows_u16")
    if not check_status:
        return packed, row_ptr, status
    if int(status.item()) & 2:
        raise _lib.GtokError(f"pack_rows_u16: capacity {cap} is too small for these rows")
    return packed, row_ptr


def unpack_rows(packed: torch.Tensor, row_ptr: Optional[torch.Tensor], ln: torch.Tensor, ld: int, pad_id: int,
                segment_rows: int = 0, segment_stride: int = 0, out: Optional[torch.Tensor] = None,
                status: Optional[torch.Tensor] = None, u16: bool = False) -> torch.Tensor:
    """packed rows -> [rows, ld] int32 slab with pad_id behind every row (gtok_unpack_rows_checked).  segment_rows /
    segment_stride: the packed buffer is the concatenation of per-rank buffers (see include/gtok.h).  row_ptr=None: the
    strided form - `packed` is itself a [rows, ld] slab of 16- or 32-bit ids (ops.sent(..., u16=True)), widened in place.
    Rows that would end beyond their segment or the buffer come out as all pad; `status` (int32 [1], zeroed by the
    caller) gets bit 1 then.  u16=True (gtok_unpack_rows_u16): the slab holds 16-bit ids (int16 storage, as ops.sent(...,
    u16=True) writes it) - half the bytes of the pass."""
    _need_gpu(packed, "unpack_rows")
    dev, rows = packed.device, int(ln.numel())
    eb = packed.element_size()
    if eb not in (2, 4):
        raise ValueError("unpack_rows expects int16 or int32 storage")
    want = torch.int16 if u16 else torch.int32
    ids = torch.empty((rows, ld), dtype=want, device=dev) if out is None else out
    if ids.dtype != want:
        raise ValueError(f"out must be {want}")
    fn = lib().gtok_unpack_rows_u16 if u16 else lib().gtok_unpack_rows_checked
    check(fn(packed.data_ptr(), eb, None if row_ptr is None else row_ptr.data_ptr(), ln.data_ptr(), rows,
             int(segment_rows), int(segment_stride), int(packed.numel()), int(pad_id), ids.data_ptr(), int(ld),
             None if status is None else status.data_ptr(), _stream(dev)), "gtok_unpack_rows")
    return ids


def unpack_rows_at(packed: torch.Tensor, row_start: torch.Tensor, ln: torch.Tensor, ld: int, pad_id: int, segment_rows: int = 0,
                   segment_stride: int = 0, out: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None,
                   u16: bool = False) -> torch.Tensor:
    """unpack_rows for rows with explicit starts (gtok_unpack_rows_at; what ops.sent(..., packed=) writes): row r's ids start at
    element (r // segment_rows) * segment_stride + row_start[r] (segment_rows == 0: at row_start[r]); a negative start or a row
    beyond its segment comes out as all pad, status bit 1."""
    _need_gpu(packed, "unpack_rows_at")
    dev, rows = packed.device, int(ln.numel())
    eb = packed.element_size()
    if eb not in (2, 4) or row_start.dtype != torch.int64 or int(row_start.numel()) < rows or not row_start.is_contiguous():
        raise ValueError("unpack_rows_at expects int16 / int32 storage and one contiguous int64 start per row")
    want = torch.int16 if u16 else torch.int32
    ids = torch.empty((rows, ld), dtype=want, device=dev) if out is None else out
    if ids.dtype != want:
        raise ValueError(f"out must be {want}")
    check(lib().gtok_unpack_rows_at(packed.data_ptr(), eb, row_start.data_ptr(), ln.data_ptr(), rows, int(segment_rows), int(segment_stride),
                                    int(packed.numel()), int(pad_id), ids.data_ptr(), 2 if u16 else 4, int(ld),
                                    None if status is None else status.data_ptr(), _stream(dev)), "gtok_unpack_rows_at")
    return ids


def collate_packed(packed: torch.Tensor, row_ptr: Optional[torch.Tensor], ln: torch.Tensor, ld: int, index: torch.Tensor,
                   pad_id: int, out_ld: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """ops.collate over the packed form: rows `index` -> (X int64 [B, out_ld], attn bool [B, out_ld]).  row_ptr=None: the
    strided form - `packed` is a [rows, ld] slab of 16- or 32-bit ids read in place (ops.sent(..., u16=True))."""
    _need_gpu(packed, "collate_packed")
    dev = packed.device
    index = index.to(dev, dtype=torch.int64).contiguous()
    B = int(index.numel())
    X = torch.empty((B, out_ld), dtype=torch.int64, device=dev)
    A = torch.empty((B, out_ld), dtype=torch.bool, device=dev)
    check(lib().gtok_collate_packed(packed.data_ptr(), packed.element_size(), None if row_ptr is None else row_ptr.data_ptr(),
                                    ln.data_ptr(), int(ld), index.data_ptr(), B, int(pad_id), X.data_ptr(), A.data_ptr(), int(out_ld),
                                    _stream(dev)), "gtok_collate_packed")
    return X, A


def collate_batch(packed: torch.Tensor, row_ptr: Optional[torch.Tensor], ln: torch.Tensor, ld: int, index_host: np.ndarray, pad_id: int,
                  out_ld: int, y: Optional[torch.Tensor] = None):
    """collate_packed for a row list that is a HOST array (int64 numpy, what a DataLoader's sampler produced): the indices travel
    in the kernel's arguments (gtok_collate_batch: no upload) and the labels y[index] come out of the same launch.
    -> (X int64 [B, out_ld], attn bool [B, out_ld], Y or None)."""
    _need_gpu(packed, "collate_batch")
    dev = packed.device
    idx = np.ascontiguousarray(index_host, dtype=np.int64)
    B = int(idx.size)
    X = torch.empty((B, out_ld), dtype=torch.int64, device=dev)
    A = torch.empty((B, out_ld), dtype=torch.bool, device=dev)
    Y = None
    if y is not None:
        if y.dim() != 1 or not y.is_contiguous() or y.element_size() not in (4, 8):
            raise ValueError("y must be a contiguous 1-D tensor of 4- or 8-byte elements")
        Y = torch.empty(B, dtype=y.dtype, device=dev)
    check(lib().gtok_collate_batch(packed.data_ptr(), packed.element_size(), None if row_ptr is None else row_ptr.data_ptr(), ln.data_ptr(), int(ld),
                                   idx.ctypes.data, B, int(ln.numel()), int(pad_id), X.data_ptr(), A.data_ptr(), int(out_ld),
                                   None if y is None else y.data_ptr(), 0 if y is None else y.element_size(), None if Y is None else Y.data_ptr(),
                                   _stream(dev)), "gtok_collate_batch")
    return X, A, Y


def collate_epoch(packed: torch.Tensor, row_ptr: Optional[torch.Tensor], ln: torch.Tensor, ld: int, order: torch.Tensor, batch_size: int,
                  pad_id: int):
    """Every batch of an epoch collated by ONE call (gtok_collate_epoch): rows `order` (int64, on the device) cut into batches of
    batch_size -> (X int64 arena, attn bool arena, lmax list[int], off list[int]): batch b is
    X.as_strided((B_b, lmax[b]), (lmax[b], 1), off[b]) - what collate_packed(index=order[b * batch_size:...], out_ld=lmax[b]) gives.
    row_ptr=None: `packed` is a [rows, ld] slab of 16- / 32-bit ids (ops.sent(..., u16=True)).  One small read-back per EPOCH
    (the batch widths), none per batch."""
    _need_gpu(packed, "collate_epoch")
    dev = packed.device
    order = order.to(dev, dtype=torch.int64).contiguous()
    n, bs = int(order.numel()), int(batch_size)
    nb = -(-n // bs) if n else 0
    lmax = torch.empty(max(nb, 1), dtype=torch.int32, device=dev)
    off = torch.empty(nb + 1, dtype=torch.int64, device=dev)
    L = lib()
    check(L.gtok_collate_epoch_plan(ln.data_ptr(), int(ld), order.data_ptr(), n, bs, lmax.data_ptr(), off.data_ptr(), _stream(dev)),
          "gtok_collate_epoch_plan")
    off_h = off.tolist()                       # the one synchronisation of the epoch: the arena is sized exactly
    total = off_h[-1]
    X = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    A = torch.empty(max(total, 1), dtype=torch.bool, device=dev)
    check(L.gtok_collate_epoch(packed.data_ptr(), packed.element_size(), None if row_ptr is None else row_ptr.data_ptr(), ln.data_ptr(), int(ld),
                               order.data_ptr(), n, bs, int(pad_id), lmax.data_ptr(), off.data_ptr(), X.data_ptr(), A.data_ptr(), total,
                               _stream(dev)), "gtok_collate_epoch")
    return X, A, lmax[:nb].tolist(), off_h


def zinc_text_tails(y: torch.Tensor, ln: torch.Tensor, max_len: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """What follows `<p>` in every ZINC text, rendered on the device (gtok_zinc_text_tails; reference
    zinc_dataset_indexbase.py:186-195, :217-221): y float32 [G] labels, ln int32 [G] ids per row (gtok_ibtt_zinc's lengths).
    Returns (take int32 [G], suf_bytes uint8 [total], suf_ptr int64 [G + 1]) on the device: ids_to_text(ids, take, strings,
    (suf_bytes, suf_ptr)) gives exactly the strings ZINCTokenizationDataset.__getitem__ returns."""
    _need_gpu(ln, "zinc_text_tails")
    if y.dtype != torch.float32 or ln.dtype != torch.int32 or y.numel() != ln.numel():
        raise ValueError("zinc_text_tails expects float32 labels and int32 lengths of one size")
    dev, G = ln.device, int(ln.numel())
    y, ln = y.to(dev).contiguous(), ln.contiguous()
    take = torch.empty(G, dtype=torch.int32, device=dev)
    slen = torch.empty(G, dtype=torch.int64, device=dev)
    L = lib()
    max_len = int(min(max_len, 2 ** 31 - 1))
    check(L.gtok_zinc_text_tails(y.data_ptr(), ln.data_ptr(), G, int(max_len), take.data_ptr(), None, None, slen.data_ptr(),
                                 _stream(dev)), "gtok_zinc_text_tails")
    sp = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    torch.cumsum(slen, 0, out=sp[1:])
    sb = torch.empty(max(56 * G, 1), dtype=torch.uint8, device=dev)      # a tail is at most 56 bytes: no size round trip
    check(L.gtok_zinc_text_tails(y.data_ptr(), ln.data_ptr(), G, int(max_len), take.data_ptr(), sp.data_ptr(), sb.data_ptr(), None,
                                 _stream(dev)), "gtok_zinc_text_tails")
    return take, sb, sp


def ids_to_text(ids: torch.Tensor, take: torch.Tensor, strings: Sequence[str], suffixes=None):
    """Rows of ids -> texts on the device (gtok_ids_to_text): row r = the strings of its first take[r] ids joined by single
    spaces + suffixes[r] verbatim.  Returns (blob uint8 [total], text_ptr int64 [rows + 1]) on the device; text r is
    blob[text_ptr[r] : text_ptr[r + 1]].  `strings[t]` is the text of id t (ASCII).  `suffixes`: one bytes object per row,
    or (suf_bytes uint8, suf_ptr int64 [rows + 1]) already on the device (zinc_text_tails)."""
    _need_gpu(ids, "ids_to_text")
    if ids.dtype != torch.int32 or ids.dim() != 2 or not ids.is_contiguous() or take.dtype != torch.int32:
        raise ValueError("ids_to_text expects a contiguous int32 [rows, ld] slab and int32 counts")
    dev, (rows, ld) = ids.device, ids.shape
    enc = [t.encode("ascii") for t in strings]
    tptr = np.zeros(len(enc) + 1, np.int32); np.cumsum([len(b) for b in enc], out=tptr[1:])
    tab = torch.frombuffer(bytearray(b"".join(enc) or b"\0"), dtype=torch.uint8).to(dev)
    tab_ptr = torch.from_numpy(tptr).to(dev)
    sb = sp = None
    if isinstance(suffixes, tuple) and len(suffixes) == 2 and torch.is_tensor(suffixes[0]):
        sb, sp = suffixes
        if sb.dtype != torch.uint8 or sp.dtype != torch.int64 or sp.numel() != rows + 1 or sb.device != dev or sp.device != dev:
            raise ValueError("device suffixes are (uint8 bytes, int64 [rows + 1] offsets) on the ids' device")
        sb, sp = sb.contiguous(), sp.contiguous()
    elif suffixes is not None:
        if len(suffixes) != rows:
            raise ValueError("one suffix per row")
        sptr = np.zeros(rows + 1, np.int64); np.cumsum(np.fromiter(map(len, suffixes), np.int64, rows), out=sptr[1:])
        sb = torch.frombuffer(bytearray(b"".join(suffixes) or b"\0"), dtype=torch.uint8).to(dev)
        sp = torch.from_numpy(sptr).to(dev)
    tlen = torch.empty(rows, dtype=torch.int64, device=dev)
    L = lib()
    args = (ids.data_ptr(), ld, take.data_ptr(), rows, tab.data_ptr(), tab_ptr.data_ptr(), len(enc),
            None if sb is None else sb.data_ptr(), None if sp is None else sp.data_ptr())
    check(L.gtok_ids_to_text(*args, None, None, tlen.data_ptr(), _stream(dev)), "gtok_ids_to_text")
    text_ptr = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(tlen, 0, out=text_ptr[1:])
    total = int(text_ptr[-1])
    blob = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    check(L.gtok_ids_to_text(*args, text_ptr.data_ptr(), blob.data_ptr(), None, _stream(dev)), "gtok_ids_to_text")
    return blob[:total], text_ptr


# ------------------------------------------------------------------------------------------------
# text -> ids (TokenDataset)
# ------------------------------------------------------------------------------------------------
NO_LABEL = -2 ** 31


def parse_graph_texts(text_bytes: torch.Tensor, text_ptr: torch.Tensor) -> Dict[str, torch.Tensor]:
    """graph-token texts -> edges / node counts / query nodes / labels on the device (the reference's
    parse_graph_from_text + parse_query_nodes_from_text + parse_label_from_text + the num_nodes rule).  Returns
    device tensors: num_edges, num_nodes, label (NO_LABEL = none), status int32 [G]; query int32 [G, 2] (-1 = none);
    edge_ptr int64 [G+1]; src, dst int32 [sum E] in text order.  status != 0 marks a text outside the canonical
    grammar: ignore its other fields and parse it on the host."""
    _need_gpu(text_bytes, "parse_graph_texts")
    dev = text_bytes.device
    text_ptr = text_ptr.to(dev, dtype=torch.int64).contiguous()
    G = int(text_ptr.numel()) - 1
    i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    ne, nn, q, lab, st = i32(G), i32(G), i32(G, 2), i32(G), i32(G)
    L = _lib.lib()
    edge_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    if not G:
        return dict(num_edges=ne, num_nodes=nn, query=q, label=lab, status=st, edge_ptr=edge_ptr, src=i32(0), dst=i32(0))
    # sizing: one streaming pass that counts the `<e>` tokens (exact for canonical texts), then ONE parse that fills
    cnt = i32(G)
    _lib.check(L.gtok_count_edge_tokens(text_bytes.data_ptr(), text_ptr.data_ptr(), G, cnt.data_ptr(), _stream(dev)),
               "gtok_count_edge_tokens")
    torch.cumsum(cnt, 0, dtype=torch.int64, out=edge_ptr[1:])
    # No host round trip for the size: an edge of a text the parser accepts is at least 8 bytes (`0 1 <e> `), so the edge
    # arrays are sized by the bytes; a corpus with enough junk to count more `<e>` than that has its ranges clamped for this
    # pass and is parsed again with exact arrays (seen in the one read-back below).
    bound = int(text_bytes.numel()) // 8 + G + 1
    clamped = edge_ptr.clamp(max=bound)

    def parse(ptr, n_edges):
        s_, d_ = i32(max(n_edges, 1)), i32(max(n_edges, 1))
        _lib.check(L.gtok_parse_graph_text(text_bytes.data_ptr(), text_ptr.data_ptr(), G, ptr.data_ptr(), s_.data_ptr(),
                                           d_.data_ptr(), ne.data_ptr(), nn.data_ptr(), q.data_ptr(), lab.data_ptr(),
                                           st.data_ptr(), _stream(dev)), "gtok_parse_graph_text")
        return s_, d_
    src, dst = parse(clamped, bound)
    # every canonical text owns exactly its first num_edges slots; normally that is its whole range (one `<e>` token per
    # edge).  Texts outside the grammar (status != 0) own nothing, and a canonical text may carry a stray `<e>` where the
    # grammar allows any word (right after <q> / <p>): then the ranges are squeezed.
    keep = torch.where(st == 0, ne, torch.zeros_like(ne))
    cap, differ = torch.stack([edge_ptr[-1], (keep != cnt).any().to(torch.int64)]).tolist()
    if cap > bound:
        src, dst = parse(edge_ptr, cap)
        keep = torch.where(st == 0, ne, torch.zeros_like(ne))
        differ = int(not torch.equal(keep, cnt))
    if not differ:
        return dict(num_edges=ne, num_nodes=nn, query=q, label=lab, status=st, edge_ptr=edge_ptr, src=src[:cap], dst=dst[:cap])
    keep = keep.to(torch.int64)
    new_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
    torch.cumsum(keep, 0, out=new_ptr[1:])
    total = int(new_ptr[-1])
    idx = torch.repeat_interleave(edge_ptr[:-1] - new_ptr[:-1], keep, output_size=total) + torch.arange(total, device=dev)
    return dict(num_edges=ne, num_nodes=nn, query=q, label=lab, status=st, edge_ptr=new_ptr, src=src[idx], dst=dst[idx])


def find_token(x: torch.Tensor, token: int) -> torch.Tensor:
    """First column of `token` in every row of an int64 [B, L] batch (-1: absent): the `<q>` search of
    trainer/train_ibtt.py:88-103 / train_agtt.py:78-114 as one launch (query nodes sit at pos + 2, pos + 3)."""
    _need_gpu(x, "find_token")
    if x.dtype != torch.int64 or x.dim() != 2 or not x.is_contiguous():
        raise ValueError("find_token expects a contiguous int64 [B, L] tensor")
    pos = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().gtok_find_token(x.data_ptr(), int(x.shape[0]), int(x.shape[1]), int(token), pos.data_ptr(),
                                          _stream(x.device)), "gtok_find_token")
    return pos


def vocab_stats_synth(batch: GraphBatch, num_ids: int, query_nodes: Optional[torch.Tensor] = None, graph_base: int = 0,
                      out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Occurrence count and first position ((graph_base + g) << 32 | position in the text) of every node-id token
    `str(i)`, i < num_ids, in the graph-token texts this batch stands for: the corpus pass of
    build_vocab_from_texts (data_loader.py:451-463) on the device.  query_nodes: int32 [G, 2] (u, v), negative = none.
    `out` = (count, first) int64 [num_ids] to ACCUMULATE into (shards / ranks); fresh tables otherwise."""
    dev = batch.device
    _need_gpu(batch.col, "vocab_stats_synth")
    if out is None:
        count = torch.zeros(num_ids, dtype=torch.int64, device=dev)
        first = torch.full((num_ids,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    else:
        count, first = out
        if count.dtype != torch.int64 or first.dtype != torch.int64 or count.numel() != num_ids or first.numel() != num_ids:
            raise ValueError("out must be two int64 [num_ids] tensors")
    if query_nodes is not None:
        if query_nodes.dtype != torch.int32 or tuple(query_nodes.shape) != (batch.num_graphs, 2):
            raise ValueError("query_nodes must be int32 [G, 2]")
        query_nodes = query_nodes.to(dev).contiguous()
    cs = batch.c_struct()
    _lib.check(_lib.lib().gtok_vocab_stats_synth(ctypes.byref(cs), None if query_nodes is None else query_nodes.data_ptr(),
                                                 int(graph_base), int(num_ids), count.data_ptr(), first.data_ptr(),
                                                 _stream(dev)), "gtok_vocab_stats_synth")
    return count, first


def vocab_stats_text(text_bytes: torch.Tensor, text_ptr: torch.Tensor, capacity: int = 1 << 16, base_offset: int = 0,
                     out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """Count and first position of every distinct whitespace-separated token of the texts (gtok_vocab_stats_text): the
    corpus pass of build_vocab_from_texts (data_loader.py:451-463) and of the ZINC dynamic-token scan
    (trainer/train_ibtt.py:361-372) on the device.  Returns the open-addressing table as device tensors
    {key uint64-as-int64, count, first int64, len int32, status int32 [1]} of `capacity` slots; pass it back as `out`
    to accumulate further shards (base_offset = the shard's byte offset in the whole corpus).  text_stats_entries()
    turns a finished table into (token, count, first) triples.  Tokens are told apart by a 64-bit hash AND verified byte
    for byte against the slot's first occurrence (status bit 2 = a collision; text_stats_entries raises) - within one
    call; with shards accumulated over several calls a cross-shard collision is not seen, and text_stats_entries needs
    the WHOLE corpus blob, since a slot's `first` may lie in any shard."""
    _need_gpu(text_bytes, "vocab_stats_text")
    dev = text_bytes.device
    text_ptr = text_ptr.to(dev, dtype=torch.int64).contiguous()
    G = int(text_ptr.numel()) - 1
    if capacity < 16 or capacity & (capacity - 1):
        raise ValueError("capacity must be a power of two >= 16")
    if out is None:
        out = dict(key=torch.zeros(capacity, dtype=torch.int64, device=dev), count=torch.zeros(capacity, dtype=torch.int64, device=dev),
                   first=torch.full((capacity,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev),
                   len=torch.zeros(capacity, dtype=torch.int32, device=dev), status=torch.zeros(1, dtype=torch.int32, device=dev))
    elif out["key"].numel() != capacity:
        raise ValueError("out was built for another capacity")
    _lib.check(_lib.lib().gtok_vocab_stats_text(text_bytes.data_ptr(), text_ptr.data_ptr(), G, int(base_offset), capacity,
                                                out["key"].data_ptr(), out["count"].data_ptr(), out["first"].data_ptr(),
                                                out["len"].data_ptr(), out["status"].data_ptr(), _stream(dev)),
               "gtok_vocab_stats_text")
    return out


def text_stats_entries(table: Dict[str, torch.Tensor], text_bytes: torch.Tensor, base_offset: int = 0):
    """[(token, count, first)] of a finished gtok_vocab_stats_text table, in Counter.most_common order (count
    descending, first appearance ascending).  The token strings are read back from the corpus blob at `first`
    (only the table and one byte range per DISTINCT token cross to the host).  Raises when the table overflowed."""
    st = int(table["status"].item())
    if st & 4:
        raise _lib.GtokError("vocab_stats_text: two different tokens share a 64-bit identity (hash collision): "
                             "build this vocab with the host function (build_vocab_from_texts)")
    if st:
        raise _lib.GtokError(f"vocab_stats_text: table {'overflowed' if st & 1 else 'was not consistent'} (status {st}): use a larger capacity")
    used = (table["key"] != 0).nonzero(as_tuple=True)[0]
    cnt, fst, ln = table["count"][used], table["first"][used], table["len"][used].to(torch.int64)
    order = torch.argsort(fst)                      # stable secondary key first ...
    order = order[torch.argsort(cnt[order], descending=True, stable=True)]
    cnt, fst, ln = cnt[order].cpu().tolist(), fst[order].cpu().tolist(), ln[order].cpu().tolist()
    # gather the token bytes on the device: one [sum len] buffer instead of a host copy of the corpus
    lens = torch.tensor(ln, dtype=torch.int64, device=text_bytes.device)
    starts = torch.tensor(fst, dtype=torch.int64, device=text_bytes.device) - base_offset
    ptr = torch.zeros(len(ln) + 1, dtype=torch.int64, device=text_bytes.device)
    torch.cumsum(lens, 0, out=ptr[1:])
    idx = torch.arange(int(ptr[-1]), device=text_bytes.device)
    tok = torch.repeat_interleave(torch.arange(len(ln), device=text_bytes.device), lens, output_size=int(ptr[-1]))
    blob = bytes(text_bytes[starts[tok] + (idx - ptr[tok])].cpu().numpy())
    p = ptr.cpu().tolist()
    return [(blob[p[i]:p[i + 1]].decode("utf-8"), cnt[i], fst[i]) for i in range(len(ln))]


def _fnv1a(b: bytes) -> int:
    h = 2166136261
    for c in b:
        h = ((h ^ c) * 16777619) & 0xFFFFFFFF
    return h


class VocabTable:
    """Open-addressing table (FNV-1a, linear probing) for gtok_text_to_ids, built on the host."""

    def __init__(self, vocab: Dict[str, int], device):
        self.pad_id = int(vocab["<pad>"])
        cap = 16
        while cap < 2 * max(1, len(vocab)):
            cap *= 2
        key_off = np.full(cap, -1, np.int32); key_len = np.zeros(cap, np.int32); ids = np.zeros(cap, np.int32)
        blob = bytearray()
        for tok, i in vocab.items():
            b = tok.encode("utf-8")
            if not b or any(c in b" \t\n\r\x0b\x0c\x1c\x1d\x1e\x1f" for c in b):
                continue  # can never come out of str.split()
            s = _fnv1a(b) & (cap - 1)
            while key_off[s] >= 0:
                s = (s + 1) & (cap - 1)
            key_off[s], key_len[s], ids[s] = len(blob), len(b), i
            blob += b
        self.capacity = cap
        t = lambda a: torch.from_numpy(a).to(device)
        self.key_off, self.key_len, self.ids = t(key_off), t(key_len), t(ids)
        self.key_bytes = torch.frombuffer(bytearray(blob or b"\0"), dtype=torch.uint8).clone().to(device)

    def c_struct(self) -> GtokVocabTable:
        return GtokVocabTable(self.capacity, self.pad_id, self.key_off.data_ptr(), self.key_len.data_ptr(),
                              self.ids.data_ptr(), self.key_bytes.data_ptr())


def pack_texts(texts: Sequence[str]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Concatenate texts into (bytes uint8 [total], text_ptr int64 [G+1]); ASCII only (str.split() on
    non-ASCII whitespace is not reproduced on the device)."""
    try:
        blob = "".join(texts).encode("ascii")        # one join + one encode: ASCII, so a text's length in bytes is len(text)
    except UnicodeEncodeError:
        raise ValueError("gtok text path handles ASCII text only") from None
    ptr = np.zeros(len(texts) + 1, np.int64)
    np.cumsum(np.fromiter(map(len, texts), np.int64, len(texts)), out=ptr[1:])
    return torch.frombuffer(bytearray(blob or b"\0"), dtype=torch.uint8), torch.from_numpy(ptr)


def text_to_ids(text_bytes: torch.Tensor, text_ptr: torch.Tensor, table: VocabTable, max_len: int,
                strip_label: bool = True, ld: Optional[int] = None, out=None) -> Tuple[torch.Tensor, torch.Tensor]:
    _need_gpu(text_bytes, "text_to_ids")
    dev = text_bytes.device
    text_ptr = text_ptr.to(dev, dtype=torch.int64).contiguous()
    G = int(text_ptr.numel()) - 1
    if ld is None:
        ld = _round4(max_len)
    ids, ln = _alloc_out(G, ld, dev, out)
    vs = table.c_struct()
    check(lib().gtok_text_to_ids(text_bytes.data_ptr(), text_ptr.data_ptr(), G, ctypes.byref(vs), int(strip_label),
                                 max_len, ids.data_ptr(), ld, ln.data_ptr(), _stream(dev)), "gtok_text_to_ids")
    return ids, ln
