"""`Graph2TrailTokenizer` with the interface trainer/train_agtt.py uses (ctor kwargs :514-530,
set_num_nodes :535, set_num_node_and_edge_types :540, idx/node/edge offsets :189-191, class attr `pad`
:286, `tokenizer(data) -> 1-D LongTensor` :250), backed by the gfx950 SENT kernel.

This is NOT upstream AutoGraph: the walk follows the spec in DESIGN.md §SENT (upstream is un-vendored and
un-pinned in the reference, so token-for-token parity with it is unpinned — SURVEY.md §8c).
"""
from typing import Optional, Sequence

import torch

# the root package is found by path, so this file works both as glearning-benchmark_amd.autograph... and as
# the top-level `autograph` package trainer/train_agtt.py:21 imports (drop-in sys.path layout)
import importlib
import os
import sys
import weakref

_PKG_DIR = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
if os.path.basename(_PKG_DIR) in sys.modules:
    _pkg = sys.modules[os.path.basename(_PKG_DIR)]
else:
    sys.path.insert(0, os.path.dirname(_PKG_DIR))
    _pkg = importlib.import_module(os.path.basename(_PKG_DIR))
_ops, _rows, GtokError, GraphBatch = _pkg.ops, _pkg.rows, _pkg.GtokError, _pkg.GraphBatch


class Graph2TrailTokenizer:
    sos, reset, ladj, radj, eos, pad = 0, 1, 2, 3, 4, 5

    def __init__(self, dataset_names: Optional[Sequence[str]] = None, max_length: int = -1,
                 truncation_length: Optional[int] = None, labeled_graph: bool = False, undirected: bool = True,
                 seed: int = 0, device=None, epochs_per_launch: Optional[int] = None, **unused):
        if dataset_names:
            raise ValueError("dataset-name tokens are not supported (the reference always passes dataset_names=[])")
        if not undirected:
            raise ValueError("only undirected=True is implemented (what the reference uses)")
        self.max_length = max_length
        self.truncation_length = truncation_length if truncation_length is not None else max_length
        self.labeled_graph = labeled_graph
        self.undirected = True
        self.idx_offset = 6
        self.max_num_nodes = None
        self.node_idx_offset = None
        self.edge_idx_offset = None
        self.num_node_types = 0
        self.num_edge_types = 0
        self.seed = seed
        self.device = device
        # K epochs of a split per gtok_sent launch (gtok_sent_params.epoch_count): None = as many as fill the chip once
        # (EPOCH_WALKS walks), 1 = an epoch per launch
        self.epochs_per_launch = epochs_per_launch
        self._calls = 0
        self.launches = 0                        # gtok_sent launches issued so far (tests / bench read it)
        self._splits = weakref.WeakKeyDictionary()    # dataset -> _Split: the split's resident CSR and its current epoch

    # ---- configuration (same call order as train_agtt.py:534-540)
    def set_num_nodes(self, max_num_nodes: int) -> None:
        self.max_num_nodes = int(max_num_nodes)
        self.node_idx_offset = self.idx_offset + self.max_num_nodes
        self.edge_idx_offset = self.node_idx_offset + self.num_node_types

    def set_num_node_and_edge_types(self, num_node_types: int, num_edge_types: int) -> None:
        if self.max_num_nodes is None:
            raise RuntimeError("call set_num_nodes() first")
        self.num_node_types, self.num_edge_types = int(num_node_types), int(num_edge_types)
        self.edge_idx_offset = self.node_idx_offset + self.num_node_types

    def __len__(self) -> int:
        """Size of the raw (un-remapped) id space."""
        return (self.edge_idx_offset or self.idx_offset) + self.num_edge_types

    def _max_len(self) -> int:
        m = self.truncation_length if self.truncation_length and self.truncation_length > 0 else self.max_length
        return int(m) if m and m > 0 else 1 << 20

    def _device(self):
        if self.device is not None:
            return torch.device(self.device)
        if not torch.cuda.is_available():
            raise GtokError("Graph2TrailTokenizer runs on the GPU and no GPU is visible; there is no CPU path")
        return torch.device("cuda", torch.cuda.current_device())

    # ---- batched fast path: one launch for a whole split / epoch
    def tokenize_batch(self, batch: "GraphBatch", epoch: int = 0, graph_base: int = 0, remap_zinc: bool = False,
                       query: Optional[torch.Tensor] = None, ld: Optional[int] = None, out=None, pad: bool = True,
                       epochs: int = 1, u16: bool = False):
        """(ids int32 [G, ld], len int32 [G]) on the device; trail g is a function of (seed, epoch, graph_base+g).
        pad=False: rows are only written up to their length (consumers that read through `len`: ops.collate).
        epochs=K > 1: epochs epoch .. epoch + K - 1 in ONE launch -> ([K, G, ld], [K, G]); u16: rows of 16-bit ids."""
        if self.max_num_nodes is None:
            raise RuntimeError("call set_num_nodes() first")
        self.launches += 1
        return _ops.sent(batch, self.max_num_nodes, self._max_len(), self.seed, epoch, labeled=self.labeled_graph,
                         num_node_types=self.num_node_types, num_edge_types=self.num_edge_types,
                         remap_zinc=remap_zinc, pad_id=self.pad, graph_base=graph_base, query=query, ld=ld, out=out, pad=pad,
                         epochs=epochs, u16=u16)

    EPOCH_WALKS = 1 << 22         # walks per launch: sixteen rounds of the 4,096 resident waves x 64 lanes of sent_lane_kernel - units that
                                  # are staged and padded while other waves walk cost less than those of a one-round launch, and the
                                  # launch's ragged end is shared by more epochs (ZINC-full as 16-bit rows, per epoch: 0.0653 ms at one
                                  # epoch per launch, 0.0573 at 4, 0.0535 at 8, 0.0516 at 16)
    EPOCH_SLAB_BYTES = 4 << 30    # ... as long as the K-epoch slab of 16-bit rows stays below this - 1.5 % of the device's 288 GB (125 k
                                  # graph-token graphs x 608 ids: 28 epochs, 0.290 / 0.327 ms per epoch on the ER / family-mix corpora against
                                  # 0.299 / 0.337 at the 14 epochs a 2 GiB bound allowed; x 1,040 ids: 16)

    @classmethod
    def epochs_for_shape(cls, num_graphs: int, ld: Optional[int] = None) -> int:
        k = cls.EPOCH_WALKS // max(1, int(num_graphs))
        if ld:
            k = min(k, cls.EPOCH_SLAB_BYTES // max(1, 2 * int(num_graphs) * int(ld)))
        return max(1, min(32, k))

    def epochs_for(self, num_graphs: int, ld: Optional[int] = None) -> int:
        """How many epochs of a split of `num_graphs` graphs (rows of `ld` ids) one launch should carry.  The reference
        re-tokenizes a split every epoch (trainer/train_agtt.py:246-250, epoch loop :676-680) and a trail depends on (seed,
        epoch, graph) only: a 12 k-molecule split (configs/agtt_zinc.yaml:4 `subset: true`) tokenizes 32 epochs in the time of
        four."""
        if self.epochs_per_launch is not None:
            return max(1, int(self.epochs_per_launch))
        return self.epochs_for_shape(num_graphs, ld)

    # ---- reference call site: one Data in, one 1-D LongTensor out, a fresh random trail per call
    def _signature(self):
        return (self.max_num_nodes, self.labeled_graph, self.num_node_types, self.num_edge_types, self._max_len(), self.seed)

    def _serve(self, owner, idx: int):
        """Row `idx` of the item's split.  The split is tokenized as a whole - ONE launch per K epochs (epochs_for) - the
        first time an item of it is asked for; an item asked for a second time starts the next epoch (in the reference
        every fetch is a new random trail: a second fetch of an item is the next epoch), which is served from the same
        launch's next slice until its K epochs are used up."""
        sp = self._splits.get(owner)
        if sp is None or sp.signature != self._signature() or sp.batch.num_graphs != len(owner):
            gb = getattr(owner, "graph_batch", None)
            if not callable(gb):
                return None
            sp = _Split(gb(device=self._device(), labeled=self.labeled_graph), self._signature())
            self._splits[owner] = sp
        if not 0 <= idx < sp.batch.num_graphs:
            return None
        row = sp.rows.take(idx) if sp.rows is not None else None
        if row is None:
            sp.epoch += 1
            if sp.slab is None or not sp.first <= sp.epoch < sp.first + sp.slab[0].shape[0]:
                K = self.epochs_for(sp.batch.num_graphs, _ops.sent_safe_ld(sp.batch, self.labeled_graph, self._max_len()))
                ids, ln = self.tokenize_batch(sp.batch, epoch=sp.epoch, pad=False, epochs=K, u16=True)
                G = sp.batch.num_graphs
                sp.slab, sp.first = (ids.view(K, G, -1), ln.view(K, G)), sp.epoch
            e = sp.epoch - sp.first
            sp.rows = _rows.EpochRows(sp.slab[0][e], sp.slab[1][e], sp.epoch)     # one packed D2H copy per epoch
            row = sp.rows.take(idx)
        return row

    def tokenize(self, data) -> torch.Tensor:
        """`tokens = tokenizer(data)` (trainer/train_agtt.py:250).  An item that came out of one of this package's
        dataset classes (they mark what they return: rows.tag_item) is served from its split's epoch - one launch
        per split and epoch however the items are fetched; any other object is tokenized on its own (one small
        launch per call)."""
        src = _rows.item_source(data)
        if src is not None:
            row = self._serve(*src)
            if row is not None:
                return row
        batch = GraphBatch.from_data_list([data], labeled=self.labeled_graph).to(self._device())
        ids, ln = self.tokenize_batch(batch, epoch=0, graph_base=self._calls)
        self._calls += 1
        return ids[0, :int(ln[0])].to(torch.long).cpu()

    __call__ = tokenize


class _Split:
    __slots__ = ("batch", "signature", "epoch", "rows", "slab", "first")

    def __init__(self, batch, signature):
        self.batch, self.signature, self.epoch, self.rows = batch, signature, -1, None
        self.slab, self.first = None, 0          # the K-epoch device slab of the last launch and the epoch of its slice 0
