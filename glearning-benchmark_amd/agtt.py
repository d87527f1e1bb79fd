"""AGTT dataset glue — `TokenizedGraphDataset` and `collate_fn` with the signatures of
trainer/train_agtt.py:150-302, tokenizing whole epochs on the GPU.

Reference behaviour kept: a fresh random trail every time an item is fetched, ZINC ids remapped to the
fixed vocabulary (remap_zinc_tokens :171-244), `[idx_offset+num_nodes, idx_offset+u, idx_offset+v]`
appended for shortest_path (:257-267), padding with Graph2TrailTokenizer.pad = 5 even after the remap
(:285-286), float32 labels iff the first label is a Python float (:296-299).
"""
from collections.abc import Sequence
from typing import Optional

import numpy as np
import torch
from torch.utils.data import Dataset

from . import ops as _ops
from . import rows as _rows
from . import csr as _csr
from .csr import GraphBatch
from .graph_data_loader.zinc_vocab import build_fixed_zinc_vocab

PAD = 5  # Graph2TrailTokenizer.pad


class TokenizedGraphDataset(Dataset):
    def __init__(self, pyg_dataset, tokenizer, task="cycle_check", remap_to_fixed_vocab=False, device=None):
        self.pyg_dataset, self.tokenizer, self.task = pyg_dataset, tokenizer, task
        self.remap_to_fixed_vocab = remap_to_fixed_vocab
        if remap_to_fixed_vocab:
            self.fixed_vocab, _ = build_fixed_zinc_vocab()
            print("[TokenizedGraphDataset] Using fixed vocabulary remapping for ZINC")
        self._batched = hasattr(tokenizer, "tokenize_batch")
        self._device = device
        self._batch: Optional[GraphBatch] = None
        self._query = None
        self._mixed_query = False
        self._epoch = -1
        self._rows = None                      # rows.EpochRows: the host view __getitem__ serves from
        self._ids = self._lens = None
        self._slab = None                      # (ids16 [K, G, ld], len [K, G], first epoch, padded): the last K-epoch launch
        self._slab_sig = None                  # ... and what it was made with (tokenizer configuration, remap switch, query table)
        self._served = None                    # __getitems__: which rows of the current epoch went out already
        self._plan = None                      # plan_epoch: the announced batches of an epoch and, once asked for, their arenas
        self._y_dev = None                     # labels on the device (batch-level fetch)
        self._lens_h = None                    # the current epoch's row lengths on the host (batch-level fetch)

    def __len__(self):
        return len(self.pyg_dataset)

    # ---- whole-epoch tokenization (one SENT launch; remap and query append fused into it)
    def _dev(self):
        if self._device is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        return torch.device(self._device)

    def _graphs(self) -> GraphBatch:
        """The split as a device-resident batch, built once: from the dataset's own `graph_batch()` (this package's
        dataset classes: collated storage, no per-item work) or from whatever storage / items it exposes."""
        if self._batch is None:
            ds = self.pyg_dataset
            gb = getattr(ds, "graph_batch", None)
            if callable(gb):
                self._batch = gb(device=self._dev(), labeled=self.tokenizer.labeled_graph)
            else:
                self._batch = GraphBatch.from_dataset(ds, labeled=self.tokenizer.labeled_graph, device=self._dev())
            if self.task == "shortest_path":
                q = getattr(ds, "queries", None)
                q = q() if callable(q) else None
                if q is not None:
                    self._query = torch.as_tensor(q, dtype=torch.int32)
                else:
                    items = [ds[i] for i in range(len(self))]
                    has = [hasattr(d, "query_u") and hasattr(d, "query_v") for d in items]
                    if items and all(has):
                        self._query = torch.tensor([[d.query_u, d.query_v] for d in items], dtype=torch.int32)
                    self._mixed_query = any(has) and not all(has)
        return self._batch

    def tokenize_epoch(self, epoch: int, pad: bool = True):
        """(ids int32 [G, ld], len int32 [G]) on the device for `epoch`.  pad=True (default): the documented slab,
        pad id 5 behind every row.  pad=False: rows are only written up to their length (rounded up to 16 ids) and
        the rest of the slab is UNINITIALISED - for readers that go through `len` (gtok_collate, gtok_pack_rows:
        what __getitem__ and device_batches use), never for code that consumes the slab whole."""
        batch = self._graphs()
        self._ids, self._lens = self.tokenizer.tokenize_batch(batch, epoch=epoch, remap_zinc=self.remap_to_fixed_vocab,
                                                              query=self._query, pad=pad)
        self._epoch, self._rows, self._served = epoch, None, None
        return self._ids, self._lens

    def tokenize_epoch_u16(self, epoch: int):
        """(ids16 int16 [G, ld], len int32 [G]) on the device for `epoch`: rows of 16-bit ids, written up to their length
        only (GTOK_SENT_U16 | GTOK_SENT_NO_PAD) - what __getitem__ / __getitems__ / device_batches read.  One launch
        carries K epochs (tokenizer.epochs_for: trails depend on (seed, epoch, graph) only, and the trainer asks for the
        same split again every epoch, trainer/train_agtt.py:676-680); the following K - 1 calls are slices of it."""
        batch = self._graphs()
        # what the cached K-epoch slab was made with: a tokenizer reconfigured since (set_num_nodes, seed, truncation_length,
        # labelled types), another remap switch or query table must not be served the old trails for up to K - 1 epochs
        sig = (self.tokenizer._signature() if hasattr(self.tokenizer, "_signature") else None, self.remap_to_fixed_vocab, id(self._query))
        sl = self._slab
        if sl is not None and self._slab_sig != sig:
            sl = self._slab = None
        if sl is None or not sl[2] <= epoch < sl[2] + sl[0].shape[0]:
            self._slab_sig = sig
            K = 1
            if hasattr(self.tokenizer, "epochs_for"):
                K = self.tokenizer.epochs_for(batch.num_graphs, _ops.sent_safe_ld(batch, self.tokenizer.labeled_graph, self.tokenizer._max_len(),
                                                                                   self._query is not None))
            ids, ln = self.tokenizer.tokenize_batch(batch, epoch=epoch, remap_zinc=self.remap_to_fixed_vocab, query=self._query,
                                                    pad=False, epochs=K, u16=True)
            G = batch.num_graphs
            sl = self._slab = (ids.view(K, G, -1), ln.view(K, G), epoch)
        e = epoch - sl[2]
        self._epoch, self._rows, self._served = epoch, None, None
        self._ids, self._lens = sl[0][e], sl[1][e]
        return self._ids, self._lens

    def remap_zinc_tokens(self, tokens: torch.Tensor, data=None) -> torch.Tensor:
        """1-D raw SENT ids -> fixed-vocab ids through gtok_remap_zinc (same table as the fused path)."""
        t = tokens.to(self._dev(), dtype=torch.int32).view(1, -1).contiguous()
        ln = torch.tensor([t.shape[1]], dtype=torch.int32, device=t.device)
        out = _ops.remap_zinc(t, ln, self.tokenizer.idx_offset, self.tokenizer.node_idx_offset,
                              self.tokenizer.edge_idx_offset)
        return out.view(-1).to(torch.long).cpu()

    def _query_tail(self, data) -> Optional[torch.Tensor]:
        if self.task == "shortest_path" and hasattr(data, "query_u") and hasattr(data, "query_v"):
            off = self.tokenizer.idx_offset
            return torch.tensor([off + data.num_nodes, off + data.query_u, off + data.query_v], dtype=torch.long)
        return None

    def __getitem__(self, idx):
        data = self.pyg_dataset[idx]
        if self._batched:
            if torch.utils.data.get_worker_info() is not None:
                # the reference runs this dataset with num_workers: 0 (configs/agtt_*.yaml:21/25, "PyG Data pickling
                # issues"); a worker process would have to launch kernels on a GPU context it must not inherit
                raise _ops._lib.GtokError("TokenizedGraphDataset tokenizes on the GPU in the training process: use "
                                     "DataLoader(num_workers=0), as the reference's AGTT configs do")
            self._graphs()
        if self._batched and not self._mixed_query:
            tokens = self._rows.take(idx) if self._rows is not None else None
            if tokens is None:                                  # first fetch, or fetched again -> a new random trail
                ids, lens = self.tokenize_epoch_u16(self._epoch + 1)
                self._rows = _rows.EpochRows(ids, lens, self._epoch)     # ONE packed D2H copy per epoch
                tokens = self._rows.take(idx)
        else:                                                   # any tokenizer object: per-item call
            tokens = self.tokenizer(data)
            if self.remap_to_fixed_vocab:
                tokens = self.remap_zinc_tokens(tokens, data)
            tail = self._query_tail(data)
            if tail is not None:
                tokens = torch.cat([tokens, tail])
        return tokens, torch.ones(tokens.size(0), dtype=torch.bool), data.y.item(), data

    def _labels_on_device(self):
        """Labels of the split as one device tensor (float32 iff the reference's collate would make floats, :296-299)."""
        if self._y_dev is None:
            ds, n = self.pyg_dataset, len(self)
            got = _csr.collated_storage(ds) or _csr.collated_storage(getattr(ds, "zinc_dataset", None))
            if got is not None and got["y"] is not None:            # labels from the collated storage: no item is touched
                yv = torch.as_tensor(got["y"]).reshape(-1)
                if got["indices"] is not None:
                    yv = yv[torch.as_tensor(got["indices"], dtype=torch.int64)]
                self._y_dev = yv.to(torch.float if yv.is_floating_point() else torch.long).to(self._dev())
            else:
                labels = [ds[i].y.item() for i in range(n)]
                is_float = n > 0 and isinstance(labels[0], float)
                self._y_dev = torch.tensor(labels, dtype=torch.float if is_float else torch.long, device=self._dev())
        return self._y_dev

    # ---- batch-level fetch: what torch.utils.data.DataLoader calls instead of __getitem__ when a dataset has it
    def __getitems__(self, indices):
        """One pre-collated batch per call: the stock DataLoader of the reference (trainer/train_agtt.py:599-607,
        `DataLoader(ds, batch_size, shuffle, num_workers=0, collate_fn=collate_fn)`) hands its index list here and the
        result to collate_fn.  The batch is collated ON THE DEVICE (gtok_collate_packed over the epoch's 16-bit slab: X
        int64 pad 5, attn bool, labels - the tensors the trainer moves `.to(device)` anyway) and wrapped in a list-like
        object: this module's collate_fn passes it through; any other collate function (the reference's own, in the
        zero-edit layout) iterates it and gets the per-item tuples of __getitem__, made on demand.  A batch that asks for
        a row already served starts the next epoch (a second fetch of an item is a new random trail in the reference)."""
        if not (self._batched and torch.utils.data.get_worker_info() is None):
            return [self[i] for i in indices]
        self._graphs()
        if self._mixed_query:
            return [self[i] for i in indices]
        plan = self._plan
        if plan is not None:
            k = plan["ids"].get(id(indices))
            if k is not None and plan["lists"][k] is indices:
                return self._planned_batch(plan, k)
        idx = np.asarray(indices, dtype=np.int64)
        if len(set(indices)) != idx.size:
            # a sampler with replacement put an item into the batch twice: in the reference every fetch of an item is a new
            # random trail, and one epoch slab holds one trail per item - the per-item path serves such a batch
            return [self[i] for i in indices]
        if self._served is None or self._ids is None or self._ids.dtype != torch.int16 or self._served[idx].any():
            self.tokenize_epoch_u16(self._epoch + 1)
            self._served = np.zeros(len(self), dtype=bool)
            self._lens_h = self._lens.cpu().numpy()
        self._served[idx] = True
        lmax = int(np.minimum(self._lens_h[idx], self._ids.shape[1]).max()) if idx.size else 0
        # the index list stays on the host: it rides in the launch's arguments, and the labels come out of the same launch
        X, A, Y = _ops.collate_batch(self._ids, None, self._lens, self._ids.shape[1], idx, PAD, lmax, self._labels_on_device())
        return CollatedBatch(self, idx, X, A, Y, self._epoch)

    def plan_epoch(self, order: np.ndarray, batch_size: int, drop_last: bool = False):
        """The batches of one epoch, announced before they are asked for (EpochBatchSampler(..., dataset=this) does it): returns the
        index lists to hand to the DataLoader; when __getitems__ later receives one of these very list objects it serves the batch out
        of ONE gtok_collate_epoch call over the whole plan (made when the plan's first batch is asked for - that is also when the
        next epoch's trails are taken, as a batch that revisits a served row would) - three views per batch, no launch.  Lists that
        are not the plan's (or a plan abandoned half-way) take the per-batch path as before."""
        order = np.ascontiguousarray(order, dtype=np.int64)
        n = order.size - (order.size % batch_size if drop_last else 0)
        lists = [order[s:s + batch_size].tolist() for s in range(0, n, batch_size)]
        self._plan = dict(order=order[:n], bs=int(batch_size), lists=lists, ids={id(l): k for k, l in enumerate(lists)}, arenas=None)
        return lists

    def _planned_batch(self, plan, k):
        ar = plan["arenas"]
        if ar is None:
            order = plan["order"]
            if len(np.unique(order)) != order.size:
                self._plan = None                           # (a plan with repeats: every fetch of an item is a new trail - per-batch path)
                return self.__getitems__(list(plan["lists"][k]))
            if self._served is None or self._ids is None or self._ids.dtype != torch.int16 or self._served[order].any():
                self.tokenize_epoch_u16(self._epoch + 1)
                self._served = np.zeros(len(self), dtype=bool)
                self._lens_h = self._lens.cpu().numpy()
            self._served[order] = True
            od = torch.from_numpy(order).to(self._ids.device)
            X, A, lmax, off = _ops.collate_epoch(self._ids, None, self._lens, self._ids.shape[1], od, plan["bs"], PAD)
            ar = plan["arenas"] = (X, A, lmax, off, self._labels_on_device()[od].split(plan["bs"]), self._epoch)
        X, A, lmax, off, ys, epoch = ar
        lst = plan["lists"][k]
        B, L, o = len(lst), lmax[k], off[k]
        return CollatedBatch(self, lst, X.as_strided((B, L), (L, 1), o), A.as_strided((B, L), (L, 1), o), ys[k], epoch)

    def device_batches(self, batch_size: int, epoch: int, shuffle: bool = False,
                       generator: Optional[torch.Generator] = None, with_data: bool = True):
        """Yield collate_fn's tuple with X/attn/labels on the device (gtok_collate_packed over the epoch's 16-bit slab).
        with_data=True: the 4th element is a lazy sequence of the batch's Data objects (collate_fn hands them on and the
        model only reads data_list[0].num_nodes, train_agtt.py:127-133: an item is made when it is asked for);
        with_data=False: an empty list."""
        ids, lens = self.tokenize_epoch_u16(epoch)
        n = len(self)
        ds = self.pyg_dataset
        y = self._labels_on_device()
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        order_d = order.to(ids.device)
        # the whole epoch collated by one call (gtok_collate_epoch): a batch is three views of the epoch's arenas - no launch, no
        # allocation, no host-side maximum per batch (round 4: one gtok_collate_packed launch + a lens[idx].max() per batch)
        X, A, lmax, off = _ops.collate_epoch(ids, None, lens, ids.shape[1], order_d, batch_size, PAD)
        ys = y[order_d].split(batch_size)                    # every batch's labels in one call
        order_l = order.tolist() if with_data else None
        for b, s in enumerate(range(0, n, batch_size)):
            B, L, o = min(batch_size, n - s), lmax[b], off[b]
            yield (X.as_strided((B, L), (L, 1), o), A.as_strided((B, L), (L, 1), o), ys[b],
                   LazyDataList(ds, order_l[s:s + B]) if with_data else [])


class EpochBatchSampler(torch.utils.data.Sampler):
    """Index lists of whole batches, cut from ONE permutation per epoch: `DataLoader(ds, batch_sampler=EpochBatchSampler(len(ds), 128,
    shuffle=True), collate_fn=collate_fn)`.  The stock `shuffle=True` loader draws its indices one Python int at a time through two
    generator frames - 38 us per batch of 128 on the bench host before the dataset is asked for anything, i.e. at most 3.4 x 10^6
    items/s whatever the dataset does (`profiles/r05/time_loader.txt`); this sampler hands the same kind of lists over at ~3 us per
    batch.  Same distribution as RandomSampler without replacement (a fresh `torch.randperm` per epoch, from `generator` if given).
    dataset=<the TokenizedGraphDataset>: the sampler also announces every epoch's batches to the dataset (`plan_epoch`), which then
    collates the whole epoch in one launch and serves each batch as views."""

    def __init__(self, num_items, batch_size: int, shuffle: bool = False, drop_last: bool = False,
                 generator: Optional[torch.Generator] = None, dataset=None):
        if dataset is None and hasattr(num_items, "plan_epoch"):        # EpochBatchSampler(ds, 128, shuffle=True): the dataset itself
            dataset, num_items = num_items, len(num_items)
        self.n, self.batch_size, self.shuffle, self.drop_last, self.generator = int(num_items), int(batch_size), shuffle, drop_last, generator
        self.dataset = dataset if hasattr(dataset, "plan_epoch") else None

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else -(-self.n // self.batch_size)

    def __iter__(self):
        order = (torch.randperm(self.n, generator=self.generator) if self.shuffle else torch.arange(self.n)).numpy()
        bs = self.batch_size
        if self.dataset is not None:
            # the dataset learns the epoch's batches before the loader asks for them: it collates them all in one call and answers
            # each list with views (TokenizedGraphDataset.plan_epoch)
            yield from self.dataset.plan_epoch(order, bs, self.drop_last)
            return
        for s in range(0, self.n - (self.n % bs if self.drop_last else 0), bs):
            yield order[s:s + bs].tolist()


class LazyDataList(Sequence):
    """The batch's Data objects as a sequence that fetches `dataset[i]` when element i is asked for: collate_fn returns
    `list(data_list)` (trainer/train_agtt.py:301) and the model reads data_list[0].num_nodes only (:127-133)."""

    def __init__(self, dataset, indices):
        self._ds, self._idx = dataset, indices

    def __len__(self):
        return len(self._idx)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return LazyDataList(self._ds, self._idx[i])
        return self._ds[self._idx[i]]


class CollatedBatch(Sequence):
    """What TokenizedGraphDataset.__getitems__ returns: the batch already collated on the device (`collated`), and -
    for a collate function that does not know about it - a sequence of the per-item tuples (tokens, mask, label, data)
    that __getitem__ would have produced for the same epoch, cut from the device batch on demand."""

    def __init__(self, owner, idx, X, A, Y, epoch):
        self._owner, self._idx, self.epoch = owner, idx, epoch
        self.collated = (X, A, Y, LazyDataList(owner.pyg_dataset, idx if isinstance(idx, list) else idx.tolist()))
        self._items = None

    def __len__(self):
        return len(self._idx)

    def pin_memory(self):
        """DataLoader(pin_memory=True) pins what the fetcher returns; this batch lives on the device already (a CUDA tensor cannot
        be pinned: without this method the loader's pin thread raises)."""
        return self

    def _materialise(self):
        if self._items is None:
            X, A, Y, datas = self.collated
            Xh, n = X.cpu(), A.sum(1).cpu().tolist()
            ys = Y.cpu().tolist()
            self._items = [(Xh[b, :n[b]].clone(), torch.ones(n[b], dtype=torch.bool), ys[b], datas[b]) for b in range(len(n))]
        return self._items

    def __getitem__(self, i):
        return self._materialise()[i]

    def __iter__(self):
        return iter(self._materialise())


def collate_fn(batch):
    """[(tokens, mask, label, data)] -> (X int64 [B,L] pad 5, attn bool [B,L], labels, list(data)) (:276-302).
    A batch that TokenizedGraphDataset.__getitems__ collated on the device already is passed through."""
    if isinstance(batch, CollatedBatch):
        return batch.collated
    toks, masks, labels, datas = zip(*batch)
    L = max(t.size(0) for t in toks)
    X = torch.full((len(toks), L), PAD, dtype=torch.long)
    A = torch.zeros((len(toks), L), dtype=torch.bool)
    for i, (t, m) in enumerate(zip(toks, masks)):
        X[i, :t.size(0)] = t
        A[i, :t.size(0)] = m
    dtype = torch.float if isinstance(labels[0], float) else torch.long
    return X, A, torch.tensor(labels, dtype=dtype), list(datas)
