"""AGTT dataset glue — `TokenizedGraphDataset` and `collate_fn` with the signatures of
trainer/train_agtt.py:150-302, tokenizing whole epochs on the GPU.

Reference behaviour kept: a fresh random trail every time an item is fetched, ZINC ids remapped to the
fixed vocabulary (remap_zinc_tokens :171-244), `[idx_offset+num_nodes, idx_offset+u, idx_offset+v]`
appended for shortest_path (:257-267), padding with Graph2TrailTokenizer.pad = 5 even after the remap
(:285-286), float32 labels iff the first label is a Python float (:296-299).
"""
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
        self._epoch, self._rows = epoch, None
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
                ids, lens = self.tokenize_epoch(self._epoch + 1, pad=False)
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

    def device_batches(self, batch_size: int, epoch: int, shuffle: bool = False,
                       generator: Optional[torch.Generator] = None, with_data: bool = True):
        """Yield collate_fn's tuple with X/attn/labels on the device (gtok_collate over the epoch's slab).
        with_data=False: the 4th element (the batch's Data objects, which collate_fn hands on and the model only reads
        for shortest_path's query nodes, train_agtt.py:127-133) is an empty list and no item object is touched."""
        ids, lens = self.tokenize_epoch(epoch, pad=False)
        n = len(self)
        ds = self.pyg_dataset
        got = _csr.collated_storage(ds) or _csr.collated_storage(getattr(ds, "zinc_dataset", None))
        if got is not None and got["y"] is not None:            # labels from the collated storage: no item is touched
            yv = torch.as_tensor(got["y"]).reshape(-1)
            if got["indices"] is not None:
                yv = yv[torch.as_tensor(got["indices"], dtype=torch.int64)]
            y = yv.to(torch.float if yv.is_floating_point() else torch.long).to(ids.device)
        else:
            labels = [ds[i].y.item() for i in range(n)]
            is_float = n > 0 and isinstance(labels[0], float)
            y = torch.tensor(labels, dtype=torch.float if is_float else torch.long, device=ids.device)
        lens_h = lens.cpu()
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        order_d = order.to(ids.device)
        for s in range(0, n, batch_size):
            idx = order[s:s + batch_size]
            idx_d = order_d[s:s + batch_size]
            X, A = _ops.collate(ids, lens, idx_d, PAD, int(lens_h[idx].max()))
            yield X, A, y[idx_d], ([ds[i] for i in idx.tolist()] if with_data else [])


def collate_fn(batch):
    """[(tokens, mask, label, data)] -> (X int64 [B,L] pad 5, attn bool [B,L], labels, list(data)) (:276-302)."""
    toks, masks, labels, datas = zip(*batch)
    L = max(t.size(0) for t in toks)
    X = torch.full((len(toks), L), PAD, dtype=torch.long)
    A = torch.zeros((len(toks), L), dtype=torch.bool)
    for i, (t, m) in enumerate(zip(toks, masks)):
        X[i, :t.size(0)] = t
        A[i, :t.size(0)] = m
    dtype = torch.float if isinstance(labels[0], float) else torch.long
    return X, A, torch.tensor(labels, dtype=dtype), list(datas)
