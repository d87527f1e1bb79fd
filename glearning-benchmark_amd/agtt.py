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
        self._epoch = -1
        self._served: Optional[np.ndarray] = None
        self._ids = self._lens = self._ids_h = self._lens_h = None

    def __len__(self):
        return len(self.pyg_dataset)

    # ---- whole-epoch tokenization (one SENT launch; remap and query append fused into it)
    def _dev(self):
        if self._device is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        return torch.device(self._device)

    def _graphs(self) -> GraphBatch:
        if self._batch is None:
            items = [self.pyg_dataset[i] for i in range(len(self))]
            gb = getattr(self.pyg_dataset, "graph_batch", None)
            host = gb() if callable(gb) else GraphBatch.from_data_list(items, labeled=self.tokenizer.labeled_graph)
            self._batch = host.to(self._dev())
            if self.task == "shortest_path" and items and all(hasattr(d, "query_u") and hasattr(d, "query_v") for d in items):
                self._query = torch.tensor([[d.query_u, d.query_v] for d in items], dtype=torch.int32)
            self._mixed_query = self.task == "shortest_path" and self._query is None and \
                any(hasattr(d, "query_u") and hasattr(d, "query_v") for d in items)
        return self._batch

    def tokenize_epoch(self, epoch: int):
        """(ids int32 [G, ld], len int32 [G]) on the device for `epoch`; also what __getitem__ serves from."""
        batch = self._graphs()
        # every reader of the slab goes through the lengths (items are cut at len, gtok_collate pads per batch as the
        # reference's collate_fn does): the pad tails - more than half of a ZINC slab - are not written
        self._ids, self._lens = self.tokenizer.tokenize_batch(batch, epoch=epoch, remap_zinc=self.remap_to_fixed_vocab,
                                                              query=self._query, pad=False)
        self._epoch, self._ids_h, self._lens_h = epoch, None, None
        self._served = np.zeros(len(self), bool)
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
            if self._served is None or self._served[idx]:       # fetched again -> a new random trail
                self.tokenize_epoch(self._epoch + 1)
            if self._ids_h is None:
                self._ids_h, self._lens_h = self._ids.cpu(), self._lens.cpu()
            self._served[idx] = True
            tokens = self._ids_h[idx, :int(self._lens_h[idx])].to(torch.long)
        else:                                                   # any tokenizer object: per-item call
            tokens = self.tokenizer(data)
            if self.remap_to_fixed_vocab:
                tokens = self.remap_zinc_tokens(tokens, data)
            tail = self._query_tail(data)
            if tail is not None:
                tokens = torch.cat([tokens, tail])
        return tokens, torch.ones(tokens.size(0), dtype=torch.bool), data.y.item(), data

    def device_batches(self, batch_size: int, epoch: int, shuffle: bool = False,
                       generator: Optional[torch.Generator] = None):
        """Yield collate_fn's tuple with X/attn/labels on the device (gtok_collate over the epoch's slab)."""
        ids, lens = self.tokenize_epoch(epoch)
        n = len(self)
        items = [self.pyg_dataset[i] for i in range(n)]
        labels = [d.y.item() for d in items]
        is_float = n > 0 and isinstance(labels[0], float)
        y = torch.tensor(labels, dtype=torch.float if is_float else torch.long, device=ids.device)
        lens_h = lens.cpu()
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        for s in range(0, n, batch_size):
            idx = order[s:s + batch_size]
            X, A = _ops.collate(ids, lens, idx, PAD, int(lens_h[idx].max()))
            yield X, A, y[idx.to(ids.device)], [items[i] for i in idx.tolist()]


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
