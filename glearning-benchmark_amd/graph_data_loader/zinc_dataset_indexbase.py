"""ZINC source for IBTT — mirror of the reference's graph_data_loader/zinc_dataset_indexbase.py.

`__getitem__` keeps the reference's {'text','label','graph_id'} contract (the trainer builds its dynamic
vocab from the strings, trainer/train_ibtt.py:364-369); `tokenize()` is the string-free path: the whole
split goes CSR -> vocab ids in one gtok_ibtt_zinc launch.
"""
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from ._root import root as _root
from .zinc_vocab import ZINC_ATOM_TYPES as ATOM_TYPE_LIST

_ops = _root().ops
GraphBatch = _root().GraphBatch

ZINC_ATOM_TYPES = dict(enumerate(ATOM_TYPE_LIST))
ZINC_BOND_TYPES = {0: "single", 1: "single", 2: "double", 3: "triple", 4: "aromatic"}
_BOND_NAME = {1: "single", 2: "double", 3: "triple", 4: "aromatic"}


def _label_token(label: float) -> str:
    """4.23 -> val_4_23, -2.1 -> val_neg2_10 (reference :192)."""
    return f"val_{label:.2f}".replace(".", "_").replace("-", "neg")


class ZINCTokenizationDataset(Dataset):
    """Same constructor as the reference (:63-81).  Offline there is no PyG/ZINC download, so a sequence
    of PyG-like Data objects (x [N,1], edge_index [2,E], edge_attr [E], y) can be handed in directly with
    `zinc_dataset=`; otherwise torch_geometric.datasets.ZINC is used exactly as the reference does."""

    def __init__(self, zinc_root: str = "./data/ZINC", split: str = "train", subset: bool = True,
                 max_vocab: int = 10000, max_len: int = 2048, zinc_dataset: Optional[Sequence] = None):
        super().__init__()
        self.zinc_root, self.split, self.subset, self.max_len = zinc_root, split, subset, max_len
        if zinc_dataset is None:
            from torch_geometric.datasets import ZINC  # needs PyG + network on first use, like the reference
            zinc_dataset = ZINC(root=zinc_root, subset=subset, split=split)
        self.zinc_dataset = zinc_dataset
        self._batch: Optional[GraphBatch] = None
        self._texts = self._labels = None
        self._bulk = True                 # render the split at once on the first fetch when a GPU is there
        print(f"Loaded ZINC {split} split: {len(self.zinc_dataset)} molecules")

    def __len__(self):
        return len(self.zinc_dataset)

    # -- string interface --------------------------------------------------------------------
    def decode_atom_features(self, node_features):
        idx = np.asarray(node_features).reshape(len(node_features), -1)[:, 0] if len(node_features) else []
        return [ZINC_ATOM_TYPES.get(int(i), "X") for i in idx]

    def decode_bond_features(self, edge_attr):
        return [_BOND_NAME.get(int(b), "unknown") for b in np.asarray(edge_attr).reshape(-1)]

    def tokenize_molecule(self, data, label):
        """'<bos> <atom> C ... <bond> single 0 1 ... <q> regression <p> val_x_xx <eos>' (reference :143-195):
        one <bond> group per undirected pair, taken from its first directed occurrence."""
        toks = ["<bos>"]
        for sym in self.decode_atom_features(data.x):
            toks += ["<atom>", sym]
        bonds = self.decode_bond_features(data.edge_attr)
        ei = np.asarray(data.edge_index).reshape(2, -1)
        seen = set()
        for i, (u, v) in enumerate(zip(ei[0].tolist(), ei[1].tolist())):
            key = (u, v) if u <= v else (v, u)
            if key in seen:
                continue
            seen.add(key)
            toks += ["<bond>", bonds[i] if i < len(bonds) else "unknown", str(u), str(v)]
        toks += ["<q>", "regression", "<p>", _label_token(label), "<eos>"]
        return " ".join(toks)

    def _item(self, idx):
        data = self.zinc_dataset[idx]
        label = data.y.item()
        text = self.tokenize_molecule(data, label)
        toks = text.split()
        if len(toks) > self.max_len:                       # reference :217-221
            text = " ".join(toks[:self.max_len - 1] + ["<eos>"])
        return {"text": text, "label": label, "graph_id": f"zinc_{self.split}_{idx}"}

    def render_all(self, device=None):
        """(texts, labels) of the WHOLE split in two launches instead of a Python loop per atom and bond: the
        serialiser kernel (gtok_ibtt_zinc) emits, per molecule, the positions of its tokens in a string table, and
        gtok_ids_to_text joins the strings.  The label token and `<eos>` (one distinct string per molecule: val_x_xx,
        reference :192) and the `max_len` cut (:217-221) are rendered per row by gtok_zinc_text_tails.  Same strings as
        `_item`."""
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        batch = self.graph_batch(device)
        G = batch.num_graphs
        strings = ["<bos>", "<eos>", "<atom>", "<bond>", "<q>", "regression", "<p>"] + list(_ops.ZINC_ATOM_SYMBOLS) \
            + list(_ops.ZINC_BOND_NAMES) + [str(i) for i in range(max(batch.max_nodes, 1))]
        lut = torch.arange(len(strings), dtype=torch.int32)        # the kernel's LUT position IS the string's index
        ids, ln = _ops.ibtt_zinc(batch, lut, 1 << 30, 0)
        y = self.labels()
        labels = y.tolist()
        if y.dtype == torch.float32:                               # the tails on the device too (gtok_zinc_text_tails)
            take, sb, sp = _ops.zinc_text_tails(y.to(ids.device), ln, self.max_len)
            blob, ptr = _ops.ids_to_text(ids, take, strings, (sb, sp))
        else:                                                      # labels of another width: Python formats them
            ln_h = ln.cpu().numpy().astype(np.int64)
            cut = ln_h + 2 > self.max_len                          # tokens = ids + [label, <eos>]
            take = np.where(cut, self.max_len - 1, ln_h)
            tail = [(b" <eos>" if k else b"<eos>") if c else (" " + _label_token(v) + " <eos>").encode("ascii")
                    for c, k, v in zip(cut.tolist(), take.tolist(), labels)]
            blob, ptr = _ops.ids_to_text(ids, torch.from_numpy(take.astype(np.int32)).to(ids.device), strings, tail)
        raw = str(memoryview(blob.cpu().numpy()), "ascii")         # one decode; the items are slices of it
        p = ptr.cpu().tolist()
        return list(map(raw.__getitem__, map(slice, p[:-1], p[1:]))), labels

    def __getitem__(self, idx):
        """{'text','label','graph_id'} (reference :197-227).  With a GPU the whole split is rendered on the first
        fetch (render_all) and items are served from that; without one - host-side tests, plumbing - the item is
        rendered on its own by the restated Python."""
        if self._texts is None and self._bulk and self.max_len >= 2 and len(self) > 1 and torch.cuda.is_available():
            self._texts, self._labels = self.render_all()
        if self._texts is None:
            return self._item(idx)
        i = idx + len(self) if idx < 0 else idx
        return {"text": self._texts[i], "label": self._labels[i], "graph_id": f"zinc_{self.split}_{idx}"}

    # -- device interface --------------------------------------------------------------------
    def graph_batch(self, device=None) -> GraphBatch:
        """The split as one batched CSR (built once per device): straight from the collated storage behind
        torch_geometric's ZINC when it is exposed (no per-item work), item by item otherwise."""
        key = None if device is None else str(device)
        if self._batch is None:
            self._batch = {}
        if key not in self._batch:
            self._batch[key] = GraphBatch.from_dataset(self.zinc_dataset, labeled=True, device=device)
        return self._batch[key]

    def labels(self) -> torch.Tensor:
        got = _root().csr.collated_storage(self.zinc_dataset)
        if got is not None and got["y"] is not None:
            y = torch.as_tensor(got["y"]).reshape(-1)
            y = y if y.dtype in (torch.float32, torch.float64) else y.to(torch.float64)   # (the reference formats y.item(): the stored width counts)
            return y if got["indices"] is None else y[torch.as_tensor(got["indices"], dtype=torch.int64)]
        ys = [self.zinc_dataset[i].y for i in range(len(self))]
        f32 = all(getattr(v, "dtype", None) == torch.float32 for v in ys)
        return torch.tensor([float(v.item()) for v in ys], dtype=torch.float32 if f32 else torch.float64)

    def tokenize(self, vocab: Dict[str, int], max_len: Optional[int] = None, device=None,
                 ld: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """ids int32 [G, ld] + lengths on the device, equal to TokenDataset(self[i]..., vocab, max_len) row by row."""
        max_len = self.max_len if max_len is None else max_len
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        batch = self.graph_batch(device)
        lut = _ops.zinc_lut(vocab, max(batch.max_nodes, 1))
        return _ops.ibtt_zinc(batch, lut, max_len, vocab["<pad>"], ld=ld)


def collate_zinc_batch(batch, pad_id):
    """Unused in the reference too (:230-252): returns the texts and float labels of a batch."""
    return [b["text"] for b in batch], torch.tensor([b["label"] for b in batch], dtype=torch.float32)
