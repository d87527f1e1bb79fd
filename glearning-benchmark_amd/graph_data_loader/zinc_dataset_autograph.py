"""ZINC source for AGTT — mirror of the reference's graph_data_loader/zinc_dataset_autograph.py."""
from typing import Optional, Sequence

from torch.utils.data import Dataset

from ._root import root as _root

GraphBatch = _root().GraphBatch
_tag_item = _root().rows.tag_item


class ZINCDatasetForAutoGraph(Dataset):
    """Returns the raw PyG Data per item (reference :51-73), flattening edge_attr [E,1] -> [E]; `graph_batch()`
    is the device-side view of the same split (batched CSR, built once)."""

    def __init__(self, zinc_root: str = "./data/ZINC", split: str = "train", subset: bool = True,
                 zinc_dataset: Optional[Sequence] = None):
        super().__init__()
        self.zinc_root, self.split, self.subset = zinc_root, split, subset
        if zinc_dataset is None:
            from torch_geometric.datasets import ZINC
            zinc_dataset = ZINC(root=zinc_root, subset=subset, split=split)
        self.zinc_dataset = zinc_dataset
        self._batches = {}
        print(f"Loaded ZINC {split} split: {len(self.zinc_dataset)} molecules")

    def __len__(self):
        return len(self.zinc_dataset)

    def __getitem__(self, idx):
        data = self.zinc_dataset[idx]
        ea = data.edge_attr
        if ea.dim() == 2 and ea.size(1) == 1:
            data.edge_attr = ea.flatten()
        return _tag_item(self, idx, data)        # lets Graph2TrailTokenizer tokenize the split this item belongs to at once

    def graph_batch(self, device=None, labeled: bool = True) -> GraphBatch:
        """The split as one batched CSR (built once per device): straight from the collated storage behind
        torch_geometric's ZINC when it is exposed (no per-item work), item by item otherwise."""
        key = (None if device is None else str(device), bool(labeled))
        if key not in self._batches:
            self._batches[key] = GraphBatch.from_dataset(self.zinc_dataset, labeled=labeled, device=device)
        return self._batches[key]


def get_zinc_num_types():
    """(atom types, bond types) the reference assumes for ZINC (:76-100)."""
    return 9, 4
