"""ZINC source for AGTT — mirror of the reference's graph_data_loader/zinc_dataset_autograph.py."""
from typing import Optional, Sequence

from torch.utils.data import Dataset

from ._root import root as _root

GraphBatch = _root().GraphBatch


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
        self._batch = None
        print(f"Loaded ZINC {split} split: {len(self.zinc_dataset)} molecules")

    def __len__(self):
        return len(self.zinc_dataset)

    def __getitem__(self, idx):
        data = self.zinc_dataset[idx]
        ea = data.edge_attr
        if ea.dim() == 2 and ea.size(1) == 1:
            data.edge_attr = ea.flatten()
        return data

    def graph_batch(self) -> GraphBatch:
        if self._batch is None:
            self._batch = GraphBatch.from_data_list([self[i] for i in range(len(self))], labeled=True)
        return self._batch


def get_zinc_num_types():
    """(atom types, bond types) the reference assumes for ZINC (:76-100)."""
    return 9, 4
