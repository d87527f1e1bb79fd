"""CPU: graphs reach the batched CSR without a per-item Python loop when the dataset exposes collated storage
(torch_geometric's InMemoryDataset layout), with the same arrays as the item-by-item builder; items are marked with
the split they come from; the graph-token dataset's cache never shadows the reference's own `data.pt`."""
import os

import numpy as np
import pytest
import torch

from _util import gtok, zinc_data_list

FIELDS = ("node_ptr", "edge_ptr", "rowptr", "col", "nattr", "eattr")


def _same(a, b):
    for k in FIELDS:
        x, y = getattr(a, k), getattr(b, k)
        assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), k
    assert (a.eorder is None) == (b.eorder is None) and (a.eorder is None or torch.equal(a.eorder, b.eorder))
    assert (a.num_graphs, a.max_nodes, a.max_edges, a.flags, a.chunk_nodes, a.chunk_edges, a.max_degree) == \
           (b.num_graphs, b.max_nodes, b.max_edges, b.flags, b.chunk_nodes, b.chunk_edges, b.max_degree)


def test_collated_storage_gives_the_arrays_of_the_item_loop():
    for coalesced in (True, False):
        d = gtok.synth.zinc_like(700, seed=31, coalesced=coalesced)
        ref = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
        ds = gtok.synth.InMemoryLike(d)
        assert gtok.csr.collated_storage(ds) is not None
        _same(gtok.GraphBatch.from_dataset(ds), ref)
        _same(gtok.GraphBatch.from_data_list([ds[i] for i in range(len(ds))]), ref)
        _same(gtok.GraphBatch.from_data_list(zinc_data_list(d)), ref)
        # unlabelled view of the same storage
        un = gtok.GraphBatch.from_dataset(ds, labeled=False)
        assert un.nattr is None and un.eattr is None and torch.equal(un.col, ref.col)
    # a subset (dataset.indices()) in its own order, graphs repeated or skipped
    idx = [5, 3, 699, 0, 7, 7, 123]
    sub = gtok.synth.InMemoryLike(d, indices=idx)
    _same(gtok.GraphBatch.from_dataset(sub), gtok.GraphBatch.from_data_list([ds[i] for i in idx]))
    # a per-item transform could change what an item holds: the storage is not trusted, items are fetched
    def drop_last_edge(item):
        item.edge_index, item.edge_attr = item.edge_index[:, :-2], item.edge_attr[:-2]
        return item
    tr = gtok.synth.InMemoryLike(d, transform=drop_last_edge)
    assert gtok.csr.collated_storage(tr) is None
    b = gtok.GraphBatch.from_dataset(tr)
    assert b.num_edges_total == ref.num_edges_total - 2 * 700


def test_from_data_list_tolerates_ragged_items():
    """edge_attr shorter than the edge list reads as 0 ('unknown', zinc_dataset_indexbase.py:183), [E,1] attrs, numpy
    and list inputs, items without num_nodes."""
    from _util import PygLike
    items = [PygLike(x=torch.tensor([[1], [2], [3]]), edge_index=torch.tensor([[0, 1, 2], [1, 2, 0]]), edge_attr=torch.tensor([[1], [2]])),
             PygLike(x=np.array([4, 5]), edge_index=[[0], [1]], edge_attr=[3]),
             PygLike(x=torch.zeros((0, 1), dtype=torch.long), edge_index=torch.empty((2, 0), dtype=torch.long),
                     edge_attr=torch.empty(0, dtype=torch.long))]
    b = gtok.GraphBatch.from_data_list(items, labeled=True)
    assert b.node_ptr.tolist() == [0, 3, 5, 5] and b.edge_ptr.tolist() == [0, 3, 4, 4]
    assert b.nattr.tolist() == [1, 2, 3, 4, 5] and b.eattr.tolist() == [1, 2, 0, 3] and b.col.tolist() == [1, 2, 0, 1]


def test_items_carry_their_split():
    d = gtok.synth.zinc_like(20, seed=2)
    gdl = gtok.graph_data_loader
    ds = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=gtok.synth.InMemoryLike(d))
    it = ds[7]
    owner, idx = gtok.rows.item_source(it)
    assert owner is ds and idx == 7 and it.edge_attr.dim() == 1
    assert gtok.rows.item_source(gtok.synth._Bag(x=1)) is None
    class Frozen:                       # an object that refuses attributes is still recognised right after the fetch
        __slots__ = ("y",)
    f = Frozen()
    gtok.rows.tag_item(ds, 3, f)
    assert gtok.rows.item_source(f) == (ds, 3) and gtok.rows.item_source(Frozen()) is None
    # the mark is plain data: an item pickles (DataLoader workers, torch.save) and is a stranger once its dataset is gone
    import pickle
    clone = pickle.loads(pickle.dumps(it))
    assert gtok.rows.item_source(clone) == (ds, 7)
    other = gdl.ZINCDatasetForAutoGraph(split="val", zinc_dataset=gtok.synth.InMemoryLike(d))
    orphan = other[2]
    del other
    import gc; gc.collect()
    assert gtok.rows.item_source(orphan) is None
    _same(ds.graph_batch(), gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"]))
    ib = gdl.ZINCTokenizationDataset(split="train", zinc_dataset=gtok.synth.InMemoryLike(d))
    _same(ib.graph_batch(), ds.graph_batch())
    assert torch.allclose(ib.labels(), torch.from_numpy(d["y"]))


def test_graph_token_cache_does_not_shadow_the_reference_file(tmp_path):
    """The reference loads `<root>/processed/<key>/data.pt` with torch.load and skips process() whenever that file
    exists (graph_token_dataset_autograph.py:233-253, :407-408).  Our cache lives beside it under another name: a
    reference-written data.pt is left byte for byte as it was, and we never create one."""
    tree = gtok.synth.graph_token_tree(12, seed=5, task="shortest_path")
    gtok.synth.write_tree(str(tmp_path), tree)
    kw = dict(root=str(tmp_path), task="shortest_path", algorithm=["er", "ba"], split="train", num_pairs_per_graph=2)
    G = gtok.graph_data_loader.GraphTokenDatasetForAutoGraph
    first = G(**kw)
    key_dir = first.processed_dir
    assert os.listdir(key_dir) == ["data_gtok.pt"] and first.processed_paths[0].endswith("data.pt")
    # what the reference would leave there
    ref_file = os.path.join(key_dir, "data.pt")
    with open(ref_file, "wb") as f:
        f.write(b"reference-owned bytes")
    again = G(**kw)                      # served from our cache, the reference's file untouched
    assert open(ref_file, "rb").read() == b"reference-owned bytes"
    assert len(again) == len(first) > 0
    for a, b in zip(first, again):
        assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.y, b.y) and a.num_nodes == b.num_nodes \
            and (a.query_u, a.query_v) == (b.query_u, b.query_v)
    # items are marked, the split's CSR comes from the collated arrays
    assert gtok.rows.item_source(again[3]) == (again, 3)
    _same(again.graph_batch(), gtok.GraphBatch.from_data_list([again[i] for i in range(len(again))], labeled=False))
    assert again.queries().shape == (len(again), 2)


@pytest.mark.parametrize("legacy", [False, True])
def test_ingestion_and_item_marks_on_objects_with_torch_geometric_semantics(legacy):
    """The reference hands over torch_geometric.datasets.ZINC (zinc_dataset_autograph.py:44, zinc_dataset_indexbase.py:79);
    torch_geometric is not installed here, so the layout and attribute semantics this package relies on are restated in
    tests/_util.py (PygDataLike / PygInMemoryLike: attribute access through a storage mapping, underscore names kept out of
    keys(), AttributeError for missing names, num_nodes inferred from x, a copy per fetch, index_select through `_indices`,
    `_data` or - before 2.3 - only `data`).  On such objects: the collated storage is ingested without touching an item,
    subsets follow `_indices`, item marks survive copy / clone and stay out of the item's keys, foreign items are strangers."""
    import copy
    from _util import PygDataLike, PygInMemoryLike
    d = gtok.synth.zinc_like(300, seed=9)
    ref = gtok.GraphBatch.from_coo(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"])
    pyg = PygInMemoryLike(d, legacy=legacy)
    got = gtok.csr.collated_storage(pyg)
    assert got is not None and got["indices"] is None and got["x"] is not None
    _same(gtok.GraphBatch.from_dataset(pyg, labeled=True), ref)
    sub = pyg[[5, 3, 250, 7]]                                    # index_select: a shallow copy with _indices
    picked = [pyg[i] for i in (5, 3, 250, 7)]
    _same(gtok.GraphBatch.from_dataset(sub, labeled=True), gtok.GraphBatch.from_data_list(picked, labeled=True))
    assert gtok.csr.collated_storage(PygInMemoryLike(d, legacy=legacy, transform=lambda it: it)) is None     # a transform: item by item
    from torch.utils.data import Subset
    nested = Subset(Subset(sub, [3, 0, 2]), [1, 2])              # torch's own subsets compose with index_select: graphs 5, 250
    assert gtok.csr.collated_storage(nested)["indices"] == [5, 250]
    _same(gtok.GraphBatch.from_dataset(nested, labeled=True), gtok.GraphBatch.from_data_list([pyg[5], pyg[250]], labeled=True))
    gdl = gtok.graph_data_loader
    ds = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=pyg)
    _same(ds.graph_batch(), ref)
    it = ds[11]
    assert isinstance(it, PygDataLike) and gtok.rows.item_source(it) == (ds, 11)
    assert sorted(it.keys()) == ["edge_attr", "edge_index", "x", "y"] and it.edge_attr.dim() == 1 and it.num_nodes == int(d["node_counts"][11])
    assert gtok.rows.item_source(copy.copy(it)) == (ds, 11) and gtok.rows.item_source(it.clone()) == (ds, 11)
    assert gtok.rows.item_source(pyg[11]) is None                # fetched from the torch_geometric dataset itself: no mark
    with pytest.raises(AttributeError):
        it.no_such_attribute
    ib = gdl.ZINCTokenizationDataset(split="train", zinc_dataset=pyg)
    _same(ib.graph_batch(), ref)
    assert ib.labels().dtype == torch.float32 and torch.equal(ib.labels(), torch.from_numpy(d["y"]))
    item = ib._item(4)                                           # the per-item Python of the IBTT dataset on such an object
    assert item["text"].startswith("<bos> <atom> ") and item["text"].endswith(" <eos>") and item["graph_id"] == "zinc_train_4"
