"""Where a batch of the stock DataLoader route goes (agtt.TokenizedGraphDataset.__getitems__ + collate_fn): the loader's own cost
(a dataset whose __getitems__ returns a constant), __getitems__ alone on sequential and on shuffled index lists, and the
pieces of __getitems__.  python profiles/tools/time_loader.py"""
import importlib, os, sys, time
import numpy as np, torch
from torch.utils.data import DataLoader, Dataset
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
gdl = gtok.graph_data_loader
dev = torch.device("cuda:0")
G = 249456
d = gtok.synth.zinc_like(G, seed=1000)
pyg = gtok.synth.InMemoryLike(d)
src = gdl.ZINCDatasetForAutoGraph(split="train", zinc_dataset=pyg)
tok = gtok.Graph2TrailTokenizer(dataset_names=[], max_length=1024, truncation_length=1024, labeled_graph=True, undirected=True, device=dev)
tok.set_num_nodes(37); tok.set_num_node_and_edge_types(*gdl.get_zinc_num_types())
ds = gtok.agtt.TokenizedGraphDataset(src, tok, task="zinc", remap_to_fixed_vocab=True, device=dev)


class Dummy(Dataset):
    def __len__(self): return G
    def __getitem__(self, i): return i
    def __getitems__(self, idx): return idx


def clock(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r


for shuffle in (False, True):
    dl = DataLoader(Dummy(), batch_size=128, shuffle=shuffle, num_workers=0, collate_fn=lambda b: b)
    clock(lambda: sum(1 for _ in dl))
    t, nb = clock(lambda: sum(1 for _ in dl))
    print(f"loader alone, shuffle={shuffle}: {t / nb * 1e6:.1f} us per batch")
    dl = DataLoader(ds, batch_size=128, shuffle=shuffle, num_workers=0, collate_fn=gtok.agtt.collate_fn)
    clock(lambda: sum(1 for _ in dl))
    t, nb = clock(lambda: sum(1 for _ in dl))
    print(f"loader + dataset, shuffle={shuffle}: {t / nb * 1e6:.1f} us per batch = {G / t:.3e} items/s")

order = np.random.default_rng(0).permutation(G)
for name, o in (("sequential", np.arange(G)), ("shuffled", order)):
    lists = [o[s:s + 128].tolist() for s in range(0, G, 128)]
    ds.__getitems__(lists[0])
    t, _ = clock(lambda: [ds.__getitems__(ix) for ix in lists])
    print(f"__getitems__ alone, {name}: {t / len(lists) * 1e6:.1f} us per batch")
    ds._graphs()
    ds.tokenize_epoch_u16(ds._epoch + 1)
    lens_h = ds._lens.cpu().numpy()
    y = ds._labels_on_device()
    t1, _ = clock(lambda: [np.asarray(ix, dtype=np.int64) for ix in lists])
    arrs = [np.asarray(ix, dtype=np.int64) for ix in lists]
    t2, _ = clock(lambda: [len(set(ix)) for ix in lists])
    t3, _ = clock(lambda: [int(np.minimum(lens_h[a], 176).max()) for a in arrs])
    t4, _ = clock(lambda: [gtok.ops.collate_batch(ds._ids, None, ds._lens, ds._ids.shape[1], a, 5, 120, y) for a in arrs])
    t5, _ = clock(lambda: [gtok.agtt.LazyDataList(src, a.tolist()) for a in arrs])
    n = len(lists)
    print(f"   pieces ({name}): asarray {t1 / n * 1e6:.1f}  set {t2 / n * 1e6:.1f}  lmax {t3 / n * 1e6:.1f}  collate_batch {t4 / n * 1e6:.1f}  lazy list {t5 / n * 1e6:.1f} us")
