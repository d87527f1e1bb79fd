import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
os.environ["GTOK_SENT_KERNEL"] = "reg"
def permuted(d, perm):
    nc, ec = d["node_counts"], d["edge_counts"]
    nptr = np.concatenate([[0], np.cumsum(nc)]); eptr = np.concatenate([[0], np.cumsum(ec)])
    take = lambda arr, ptr: np.concatenate([arr[ptr[g]:ptr[g+1]] for g in perm])
    return dict(node_counts=nc[perm], edge_counts=ec[perm], src=take(d["src"], eptr), dst=take(d["dst"], eptr), x=take(d["x"], nptr), edge_attr=take(d["edge_attr"], eptr))
for G in (12000, 31182):
    d = gtok.synth.zinc_like(G, seed=1000)
    order = np.argsort(-(d["node_counts"] + 0.001 * d["edge_counts"]), kind="stable")
    half = G // 2
    big, small = order[:half], order[::-1][:G - half]
    pos = np.empty(G, np.int64)
    i = np.arange(half); pos[(i // 4) * 8 + (i % 4)] = big[i] if True else 0
    j = np.arange(G - half)
    slots = (j // 4) * 8 + 4 + (j % 4)
    ok = slots < G
    perm = np.full(G, -1, np.int64)
    perm[(i // 4) * 8 + (i % 4)] = big
    perm[slots[ok]] = small[ok]
    left = [g for g in small[~ok]]
    holes = np.nonzero(perm < 0)[0]
    perm[holes] = left[:len(holes)]
    assert sorted(perm.tolist()) == list(range(G))
    for name, dd in (("dataset order", d), ("sorted desc", permuted(d, order)), ("big+small per wave", permuted(d, perm))):
        b = gtok.GraphBatch.from_coo_device(dd["node_counts"], dd["edge_counts"], dd["src"], dd["dst"], dd["x"], dd["edge_attr"], device=dev)
        ids = torch.empty((G, 208), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
        for _ in range(5): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(100): gtok.ops.sent(b, 37, 1024, 0, k, ld=208, out=(ids, ln), **kw)
        e.record(); torch.cuda.synchronize()
        print(f"G={G} {name:20s} {s.elapsed_time(e) / 100:.4f} ms", flush=True)
