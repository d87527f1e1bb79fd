import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda:0")
G, ld = 249456, 176
d = gtok.synth.zinc_like(G, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
out = []
for K in (8, 16, 24, 32):
    ids = torch.empty((K * G, ld), dtype=torch.int16, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
    f = lambda k: gtok.ops.sent(b, 37, 1024, 0, k * K, ld=ld, out=(ids, ln), pad=False, epochs=K, u16=True, **kw)
    for i in range(6): f(i)
    torch.cuda.synchronize(); best = 1e9
    for rep in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(6): f(k)
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / 6 / K)
    out.append(f"K={K}: {best:.5f}")
    del ids, ln
print("zinc_full u16 unpadded ms per epoch:", "  ".join(out), flush=True)
