import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
d = gtok.synth.zinc_like(249456, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
G = 249456
for K in (4, 8, 16, 4, 8):
    ids = torch.empty((K * G, 208), dtype=torch.int16, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
    n = max(8, 160 // K)
    for _ in range(3): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), epochs=K, u16=True, pad=False, **kw)
    torch.cuda.synchronize(); best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(n): gtok.ops.sent(b, 37, 1024, 0, k * K, ld=208, out=(ids, ln), epochs=K, u16=True, pad=False, **kw)
        e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / n / K)
    print(f"zinc_full x K={K}: {best*1e3:.2f} us per epoch", flush=True)
    del ids, ln
