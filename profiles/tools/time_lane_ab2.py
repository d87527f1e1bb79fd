"""Two numbers for an A/B of sent_lane_kernel on the ZINC-full corpus: the headline form (one epoch per launch, int32 padded slab) and the
dataset classes' form (4 epochs per launch, 16-bit rows, no padding: the walk at its vector-issue bound).  Best of 3 x 100 launches."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
d = gtok.synth.zinc_like(249456, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
G = b.num_graphs
def run(K, u16, pad, n=100):
    ids = torch.empty((K * G, 208), dtype=torch.int16 if u16 else torch.int32, device=dev); ln = torch.empty(K * G, dtype=torch.int32, device=dev)
    for _ in range(10): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), epochs=K, u16=u16, pad=pad, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(n): gtok.ops.sent(b, 37, 1024, 0, k * K, ld=208, out=(ids, ln), epochs=K, u16=u16, pad=pad, **kw)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n / K)
    return best, int(ln.sum()) ^ int(ids[:1000].to(torch.int64).sum())
a = run(1, False, True); c = run(4, True, False)
print(f"{os.path.basename(gtok._lib.LIB_PATH)}: int32 padded K=1 {a[0]:.4f} ms; 16-bit rows K=4 {c[0]:.4f} ms per epoch  (checksums {a[1]} {c[1]})", flush=True)
