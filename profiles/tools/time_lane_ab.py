"""One number: sent_lane_kernel (labelled + remap, reordered copy) on the ZINC-full corpus, best of 3 x 200 back-to-back launches."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
d = gtok.synth.zinc_like(249456, seed=1000)
b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
ids = torch.empty((b.num_graphs, 208), dtype=torch.int32, device=dev); ln = torch.empty(b.num_graphs, dtype=torch.int32, device=dev)
for _ in range(20): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), **kw)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(200): gtok.ops.sent(b, 37, 1024, 0, k, ld=208, out=(ids, ln), **kw)
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 200)
print(f"{os.path.basename(gtok._lib.LIB_PATH)}: {best:.4f} ms  (checksum {int(ids.sum())})", flush=True)
