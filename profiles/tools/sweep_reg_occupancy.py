"""sent_reg_kernel at BASELINE config 2 / config 4's 8-GPU share: GTOK_MAX_BLOCKS_PER_CU sweep (molecules per wave vs waves in flight)."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
os.environ["GTOK_SENT_KERNEL"] = "reg"
for G in (12000, 31182):
    d = gtok.synth.zinc_like(G, seed=1000)
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
    ids = torch.empty((G, 208), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
    for cap in ("1", "2", "3", "4", "5", "6", "7", "8"):
        os.environ["GTOK_MAX_BLOCKS_PER_CU"] = cap
        for _ in range(5): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(100): gtok.ops.sent(b, 37, 1024, 0, k, ld=208, out=(ids, ln), **kw)
        e.record(); torch.cuda.synchronize()
        print(f"G={G} blocks/CU<={cap}: {s.elapsed_time(e) / 100:.4f} ms", flush=True)
