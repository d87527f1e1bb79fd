"""sent_reg_kernel (labelled + remap) at 12 k and 31 k molecules, best of 3 x 300 back-to-back launches; GTOK_MAX_BLOCKS_PER_CU from the environment."""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
gtok = importlib.import_module("glearning-benchmark_amd")
dev = torch.device("cuda", 0)
kw = dict(labeled=True, num_node_types=9, num_edge_types=4, remap_zinc=True)
os.environ["GTOK_SENT_KERNEL"] = "reg"
out = []
for G in (12000, 31182):
    d = gtok.synth.zinc_like(G, seed=1000)
    b = gtok.GraphBatch.from_coo_device(d["node_counts"], d["edge_counts"], d["src"], d["dst"], d["x"], d["edge_attr"], device=dev)
    ids = torch.empty((G, 208), dtype=torch.int32, device=dev); ln = torch.empty(G, dtype=torch.int32, device=dev)
    for _ in range(20): gtok.ops.sent(b, 37, 1024, 0, 0, ld=208, out=(ids, ln), **kw)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for k in range(300): gtok.ops.sent(b, 37, 1024, 0, k, ld=208, out=(ids, ln), **kw)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 300)
    out.append(f"G={G}: {best:.4f} ms (checksum {int(ids.sum())})")
print(os.path.basename(gtok._lib.LIB_PATH), os.environ.get("GTOK_MAX_BLOCKS_PER_CU", "-"), " | ".join(out), flush=True)
